"""tools/bench_channels.py [--steps K] -- config C forward + backward with 3, 4 and 6 colour channels (colors_precomp [P,C]):
ms per view and the HIP-event time of both blends per channel count (gsr_profile_*)."""
import argparse, ctypes, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, scene

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=40); ap.add_argument("--warmup", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
gs, cams, bg = scene.config_C()
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
means3D, opac, scales, rots = (t(x).requires_grad_(True) for x in (gs.means3D, gs.opacities, gs.scales, gs.rotations))
means2D = torch.zeros(gs.P, 3, device=dev, requires_grad=True)
W, H = cams[0].W, cams[0].H
g = torch.Generator(device=dev).manual_seed(7)
nst = lib.gsr_num_stages(); names = [lib.gsr_stage_name(i).decode() for i in range(nst)]
out = {}
for C in (3, 4, 6):
    col = torch.rand(gs.P, C, device=dev, generator=g).requires_grad_(True)
    d = torch.randn(C, H, W, device=dev, generator=g)
    bgc = torch.rand(C, device=dev, generator=g)
    rs = []
    for cam in cams[:args.steps + args.warmup]:
        vm, pm, cp = t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos)
        rs.append(GaussianRasterizer(GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, bgc, 1.0, vm, pm, 0, cp, False, False)))
    def step(i):
        for p in (means3D, opac, scales, rots, col, means2D):
            p.grad = None
        img, _ = rs[i](means3D=means3D, means2D=means2D, opacities=opac, colors_precomp=col, scales=scales, rotations=rots)
        img.backward(d)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    lib.gsr_profile_enable(1)
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    ms = (ctypes.c_float * nst)(); cnt = (ctypes.c_int * nst)()
    lib.gsr_profile_read(ms, cnt, 1); lib.gsr_profile_enable(0)
    k = {n.replace("_kernel", ""): round(ms[i] / max(cnt[i], 1), 4) for i, n in enumerate(names) if cnt[i]}
    out[f"C={C}"] = {"ms_per_view": round(dt * 1e3, 4), "blend_fwd": k.get("blend_fwd"), "blend_bwd": k.get("blend_bwd"), "geom_bwd": k.get("geom_bwd")}
print(json.dumps(out))

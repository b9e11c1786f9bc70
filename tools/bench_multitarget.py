"""tools/bench_multitarget.py [--steps K] -- one training iteration's rasterizer work on config C, two ways:
(a) the reference's pattern: an RGB render and a depth-as-colour render (bg = 10), each forward + backward
    (gaustar_trainers/refine.py:552, :607);
(b) one 6-channel render (colors_precomp [P,6], bg [6]) forward + backward: gsr_forward_stage2_mt / gsr_backward_mt.
Prints ms per iteration for both and the per-kernel HIP-event times of (b)."""
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
P = gs.P
means3D, opac, scales, rots, rgb = (t(x).requires_grad_(True) for x in (gs.means3D, gs.opacities, gs.scales, gs.rotations, gs.colors_precomp))
means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
W, H = cams[0].W, cams[0].H
g = torch.Generator(device=dev).manual_seed(7)
d_rgb = torch.randn(3, H, W, device=dev, generator=g)
d_dep = torch.randn(3, H, W, device=dev, generator=g); d_dep[1:] = 0
d6 = torch.cat([d_rgb, d_dep])
bg3, bg10 = t(bg), torch.full((3,), 10.0, device=dev)
views = []
for cam in cams[:args.steps + args.warmup]:
    vm, pm, cp = t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos)
    mk = lambda b: GaussianRasterizer(GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, b, 1.0, vm, pm, 0, cp, False, False))
    dep = t(scene.view_depth_colors(gs, cam))   # per-view constant here; the trainer recomputes it from the points
    views.append((mk(bg3), mk(bg10), mk(torch.cat([bg3, bg10])), dep))

def zero():
    for p in (means3D, opac, scales, rots, rgb, means2D):
        p.grad = None

def two_passes(i):
    r_rgb, r_dep, _, dep = views[i]
    zero()
    img, _ = r_rgb(means3D=means3D, means2D=means2D, opacities=opac, colors_precomp=rgb, scales=scales, rotations=rots)
    img.backward(d_rgb)
    img2, _ = r_dep(means3D=means3D, means2D=means2D, opacities=opac, colors_precomp=dep, scales=scales, rotations=rots)
    img2.backward(d_dep)

def one_pass(i):
    _, _, r6, dep = views[i]
    zero()
    img, _ = r6(means3D=means3D, means2D=means2D, opacities=opac, colors_precomp=torch.cat([rgb, dep], 1), scales=scales, rotations=rots)
    img.backward(d6)

def timed(fn):
    for i in range(args.warmup): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e3

res = {"two_3ch_renders_ms": round(timed(two_passes), 4), "one_6ch_render_ms": round(timed(one_pass), 4)}
nst = lib.gsr_num_stages(); names = [lib.gsr_stage_name(i).decode() for i in range(nst)]
ms = (ctypes.c_float * nst)(); cnt = (ctypes.c_int * nst)()
lib.gsr_profile_enable(1); timed(one_pass); lib.gsr_profile_read(ms, cnt, 1); lib.gsr_profile_enable(0)
res["kernels_6ch_ms"] = {n.replace("_kernel", ""): round(ms[i] / max(cnt[i], 1), 4) for i, n in enumerate(names) if cnt[i]}
res["speedup"] = round(res["two_3ch_renders_ms"] / res["one_6ch_render_ms"], 3)
print(json.dumps(res))

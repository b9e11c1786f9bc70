"""tools/bench_config.py {A,B,C,D} [--depth] -- fwd+bwd timing and per-kernel HIP-event times of one BASELINE config
(bench.py itself always measures config C, the configuration the metric is quoted on)."""
import argparse, ctypes, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, scene

ap = argparse.ArgumentParser(); ap.add_argument("config"); ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
gs, cams, bg = getattr(scene, "config_" + a.config)()
cams = cams if isinstance(cams, (list, tuple)) else [cams]
dev = torch.device("cuda:0"); lib = _lib.load()
t = lambda x, g=False: None if x is None else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev).requires_grad_(g)
m3, op, sc, ro = t(gs.means3D, True), t(gs.opacities, True), t(gs.scales, True), t(gs.rotations, True)
shs, cols = t(gs.shs, True), t(gs.colors_precomp, True)
m2 = torch.zeros(gs.P, 3, device=dev, requires_grad=True)
rs = [GaussianRasterizer(GaussianRasterizationSettings(c.H, c.W, c.tanfovx, c.tanfovy, t(bg), 1.0, t(c.viewmatrix), t(c.projmatrix),
                                                       gs.sh_degree, t(c.campos), False, False)) for c in cams[:a.steps + 3]]
dp = torch.randn(3, cams[0].H, cams[0].W, device=dev)
def step(i):
    for p in (m3, op, sc, ro, shs, cols, m2):
        if p is not None: p.grad = None
    img, _ = rs[i % len(rs)](means3D=m3, means2D=m2, opacities=op, shs=shs, colors_precomp=cols, scales=sc, rotations=ro)
    img.backward(dp)
for i in range(3): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(a.steps): step(i)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / a.steps * 1e3
nst = lib.gsr_num_stages(); names = [lib.gsr_stage_name(i).decode() for i in range(nst)]
msv = (ctypes.c_float * nst)(); cnt = (ctypes.c_int * nst)()
lib.gsr_profile_enable(1)
for i in range(a.steps): step(i)
lib.gsr_profile_read(msv, cnt, 1); lib.gsr_profile_enable(0)
print(json.dumps({"config": a.config, "gaussians": gs.P, "sh_coeffs": 0 if gs.shs is None else int(gs.shs.shape[1]), "ms_per_view": round(ms, 4),
                  "views_per_s": round(1e3 / ms, 1),
                  "kernels_ms": {n.replace("_kernel", ""): round(msv[i] / max(cnt[i], 1), 4) for i, n in enumerate(names) if cnt[i]}}))

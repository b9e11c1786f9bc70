"""tools/plan_diag.py CONFIG -- which path each visit of each camera takes (planned / exact / misfit) and what a view costs."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from gaustar_amd import scene
from gaustar_amd import rasterizer as R
name = sys.argv[1] if len(sys.argv) > 1 else "B"
gs, cams, bg = {"B": scene.config_B, "C": scene.config_C, "D": scene.config_D}[name]()
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
m3, op, sc, rot = t(gs.means3D), t(gs.opacities), t(gs.scales), t(gs.rotations)
cols = t(gs.colors_precomp) if gs.colors_precomp is not None else None
e = torch.Tensor([])
n = 20
if not isinstance(cams, (list, tuple)): cams = [cams] * n
camt = [(t(cams[0].viewmatrix), t(cams[0].projmatrix), t(cams[0].campos))] * n if cams[0] is cams[-1] else [(t(c.viewmatrix), t(c.projmatrix), t(c.campos)) for c in cams[:n]]
if cols is None:
    print("SH config: using depth colours"); cols = t(scene.view_depth_colors(gs, cams[0]))
for epoch in range(6):
    before = dict(R.PLAN_STATS)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        c = cams[i]; vm, pm, cp = camt[i]
        out = R.rasterize_gaussians_native(t(bg) if epoch < 0 else camt[0][2].new_tensor(bg), m3, cols, op, sc, rot, 1.0, e, vm, pm, c.tanfovx, c.tanfovy, c.H, c.W, e, 0, cp, False, False, need_backward=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    d = {k: R.PLAN_STATS[k] - before[k] for k in before}
    with R._HINT_LOCK:
        infos = [[v.info[i] for i in range(5)] + [v.skip] for v in R._PLANS.values()]
    print(f"epoch {epoch}: {dt:.4f} ms per forward  {d}  valid plans {sum(1 for i_ in infos if i_[0] == 1)} of {len(infos)}  e.g. {infos[:3]}  R {out[0]} max {out[6]}")

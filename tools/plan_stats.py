"""tools/plan_stats.py -- per camera of config C's rig: is the view plannable, the plan's capacities against the exact counts
(R_cap / R, U_cap / U, longest list), and which path the second and third visit take.  GPU."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from gaustar_amd import scene
from gaustar_amd import rasterizer as R
gs, cams, bg = scene.config_C()
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
m3, op, cols, sc, rot = t(gs.means3D), t(gs.opacities), t(gs.colors_precomp), t(gs.scales), t(gs.rotations)
e = torch.Tensor([])
step = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rows = []
camt = [(t(c.viewmatrix), t(c.projmatrix), t(c.campos)) for c in cams]
for epoch in range(3):
    for i in range(0, len(cams), step):
        c = cams[i]; vm, pm, cp = camt[i]
        before = dict(R.PLAN_STATS)
        out = R.rasterize_gaussians_native(t(bg), m3, cols, op, sc, rot, 1.0, e, vm, pm, c.tanfovx, c.tanfovy, c.H, c.W, e, 0, cp, False, False)
        d = {k: R.PLAN_STATS[k] - before[k] for k in before}
        if epoch == 0:
            rows.append([i, out[0], out[7], out[6]])
        else:
            rows[i // step] += [out[0], out[7], d["planned"], d["misfit"]]
torch.cuda.synchronize()
a = np.array([r[:4] + r[4:8] + r[8:12] for r in rows], dtype=np.int64)
print("cameras", len(rows), " planned on visit 2:", int(a[:, 6].sum()), " on visit 3:", int(a[:, 10].sum()), " misfits:", int(a[:, 7].sum() + a[:, 11].sum()))
pl = a[a[:, 10] == 1]
if len(pl):
    print("planned views: R_cap / R  min/median/max %.3f %.3f %.3f   U_cap / U %.3f %.3f %.3f" % (
        (pl[:, 8] / pl[:, 1]).min(), np.median(pl[:, 8] / pl[:, 1]), (pl[:, 8] / pl[:, 1]).max(),
        (pl[:, 9] / pl[:, 2]).min(), np.median(pl[:, 9] / pl[:, 2]), (pl[:, 9] / pl[:, 2]).max()))
npl = a[a[:, 10] == 0]
print("not planned: cameras", npl[:, 0].tolist()[:40], " their longest lists", npl[:, 3].tolist()[:40])
print("longest list over the rig: min/median/max", a[:, 3].min(), int(np.median(a[:, 3])), a[:, 3].max())
with R._HINT_LOCK:
    infos = [[v.info[i] for i in range(5)] for v in R._PLANS.values()]
print("plan infos (first 8):", infos[:8])

import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from gaustar_amd import scene
from gaustar_amd import rasterizer as R
gs, cams, bg = scene.config_C()
cam = cams[0]
sc_ = np.array(gs.scales, dtype=np.float32, copy=True)
sc_[:, 1:] *= np.exp(np.random.default_rng(7).normal(0.0, 1.0, size=(gs.P, 1))).astype(np.float32)
sc_[:, 1:] = np.maximum(sc_[:, 1:].mean(), sc_[:, 1:])
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
m3, op, cols, sc, rot = t(gs.means3D), t(gs.opacities), t(gs.colors_precomp), t(sc_), t(gs.rotations)
vm, pm, cp, bgt = t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), t(bg)
e = torch.Tensor([])
dp = torch.ones(3, cam.H, cam.W, device=dev)
for i in range(6):
    b = dict(R.PLAN_STATS)
    out = R.rasterize_gaussians_native(bgt, m3, cols, op, sc, rot, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, cam.H, cam.W, e, 0, cp, False, False)
    torch.cuda.synchronize()
    print(i, "forward ok", out[0], out[6], out[7], {k: R.PLAN_STATS[k] - b[k] for k in b}, [list(v.info[j] for j in range(5)) for v in R._PLANS.values()], flush=True)
    g = R.rasterize_gaussians_backward_native(bgt, m3, out[2], cols, sc, rot, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, dp, e, 0, cp, out[3], out[0], out[4], out[5], False, num_segments=out[7])
    torch.cuda.synchronize()
    print(i, "backward ok", flush=True)

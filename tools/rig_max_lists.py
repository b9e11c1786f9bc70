import sys, os, numpy as np, torch, time
sys.path.insert(0, os.getcwd())
from gaustar_amd import scene, _lib
from gaustar_amd import rasterizer as R
gs, cams, bg = scene.config_C()
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
m3, op, cols, sc, rot = t(gs.means3D), t(gs.opacities), t(gs.colors_precomp), t(gs.scales), t(gs.rotations)
e = torch.Tensor([])
mx = []
for c in cams:
    out = R.rasterize_gaussians_native(t(bg), m3, cols, op, sc, rot, 1.0, e, t(c.viewmatrix), t(c.projmatrix), c.tanfovx, c.tanfovy, c.H, c.W, e, 0, t(c.campos), False, False)
    mx.append((out[6], out[0]))
a = np.array(mx)
print("max list per view: min/median/max", a[:,0].min(), np.median(a[:,0]), a[:,0].max(), " views above 1792:", int((a[:,0] > 1792).sum()), "above 1536:", int((a[:,0]>1536).sum()), "above 2048:", int((a[:,0]>2048).sum()))
print(sorted(a[:,0].tolist())[-12:])

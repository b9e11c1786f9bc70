"""tools/dead_units.py [view] -- config C: how many backward work items (unit, block) lie wholly behind the last contributor of
(a) every pixel of their TILE (the unit is dead for all four blocks), (b) every pixel of their BLOCK only."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaustar_amd import _lib, scene
from gaustar_amd import rasterizer as R
view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gs, cams, bg = scene.config_C(); cam = cams[view]
dev = torch.device("cuda:0"); lib = _lib.load()
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
W, H = cam.W, cam.H; gx, gy = (W + 15) // 16, (H + 15) // 16; T = gx * gy
e = torch.Tensor([])
out = R.rasterize_gaussians_native(t(bg), t(gs.means3D), t(gs.colors_precomp), t(gs.opacities), t(gs.scales), t(gs.rotations), 1.0, e,
                                   t(cam.viewmatrix), t(cam.projmatrix), cam.tanfovx, cam.tanfovy, H, W, e, 0, t(cam.campos), False, False, use_plan=False)
Rn, _, _, geom, binning, img, maxc, U = out
P = gs.P
rng_t = torch.zeros(T, 2, dtype=torch.int32, device=dev); pl_t = torch.zeros(max(Rn, 1), dtype=torch.int32, device=dev)
m2 = torch.zeros(P, 2, device=dev); co = torch.zeros(P, 4, device=dev); fT = torch.zeros(H, W, device=dev); nc = torch.zeros(H, W, dtype=torch.int32, device=dev)
pp = lambda x: ctypes.c_void_p(x.data_ptr())
_lib.check(lib.gsr_debug_export(P, Rn, 1, W, H, pp(geom), pp(binning), pp(img), pp(m2), pp(co), None, None, pp(rng_t), pp(pl_t), pp(fT), pp(nc), None), "export")
torch.cuda.synchronize()
rg = rng_t.cpu().numpy().astype(np.int64); n = rg[:, 1] - rg[:, 0]
ncn = np.zeros((gy * 16, gx * 16), np.int64); ncn[:H, :W] = nc.cpu().numpy()
blk = ncn.reshape(gy, 2, 8, gx, 2, 8).max(axis=(2, 5))            # [gy, by, gx, bx] last contributor (1-based) per block
blk = blk.transpose(0, 2, 1, 3).reshape(T, 4)
tile_max = blk.max(1)
units = (n + 63) // 64
tot = int(units.sum()) * 4
live_units_tile = (np.minimum(tile_max, n) + 63) // 64
live_blocks = ((np.minimum(blk, n[:, None]) + 63) // 64).sum()
print(f"R {Rn}  units {units.sum()} (U = {U})  work items {tot}")
print(f"work items behind their TILE's last contributor: {int((units - live_units_tile).sum()) * 4} ({100.0 * (units - live_units_tile).sum() / units.sum():.1f} % of the units)")
print(f"work items behind their BLOCK's last contributor: {tot - int(live_blocks)} ({100.0 * (tot - live_blocks) / tot:.1f} %)")

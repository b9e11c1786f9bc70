"""tools/rig_R.py -- num_rendered of every camera of config C's rig (exact binning): how representative the first K cameras are.  GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gaustar_amd import rasterizer as rz
dev = torch.device("cuda:0")
gs, cams, bg, params, means2D, rasters, dpix = bench.build_workload(dev, 0)
e = torch.Tensor([])
R = []
for r in rasters:
    rs = r.raster_settings
    R.append(rz.rasterize_gaussians_native(rs.bg, params["means3D"].detach(), params["colors"].detach(), params["opacities"].detach(),
                                           params["scales"].detach(), params["rotations"].detach(), 1.0, e, rs.viewmatrix, rs.projmatrix,
                                           rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, e, 0, rs.campos, False, False, use_plan=False)[0])
R = np.array(R, np.float64)
print(f"rig mean R {R.mean():.0f} (min {R.min():.0f}, max {R.max():.0f}); first 20 cameras {R[:20].mean():.0f} ({R[:20].mean() / R.mean():.3f} of the mean); "
      f"first 40 {R[:40].mean():.0f}")
for seed in (0, 1, 2):
    p = np.random.default_rng(seed).permutation(len(R))
    print(f"seeded permutation {seed}: first 20 {R[p[:20]].mean():.0f} ({R[p[:20]].mean() / R.mean():.3f})")
print("by block of 20:", [int(R[i:i + 20].mean()) for i in range(0, len(R), 20)])

"""tools/plan_caps.py [views...] -- a camera's exact (R, U) against its plan's capacities (R_cap, U_cap) under the library in use
(GSR_LIB_PATH) and GSR_PLAN_LEVEL: what the slack costs in address space and in empty work items of the backward.  GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_planned as tp
from gaustar_amd import rasterizer as rz
dev = torch.device("cuda:0")
views = [int(v) for v in sys.argv[1:]] or [0, 21, 90]
for v in views:
    gs, cam, bg = tp._scene("C", v)
    ps, cam_t, bg_t, dpix = tp._inputs(dev, gs, cam, bg)
    rz.drop_plans()
    ex = tp._render(dev, ps, cam_t, bg_t, cam, dpix, use_plan=False)
    for i in range(4):
        r = tp._render(dev, ps, cam_t, bg_t, cam, dpix)
    pl = list(rz._PLANS.values())[0]
    print(f"view {v}: exact R {ex[4][0]} U {ex[4][1]} | plan level {pl.info[4]} R_cap {pl.info[1]} ({pl.info[1] / ex[4][0]:.2f} R) "
          f"U_cap {pl.info[2]} ({pl.info[2] / ex[4][1]:.2f} U) last {r[3]}", flush=True)

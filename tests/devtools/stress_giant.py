"""Dev tool: the giant-splat view of tests/test_gpu_planned.py rendered N times under plans (every view misfits and falls back);
reports image mismatches and the spread of every gradient against the exact path's.  python tests/devtools/stress_giant.py [N]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_planned as T
from gaustar_amd import rasterizer as rz, scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
gs, cams, bg = scene.config_C()
cam = cams[0]
sc = np.array(gs.scales, dtype=np.float32, copy=True)
sc[:, 1:] *= np.exp(np.random.default_rng(7).normal(0.0, 1.0, size=(gs.P, 1))).astype(np.float32)
sc[:, 1:] = np.maximum(sc[:, 1:].mean(), sc[:, 1:])
gs.scales = sc
ps, cam_t, bg_t, dpix = T._inputs(dev, gs, cam, bg)
rz.drop_plans()
exact = T._render(dev, ps, cam_t, bg_t, cam, dpix, use_plan=False)
worst = {}
bad_img = 0
for mode in ("exact", "planned"):
    for it in range(n):
        r = T._render(dev, ps, cam_t, bg_t, cam, dpix, use_plan=(mode == "planned"))
        if not torch.equal(r[0], exact[0]) or not torch.equal(r[1], exact[1]):
            bad_img += 1
            print(mode, it, "IMAGE/RADII differ: max", float((r[0] - exact[0]).abs().max()), "pixels", int((r[0] != exact[0]).any(0).sum()), r[3], r[4], exact[4])
        for i, (ga, gb) in enumerate(zip(r[2], exact[2])):
            if ga is None or ga.numel() == 0:
                continue
            e = float((ga - gb).abs().max() / gb.abs().max())
            if e > worst.get((mode, i), 0.0):
                worst[(mode, i)] = e
            if e > 2e-5:
                print(mode, it, "gradient", i, "differs by", e, r[3])
print("image mismatches", bad_img)
for k in sorted(worst):
    print(k, f"{worst[k]:.3e}")
print(rz.PLAN_STATS)

import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import test_gpu_planned as tp
from gaustar_amd import rasterizer as rz
dev = torch.device("cuda:0")
gs, cam, bg = tp._scene("C", 21)
ps, cam_t, bg_t, dpix = tp._inputs(dev, gs, cam, bg)
rz.drop_plans()
for i in range(8):
    moved = dict(ps)
    moved["scales"] = (ps["scales"] * (1.0 + 0.03 * i)).contiguous()
    moved["means3D"] = (ps["means3D"] + torch.tensor([0.004 * i, -0.003 * i, 0.0], device=dev)).contiguous()
    r = tp._render(dev, moved, cam_t, bg_t, cam, dpix)
    pl = list(rz._PLANS.values())[0]
    T = 8160
    hb = pl.buf.numel() // 2
    for h in (0, 1):
        words = pl.buf[h * hb:(h + 1) * hb].view(torch.int32)
        rng = words[64:64 + 2 * T].view(T, 2)
        print("   half", h, "hdr", words[:8].tolist(), "nonzero caps", int((rng[:, 1] != 0).sum()), "sum caps", int(rng[:, 1].sum()), "max cap", int(rng[:,1].max()))
    print(i, r[3], "R,U", r[4], "info", [pl.info[k] for k in range(8)], "hdr", [pl.info[k] for k in range(8, 16)], "seq", pl.info[16], pl.info[17], flush=True)

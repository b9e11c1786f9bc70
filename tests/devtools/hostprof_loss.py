import os, sys, time, ctypes
sys.path.insert(0, os.getcwd())
import torch
from gaustar_amd import losses, _lib, _host
from gaustar_amd.losses import _crop, _chw_view, _scale_ptr, _stream
dev = torch.device("cuda:0")
H, W = 96, 160
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand(4, H, W, device=dev, generator=g); gt = torch.rand(3, H, W, device=dev, generator=g); gd = torch.rand(H, W, device=dev, generator=g) * 5
lib = _lib.load()
T = {}
def tick(name, t0):
    t1 = time.perf_counter(); T.setdefault(name, []).append(t1 - t0); return t1
x = img; g_full = gt
one = torch.ones((), device=dev)
for it in range(300):
    t = time.perf_counter()
    p, gg = _crop(x[:3], None), _crop(g_full, None); t = tick("crop x2", t)
    C, Hh, Ww = (int(v) for v in p.shape); d = x[3]; Hd, Wd = (int(v) for v in d.shape); t = tick("shapes + x[3]", t)
    vp = lambda t_: ctypes.c_void_p(t_.data_ptr())
    with _host.on_device(dev):
        t = tick("on_device enter", t)
        ws = torch.empty(lib.gsr_l1_ssim_workspace_bytes(C, Hh, Ww), dtype=torch.uint8, device=dev)
        wd = torch.empty(lib.gsr_depth_l1_workspace_bytes(), dtype=torch.uint8, device=dev)
        out = torch.empty(8, dtype=torch.float32, device=dev); t = tick("3 x torch.empty + 2 size calls", t)
        args = (C, Hh, Ww, vp(p), p.stride(0), p.stride(1), p.stride(2), vp(gg), gg.stride(0), gg.stride(1), gg.stride(2), 0.2, vp(ws), Hd, Wd, vp(d), d.stride(0), d.stride(1), vp(gd), gd.stride(0), gd.stride(1), 10.0, 1.0, 0.5, vp(wd), vp(out), _stream()); t = tick("arg tuple (fwd)", t)
        lib.gsr_rgb_depth_loss(*args); t = tick("ctypes call fwd (2 launches)", t)
    t = tick("on_device exit", t)
    grad6 = torch.empty_like(x, memory_format=torch.contiguous_format); t = tick("empty_like", t)
    gv, gdv = _crop(grad6[:3], None), grad6[3]; t = tick("crop grad + [3]", t)
    keep, sp = _scale_ptr(one, dev); t = tick("_scale_ptr", t)
    args = (C, Hh, Ww, vp(p), p.stride(0), p.stride(1), p.stride(2), vp(gg), gg.stride(0), gg.stride(1), gg.stride(2), 0.2, vp(ws), Hd, Wd, vp(d), d.stride(0), d.stride(1), vp(gd), gd.stride(0), gd.stride(1), 10.0, 1.0, 0.5, ctypes.c_void_p(ws.data_ptr() + ws.numel() - 256), sp, vp(gv), gv.stride(0), gv.stride(1), gv.stride(2), vp(gdv), gdv.stride(0), gdv.stride(1), _stream()); t = tick("arg tuple (bwd)", t)
    lib.gsr_rgb_depth_loss_backward(*args); t = tick("ctypes call bwd (1 launch)", t)
    if it % 50 == 0: torch.cuda.synchronize()
import numpy as np
for k, v in T.items(): print(f"{k:40s} {np.median(v[50:]) * 1e6:7.1f} us")

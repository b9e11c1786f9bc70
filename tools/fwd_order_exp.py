"""tools/fwd_order_exp.py [view] -- does an XCD-aware launch order shorten the planned forward blend?  (GPU)
Workgroup i of the launch runs on XCD i % 8; the plan's `order` (tiles by list length, longest first) is rewritten in place in the
camera's plan buffer so that, band by band of similar lengths, slot i takes a tile of macro-block region i % 8 (neighbouring tiles ->
one L2: they gather many of the same Gaussian records).  Timed by the library's HIP-event brackets, interleaved rounds."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from gaustar_amd import _lib, scene
from gaustar_amd import rasterizer as R

view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gs, cams, bg = scene.config_C(); cam = cams[view]
dev = torch.device("cuda:0"); lib = _lib.load()
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
W, H = cam.W, cam.H; gx, gy = (W + 15) // 16, (H + 15) // 16; T = gx * gy
args = (t(bg), t(gs.means3D), t(gs.colors_precomp), t(gs.opacities), t(gs.scales), t(gs.rotations), 1.0, torch.Tensor([]), t(cam.viewmatrix),
        t(cam.projmatrix), cam.tanfovx, cam.tanfovy, H, W, torch.Tensor([]), 0, t(cam.campos), False, False)
def fwd():
    return R.rasterize_gaussians_native(*args, plan_key="exp")
for _ in range(4): fwd()
torch.cuda.synchronize()
print(R.PLAN_STATS)
plan = [v for k, v in R._PLANS.items() if k[-1] == "exp"][0]
al = lambda x: (x + 255) & ~255
o_ranges = al(64); o_seg = al(o_ranges + 8 * T); o_order = al(o_seg + 4 * (T + 1))
buf = plan.buf
order_t = buf[o_order:o_order + 4 * T].view(torch.int32)
ranges = buf[o_ranges:o_ranges + 8 * T].view(torch.int32).view(T, 2).cpu().numpy()
order0 = order_t.cpu().numpy().copy()
assert sorted(order0.tolist()) == list(range(T)), "not the order array"
cap = ranges[:, 1]

def xcd_order(base, band, mb):
    out = np.empty_like(base)
    nbx = (gx + mb - 1) // mb
    for b0 in range(0, T, band):
        tiles = base[b0:b0 + band]
        reg = ((tiles // gx) // mb * nbx + (tiles % gx) // mb) % 8
        lists = [list(tiles[reg == r]) for r in range(8)]
        res = []
        for p in range(len(tiles)):
            r = (b0 + p) % 8
            if not lists[r]:
                r = int(np.argmax([len(l) for l in lists]))
            res.append(lists[r].pop(0))
        out[b0:b0 + band] = res
    return out

nst = lib.gsr_num_stages(); names = [lib.gsr_stage_name(i).decode() for i in range(nst)]
ib = names.index("blend_fwd_kernel")
ms = (ctypes.c_float * nst)(); cnt = (ctypes.c_int * nst)()
def timed(order, n=30):
    order_t.copy_(torch.from_numpy(order).to(dev)); torch.cuda.synchronize()
    for _ in range(3): fwd()
    torch.cuda.synchronize()
    lib.gsr_profile_enable(1); lib.gsr_profile_read(ms, cnt, 1)
    for _ in range(n): fwd()
    torch.cuda.synchronize()
    lib.gsr_profile_read(ms, cnt, 1); lib.gsr_profile_enable(0)
    return ms[ib] / max(cnt[ib], 1) * 1e3
variants = {"product order": order0}
for band in (64, 256, 1024):
    for mb in (2, 4, 8):
        variants[f"xcd band {band} macro {mb}x{mb}"] = xcd_order(order0, band, mb)
res = {k: [] for k in variants}
for rnd in range(3):
    for k, o in variants.items():
        res[k].append(timed(o))
for k, v in res.items():
    print(f"{k:32s} blend_fwd us: {[round(x, 1) for x in v]}  median {np.median(v):.1f}")
print(R.PLAN_STATS)

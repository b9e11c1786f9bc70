"""tools/bwd_order_exp.py [view] -- does a launch order built from MEASURED unit lives shorten blend_bwd?  (GPU)
Lives per unit come from the product kernel's own trace stamps (gsr_debug_set_trace: start of the unit's first wave, end of its
last); orders are handed to the kernel through gsr_debug_set_bwd_order; every variant is timed by the library's HIP-event
brackets over interleaved rounds."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from gaustar_amd import _lib, scene
from gaustar_amd import rasterizer as R

view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gs, cams, bg = scene.config_C()
cam = cams[view]
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
lib = _lib.load()
W, H = cam.W, cam.H
T = ((W + 15) // 16) * ((H + 15) // 16)
m3, op, cols, sc, rot = t(gs.means3D), t(gs.opacities), t(gs.colors_precomp), t(gs.scales), t(gs.rotations)
vm, pm, cp, bgt = t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), t(bg)
e = torch.Tensor([])
dp = torch.randn(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
out = R.rasterize_gaussians_native(bgt, m3, cols, op, sc, rot, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, H, W, e, 0, cp, False, False, use_plan=False)
Rn, _, radii, geom, binning, img, maxc, U = out

def bwd():
    return R.rasterize_gaussians_backward_native(bgt, m3, radii, cols, sc, rot, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, dp, e, 0, cp, geom, Rn,
                                                 binning, img, False, num_segments=U)
for _ in range(3):
    bwd()
torch.cuda.synchronize()
trace = torch.zeros(2 * T + 2 * U + 64, dtype=torch.int64, device=dev)
lives = []
for _ in range(5):
    trace.zero_()
    lib.gsr_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
    bwd(); torch.cuda.synchronize()
    lib.gsr_debug_set_trace(None)
    tr = trace.cpu().numpy()[2 * T:2 * T + 2 * U].reshape(U, 2).astype(np.float64)
    # (the first wave of a unit stamps its start only if it has work; a unit's other blocks may still have some: such units
    # are given the span from the kernel's median start of their neighbours -- a coarse stand-in, they are light)
    ok = (tr[:, 0] > 0) & (tr[:, 1] > 0)
    lf = np.where(ok, (tr[:, 1] - tr[:, 0]) / 100.0, np.where(tr[:, 1] > 0, 4.0, 0.5))
    lives.append(lf)
    span = (tr[:, 1].max() - tr[:, 0][tr[:, 0] > 0].min()) / 100.0
lives = np.array(lives)
life = np.median(lives, axis=0)
print(f"view {view}: U {U}, span {span:.1f} us; unit life us p10/p50/p90/p99/max {[round(float(np.percentile(life, q)), 1) for q in (10, 50, 90, 99, 100)]}; "
      f"run-to-run |dev| / life median {np.median(np.abs(lives - life) / np.maximum(life, 0.1)):.3f}; sum {life.sum() / 1e3:.1f} ms.us")

nst = lib.gsr_num_stages()
names = [lib.gsr_stage_name(i).decode() for i in range(nst)]
ib = names.index("blend_bwd_kernel")
ms = (ctypes.c_float * nst)(); cnt = (ctypes.c_int * nst)()

def timed(order, n=24):
    lib.gsr_debug_set_bwd_order(None if order is None else ctypes.c_void_p(order.data_ptr()))
    for _ in range(2):
        bwd()
    torch.cuda.synchronize()
    lib.gsr_profile_enable(1); lib.gsr_profile_read(ms, cnt, 1)
    for _ in range(n):
        bwd()
    torch.cuda.synchronize()
    lib.gsr_profile_read(ms, cnt, 1); lib.gsr_profile_enable(0)
    lib.gsr_debug_set_bwd_order(None)
    return ms[ib] / max(cnt[ib], 1) * 1e3

ident = np.arange(U)
def dev_order(o):
    return torch.from_numpy(np.ascontiguousarray(o.astype(np.int32))).to(dev)
orders = {"none": None, "identity table": dev_order(ident), "life descending": dev_order(np.argsort(-life, kind="stable"))}
# life classes of 1 us, tile order kept inside a class
cls = np.minimum(63, (life / 1.0).astype(np.int64))
orders["1 us classes, stable"] = dev_order(np.argsort(-cls, kind="stable"))
cls4 = np.minimum(15, (life / 4.0).astype(np.int64))
orders["4 us classes, stable"] = dev_order(np.argsort(-cls4, kind="stable"))
# heavy tail first only: units above the 80th percentile in front, the rest in place
thr = np.percentile(life, 80)
orders["top 20 % first, rest in place"] = dev_order(np.concatenate([np.nonzero(life >= thr)[0], np.nonzero(life < thr)[0]]))
# whole tiles: a tile's units stay together and in order (they share the tile's pixel state in L2); tiles by their heaviest unit
rng_ = torch.zeros(T, 2, dtype=torch.int32, device=dev)
lib.gsr_debug_export(gs.P, Rn, U, W, H, ctypes.c_void_p(geom.data_ptr()), ctypes.c_void_p(binning.data_ptr()), ctypes.c_void_p(img.data_ptr()),
                     None, None, None, None, ctypes.c_void_p(rng_.data_ptr()), None, None, None, None)
torch.cuda.synchronize()
rg = rng_.cpu().numpy().astype(np.int64)
upt = (rg[:, 1] - rg[:, 0] + 63) // 64
tile_of = np.repeat(np.arange(T), upt)
assert len(tile_of) == U
tmax = np.zeros(T); np.maximum.at(tmax, tile_of, life)
tsum = np.zeros(T); np.add.at(tsum, tile_of, life)
orders["tiles by heaviest unit"] = dev_order(np.lexsort((ident, -tmax[tile_of])))
orders["tiles by summed life"] = dev_order(np.lexsort((ident, -tsum[tile_of])))
# runs of 8 units (what one XCD takes in a row) by their heaviest unit
run = ident // 8
rmax = np.zeros(run.max() + 1); np.maximum.at(rmax, run, life)
orders["runs of 8 by heaviest unit"] = dev_order(np.lexsort((ident, -rmax[run])))
res = {k: [] for k in orders}
for rnd in range(4):
    for k, o in orders.items():
        res[k].append(timed(o))
for k, v in res.items():
    print(f"{k:32s} blend_bwd us (HIP events) median {np.median(v):7.1f}   rounds {[round(x, 1) for x in v]}")

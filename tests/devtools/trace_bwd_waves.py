"""Dev tool: per-WAVE timeline of blend_bwd on one view of config C, from a build with -DGSR_TRACE_DETAIL
(python -m gaustar_amd.build --variant trace -DGSR_TRACE_DETAIL; GSR_LIB_PATH=.../libgsr_hip_trace.so).
Per wave: start, end of the unit's head (all loads landed, keep-set known), end; plus HW_ID / XCC_ID."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, scene
from gaustar_amd import rasterizer as R

cam_i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gs, cams, bg = scene.config_C()
cam = cams[cam_i]
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
lib = _lib.load()
W, H = cam.W, cam.H
T = ((W + 15) // 16) * ((H + 15) // 16)
m3, m2, op = t(gs.means3D).requires_grad_(True), torch.zeros(gs.P, 3, device=dev, requires_grad=True), t(gs.opacities).requires_grad_(True)
cols, sc, rot = t(gs.colors_precomp).requires_grad_(True), t(gs.scales).requires_grad_(True), t(gs.rotations).requires_grad_(True)
s = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, t(bg), 1.0, t(cam.viewmatrix), t(cam.projmatrix), 0, t(cam.campos), False, False)
rast = GaussianRasterizer(s)
dp = torch.randn(3, H, W, device=dev)
for _ in range(3):
    c, r = rast(m3, m2, op, None, cols, sc, rot, None); c.backward(dp)
e = torch.Tensor([])
out = R.rasterize_gaussians_native(t(bg), m3.detach(), cols.detach(), op.detach(), sc.detach(), rot.detach(), 1.0, e,
                                   t(cam.viewmatrix), t(cam.projmatrix), cam.tanfovx, cam.tanfovy, H, W, e, 0, t(cam.campos), False, False, use_plan=False)
Rn, _, _, geom, binning, img, maxc, U = out
trace = torch.zeros(2 * T + 16 * U, dtype=torch.int64, device=dev)
lib.gsr_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
c, r = rast(m3, m2, op, None, cols, sc, rot, None); c.backward(dp)
torch.cuda.synchronize()
lib.gsr_debug_set_trace(None)
tr = trace.cpu().numpy()[2 * T:].reshape(4 * U, 4)
ok = tr[:, 0] > 0
st, hd, en = (tr[ok, i].astype(np.float64) / 100.0 for i in range(3))   # microseconds (100 MHz clock)
hw = tr[ok, 3]
t0 = st.min(); st -= t0; hd -= t0; en -= t0
span = en.max()
print(f"waves {ok.sum()} of {4 * U}; span {span:.1f} us")
print("head (start -> loads landed) us  p10/p50/p90/p99:", [round(float(np.percentile(hd - st, q)), 2) for q in (10, 50, 90, 99)])
body = en - hd
live = body > 0
print(f"waves with work {live.sum()}; body us p10/p50/p90/p99:", [round(float(np.percentile(body[live], q)), 2) for q in (10, 50, 90, 99)])
print(f"sum head {np.sum(hd - st):.0f} us, sum body {body.sum():.0f} us; mean waves in head {np.sum(hd - st) / span:.0f}, in body {body.sum() / span:.0f}")
# per SIMD: how many of its resident waves are in the head phase at a time (sampled)
simd = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 8) & 0xf) * 8 + ((hw >> 4) & 3)   # xcc, se, cu, simd
ids = np.unique(simd)
print("distinct (xcc, se, cu, simd):", len(ids))
ts = np.linspace(0.1 * span, 0.9 * span, 41)
frac = []
hist = np.zeros(10, int)
for sid in ids[:: max(1, len(ids) // 128)]:
    m = simd == sid
    for tt in ts:
        inhead = np.sum((st[m] <= tt) & (hd[m] > tt)); inbody = np.sum((hd[m] <= tt) & (en[m] > tt))
        hist[min(9, inhead)] += 1
        frac.append((inhead, inbody))
frac = np.array(frac)
print("per SIMD sample: mean waves in head %.2f, in body %.2f" % (frac[:, 0].mean(), frac[:, 1].mean()))
print("histogram of #waves in head per SIMD sample:", hist.tolist())
print("histogram of #waves in body per SIMD sample:", np.bincount(frac[:, 1], minlength=8).tolist())
# residency over the span: mean resident waves per SIMD slot in each twentieth of the kernel's span
edges = np.linspace(0.0, span, 21)
res = []
for a, b in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(en, b) - np.maximum(st, a), 0.0, None).sum() / (b - a)
    res.append(ov / len(ids))
print("resident waves per SIMD by twentieth of the span:", [round(float(x), 2) for x in res])
order = np.argsort(st)
print("start time of the last wave %.1f us; waves ending in the last 10 us: %d" % (st.max(), int(np.sum(en > span - 10.0))))
# by launch order: wave index = row of the trace = unit * 4 + block (units are in tile launch order, longest lists first)
idx = np.nonzero(ok)[0]
dec = np.array_split(np.arange(len(idx)), 10)
print("by tenth of the unit order: mean start / head / body (us), share of waves with work")
for d in dec:
    b_ = (en - hd)[d]
    print("   start %6.1f  head %5.2f  body %5.2f  with work %.2f" % (st[d].mean(), (hd - st)[d].mean(), b_[b_ > 0].mean() if (b_ > 0).any() else 0.0, (b_ > 0).mean()))
# what a different launch order could buy: greedy list scheduling of the measured wave lives (start -> end) onto as many slots as
# were busy on average, in the order launched, sorted longest-first (the best any reordering can do) and shortest-first
import heapq
life = (en - st)
def makespan(order, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for i in order:
        t = heapq.heappop(h) + life[i]
        end = max(end, t)
        heapq.heappush(h, t)
    return end
slots = int(round(res[len(res) // 2] * len(ids)))
launch = np.argsort(st, kind="stable")
print(f"list scheduling on {slots} slots: launch order {makespan(launch, slots):.1f} us, longest first {makespan(np.argsort(-life), slots):.1f} us, "
      f"shortest first {makespan(np.argsort(life), slots):.1f} us; sum of lives / slots = {life.sum() / slots:.1f} us")
# proxies a launch order could be built from before the backward runs: the tile's list length, the unit's depth in its tile, the
# entries the unit holds
rng_t = torch.zeros(T, 2, dtype=torch.int32, device=dev)
pv = lambda x: ctypes.c_void_p(x.data_ptr())
lib.gsr_debug_export(gs.P, int(Rn), int(U), W, H, pv(geom), pv(binning), pv(img), None, None, None, None, pv(rng_t), None, None, None, None)
rg = rng_t.cpu().numpy().astype(np.int64)
tl = rg[:, 1] - rg[:, 0]
nun = (tl + 63) // 64
u_tile = np.repeat(np.arange(T), nun); u_k = np.concatenate([np.arange(c) for c in nun]) if nun.sum() else np.zeros(0, int)
assert len(u_tile) == U, (len(u_tile), U)
u_n = tl[u_tile]; u_ent = np.minimum(64, u_n - 64 * u_k)
w_unit = idx // 4
for name, key in (("tile length, longest first", -u_n[w_unit] * 100 + u_k[w_unit]), ("unit depth in tile, front units first", u_k[w_unit]),
                  ("entries of the unit, fullest first", -u_ent[w_unit]), ("front units first, then fullest", u_k[w_unit] * 100 - u_ent[w_unit]),
                  ("fullest first, then front units", -u_ent[w_unit] * 100 + u_k[w_unit])):
    print(f"   order by {name}: {makespan(np.argsort(key, kind='stable'), slots):.1f} us")
print("   mean life by unit depth k = 0, 1, 2, 3, 4+:", [round(float(life[u_k[w_unit] == k].mean()), 1) for k in range(4)], round(float(life[u_k[w_unit] >= 4].mean()), 1),
      " by entries <=16 / <=32 / <=48 / <64 / 64:", [round(float(life[(u_ent[w_unit] > a) & (u_ent[w_unit] <= b)].mean()), 1) for a, b in ((0, 16), (16, 32), (32, 48), (48, 63), (63, 64))])
o = np.argsort(-u_ent[w_unit], kind="stable")
tail_ = o[-slots:]
print("   fullest first: last generation mean life %.1f, max %.1f, p99 %.1f; entries there min/max %d/%d; count of 64-entry waves %d of %d" % (
    life[tail_].mean(), life[tail_].max(), np.percentile(life[tail_], 99), u_ent[w_unit][tail_].min(), u_ent[w_unit][tail_].max(), int((u_ent[w_unit] == 64).sum()), len(life)))
cs = np.cumsum(life[o]) / slots
print("   cumulative work / slots at 25/50/75/100 %% of that order: %.1f %.1f %.1f %.1f" % tuple(cs[[len(cs) // 4, len(cs) // 2, 3 * len(cs) // 4, -1]]))
# stragglers: the slowest 2 % of the waves -- where do they lose their time, and do they share anything?
thr = np.percentile(life, 98)
sl = life >= thr
xcc = (hw >> 32) & 0xf
print("stragglers (life >= %.1f us, %d waves): head %.1f us (all: %.1f), body %.1f (all with work: %.1f); entries mean %.0f (all %.0f); unit depth mean %.1f (all %.1f)" % (
    thr, int(sl.sum()), (hd - st)[sl].mean(), (hd - st).mean(), (en - hd)[sl].mean(), body[live].mean(), u_ent[w_unit][sl].mean(), u_ent[w_unit].mean(),
    u_k[w_unit][sl].mean(), u_k[w_unit].mean()))
print("   per XCD share of stragglers:", [round(float((xcc[sl] == x).mean()), 3) for x in range(8)], " start time mean %.1f (all %.1f)" % (st[sl].mean(), st.mean()))
tiles_sl = u_tile[w_unit][sl]
uq, cn = np.unique(tiles_sl, return_counts=True)
print("   distinct tiles among stragglers: %d; tiles with >= 4 straggling waves: %d; list length of straggler tiles mean %.0f (all tiles with units: %.0f)" % (
    len(uq), int((cn >= 4).sum()), tl[uq].mean(), tl[tl > 0].mean()))
blk = idx % 4
print("   by block:", [int(((blk == b) & sl).sum()) for b in range(4)], " same (unit) with >= 2 straggling blocks:", int((np.unique(w_unit[sl], return_counts=True)[1] >= 2).sum()))
print("   per XCD: waves", [int((xcc == x).sum()) for x in range(8)])
print("   per XCD: sum of lives (ms-us)", [round(float(life[xcc == x].sum()) / 1e3, 1) for x in range(8)], " last end (us)", [round(float(en[xcc == x].max()), 1) for x in range(8)],
      " last start", [round(float(st[xcc == x].max()), 1) for x in range(8)])

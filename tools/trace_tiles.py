"""Dev tool: per-workgroup timeline of the two blend kernels on one view of a BASELINE config (default C)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, scene
from gaustar_amd import rasterizer as R

cam_i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = sys.argv[2] if len(sys.argv) > 2 else "C"     # tools/trace_tiles.py [view] [A|B|C|D]
gs, cams, bg = getattr(scene, "config_" + cfg)()
cams = cams if isinstance(cams, (list, tuple)) else [cams]
cam = cams[cam_i]
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
lib = _lib.load()
W, H = cam.W, cam.H
T = ((W + 15) // 16) * ((H + 15) // 16)
m3, m2, op = t(gs.means3D).requires_grad_(True), torch.zeros(gs.P, 3, device=dev, requires_grad=True), t(gs.opacities).requires_grad_(True)
cols_np = gs.colors_precomp if gs.colors_precomp is not None else np.random.default_rng(0).random((gs.P, 3), dtype=np.float32)   # (SH configs: any colours, the lists are the same)
cols, sc, rot = t(cols_np).requires_grad_(True), t(gs.scales).requires_grad_(True), t(gs.rotations).requires_grad_(True)
s = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, t(bg), 1.0, t(cam.viewmatrix), t(cam.projmatrix), 0, t(cam.campos), False, False)
rast = GaussianRasterizer(s)
dp = torch.randn(3, H, W, device=dev)
for _ in range(3):
    c, r = rast(m3, m2, op, None, cols, sc, rot, None); c.backward(dp)
e = torch.Tensor([])
out = R.rasterize_gaussians_native(t(bg), m3.detach(), cols.detach(), op.detach(), sc.detach(), rot.detach(), 1.0, e,
                                   t(cam.viewmatrix), t(cam.projmatrix), cam.tanfovx, cam.tanfovy, H, W, e, 0, t(cam.campos), False, False, use_plan=False)
Rn, _, _, geom, binning, img, maxc, U = out
trace = torch.zeros(2 * T + 2 * U, dtype=torch.int64, device=dev)
lib.gsr_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
c, r = rast(m3, m2, op, None, cols, sc, rot, None); c.backward(dp)
torch.cuda.synchronize()
lib.gsr_debug_set_trace(None)
tr = trace.cpu().numpy().astype(np.float64) / 100.0   # microseconds
print("R", Rn, "max", maxc, "units", U)
for name, arr in (("fwd", tr[:2 * T].reshape(T, 2)), ("bwd", tr[2 * T:].reshape(U, 2))):
    st, en = arr[:, 0], arr[:, 1]
    ok = (en > 0) & (st > 0)   # a unit whose first block left early (nothing to replay) has no start stamp
    t0 = st[ok].min()
    st, en = st - t0, en - t0
    dur = en - st
    span = en[ok].max()
    cap = 1792
    print(f"== {name}: span {span:.1f} us; WGs {ok.sum()}; sum of durations {dur[ok].sum():.0f} us; /{cap} = {dur[ok].sum() / cap:.1f} us; mean concurrency {dur[ok].sum() / span:.1f}")
    edges = np.linspace(0, span, 11)
    conc = [(np.minimum(en[ok], b) - np.maximum(st[ok], a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    print("   concurrency per decile:", [int(c_) for c_ in conc])
    print("   duration percentiles (us) p50/p90/p99/max:", [round(float(np.percentile(dur[ok], q)), 1) for q in (50, 90, 99, 100)])
# the slowest forward workgroups: launch slot, duration, list length of their tile
e2 = torch.Tensor([])
rng_t = torch.zeros(T, 2, dtype=torch.int32, device=dev); pl_t = torch.zeros(max(Rn, 1), dtype=torch.int32, device=dev)
P_ = gs.P
m2_ = torch.zeros(P_, 2, device=dev); co_ = torch.zeros(P_, 4, device=dev); fT_ = torch.zeros(H, W, device=dev); nc_ = torch.zeros(H, W, dtype=torch.int32, device=dev)
pp = lambda x: ctypes.c_void_p(x.data_ptr())
_lib.check(lib.gsr_debug_export(P_, Rn, 1, W, H, pp(geom), pp(binning), pp(img), pp(m2_), pp(co_), None, None, pp(rng_t), pp(pl_t), pp(fT_), pp(nc_), None), "export")
torch.cuda.synchronize()
rg = rng_t.cpu().numpy().astype(np.int64); lens = rg[:, 1] - rg[:, 0]
order = np.argsort(-lens, kind="stable")      # approximately the launch order (32-entry buckets, snake)
fw = tr[:2 * T].reshape(T, 2); dur = fw[:, 1] - fw[:, 0]
top = np.argsort(-dur)[:8]
print("slowest forward slots (slot, us, list length of the tile at that rank of the length order):", [(int(s_), round(float(dur[s_]), 1), int(lens[order[s_]])) for s_ in top])
print("deepest n_contrib per tile of the longest lists:", [(int(lens[t_]), int(nc_.cpu().numpy()[(t_ // ((W + 15) // 16)) * 16:(t_ // ((W + 15) // 16)) * 16 + 16, (t_ % ((W + 15) // 16)) * 16:(t_ % ((W + 15) // 16)) * 16 + 16].max())) for t_ in order[:8]])

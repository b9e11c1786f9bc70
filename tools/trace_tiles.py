"""Dev tool: per-workgroup timeline of the two blend kernels on one view of config C."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
params = [t(gs.means3D).requires_grad_(True), torch.zeros(gs.P, 3, device=dev, requires_grad=True), t(gs.opacities).requires_grad_(True)]
cols, sc, rot = t(gs.colors_precomp).requires_grad_(True), t(gs.scales).requires_grad_(True), t(gs.rotations).requires_grad_(True)
s = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, t(bg), 1.0, t(cam.viewmatrix), t(cam.projmatrix), 0, t(cam.campos), False, False)
rast = GaussianRasterizer(s)
dp = torch.randn(3, H, W, device=dev)
for _ in range(3):
    c, r = rast(params[0], params[1], params[2], None, cols, sc, rot, None); c.backward(dp)
trace = torch.zeros(4 * T, dtype=torch.int64, device=dev)
lib.gsr_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
c, r = rast(params[0], params[1], params[2], None, cols, sc, rot, None); c.backward(dp)
torch.cuda.synchronize()
lib.gsr_debug_set_trace(None)
tr = trace.cpu().numpy().reshape(2, T, 2).astype(np.float64) / 100.0   # microseconds
# tile lengths in launch order
e = torch.Tensor([])
out = R.rasterize_gaussians_native(t(bg), params[0].detach(), cols.detach(), params[2].detach(), sc.detach(), rot.detach(), 1.0, e,
                                   t(cam.viewmatrix), t(cam.projmatrix), cam.tanfovx, cam.tanfovy, H, W, e, 0, t(cam.campos), False, False)
Rn, _, _, geom, binning, img, maxc = out
rng_ = torch.zeros(T, 2, dtype=torch.int32, device=dev)
p = lambda x: ctypes.c_void_p(x.data_ptr())
lib.gsr_debug_export(gs.P, Rn, W, H, p(geom), p(binning), p(img), None, None, None, None, p(rng_), None, None, None, None)
torch.cuda.synchronize()
lens = (rng_[:, 1] - rng_[:, 0]).cpu().numpy()
lens_sorted = np.sort(lens)[::-1]
print("R", Rn, "max", maxc, "nonempty", (lens > 0).sum(), "top lens", lens_sorted[:8], "median nonempty", np.median(lens[lens > 0]))
for name, k in (("fwd", 0), ("bwd", 1)):
    st, en = tr[k, :, 0], tr[k, :, 1]
    ok = en > 0
    t0 = st[ok].min()
    st, en = st - t0, en - t0
    dur = en - st
    span = en[ok].max()
    print(f"== {name}: span {span:.1f} us; WGs traced {ok.sum()}; sum of durations {dur[ok].sum():.0f} us; mean concurrency {dur[ok].sum() / span:.1f}")
    idx = np.argsort(-dur * ok)[:6]
    print("   longest WGs (launch idx, start, dur):", [(int(i), round(st[i], 1), round(dur[i], 1)) for i in idx])
    # concurrency over time
    edges = np.linspace(0, span, 11)
    conc = [(np.minimum(en[ok], b) - np.maximum(st[ok], a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    print("   concurrency per decile:", [round(c_, 0) for c_ in conc])
    ne = ok & (dur > 1.0)
    print(f"   WGs with dur>1us: {ne.sum()}, last start {st[ne].max():.1f}, first-launched long WG ends at {en[0]:.1f}")

order = None
for name, k in (("fwd", 0), ("bwd", 1)):
    st, en = tr[k, :, 0], tr[k, :, 1]
    t0 = st.min(); st, en = st - t0, en - t0
    dur = en - st
    print(f"== {name}: duration by launch-index band (mean dur us, mean start us, mean end us)")
    for a in range(0, 2048, 256):
        sl = slice(a, a + 256)
        print(f"   [{a:4d},{a+256:4d}) dur {dur[sl].mean():7.1f}  start {st[sl].mean():6.1f}  end {en[sl].mean():6.1f}  max end {en[sl].max():6.1f}")
    print("   empties: dur mean", dur[2048:].mean(), "start mean", st[2048:].mean())

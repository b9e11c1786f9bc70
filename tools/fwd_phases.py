"""tools/fwd_phases.py [view] -- where a forward tile's time goes (sort / park + words / walk per chunk), from a -DGSR_FWD_PHASES build:
python -m gaustar_amd.build --variant fwdphases -DGSR_FWD_PHASES; GSR_LIB_PATH=gaustar_amd/libgsr_hip_fwdphases.so python tools/fwd_phases.py"""
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
W, H = cam.W, cam.H; T = ((W + 15) // 16) * ((H + 15) // 16)
args = (t(bg), t(gs.means3D), t(gs.colors_precomp), t(gs.opacities), t(gs.scales), t(gs.rotations), 1.0, torch.Tensor([]), t(cam.viewmatrix),
        t(cam.projmatrix), cam.tanfovx, cam.tanfovy, H, W, torch.Tensor([]), 0, t(cam.campos), False, False)
for _ in range(3):
    out = R.rasterize_gaussians_native(*args, use_plan=False)
trace = torch.zeros(2 * T + 2 * 65536 + 16 * T + 64, dtype=torch.int64, device=dev)
lib.gsr_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
out = R.rasterize_gaussians_native(*args, use_plan=False)
torch.cuda.synchronize(); lib.gsr_debug_set_trace(None)
tr = trace.cpu().numpy().astype(np.float64) / 100.0
se = tr[:2 * T].reshape(T, 2)
ph = tr[2 * T + 2 * 65536:2 * T + 2 * 65536 + 16 * T].reshape(T, 16)
dur = se[:, 1] - se[:, 0]
t0 = se[:, 0][se[:, 0] > 0].min()
print(f"span {se[:, 1].max() - t0:.1f} us, sum of tile durations {dur[dur > 0].sum() / 1e3:.1f} ms.us")
# stamps: start, set-up done, [runs sorted, one per merge level ...], ids out + barrier, then (parked + words, walked) per chunk
for b in np.argsort(-dur)[:8]:
    p = ph[b]; n = int((p > 0).sum())
    print(f"slot {b}: {dur[b]:.1f} us; deltas between stamps: {[round(p[k + 1] - p[k], 1) for k in range(n - 1)]}; tail {se[b, 1] - p[n - 1]:.1f}")
mid = np.argsort(-dur)[1500:1504]
for b in mid:
    p = ph[b]; n = int((p > 0).sum())
    print(f"slot {b}: {dur[b]:.1f} us; deltas: {[round(p[k + 1] - p[k], 1) for k in range(n - 1)]}; tail {se[b, 1] - p[n - 1]:.1f}")
# What pre-sorting the K longest tiles could buy (VERDICT r5 task 7): the launch's span is its latest end; a tile that skips its
# in-kernel sort ends that much earlier (the first thousand tiles all start at ~0).  Sort time = stamps [1] .. the stamp before
# the first chunk's pair; approximated here by the known per-level cost: listed are the durations and ends by rank.
ends = se[:, 1] - t0
order_d = np.argsort(-ends)
ranks = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024]
print("ends by rank (us):", {r: round(float(ends[order_d[r - 1]]), 1) for r in ranks})
print("durations of the same tiles (us):", {r: round(float(dur[order_d[r - 1]]), 1) for r in ranks})
print("starts of the same tiles (us):", {r: round(float(se[order_d[r - 1], 0] - t0), 1) for r in ranks})

"""Dev tool: phase stamps of ONE forward workgroup (launch slot GSR_FWD_PHASES - 1; slot 0 = the longest tile) on config C,
from a build with -DGSR_FWD_PHASES=<slot + 1>.  Stamps: start, after the sort, then per chunk: records+masks parked /
barrier passed / walk done."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, scene

cam_i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = sys.argv[2] if len(sys.argv) > 2 else "C"
gs, cams, bg = getattr(scene, "config_" + cfg)()
cams = cams if isinstance(cams, (list, tuple)) else [cams]
cam = cams[cam_i]
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
lib = _lib.load()
W, H = cam.W, cam.H
T = ((W + 15) // 16) * ((H + 15) // 16)
m3, m2, op = t(gs.means3D).requires_grad_(True), torch.zeros(gs.P, 3, device=dev, requires_grad=True), t(gs.opacities).requires_grad_(True)
cols_np = gs.colors_precomp if gs.colors_precomp is not None else np.random.default_rng(0).random((gs.P, 3), dtype=np.float32)
cols, sc, rot = t(cols_np).requires_grad_(True), t(gs.scales).requires_grad_(True), t(gs.rotations).requires_grad_(True)
s = GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, t(bg), 1.0, t(cam.viewmatrix), t(cam.projmatrix), 0, t(cam.campos), False, False)
rast = GaussianRasterizer(s)
for _ in range(3):
    c, r = rast(m3, m2, op, None, cols, sc, rot, None)
trace = torch.zeros(2 * T + 256, dtype=torch.int64, device=dev)
lib.gsr_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
c, r = rast(m3, m2, op, None, cols, sc, rot, None)
torch.cuda.synchronize()
lib.gsr_debug_set_trace(None)
tr = trace.cpu().numpy()
ph = tr[2 * T:]
ph = ph[ph > 0].astype(np.float64) / 100.0
fw = tr[:2 * T].reshape(T, 2).astype(np.float64) / 100.0
t0 = fw[fw[:, 0] > 0, 0].min()
print("kernel span %.1f us" % (fw[:, 1].max() - t0))
ph -= t0
print("start %.1f  sorted %.1f" % (ph[0], ph[1]))
for i, k in enumerate(range(2, len(ph) - 2, 3)):
    print("chunk %d: parked +%.1f  barrier +%.1f  walk +%.1f   (ends at %.1f)" % (i, ph[k] - ph[k - 1], ph[k + 1] - ph[k], ph[k + 2] - ph[k + 1], ph[k + 2]))

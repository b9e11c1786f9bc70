"""tools/host_floor.py -- the host's share of a forward + backward step: bench.py's step on a workload whose GPU work is
negligible (300 Gaussians, 64 x 48 pixels), so that what is timed is Python, autograd, ctypes and the launches."""
import ctypes, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, scene

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
gs = scene.random_gaussians(300, rng, sh_degree=0, with_sh=False)
cam = scene.look_at_camera((0.0, 0.0, -4.0), (0, 0, 0), 64, 48, fovx=0.9, znear=0.01)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
params = [t(gs.means3D), t(gs.opacities), t(gs.colors_precomp), t(gs.scales), t(gs.rotations)]
for p in params:
    p.requires_grad_(True)
means2D = torch.zeros(gs.P, 3, device=dev, requires_grad=True)
r = GaussianRasterizer(GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, t(np.zeros(3)), 1.0, t(cam.viewmatrix),
                                                     t(cam.projmatrix), 0, t(cam.campos), False, False))
dpix = torch.randn(3, cam.H, cam.W, device=dev)


def step():
    for p in params:
        p.grad = None
    means2D.grad = None
    c, _ = r(means3D=params[0], means2D=means2D, opacities=params[1], colors_precomp=params[2], scales=params[3], rotations=params[4])
    c.backward(dpix)


for _ in range(20):
    step()
lib = _lib.load()
lib.gsr_debug_host_wait(None, None, 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 500
for _ in range(N):
    step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
w, c = ctypes.c_longlong(0), ctypes.c_longlong(0)
lib.gsr_debug_host_wait(ctypes.byref(w), ctypes.byref(c), 1)
# With almost nothing to compute the step is a chain of latencies: the host queues preprocess + scan and WAITS for the scan's
# totals (two dependent kernel launches' worth of GPU latency it cannot overlap with anything), then queues the rest.  The
# step minus that wait is what the host itself spends per step: Python, autograd, ctypes, launches.
print(json.dumps({"host_floor_ms_per_step": round(dt / N * 1e3, 4), "of_which_waiting_for_the_scan_ms": round(w.value / 1e6 / N, 4),
                  "host_work_ms_per_step": round(dt / N * 1e3 - w.value / 1e6 / N, 4)}))

"""tools/ab_lib.py A.so B.so [rounds] -- interleaved A/B of two builds of libgsr_hip.so inside ONE process on ONE box
(alternating rounds of 160 bench steps); box-to-box and run-to-run spread (+-3 %) drops out, differences of 0.1 % show.
Build variants with `python -m gaustar_amd.build --variant NAME -DFLAG ...`."""
import ctypes, json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaustar_amd import _lib, dist as gdist
paths = [os.path.abspath(p) for p in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
libs = []
for p in paths:
    _lib._lib, _lib.LIB_PATH = None, p
    libs.append(_lib.load())
import bench
def use(i):
    _lib._lib = libs[i]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
gdist.bind_to_local_cpus(0)
use(0)
gs, cams, bg, params, means2D, rasters, dpix = bench.build_workload(dev, 0)
if os.environ.get("AB_CHANNELS", "3") != "3":   # (AB_CHANNELS=4: the refinement loop's RGB + depth render)
    from gaustar_amd import GaussianRasterizer
    C = int(os.environ["AB_CHANNELS"])
    g = torch.Generator(device=dev).manual_seed(11)
    params["colors"] = torch.rand(params["colors"].shape[0], C, device=dev, generator=g).requires_grad_(True)
    dpix = torch.randn(C, dpix.shape[1], dpix.shape[2], device=dev, generator=g)
    bgc = torch.rand(C, device=dev, generator=g)
    rasters = [GaussianRasterizer(r.raster_settings._replace(bg=bgc)) for r in rasters]
step = lambda s: bench.one_step(s, 0, 1, params, means2D, rasters, dpix)
for i in (0, 1):
    use(i)
    for s in range(160): step(s)
res = [[], []]
for r in range(rounds):
    for i in ((0, 1) if r % 2 == 0 else (1, 0)):
        use(i)
        for s in range(8): step(s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(160): step(s)
        torch.cuda.synchronize(); res[i].append(round((time.perf_counter() - t0) / 160 * 1e3, 4))
print(json.dumps({"A": os.path.basename(paths[0]), "B": os.path.basename(paths[1]), "A_ms": res[0], "B_ms": res[1],
                  "median_A": statistics.median(res[0]), "median_B": statistics.median(res[1])}))

"""tools/ab_env.py VAR A B [rounds] -- interleaved A/B of one environment knob that the library reads per call, inside ONE
process on ONE box: alternating rounds of 160 bench steps with VAR=A / VAR=B; prints the per-round ms and the medians."""
import ctypes, json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaustar_amd import dist as gdist
var, a, b = sys.argv[1:4]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
libc = ctypes.CDLL(None)   # os.environ alone does not reach getenv() of an already running process on every libc
def setenv(v):
    os.environ[var] = v
    libc.setenv(var.encode(), v.encode(), 1)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
gdist.bind_to_local_cpus(0)
gs, cams, bg, params, means2D, rasters, dpix = bench.build_workload(dev, 0)
if os.environ.get("AB_CHANNELS", "3") != "3":   # (AB_CHANNELS=4 / 6: the multi-target renders)
    from gaustar_amd import GaussianRasterizer
    C = int(os.environ["AB_CHANNELS"])
    g = torch.Generator(device=dev).manual_seed(11)
    params["colors"] = torch.rand(params["colors"].shape[0], C, device=dev, generator=g).requires_grad_(True)
    dpix = torch.randn(C, dpix.shape[1], dpix.shape[2], device=dev, generator=g)
    bgc = torch.rand(C, device=dev, generator=g)
    rasters = [GaussianRasterizer(r.raster_settings._replace(bg=bgc)) for r in rasters]
step = lambda s: bench.one_step(s, 0, 1, params, means2D, rasters, dpix)
for v in (a, b):
    setenv(v)
    for s in range(160): step(s)
res = {a: [], b: []}
for r in range(rounds):
    for v in ((a, b) if r % 2 == 0 else (b, a)):
        setenv(v)
        for s in range(8): step(s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(160): step(s)
        torch.cuda.synchronize(); res[v].append(round((time.perf_counter() - t0) / 160 * 1e3, 4))
print(json.dumps({"var": var, a: res[a], b: res[b], "median_" + a: statistics.median(res[a]), "median_" + b: statistics.median(res[b])}))

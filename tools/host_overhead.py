"""tools/host_overhead.py -- how close is bench.py's step to being host-bound?  Times the Python enqueue loop of config C
(no synchronisation inside) against the same loop including the final synchronize."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
gs, cams, bg, params, means2D, rasters, dpix = bench.build_workload(dev, 0)
step = lambda s: bench.one_step(s, 0, 1, params, means2D, rasters, dpix)
for s in range(10): step(s)
out = {}
for n in (50, 200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(n): step(s)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    out[f"steps_{n}"] = {"enqueue_ms_per_step": round((t1 - t0) / n * 1e3, 4), "total_ms_per_step": round((t2 - t0) / n * 1e3, 4)}
print(json.dumps(out))

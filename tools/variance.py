"""tools/variance.py -- run-to-run spread of bench.py's step inside ONE process: R rounds of K steps each."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
gs, cams, bg, params, means2D, rasters, dpix = bench.build_workload(dev, 0)
step = lambda s: bench.one_step(s, 0, 1, params, means2D, rasters, dpix)
for s in range(20): step(s)
out = []
for r in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(160): step(s)
    torch.cuda.synchronize(); out.append(round((time.perf_counter() - t0) / 160 * 1e3, 4))
print(json.dumps({"ms_per_step_rounds": out, "cpu": os.sched_getaffinity(0).__len__()}))

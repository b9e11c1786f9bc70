import argparse, ctypes, json, os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from gaustar_amd import _lib, rasterizer
import bench_window
lib = _lib.load()
orig = bench_window.losses.rgb_depth_loss
st = {"n": 0, "t": time.perf_counter(), "rows": []}
def hook(*a, **k):
    torch.cuda.synchronize()
    now = time.perf_counter()
    w = ctypes.c_longlong(0); c = ctypes.c_longlong(0)
    lib.gsr_debug_host_wait(ctypes.byref(w), ctypes.byref(c), 1)
    ms = torch.cuda.memory_stats()
    st["rows"].append((st["n"], round((now - st["t"]) * 1e3, 2), round(w.value / 1e6, 3), ms["num_alloc_retries"], ms["num_device_alloc"], ms["num_device_free"], dict(rasterizer._BINNING_HINT)))
    st["n"] += 1; st["t"] = time.perf_counter()
    return orig(*a, **k)
bench_window.losses.rgb_depth_loss = hook
r = bench_window.run(argparse.Namespace(frames=3, iters=50, level=6, width=1920, height=1080, cameras=160))
rows = st["rows"]
for i in list(range(116, 126)):
    print(rows[i])
print(r["frames"])

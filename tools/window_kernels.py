import argparse, ctypes, json, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch
from gaustar_amd import _lib
import bench_window
lib = _lib.load()
nst = lib.gsr_num_stages(); names = [lib.gsr_stage_name(i).decode() for i in range(nst)]
orig = bench_window.losses.rgb_depth_loss
state = {"n": 0}
def hook(*a, **k):
    state["n"] += 1
    if state["n"] % 50 == 0:
        ms = (ctypes.c_float * nst)(); cnt = (ctypes.c_int * nst)()
        lib.gsr_profile_read(ms, cnt, 1)
        print("after", state["n"], {n.replace("_kernel",""): round(ms[i] / max(cnt[i], 1), 4) for i, n in enumerate(names) if cnt[i]}, flush=True)
    return orig(*a, **k)
bench_window.losses.rgb_depth_loss = hook
lib.gsr_profile_enable(1)
print(bench_window.run(argparse.Namespace(frames=3, iters=50, level=6, width=1920, height=1080, cameras=160)))

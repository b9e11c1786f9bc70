#!/usr/bin/env python
"""Independent view pipelines on one GPU: T host threads, each with its own HIP stream, its own leaf tensors and its own
forward + backward loop over every T-th camera of config C (no cross-stream synchronisation at all until the end).  The
small latency-bound kernels of one view (preprocess, scan, scatter) then overlap the blends of another -- what a
view-parallel job with T ranks PER GPU would see.

    python tools/bench_threads.py [--steps 160] [--threads 1 2 3]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=160)
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 2, 3])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    gs, cams, bg, params0, means2D0, rasters, dpix = bench.build_workload(dev, 0)
    for T in args.threads:
        streams = [torch.cuda.Stream(dev) for _ in range(T)]
        sets = []
        for t in range(T):
            ps = {k: v.detach().clone().requires_grad_(True) for k, v in params0.items()}
            sets.append((ps, torch.zeros_like(means2D0).requires_grad_(True)))
        barrier = threading.Barrier(T + 1)
        n_each = args.steps // T

        def worker(t):
            ps, m2 = sets[t]
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[t]):
                for s in range(3):
                    bench.one_step(s * T + t, 0, 1, ps, m2, rasters, dpix)
                streams[t].synchronize()
                barrier.wait()
                for s in range(n_each):
                    bench.one_step(s * T + t, 0, 1, ps, m2, rasters, dpix)
                streams[t].synchronize()
                barrier.wait()

        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        barrier.wait()
        t0 = time.perf_counter()
        barrier.wait()
        dt = time.perf_counter() - t0
        for x in th:
            x.join()
        n = n_each * T
        print(f"{T} thread(s) x stream(s): {dt / n * 1e3:.4f} ms per view, {n / dt:.0f} views/s", flush=True)


if __name__ == "__main__":
    main()

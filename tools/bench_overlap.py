#!/usr/bin/env python
"""Views in flight: config C, V views per step on V streams (forward of each on its own stream, ONE autograd backward over
all of them, so every view's backward runs on the stream of its forward and the gradients of the shared inputs are summed
by autograd), against the same views one after the other.

    python tools/bench_overlap.py [--steps 80] [--views 1 2 3]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--views", type=int, nargs="+", default=[1, 2, 3])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    gs, cams, bg, params, means2D, rasters, dpix = bench.build_workload(dev, 0)
    for V in args.views:
        streams = [torch.cuda.Stream(dev) for _ in range(V)]

        def step(s):
            for p in params.values():
                p.grad = None
            means2D.grad = None
            main_s = torch.cuda.current_stream()
            colors = []
            for v in range(V):
                r = rasters[(s * V + v) % len(rasters)]
                if V == 1:
                    c, _ = r(means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                             colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
                else:
                    streams[v].wait_stream(main_s)
                    with torch.cuda.stream(streams[v]):
                        c, _ = r(means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                                 colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
                colors.append(c)
            torch.autograd.backward(colors, [dpix] * V)

        for s in range(5):
            step(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(args.steps):
            step(s)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{V} view(s) in flight: {dt / args.steps * 1e3:.4f} ms per step, {dt / args.steps / V * 1e3:.4f} ms per view, "
              f"{args.steps * V / dt:.0f} views/s", flush=True)
        g = params["means3D"].grad
        print("   |grad means3D| sum", float(g.abs().sum()))


if __name__ == "__main__":
    main()

"""Dev tool (GPU): gradients of one config-C view (seeded dL_dpix) written to an .npz -- run once per library variant
(GSR_LIB_PATH) and compare with `python tests/devtools/dump_grads.py --compare a.npz b.npz`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def compare(a, b):
    A, B = np.load(a), np.load(b)
    for k in A.files:
        x, y = A[k].astype(np.float64), B[k].astype(np.float64)
        m = np.abs(y).max()
        d = np.abs(x - y)
        print(f"{k:12s} max|ref| {m:.3e}  max|diff| / max|ref| {d.max() / max(m, 1e-30):.3e}   "
              f"mean|diff| / mean|ref| {d.mean() / max(np.abs(y).mean(), 1e-30):.3e}  bit-identical {float((x == y).mean()):.4f}")


def main():
    if sys.argv[1] == "--compare":
        return compare(sys.argv[2], sys.argv[3])
    import torch
    import bench
    dev = torch.device("cuda", 0)
    gs, cams, bg, params, means2D, rasters, dpix = bench.build_workload(dev, 0)
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    r = rasters[view]
    c, _ = r(means3D=params["means3D"], means2D=means2D, opacities=params["opacities"], colors_precomp=params["colors"],
             scales=params["scales"], rotations=params["rotations"])
    c.backward(dpix)
    torch.cuda.synchronize()
    out = {k: v.grad.cpu().numpy() for k, v in params.items()}
    out["means2D"] = means2D.grad.cpu().numpy()
    np.savez(sys.argv[1], **out)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    main()

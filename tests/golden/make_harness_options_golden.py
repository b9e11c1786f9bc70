"""Generate tests/golden/harness_options_kat.npz by IMPORTING the reference's `eval_sh`
(/root/reference/gaustar_utils/spherical_harmonics.py -- importable in the dev container) on GIVEN, non-unit view directions:
the `directions=` form of SuGaR.get_points_rgb (sugar_model.py:700-716), which render_image_gaussian_rasterizer takes with
`sh_rotations` (:1200-1205).  Values and autograd gradients for sh_levels 1..5.  Runs only where /root/reference exists;
the fixture (data only) is committed.  Pins gaustar_amd.producers.points_rgb_from_directions.

    python tests/golden/make_harness_options_golden.py
"""
import importlib.util
import os

import numpy as np
import torch

REF = os.environ.get("GSR_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness_options_kat.npz")


def main():
    spec = importlib.util.spec_from_file_location("spherical_harmonics", os.path.join(REF, "gaustar_utils", "spherical_harmonics.py"))
    sh_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh_mod)
    g = torch.Generator().manual_seed(7)
    d = {}
    for lv in (1, 2, 3, 4, 5):
        P = 40
        dirs = (torch.randn(P, 3, generator=g) * 1.2).requires_grad_(True)        # NOT unit vectors: used as given
        sh = (torch.randn(P, 25, 3, generator=g) * 0.6).requires_grad_(True)
        # sugar_model.py:708-716
        coords = sh[:, :lv ** 2]
        shs_view = coords.transpose(-1, -2).view(-1, 3, lv ** 2)
        colors = torch.clamp_min(sh_mod.eval_sh(lv - 1, shs_view, dirs) + 0.5, 0.0).view(-1, 3)
        w = torch.randn(P, 3, generator=g)
        (colors * w).sum().backward()
        d[f"dirs_{lv}"], d[f"sh_{lv}"], d[f"w_{lv}"] = dirs.detach().numpy(), sh.detach().numpy(), w.numpy()
        d[f"colors_{lv}"] = colors.detach().numpy()
        d[f"ddirs_{lv}"] = dirs.grad.numpy() if dirs.grad is not None else np.zeros((P, 3), np.float32)
        d[f"dsh_{lv}"] = sh.grad.numpy()
    np.savez_compressed(OUT, **d)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

"""Generate tests/golden/utils_kat.npz by IMPORTING the reference's Python helpers
(/root/reference/gaustar_utils/{graphics_utils,spherical_harmonics}.py -- importable in the dev
container: they need only torch/numpy).  Runs only where /root/reference exists; the fixture (data
only) is committed.  Pins gaustar_amd.scene's camera-matrix and SH-colour restatements.

    python tests/golden/make_utils_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = os.environ.get("GSR_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "utils_kat.npz")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "gaustar_utils", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    gu, sh = _load("graphics_utils"), _load("spherical_harmonics")
    rng = np.random.default_rng(0)
    d = {}
    # projection matrices / fov helpers
    cases = [(0.01, 100.0, 0.9, 0.7), (1e-4, 100.0, 1.3494, 0.8448), (0.1, 10.0, 0.3, 1.2)]
    d["proj_args"] = np.array(cases, np.float64)
    d["proj_out"] = np.stack([gu.getProjectionMatrix(*c).numpy() for c in cases])
    d["focal_in"] = np.array([[1200.0, 1920.0], [1200.0, 1080.0], [500.0, 512.0]])
    d["focal2fov_out"] = np.array([gu.focal2fov(f, p) for f, p in d["focal_in"]])
    # world->view
    A = rng.normal(size=(3, 3))
    Q, _ = np.linalg.qr(A)
    t = rng.normal(size=3)
    d["w2v_R"], d["w2v_t"] = Q, t
    d["w2v_out"] = gu.getWorld2View(Q, t)
    # SH evaluation deg 0..3: eval_sh takes sh [..., C, (deg+1)^2] and unit dirs
    N = 64
    shc = rng.uniform(-1, 1, size=(N, 16, 3)).astype(np.float32)       # our layout [N, M, 3]
    dirs = rng.normal(size=(N, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    d["sh_coeffs"], d["sh_dirs"] = shc, dirs
    for deg in range(4):
        m = (deg + 1) ** 2
        res = sh.eval_sh(deg, torch.from_numpy(shc[:, :m]).transpose(-1, -2), torch.from_numpy(dirs))
        d[f"sh_rgb_deg{deg}"] = torch.clamp_min(res + 0.5, 0.0).numpy()   # sugar_model.py:714-716
    np.savez_compressed(OUT, **d)
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())

"""gaustar_amd -- MI355X-native differentiable surface-Gaussian rasterizer (GauSTAR hot path).

    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer

is the same API as the reference's `diff_gaussian_rasterization` package; the compute path is
hand-written HIP for gfx950 behind the C ABI of include/gsr.h.  `scene` (numpy only) builds the
synthetic cameras / mesh-bound Gaussians used by tests and bench.py; `dist` is the view-parallel
gradient all-reduce.  Either side of the rasterizer: `producers` (SH -> RGB, mesh-bound
means / scales / quaternions), `losses` (l1 + dssim, masked depth L1), `sweep` (forward-only camera sweeps) and
`formats` (cameras.json, 3DGS PLY, SuGaR .pt) -- import them as submodules.
"""
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]

"""Import-name shim: `from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer` (gaustar_scene/sugar_model.py:10, gaustar_scene/sugar_compositor.py:4,
gaussian_splatting/gaussian_renderer/__init__.py:14) resolves to the MI355X-native package when
this repository's root is on sys.path, so the reference's callers run unchanged."""
from gaustar_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                    _RasterizeGaussians, cpu_deep_copy_tuple, rasterize_gaussians)

"""Generates tests/golden/producers_kat.npz by IMPORTING the reference's own gaustar_utils/spherical_harmonics.py
(build container only).  The surrounding three lines of SuGaR.get_points_rgb (sugar_model.py:700, :711-716:
normalize, slice/transposes, clamp_min(+0.5)) are inline model code (sugar_model.py itself cannot be imported:
open3d / pytorch3d are absent) and are applied here on top of the imported eval_sh.  Inputs + expected outputs."""
import os, sys
import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from gaustar_utils.spherical_harmonics import eval_sh   # noqa: E402

out = {}
g = torch.Generator().manual_seed(0)
for sh_levels, M in ((1, 1), (2, 4), (3, 16), (4, 16), (5, 25)):   # (5, 25): degree 4, the highest eval_sh accepts
    P = 700
    pos = (torch.rand(P, 3, generator=g) * 2 - 1).requires_grad_(True)
    cam = torch.tensor([[0.3, 1.4, -3.0]])
    sh = (torch.rand(P, M, 3, generator=g) - 0.5)
    sh[:, 0] = torch.rand(P, 3, generator=g) * 6 - 3            # DC wide enough that the clamp at 0 is hit
    sh.requires_grad_(True)
    dirs = torch.nn.functional.normalize(pos - cam, dim=-1)
    coords = sh[:, :sh_levels ** 2]
    shs_view = coords.transpose(-1, -2).view(-1, 3, sh_levels ** 2)
    colors = torch.clamp_min(eval_sh(sh_levels - 1, shs_view, dirs) + 0.5, 0.0).view(-1, 3)
    dL = torch.randn(P, 3, generator=g)
    colors.backward(dL)
    k = f"l{sh_levels}"
    out.update({f"{k}_pos": pos.detach().numpy(), f"{k}_cam": cam.numpy(), f"{k}_sh": sh.detach().numpy(),
                f"{k}_colors": colors.detach().numpy(), f"{k}_dL": dL.numpy(), f"{k}_dpos": (pos.grad if pos.grad is not None else torch.zeros_like(pos)).numpy(),
                f"{k}_dsh": sh.grad.numpy()})
    assert (colors == 0).any()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "producers_kat.npz"), **out)
print("wrote producers_kat.npz")

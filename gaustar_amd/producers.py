"""Producers of rasterizer inputs as fused HIP ops.

`points_rgb` mirrors SuGaR.get_points_rgb (gaustar_scene/sugar_model.py:674-718): view-dependent colours from
spherical-harmonic coefficients, `clamp_min(eval_sh(...) + 0.5, 0)`, with gradients to the positions (through the
normalised view direction) and to the coefficients.  One kernel forward, one backward, instead of ~60 elementwise
kernels each way.  No CPU path."""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _PointsRGB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, positions, camera_center, sh_coordinates, sh_levels):
        lib = _lib.load()
        if not positions.is_cuda:
            raise RuntimeError("gaustar_amd.producers: positions must live on a HIP (cuda) device -- there is no CPU path")
        if positions.dim() != 2 or positions.size(1) != 3:
            raise RuntimeError("positions must have dimensions (num_points, 3)")
        if sh_coordinates.dim() != 3 or sh_coordinates.size(2) != 3 or sh_coordinates.size(0) != positions.size(0):
            raise RuntimeError("sh_coordinates must have dimensions (num_points, n_coeffs, 3)")
        D = int(sh_levels) - 1
        M = int(sh_coordinates.size(1))
        if D < 0 or D > 3 or (D + 1) ** 2 > M:
            raise RuntimeError(f"sh_levels must be 1..4 and sh_levels**2 <= n_coeffs ({M})")
        dev = positions.device
        pos = positions.detach().to(torch.float32).contiguous()
        cam = camera_center.detach().to(dev, torch.float32).reshape(-1)[:3].contiguous()
        if camera_center.numel() != 3:
            raise RuntimeError("camera_center must hold one 3-vector (shape (3,) or (1, 3))")
        sh = sh_coordinates.detach().to(torch.float32).contiguous()
        P = int(pos.size(0))
        rgb = torch.empty(P, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.gsr_sh_to_rgb(P, D, M, _p(pos), _p(cam), _p(sh), _p(rgb), _stream()), "gsr_sh_to_rgb")
        ctx.save_for_backward(pos, cam, sh)
        ctx.D = D
        return rgb

    @staticmethod
    def backward(ctx, dL_drgb):
        lib = _lib.load()
        pos, cam, sh = ctx.saved_tensors
        P, M = int(pos.size(0)), int(sh.size(1))
        g = dL_drgb.to(torch.float32).contiguous()
        dsh = torch.empty_like(sh)
        dpos = torch.empty_like(pos)
        with torch.cuda.device(pos.device):
            _lib.check(lib.gsr_sh_to_rgb_backward(P, ctx.D, M, _p(pos), _p(cam), _p(sh), _p(g), _p(dsh), _p(dpos),
                                                  _stream()), "gsr_sh_to_rgb_backward")
        return dpos, None, dsh, None


def points_rgb(positions: torch.Tensor, camera_centers: torch.Tensor, sh_coordinates: torch.Tensor,
               sh_levels: int) -> torch.Tensor:
    """colors[P,3] = clamp_min(eval_sh(sh_levels-1, sh_coordinates[:, :sh_levels**2], normalize(positions -
    camera_centers)) + 0.5, 0), sugar_model.py:698-716 with one camera centre ((3,) or (1,3)), sh_coordinates
    [P, n_coeffs, 3] as SuGaR stores them (sugar_model.py:449-450)."""
    return _PointsRGB.apply(positions, camera_centers, sh_coordinates, int(sh_levels))

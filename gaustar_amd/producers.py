"""Producers of rasterizer inputs as fused HIP ops.

`points_rgb` mirrors SuGaR.get_points_rgb (gaustar_scene/sugar_model.py:674-718): view-dependent colours from
spherical-harmonic coefficients, `clamp_min(eval_sh(...) + 0.5, 0)`, with gradients to the positions (through the
normalised view direction) and to the coefficients.  One kernel forward, one backward, instead of ~60 elementwise
kernels each way.  No CPU path."""
from __future__ import annotations

import ctypes

import torch

from . import _host, _lib


def _stream():
    return _host.raw_stream(torch._C._cuda_getDevice())


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _op(t):
    return None if t is None else _p(t)


def _sh_forward_raw(pos, cam, sh, sh_rest, D, M, view, depth_channels, densities=None):
    """The colour producer on prepared tensors (float32, contiguous, detached, one device): sh [P,M,3] with sh_rest None,
    or sh = `_sh_coordinates_dc` [P,1,3] and sh_rest = `_sh_coordinates_rest` [P,M-1,3].  -> (colors, opacity or None)."""
    lib = _lib.load()
    P = int(pos.size(0))
    dev = pos.device
    colors = torch.empty(P, 3 + (int(depth_channels) if view is not None else 0), dtype=torch.float32, device=dev)
    opacity = torch.empty(P, 1, dtype=torch.float32, device=dev) if densities is not None else None
    with _host.on_device(dev):
        if sh_rest is None and densities is None:
            if view is None:
                _lib.check(lib.gsr_sh_to_rgb(P, D, M, _p(pos), _p(cam), _p(sh), _p(colors), _stream()), "gsr_sh_to_rgb")
            else:
                _lib.check(lib.gsr_sh_to_rgbd(P, D, M, _p(pos), _p(cam), _p(sh), _p(view), int(depth_channels), _p(colors), _stream()),
                           "gsr_sh_to_rgbd")
        else:
            if sh_rest is None and M != 1:
                raise RuntimeError("opacities ride along only with the two-array coefficient layout")
            _lib.check(lib.gsr_sh_colors_split(P, D, M, _p(pos), _p(cam), _p(sh), _op(sh_rest if M > 1 else None), _op(view),
                                               int(depth_channels) if view is not None else 0, _op(densities), _p(colors),
                                               _op(opacity), _stream()), "gsr_sh_colors_split")
    return colors, opacity


def _sh_backward_raw(pos, cam, sh, sh_rest, D, M, view, depth_channels, g, opacity=None, dL_dopacity=None, dpos_inout=None,
                     out=(None, None, None)):
    """-> (dL_dsh, dL_dsh_rest or None, dL_dpos, dL_ddensities or None).  dpos_inout: a [P,3] gradient w.r.t. the positions
    that takes this producer's on top (in place) instead of a fresh array.  out: contiguous float32 tensors to write
    dL_dsh / dL_dsh_rest / dL_ddensities into instead of fresh ones (e.g. views of an optimiser's flat gradient buffer)."""
    lib = _lib.load()
    P = int(pos.size(0))
    dsh = out[0] if out[0] is not None else torch.empty_like(sh)
    drest = (out[1] if out[1] is not None else torch.empty_like(sh_rest)) if sh_rest is not None else None
    dpos = dpos_inout if dpos_inout is not None else torch.empty_like(pos)
    ddens = (out[2] if out[2] is not None else torch.empty_like(opacity)) if opacity is not None else None
    with _host.on_device(pos.device):
        if sh_rest is None and opacity is None and dpos_inout is None:
            if view is None:
                _lib.check(lib.gsr_sh_to_rgb_backward(P, D, M, _p(pos), _p(cam), _p(sh), _p(g), _p(dsh), _p(dpos), _stream()),
                           "gsr_sh_to_rgb_backward")
            else:
                _lib.check(lib.gsr_sh_to_rgbd_backward(P, D, M, _p(pos), _p(cam), _p(sh), _p(view), int(depth_channels), _p(g),
                                                       _p(dsh), _p(dpos), _stream()), "gsr_sh_to_rgbd_backward")
        else:
            if sh_rest is None and M != 1:
                raise RuntimeError("the fused backward takes the two-array coefficient layout")
            _lib.check(lib.gsr_sh_colors_split_backward(
                P, D, M, _p(pos), _p(cam), _p(sh), _op(sh_rest if M > 1 else None), _op(view),
                int(depth_channels) if view is not None else 0, _p(g), _op(opacity), _op(dL_dopacity), _p(dsh),
                _op(drest if M > 1 else None), _p(dpos), int(dpos_inout is not None), _op(ddens), _stream()),
                "gsr_sh_colors_split_backward")
    return dsh, drest, dpos, ddens


class _PointsRGB(torch.autograd.Function):
    """view is None: rgb [P,3]; otherwise rgb + view-space depth as a second colour target [P,6] (points_rgb_depth)."""

    @staticmethod
    def forward(ctx, positions, camera_center, sh_coordinates, sh_levels, view=None, depth_channels=3):
        lib = _lib.load()
        if not positions.is_cuda:
            raise RuntimeError("gaustar_amd.producers: positions must live on a HIP (cuda) device -- there is no CPU path")
        if positions.dim() != 2 or positions.size(1) != 3:
            raise RuntimeError("positions must have dimensions (num_points, 3)")
        if sh_coordinates.dim() != 3 or sh_coordinates.size(2) != 3 or sh_coordinates.size(0) != positions.size(0):
            raise RuntimeError("sh_coordinates must have dimensions (num_points, n_coeffs, 3)")
        D = int(sh_levels) - 1
        M = int(sh_coordinates.size(1))
        if D < 0 or D > 4 or (D + 1) ** 2 > M:   # eval_sh asserts deg <= 4 (spherical_harmonics.py:130)
            raise RuntimeError(f"sh_levels must be 1..5 and sh_levels**2 <= n_coeffs ({M})")
        dev = positions.device
        pos = positions.detach().to(torch.float32).contiguous()
        cam = camera_center.detach().to(dev, torch.float32).reshape(-1)[:3].contiguous()
        if camera_center.numel() != 3:
            raise RuntimeError("camera_center must hold one 3-vector (shape (3,) or (1, 3))")
        sh = sh_coordinates.detach().to(torch.float32).contiguous()
        P = int(pos.size(0))
        if view is not None:
            if tuple(view.shape) != (4, 4):
                raise RuntimeError("viewmatrix must be (4, 4), as handed to the rasterizer")
            view = view.detach().to(dev, torch.float32).contiguous()
        if view is not None and depth_channels not in (1, 3):
            raise RuntimeError("depth_channels must be 1 (colours [P,4]) or 3 (colours [P,6])")
        rgb, _ = _sh_forward_raw(pos, cam, sh, None, D, M, view, depth_channels)
        ctx.save_for_backward(pos, cam, sh, view)
        ctx.D = D
        ctx.depth_channels = int(depth_channels)
        return rgb

    @staticmethod
    def backward(ctx, dL_drgb):
        pos, cam, sh, view = ctx.saved_tensors
        M = int(sh.size(1))
        g = dL_drgb.to(torch.float32).contiguous()
        dsh, _, dpos, _ = _sh_backward_raw(pos, cam, sh, None, ctx.D, M, view, ctx.depth_channels, g)
        return dpos, None, dsh, None, None, None


def points_rgb(positions: torch.Tensor, camera_centers: torch.Tensor, sh_coordinates: torch.Tensor,
               sh_levels: int) -> torch.Tensor:
    """colors[P,3] = clamp_min(eval_sh(sh_levels-1, sh_coordinates[:, :sh_levels**2], normalize(positions -
    camera_centers)) + 0.5, 0), sugar_model.py:698-716 with one camera centre ((3,) or (1,3)), sh_coordinates
    [P, n_coeffs, 3] as SuGaR stores them (sugar_model.py:449-450)."""
    return _PointsRGB.apply(positions, camera_centers, sh_coordinates, int(sh_levels))


def points_rgb_depth(positions: torch.Tensor, camera_centers: torch.Tensor, sh_coordinates: torch.Tensor, sh_levels: int,
                     viewmatrix: torch.Tensor, depth_channels: int = 3) -> torch.Tensor:
    """colors[P,6] for the one-pass RGB + depth render: columns 0-2 = points_rgb(...), columns 3-5 = the view-space
    depth of every position, three times -- the `point_depth.expand(-1, 3)` GauSTAR renders as colours
    (gaustar_trainers/refine.py:603-605).  viewmatrix: the (4, 4) world-to-view matrix handed to the rasterizer
    (row-vector convention, sugar_model.py:1149).  One kernel each way instead of eval_sh + a skinny matmul + cat and
    their autograd mirror; the gradient w.r.t. viewmatrix is not provided (cameras are fixed in the trainer).
    depth_channels = 1 gives colors[P,4] = {rgb, z} for the 4-channel render (RGB + one scalar target)."""
    return _PointsRGB.apply(positions, camera_centers, sh_coordinates, int(sh_levels), viewmatrix, int(depth_channels))


class _PointsColorsSplit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, positions, camera_center, sh_dc, sh_rest, sh_levels, view, depth_channels, densities):
        if not positions.is_cuda:
            raise RuntimeError("gaustar_amd.producers: positions must live on a HIP (cuda) device -- there is no CPU path")
        P = int(positions.size(0)) if positions.dim() == 2 else -1
        if positions.dim() != 2 or positions.size(1) != 3:
            raise RuntimeError("positions must have dimensions (num_points, 3)")
        if tuple(sh_dc.shape) != (P, 1, 3) or sh_rest.dim() != 3 or sh_rest.size(0) != P or sh_rest.size(2) != 3:
            raise RuntimeError("sh_dc must be (num_points, 1, 3) and sh_rest (num_points, n_coeffs - 1, 3)")
        D, M = int(sh_levels) - 1, 1 + int(sh_rest.size(1))
        if D < 0 or D > 4 or (D + 1) ** 2 > M:
            raise RuntimeError(f"sh_levels must be 1..5 and sh_levels**2 <= n_coeffs ({M})")
        if camera_center.numel() != 3:
            raise RuntimeError("camera_center must hold one 3-vector (shape (3,) or (1, 3))")
        if view is not None and (tuple(view.shape) != (4, 4) or depth_channels not in (1, 3)):
            raise RuntimeError("viewmatrix must be (4, 4) and depth_channels 1 or 3")
        if densities is not None and densities.numel() != P:
            raise RuntimeError("densities must hold one value per point")
        dev = positions.device
        f32 = lambda t: None if t is None else t.detach().to(dev, torch.float32).contiguous()
        pos, dc, rest, view, dens = f32(positions), f32(sh_dc), f32(sh_rest), f32(view), f32(densities)
        cam = camera_center.detach().to(dev, torch.float32).reshape(-1).contiguous()
        colors, opacity = _sh_forward_raw(pos, cam, dc, rest, D, M, view, depth_channels if view is not None else 0, dens)
        ctx.save_for_backward(pos, cam, dc, rest, view, opacity)
        ctx.cfg = (D, M, int(depth_channels) if view is not None else 0)
        ctx.dens_shape = None if densities is None else tuple(densities.shape)
        if opacity is None:
            return colors
        return colors, opacity

    @staticmethod
    def backward(ctx, dL_dcolors, dL_dopacity=None):
        pos, cam, dc, rest, view, opacity = ctx.saved_tensors
        D, M, dch = ctx.cfg
        g = dL_dcolors.to(torch.float32).contiguous()
        if opacity is not None:
            dL_dopacity = torch.zeros_like(opacity) if dL_dopacity is None else dL_dopacity.to(torch.float32).contiguous()
        ddc, drest, dpos, ddens = _sh_backward_raw(pos, cam, dc, rest, D, M, view, dch, g, opacity, dL_dopacity)
        if ddens is not None:
            ddens = ddens.view(ctx.dens_shape)
        return dpos, None, ddc, drest, None, None, None, ddens


def points_colors_split(positions: torch.Tensor, camera_centers: torch.Tensor, sh_dc: torch.Tensor, sh_rest: torch.Tensor,
                        sh_levels: int, viewmatrix: torch.Tensor = None, depth_channels: int = 1, densities: torch.Tensor = None):
    """points_rgb / points_rgb_depth reading SuGaR's coefficients where they live -- `_sh_coordinates_dc` [P,1,3] and
    `_sh_coordinates_rest` [P,n-1,3] (sugar_model.py:449-450 concatenates them on every access) -- and, with `densities`
    ([P,1] `all_densities`), returning SuGaR.strengths (sigmoid, sugar_model.py:442-447) from the same kernel:
    -> colors [P,3] (viewmatrix None) or [P, 3 + depth_channels]; with densities: (colors, opacities [P,1]).
    Bit-identical to points_rgb_depth(torch.cat([sh_dc, sh_rest], 1)) and torch.sigmoid, forward and backward."""
    return _PointsColorsSplit.apply(positions, camera_centers, sh_dc, sh_rest, int(sh_levels), viewmatrix, int(depth_channels),
                                    densities)


def _mesh_forward_raw(v, fc, bc, rs, rc, thickness, lo, hi, dt, dr, clear=None):
    """The mesh producer on prepared tensors (float32 / int64, contiguous, detached, one device) -> (points, scaling, quats).
    clear: a contiguous float32 [V,3] tensor the launch also sets to zero -- the backward's vertex-gradient accumulator, handed
    to _mesh_backward_raw as out[0] with verts_cleared=True."""
    lib = _lib.load()
    dev = v.device
    F, G = int(fc.size(0)), int(bc.size(0))
    N = F * G
    points = torch.empty(N, 3, dtype=torch.float32, device=dev)
    scaling = torch.empty(N, 3, dtype=torch.float32, device=dev)
    quats = torch.empty(N, 4, dtype=torch.float32, device=dev)
    with _host.on_device(dev):
        _lib.check(lib.gsr_mesh_gaussians(F, G, _p(v), _p(fc), _p(bc), _p(rs), _p(rc), float(thickness), lo, hi, _op(dt),
                                          _op(dr), _p(points), _p(scaling), _p(quats), _op(clear), int(v.size(0)), _stream()),
                   "gsr_mesh_gaussians")
    return points, scaling, quats


def _mesh_backward_raw(v, fc, bc, rs, rc, dr, lo, hi, has_dt, g_points, g_scaling, g_quats, out=(None,) * 5, verts_cleared=False):
    """-> (d_verts, d_raw_scales, d_raw_complex, d_delta_t or None, d_delta_r or None); absent output gradients are None.
    out: contiguous float32 tensors to write the five gradients into instead of fresh ones."""
    lib = _lib.load()
    dev = v.device
    F, G, V = int(fc.size(0)), int(bc.size(0)), int(v.size(0))
    pick = lambda o, make: o if o is not None else make()
    d_verts = pick(out[0], lambda: torch.empty_like(v))
    d_rs, d_rc = pick(out[1], lambda: torch.empty_like(rs)), pick(out[2], lambda: torch.empty_like(rc))
    d_dt = pick(out[3], lambda: torch.empty(F * G, 3, dtype=torch.float32, device=dev)) if has_dt else None
    d_dr = pick(out[4], lambda: torch.empty_like(dr)) if dr is not None else None
    with _host.on_device(dev):
        _lib.check(lib.gsr_mesh_gaussians_backward(
            F, G, V, _p(v), _p(fc), _p(bc), _p(rs), _p(rc), lo, hi, _op(dr), _op(g_points), _op(g_scaling), _op(g_quats),
            _p(d_verts), _p(d_rs), _p(d_rc), _op(d_dt), _op(d_dr), int(bool(verts_cleared and out[0] is not None)), _stream()),
            "gsr_mesh_gaussians_backward")
    return d_verts, d_rs, d_rc, d_dt, d_dr


class _MeshGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces, bary, raw_scales, raw_complex, thickness, min_scale, max_scale, delta_t, delta_r):
        if not verts.is_cuda:
            raise RuntimeError("gaustar_amd.producers: verts must live on a HIP (cuda) device -- there is no CPU path")
        dev = verts.device
        f32 = lambda t: None if t is None else t.detach().to(dev, torch.float32).contiguous()
        v, rs, rc, dt, dr = f32(verts), f32(raw_scales), f32(raw_complex), f32(delta_t), f32(delta_r)
        fc = faces.detach().to(dev, torch.int64).contiguous()
        bc = f32(bary).reshape(-1, 3)
        F, G = int(fc.size(0)), int(bc.size(0))
        N = F * G
        if v.dim() != 2 or v.size(1) != 3 or fc.dim() != 2 or fc.size(1) != 3:
            raise RuntimeError("verts must be (V, 3) and faces (F, 3)")
        if tuple(rs.shape) != (N, 2) or tuple(rc.shape) != (N, 2):
            raise RuntimeError(f"raw scales and raw 2-D rotations must both be ({N}, 2)")
        if (dt is not None and tuple(dt.shape) != (N, 3)) or (dr is not None and tuple(dr.shape) != (N, 4)):
            raise RuntimeError(f"delta_t must be ({N}, 3) and delta_r ({N}, 4)")
        lo = float("-inf") if min_scale is None else float(min_scale)
        hi = float("inf") if max_scale is None else float(max_scale)
        # (the backward's vertex-gradient accumulator is cleared by the forward's launch; good for ONE backward)
        ctx.d_verts = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        points, scaling, quats = _mesh_forward_raw(v, fc, bc, rs, rc, thickness, lo, hi, dt, dr, clear=ctx.d_verts)
        ctx.save_for_backward(v, fc, bc, rs, rc, dr)
        ctx.dims = (lo, hi, dt is not None)
        return points, scaling, quats

    @staticmethod
    def backward(ctx, g_points, g_scaling, g_quats):
        v, fc, bc, rs, rc, dr = ctx.saved_tensors
        lo, hi, has_dt = ctx.dims
        c = lambda t: None if t is None else t.to(torch.float32).contiguous()
        pre, ctx.d_verts = ctx.d_verts, None
        d_verts, d_rs, d_rc, d_dt, d_dr = _mesh_backward_raw(v, fc, bc, rs, rc, dr, lo, hi, has_dt, c(g_points), c(g_scaling),
                                                             c(g_quats), out=(pre, None, None, None, None), verts_cleared=pre is not None)
        return d_verts, None, None, d_rs, d_rc, None, None, None, d_dt, d_dr


def mesh_bound_gaussians(verts: torch.Tensor, faces: torch.Tensor, bary_coords: torch.Tensor, raw_scales: torch.Tensor,
                         raw_complex: torch.Tensor, thickness: float, min_scale=None, max_scale=None,
                         delta_t: torch.Tensor = None, delta_r: torch.Tensor = None):
    """-> (points [N,3], scaling [N,3], quaternions [N,4]) of the N = F*G Gaussians bound to a triangle mesh:
    SuGaR.points / .scaling / .quaternions (sugar_model.py:417-435, :457-476, :478-508) in one kernel, one more
    for the backward.  Arguments are the model's `_points`, `_surface_mesh_faces`,
    `surface_triangle_bary_coords` ([G,3] or [G,3,1]), `_scales`, `_quaternions` (the 2-D rotation),
    `surface_mesh_thickness`, `min_gaussian_scale`, `max_gaussian_scale`, and the loose-bind `_delta_t`, `_delta_r`."""
    _check_faces(faces, int(verts.shape[0]))
    return _MeshGaussians.apply(verts, faces, bary_coords, raw_scales, raw_complex, float(thickness), min_scale,
                                max_scale, delta_t, delta_r)


_FACES_ATTR = "_gsr_checked_faces"   # set on the tensor OBJECT: (version, number of vertices the indices were checked against)


def _check_faces(faces: torch.Tensor, n_verts: int) -> None:
    """A face index outside [0, V) would read out of bounds in the forward kernel and ADD out of bounds in the backward
    (silent corruption; the reference's `self._points[self._surface_mesh_faces]` raises).  Checked once per faces tensor
    OBJECT (one device reduction) and remembered on the object itself until it is modified in place -- not keyed on the
    data pointer, which the caching allocator hands to the next tensor of the same size."""
    if faces.numel() == 0:
        return
    if getattr(faces, _FACES_ATTR, None) == (faces._version, n_verts):
        return
    lo, hi = int(faces.min()), int(faces.max())
    if lo < 0 or hi >= n_verts:
        raise IndexError(f"faces hold vertex indices in [{lo}, {hi}] but there are {n_verts} vertices")
    try:
        setattr(faces, _FACES_ATTR, (faces._version, n_verts))
    except AttributeError:   # (a tensor subclass with __slots__: checked every call)
        pass


# ------------------------------------------------------------------------------------------------------------------
# The `directions=` form of SuGaR.get_points_rgb (sugar_model.py:700-716): colours from GIVEN view directions, used as they
# are (not re-normalised) -- the form render_image_gaussian_rasterizer takes with `sh_rotations` (sugar_model.py:1200-1205).
# A rarely used option: plain torch operations on the caller's (device) tensors, autograd included; the constants and
# polynomials are those of gaustar_utils/spherical_harmonics.py:16-33, :134-171 (= auxiliary.h:22-39 up to degree 3).
# ------------------------------------------------------------------------------------------------------------------
_SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)
_SH_C4 = (2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431,
          -0.6690465435572892, 0.47308734787878004, -1.7701307697799304, 0.6258357354491761)


def sh_basis_torch(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """[P, (deg+1)^2] values of the real SH basis polynomials at `dirs` [P,3] (spherical_harmonics.py:134-171; degree <= 4)."""
    if deg < 0 or deg > 4:
        raise ValueError("sh degree must be 0..4")
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    b = [torch.full_like(x, _SH_C0)]
    if deg > 0:
        b += [-_SH_C1 * y, _SH_C1 * z, -_SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [_SH_C2[0] * xy, _SH_C2[1] * yz, _SH_C2[2] * (2.0 * zz - xx - yy), _SH_C2[3] * xz, _SH_C2[4] * (xx - yy)]
        if deg > 2:
            b += [_SH_C3[0] * y * (3 * xx - yy), _SH_C3[1] * xy * z, _SH_C3[2] * y * (4 * zz - xx - yy),
                  _SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), _SH_C3[4] * x * (4 * zz - xx - yy), _SH_C3[5] * z * (xx - yy),
                  _SH_C3[6] * x * (xx - 3 * yy)]
            if deg > 3:
                b += [_SH_C4[0] * xy * (xx - yy), _SH_C4[1] * yz * (3 * xx - yy), _SH_C4[2] * xy * (7 * zz - 1),
                      _SH_C4[3] * yz * (7 * zz - 3), _SH_C4[4] * (zz * (35 * zz - 30) + 3), _SH_C4[5] * xz * (7 * zz - 3),
                      _SH_C4[6] * (xx - yy) * (7 * zz - 1), _SH_C4[7] * xz * (xx - 3 * yy),
                      _SH_C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(b, dim=-1)


def points_rgb_from_directions(directions: torch.Tensor, sh_coordinates: torch.Tensor, sh_levels: int) -> torch.Tensor:
    """colors[P,3] = clamp_min(eval_sh(sh_levels-1, sh_coordinates[:, :sh_levels**2], directions) + 0.5, 0) with the
    directions taken as given (sugar_model.py:702-716)."""
    n = int(sh_levels) ** 2
    if sh_coordinates.dim() != 3 or sh_coordinates.size(1) < n or sh_coordinates.size(2) != 3:
        raise RuntimeError(f"sh_coordinates must be (num_points, >= {n}, 3)")
    basis = sh_basis_torch(int(sh_levels) - 1, directions)                       # [P, n]
    return torch.clamp_min((basis.unsqueeze(-1) * sh_coordinates[:, :n]).sum(dim=1) + 0.5, 0.0)


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.quaternion_to_matrix (real part first; the published algorithm: two_s = 2 / |q|^2), as
    render_image_gaussian_rasterizer uses it for `compute_covariance_in_rasterizer=False` (sugar_model.py:1239)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def covariance_3d(scales: torch.Tensor, quaternions: torch.Tensor) -> torch.Tensor:
    """cov3D[P,6] = upper triangle {xx, xy, xz, yy, yz, zz} of R diag(s^2) R^T (sugar_model.py:1237-1257)."""
    R = quaternion_to_matrix(quaternions)
    M = R * (scales * scales).unsqueeze(-2)                                      # R diag(s^2)
    S = M @ R.transpose(-1, -2)
    return torch.stack((S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]), dim=-1)

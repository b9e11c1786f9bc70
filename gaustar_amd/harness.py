"""Host-side counterpart of the reference's render harness:
`SuGaR.render_image_gaussian_rasterizer` and the properties it reads (gaustar_scene/sugar_model.py:1065-1311, :417-508,
:442-450, :674-718), assembled from this package's fused ops.  The reference module itself cannot be imported here
(open3d / pytorch3d), so this mirrors its interface -- same parameter names in the state dict, same argument names
and return conventions of the render call -- for the mesh-bound ("binded_to_surface_mesh") case GauSTAR uses.

    model = SurfaceGaussians.from_checkpoint(formats.load_sugar_checkpoint("2000.pt"), device)
    image = model.render_image_gaussian_rasterizer(camera=cam, bg_color=[0, 1, 0], sh_deg=3)          # [H,W,3]
    rgb, depth = model.render_rgb_depth(camera=cam, bg_color=[0, 1, 0], max_depth=10.0, sh_deg=3)     # one pass

What differs from the reference, by design: per-camera matrices are built once and cached on the device (the reference
redoes a numpy inverse and three uploads per call, :1129-1163); points / scaling / quaternions come from one fused kernel
(and one backward) per parameter version instead of ~50 elementwise kernels per property access.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import producers, rasterizer as _rasterizer, scene
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

BARY_COORDS = {  # sugar_model.py:186-226
    1: [[1 / 3, 1 / 3, 1 / 3]],
    3: [[1 / 2, 1 / 4, 1 / 4], [1 / 4, 1 / 2, 1 / 4], [1 / 4, 1 / 4, 1 / 2]],
    4: [[1 / 3, 1 / 3, 1 / 3], [2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3]],
    6: [[2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3], [1 / 6, 5 / 12, 5 / 12], [5 / 12, 1 / 6, 5 / 12],
        [5 / 12, 5 / 12, 1 / 6]],
}


@dataclass
class NerfCamera:
    """One camera as CamerasWrapper holds it: `c2w` is the [3,4] (or [4,4]) NeRF 'transform_matrix' -- camera-to-world
    with OpenGL/Blender axes (Y up, Z back) --, pinhole intrinsics in pixels."""
    c2w: np.ndarray
    fx: float
    fy: float
    width: int
    height: int
    znear: float = 1e-4
    zfar: float = 100.0
    principal_ndc: Sequence[float] = (0.0, 0.0)   # p3d K[0,0,2], K[0,1,2]; 0 when cx = W/2, cy = H/2 (cameras.py:276-277)
    # camera centre handed to the rasterizer instead of c2w[:3, 3] (with_extrinsic: the reference's -T R^-1 differs from it
    # for a non-rigid extrinsic); a declared field, so dataclasses.replace / copy / pickle keep it
    center_override: Optional[np.ndarray] = None
    _cache: Dict = field(default_factory=dict, repr=False, compare=False)

    def rasterizer_camera(self) -> scene.Camera:
        """sugar_model.py:1129-1163: flip to COLMAP axes, invert, R stored transposed, world_view = getWorld2View(R, T)^T,
        proj = getProjectionMatrix(...)^T with the principal-point entries, full = world_view @ proj."""
        c2w = np.eye(4)
        c2w[:np.asarray(self.c2w).shape[0], :] = np.asarray(self.c2w, dtype=np.float64)
        c2w[:3, 1:3] *= -1                                    # :1134
        w2c = np.linalg.inv(c2w)                              # :1138
        R, T = np.transpose(w2c[:3, :3]), w2c[:3, 3]          # :1139-1140
        fovx, fovy = scene.focal2fov(self.fx, self.width), scene.focal2fov(self.fy, self.height)
        view_t = scene.get_world2view(R, T).transpose()       # :1149-1150
        proj_t = scene.get_projection_matrix(self.znear, self.zfar, fovx, fovy).transpose().copy()
        proj_t[2, 0] = -float(self.principal_ndc[0])          # :1159-1160
        proj_t[2, 1] = -float(self.principal_ndc[1])
        full_t = (view_t @ proj_t).astype(np.float32)
        campos = np.asarray(self.c2w, dtype=np.float64)[:3, 3]    # p3d get_camera_center() = camera position
        if self.center_override is not None:                  # (with_extrinsic: the reference's own formula)
            campos = np.asarray(self.center_override, dtype=np.float64)
        return scene.Camera(W=int(self.width), H=int(self.height), tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5),
                            viewmatrix=np.ascontiguousarray(view_t, dtype=np.float32), projmatrix=np.ascontiguousarray(full_t),
                            campos=campos.astype(np.float32))

    def with_extrinsic(self, extr) -> "NerfCamera":
        """The camera `overwrite_extr` turns this one into (sugar_model.py:1119-1127 and :1141-1147): `extr` is a [4,4]
        WORLD-TO-CAMERA matrix in COLMAP axes (X right, Y down, Z forward).  The reference stores R = inverse(extr[:3,:3])
        and T = extr[:3,3] in its pytorch3d camera with the first two axes negated, negates them back, and hands
        getWorld2View(R, T) = [R^T | T] to the rasterizer; its camera centre is pytorch3d's -T R^-1 of the stored pair, i.e.
        -extr[:3,:3]^T ... for a rotation, the camera position.  Intrinsics, image size and clip planes are kept."""
        E = np.asarray(extr.detach().cpu().numpy() if isinstance(extr, torch.Tensor) else extr, dtype=np.float64).reshape(4, 4)
        key = ("extr", E.tobytes())                          # (the derived camera and its device tensors are built once per pose)
        hit = self._cache.get(key)
        if hit is not None:
            return hit
        R = np.linalg.inv(E[:3, :3])                          # :1121 (p3d_camera.R, the two sign flips of :1122 / :1143 cancel)
        T = E[:3, 3].copy()                                   # :1123
        w2c = np.eye(4)
        w2c[:3, :3] = R.transpose()                           # getWorld2View: Rt[:3,:3] = R^T, Rt[:3,3] = t
        w2c[:3, 3] = T
        c2w = np.linalg.inv(w2c)
        c2w[:3, 1:3] *= -1                                    # back to the NeRF axes rasterizer_camera() starts from
        # pytorch3d's get_camera_center() of the stored (R, T) pair: C = -T_p3d R_p3d^-1 with R_p3d = R D, T_p3d = T D
        # (D = diag(-1, -1, 1)) = -T R^-1 -- equal to c2w[:3, 3] for a rigid `extr`, and what the reference uses otherwise
        cam = NerfCamera(c2w=c2w[:3, :], fx=self.fx, fy=self.fy, width=self.width, height=self.height, znear=self.znear,
                         zfar=self.zfar, principal_ndc=self.principal_ndc,
                         center_override=(-(T @ np.linalg.inv(R))).astype(np.float32))
        if len(self._cache) > 64:                             # (a pose optimised per iteration must not grow the cache for ever)
            for k_ in [k_ for k_ in self._cache if isinstance(k_, tuple) and k_[0] == "extr"]:
                del self._cache[k_]
        self._cache[key] = cam
        return cam

    def on_device(self, device):
        """(Camera, viewmatrix, full_proj, campos) with the three tensors resident on `device`, built once."""
        key = str(device)
        if key not in self._cache:
            cam = self.rasterizer_camera()
            t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)
            self._cache[key] = (cam, t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos))
        return self._cache[key]


def nerf_camera_from_scene(cam: scene.Camera, znear: float = 1e-4, zfar: float = 100.0) -> NerfCamera:
    """The NeRF-convention camera that produces `cam` (tests: scene.look_at_camera builds COLMAP-axes cameras)."""
    w2c = np.asarray(cam.viewmatrix, dtype=np.float64).T
    c2w = np.linalg.inv(w2c)
    c2w[:3, 1:3] *= -1
    fx = cam.W / (2.0 * cam.tanfovx)
    fy = cam.H / (2.0 * cam.tanfovy)
    return NerfCamera(c2w=c2w[:3, :], fx=fx, fy=fy, width=cam.W, height=cam.H, znear=znear, zfar=zfar)


class _PlainCtx:
    """Stand-in for an autograd context: lets SurfaceGaussians.rgbd_step run the Functions' forward / backward bodies directly."""
    __slots__ = ("needs_input_grad", "saved_tensors", "__dict__")

    def __init__(self, needs_input_grad):
        self.needs_input_grad, self.saved_tensors = needs_input_grad, ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


class _RenderMeshBound(torch.autograd.Function):
    """A render of mesh-bound Gaussians as ONE autograd node: the model's parameters in, the image out.  Forward = mesh
    producer -> colour producer (coefficients read from `_sh_coordinates_dc` / `_sh_coordinates_rest` where they live,
    sigmoid of the densities from the same kernel) -> rasterizer; backward = the three backward calls in reverse, the
    colour producer's gradient w.r.t. the positions added in place to the rasterizer's.  The same kernels on the same
    numbers as the composition of autograd nodes in render_image_gaussian_rasterizer -- what goes away is what autograd puts
    between them per iteration at config-C size: torch.cat of the coefficients and the two copies that split its gradient
    (53 MB each way), sigmoid and its backward, the add of the two position gradients, the zero fill of the screen-space
    gradient carrier, and five graph nodes' worth of host work (tools/window_phases.py)."""

    @staticmethod
    def forward(ctx, verts, raw_scales, raw_complex, densities, sh_dc, sh_rest, delta_t, delta_r, cfg):
        dev = verts.device
        f32 = lambda t: None if t is None else _rasterizer._dev_f32(t.detach(), dev)
        v, rs, rc, dens, dc, rest, dt, dr = (f32(t) for t in (verts, raw_scales, raw_complex, densities, sh_dc, sh_rest, delta_t, delta_r))
        st = cfg["settings"]
        D, M = cfg["sh_levels"] - 1, 1 + int(rest.size(1))
        view = st.viewmatrix if cfg["depth_channels"] else None
        # A sharded optimiser may have left the all-gather of some parameters in flight (dist.ShardedAdam(gather_first=...)):
        # each producer's inputs are fenced right before it is launched, so the mesh producer runs under the SH buckets' gather
        fence = getattr(cfg.get("sink"), "wait_params", None)
        p_verts, p_rs, p_rc, p_dens, p_dc, p_rest, p_dt, p_dr = cfg["params"]
        if fence is not None:
            fence((p_verts, p_rs, p_rc, p_dt, p_dr))
        need_bwd = any(ctx.needs_input_grad) and cfg.get("grad", True)
        # (without a gradient sink the backward's vertex-gradient accumulator is made here and cleared by the forward's launch)
        ctx.d_verts = torch.empty_like(v) if need_bwd and ctx.needs_input_grad[0] and cfg.get("sink") is None else None
        points, scaling, quats = producers._mesh_forward_raw(v, cfg["faces"], cfg["bary"], rs, rc, cfg["thickness"], cfg["lo"],
                                                             cfg["hi"], dt, dr, clear=ctx.d_verts)
        if fence is not None:
            fence((p_dc, p_rest, p_dens))
        colors, opac = producers._sh_forward_raw(points, st.campos, dc, rest, D, M, view, cfg["depth_channels"], dens)
        box = []
        out = _rasterizer.rasterize_gaussians_native(
            st.bg, points, colors, opac, scaling, quats, st.scale_modifier, None, st.viewmatrix, st.projmatrix, st.tanfovx,
            st.tanfovy, st.image_height, st.image_width, None, 0, st.campos, st.prefiltered, st.debug, need_backward=need_bwd,
            scratch_box=box)
        num_rendered, color, radii, geom, binning, img, _max_tile, num_segments = out
        ctx.save_for_backward(v, rs, rc, dr, dc, rest, points, scaling, quats, colors, opac, radii, geom, binning, img)
        ctx.cfg, ctx.counts = cfg, (num_rendered, num_segments, D, M, dt is not None, tuple(densities.shape))
        ctx.zeroed_scratch = box[0] if box else None
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, radii

    @staticmethod
    def backward(ctx, grad_color, _):
        if grad_color is None:
            return (None,) * 9
        v, rs, rc, dr, dc, rest, points, scaling, quats, colors, opac, radii, geom, binning, img = ctx.saved_tensors
        cfg, (num_rendered, num_segments, D, M, has_dt, dens_shape) = ctx.cfg, ctx.counts
        st = cfg["settings"]
        zeroed, ctx.zeroed_scratch = ctx.zeroed_scratch, None
        _dm2d, d_colors, d_opac, d_points, _dcov, _dsh, d_scaling, d_quats = _rasterizer.rasterize_gaussians_backward_native(
            st.bg, points, radii, colors, scaling, quats, st.scale_modifier, None, st.viewmatrix, st.projmatrix, st.tanfovx,
            st.tanfovy, grad_color, None, 0, st.campos, geom, num_rendered, binning, img, st.debug, num_segments=num_segments,
            zeroed_scratch=zeroed)
        view = st.viewmatrix if cfg["depth_channels"] else None
        # A gradient sink (dist.ShardedAdam: views of its flat gradient buffer) takes the parameter gradients where the
        # reduce-scatter reads them, and hears about the colour producer's -- 70 % of the bytes -- BEFORE the mesh producer's
        # backward is launched, so their buckets leave while that one still runs.  Only parameters without a gradient yet.
        sink, params = cfg.get("sink"), cfg["params"]
        views = sink.grad_views() if sink is not None else {}
        # (accepts(): first delivery of this step only -- a second render under the same loss goes through autograd's own addition)
        buf = lambda p: views.get(id(p)) if (p is not None and sink is not None and sink.accepts(p)) else None
        p_verts, p_rs, p_rc, p_dens, p_dc, p_rest, p_dt, p_dr = params
        o_dc, o_rest, o_dens = buf(p_dc), buf(p_rest), buf(p_dens)
        d_dc, d_rest, _, d_dens = producers._sh_backward_raw(points, _rasterizer._dev_f32(st.campos, points.device), dc, rest, D, M,
                                                             _rasterizer._dev_f32(view, points.device), cfg["depth_channels"], d_colors, opac, d_opac,
                                                             dpos_inout=d_points, out=(o_dc, o_rest if M > 1 else None, o_dens))
        if sink is not None:
            sink.written([p for p, o in ((p_dc, o_dc), (p_rest, o_rest if M > 1 else None), (p_dens, o_dens)) if o is not None])
        o_mesh = (buf(p_verts), buf(p_rs), buf(p_rc), buf(p_dt), buf(p_dr))
        pre, ctx.d_verts = getattr(ctx, "d_verts", None), None      # (cleared by the forward: good for one backward)
        cleared = o_mesh[0] is None and pre is not None
        d_verts, d_rs, d_rc, d_dt, d_dr = producers._mesh_backward_raw(v, cfg["faces"], cfg["bary"], rs, rc, dr, cfg["lo"], cfg["hi"],
                                                                       has_dt, d_points, d_scaling, d_quats,
                                                                       out=(pre,) + o_mesh[1:] if cleared else o_mesh, verts_cleared=cleared)
        if sink is not None:
            sink.written([p for p, o in zip((p_verts, p_rs, p_rc, p_dt, p_dr), o_mesh) if o is not None and p is not None])
        # (sink views go back to autograd as FRESH tensor objects: AccumulateGrad adopts an incoming gradient as p.grad only
        # if nobody else holds that tensor object -- the optimiser's cached views would make it clone all 77 MB instead)
        new = lambda t, o: t if (t is None or o is None) else t.view(t.shape)
        return (new(d_verts, o_mesh[0]), new(d_rs, o_mesh[1]), new(d_rc, o_mesh[2]), d_dens.view(dens_shape), new(d_dc, o_dc),
                new(d_rest, o_rest), new(d_dt, o_mesh[3]), new(d_dr, o_mesh[4]), None)


class SurfaceGaussians(nn.Module):
    """Gaussians bound to a triangle mesh: SuGaR with binded_to_surface_mesh = True (sugar_model.py:160-403), parameters
    under the reference's names so that a reference state dict loads with load_state_dict."""

    def __init__(self, verts: torch.Tensor, faces: torch.Tensor, n_gaussians_per_surface_triangle: int = 6,
                 sh_levels: int = 4, surface_mesh_thickness: float = 1e-6, min_gaussian_scale: Optional[float] = None,
                 max_gaussian_scale: Optional[float] = None, loose_bind: bool = False):
        super().__init__()
        G = int(n_gaussians_per_surface_triangle)
        if G not in BARY_COORDS:
            raise ValueError("n_gaussians_per_surface_triangle must be 1, 3, 4 or 6")
        F = int(faces.shape[0])
        N = F * G
        dev = verts.device
        self.n_gaussians_per_surface_triangle = G
        self.sh_levels = int(sh_levels)
        self.min_gaussian_scale, self.max_gaussian_scale = min_gaussian_scale, max_gaussian_scale
        self.return_one_densities = False
        self._points = nn.Parameter(verts.detach().clone().float())
        self.register_buffer("_surface_mesh_faces", faces.detach().long().contiguous().clone())   # (clone() keeps strides: the kernels read it as a dense [F, 3] int64 array)
        self.register_buffer("surface_triangle_bary_coords", torch.tensor(BARY_COORDS[G], dtype=torch.float32, device=dev)[..., None])
        self.register_buffer("surface_mesh_thickness", torch.tensor(float(surface_mesh_thickness), device=dev))
        # initial in-plane scale: the inscribed-circle radius of the face (sugar_model.py:216, :357)
        fv = self._points.detach()[self._surface_mesh_faces]
        edge = torch.stack([(fv[:, 0] - fv[:, 1]).norm(dim=-1), (fv[:, 1] - fv[:, 2]).norm(dim=-1), (fv[:, 2] - fv[:, 0]).norm(dim=-1)], -1)
        radius = {1: 1 / (2 * math.sqrt(3)), 3: 1 / (2 * (math.sqrt(3) + 1)), 4: 1 / (4 * math.sqrt(3)), 6: 1 / (4 + 2 * math.sqrt(3))}[G]
        init = (edge.min(dim=-1).values * radius).clamp(min=1e-8).log()
        self._scales = nn.Parameter(init[:, None, None].expand(F, G, 2).reshape(N, 2).contiguous())
        self._quaternions = nn.Parameter(torch.tensor([1.0, 0.0], device=dev).repeat(N, 1))          # 2-D rotation, identity
        self.all_densities = nn.Parameter(torch.full((N, 1), 2.2, device=dev))                      # sigmoid ~ 0.9
        self._sh_coordinates_dc = nn.Parameter(torch.zeros(N, 1, 3, device=dev))
        self._sh_coordinates_rest = nn.Parameter(torch.zeros(N, self.sh_levels ** 2 - 1, 3, device=dev))
        self._loose_bind = bool(loose_bind)
        if loose_bind:
            self._delta_t = nn.Parameter(torch.zeros(N, 3, device=dev))
            self._delta_r = nn.Parameter(torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev).repeat(N, 1))
        self._geom_cache = None
        # optional: an object with grad_views() / written(params) (gaustar_amd.dist.ShardedAdam) that receives the parameter
        # gradients of render_channels' backward in place -- see _RenderMeshBound.backward
        self.grad_sink = None

    def __getstate__(self):
        """copy.deepcopy / pickle of the model: the gradient sink (an optimiser with process-group handles and hooks on THIS
        model's parameters) and the per-object caches stay behind."""
        d = self.__dict__.copy()
        for k in ("grad_sink", "_geom_cache", "_thickness_cache", "_bary_rows_cache"):
            if k in d:
                d[k] = None
        return d

    # -------------------------------------------------------------------------------- construction from a checkpoint
    @classmethod
    def from_checkpoint(cls, ckpt: Dict, device, **kw) -> "SurfaceGaussians":
        """`ckpt` = formats.load_sugar_checkpoint(path)."""
        N, F = ckpt["raw_scales"].shape[0], ckpt["faces"].shape[0]
        m = cls(ckpt["verts"].to(device), ckpt["faces"].to(device), n_gaussians_per_surface_triangle=N // F,
                sh_levels=int(round(math.sqrt(ckpt["sh"].shape[1]))), loose_bind=ckpt.get("delta_r") is not None,
                surface_mesh_thickness=ckpt["thickness"] if ckpt.get("thickness") is not None else 1e-6, **kw)
        with torch.no_grad():
            m._scales.copy_(ckpt["raw_scales"]); m._quaternions.copy_(ckpt["raw_complex"])
            m.all_densities.copy_(ckpt["densities"].view(-1, 1))
            m._sh_coordinates_dc.copy_(ckpt["sh"][:, :1]); m._sh_coordinates_rest.copy_(ckpt["sh"][:, 1:])
            if m._loose_bind:
                m._delta_t.copy_(ckpt["delta_t"]); m._delta_r.copy_(ckpt["delta_r"])
        return m

    def grad_ready_order(self):
        """The optimiser's parameters (sugar_optimizer.py:67-87) in the order their gradients become final in the backward
        of a render: the rasterizer's backward feeds the SH producer's backward first (`_sh_coordinates_*`, 70 % of the
        bytes) and `all_densities` (one sigmoid), then the mesh producer's backward (`_points`, `_scales`,
        `_quaternions`, `_delta_*`).  dist.GradAllReducer buckets in this order and starts a bucket's all-reduce as soon
        as its last gradient lands."""
        ps = [self._sh_coordinates_rest, self._sh_coordinates_dc, self.all_densities, self._scales, self._quaternions]
        if self._loose_bind:
            ps += [self._delta_t, self._delta_r]
        return ps + [self._points]

    def mesh_parameters(self):
        """The parameters the mesh producer reads -- the FIRST thing a render needs (dist.ShardedAdam(gather_first=...): their
        all-gather is waited for in step(), the SH coefficients' runs under the next render's mesh producer)."""
        ps = [self._points, self._scales, self._quaternions]
        if self._loose_bind:
            ps += [self._delta_t, self._delta_r]
        return ps

    def _fence(self, *params) -> None:
        """Readers of parameters outside the fused render: if the optimiser left their all-gather in flight
        (dist.ShardedAdam(gather_first=...)), order this stream behind it.  No sink, nothing pending: one attribute look-up."""
        fence = getattr(self.grad_sink, "wait_params", None)
        if fence is not None:
            fence(params if params else None)

    def state_dict(self, *a, **kw):
        """nn.Module.state_dict behind a fence on every lazily gathered parameter (a checkpoint must not read torn shards)."""
        self._fence()
        return super().state_dict(*a, **kw)

    # -------------------------------------------------------------------------------- the reference's properties
    @property
    def device(self):
        return self._points.device

    @property
    def n_points(self) -> int:
        return int(self._scales.shape[0])

    def _thickness(self) -> float:
        """surface_mesh_thickness as a Python float, read back from the device ONCE per value: `float(buffer)` is a
        device-to-host copy that waits for everything queued before it -- as a per-render call it drained the GPU once per
        iteration (the host could never run ahead of the previous iteration's optimiser step)."""
        t = self.surface_mesh_thickness
        c = getattr(self, "_thickness_cache", None)
        if c is None or c[0] is not t or c[1] != t._version:
            c = (t, t._version, float(t))
            self._thickness_cache = c
        return c[2]

    def _geometry(self):
        """(points, scaling, quaternions) from one fused call, shared by the three properties while no parameter
        changes (the reference recomputes each property from scratch on every access)."""
        params = [self._points, self._scales, self._quaternions] + ([self._delta_t, self._delta_r] if self._loose_bind else [])
        self._fence(*params)
        key = (tuple(p._version for p in params), torch.is_grad_enabled())
        if self._geom_cache is None or self._geom_cache[0] != key:
            out = producers.mesh_bound_gaussians(
                self._points, self._surface_mesh_faces, self.surface_triangle_bary_coords[..., 0], self._scales,
                self._quaternions, self._thickness(), self.min_gaussian_scale, self.max_gaussian_scale,
                self._delta_t if self._loose_bind else None, self._delta_r if self._loose_bind else None)
            self._geom_cache = (key, out)
        return self._geom_cache[1]

    @property
    def points(self):                 # sugar_model.py:417-435
        return self._geometry()[0]

    @property
    def scaling(self):                # :457-476
        return self._geometry()[1]

    @property
    def quaternions(self):            # :478-508
        return self._geometry()[2]

    @property
    def strengths(self):              # :442-447
        self._fence(self.all_densities)
        if self.return_one_densities:
            return torch.ones_like(self.all_densities.view(-1, 1))
        return torch.sigmoid(self.all_densities.view(-1, 1))

    @property
    def sh_coordinates(self):         # :449-450
        self._fence(self._sh_coordinates_dc, self._sh_coordinates_rest)
        return torch.cat([self._sh_coordinates_dc, self._sh_coordinates_rest], dim=1)

    def get_points_rgb(self, positions=None, camera_centers=None, directions=None, sh_levels=None, sh_coordinates=None):
        """sugar_model.py:674-718: colours for one camera centre (fused HIP producer) or, when `camera_centers` is None, for
        the given `directions`, taken as they are (torch operations, producers.points_rgb_from_directions)."""
        sh = self.sh_coordinates if sh_coordinates is None else sh_coordinates
        levels = self.sh_levels if sh_levels is None else int(sh_levels)
        if camera_centers is not None:                                        # :698-699 (takes precedence, as in the reference)
            positions = self.points if positions is None else positions
            return producers.points_rgb(positions, camera_centers, sh, levels)
        if directions is not None:                                            # :700-701
            return producers.points_rgb_from_directions(directions, sh, levels)
        raise ValueError("Either camera_centers or directions must be provided.")   # :703

    # -------------------------------------------------------------------------------- rendering
    def _settings(self, camera: NerfCamera, bg: torch.Tensor, sh_degree: int):
        cam, view, proj, campos = camera.on_device(self.device)
        return GaussianRasterizationSettings(image_height=cam.H, image_width=cam.W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg,
                                             scale_modifier=1.0, viewmatrix=view, projmatrix=proj, sh_degree=int(sh_degree),
                                             campos=campos, prefiltered=False, debug=False), view, campos

    def _scales_for_render(self, use_solid_surface: bool, use_same_scale_in_all_directions: bool):
        scales = self.scaling
        if use_same_scale_in_all_directions:                                  # :1226-1228
            scales = scales.mean(dim=-1, keepdim=True).expand(-1, 3)
        if use_solid_surface:                                                 # :1230-1232
            scales = scales.clone()
            mean_scale = scales[..., 1:].mean()
            scales[..., 1:] = torch.maximum(mean_scale, scales[..., 1:])
        return scales

    def render_image_gaussian_rasterizer(self, camera: NerfCamera, bg_color=None, sh_deg: Optional[int] = None,
                                         sh_rotations=None, compute_color_in_rasterizer: bool = False,
                                         compute_covariance_in_rasterizer: bool = True, return_2d_radii: bool = False,
                                         quaternions=None, use_solid_surface: bool = False,
                                         use_same_scale_in_all_directions: bool = False, return_opacities: bool = False,
                                         return_colors: bool = False, positions=None, point_colors=None, overwrite_extr=None):
        """sugar_model.py:1065-1311 with `camera` in place of (nerf_cameras, camera_indices).  Returns the image [H,W,3]
        or, with return_2d_radii / return_opacities / return_colors, the reference's dict.  `overwrite_extr` (a [4,4]
        world-to-camera matrix, COLMAP axes; :1119-1127, :1141-1147) replaces the camera's pose and keeps its intrinsics;
        `sh_rotations` ([P,3,3]) turns every view direction before the SH evaluation (:1200-1205);
        `compute_covariance_in_rasterizer=False` hands the rasterizer R diag(s^2) R^T instead of scales + quaternions
        (:1237-1260)."""
        if overwrite_extr is not None:
            camera = camera.with_extrinsic(overwrite_extr)
        dev = self.device
        bg = torch.zeros(3, device=dev) if bg_color is None else torch.as_tensor(bg_color, dtype=torch.float32, device=dev)
        sh_deg = self.sh_levels - 1 if sh_deg is None else int(sh_deg)
        if (positions is None and point_colors is None and quaternions is None and not compute_color_in_rasterizer
                and not use_solid_surface and not use_same_scale_in_all_directions and sh_rotations is None
                and compute_covariance_in_rasterizer
                and not (return_2d_radii or return_opacities or return_colors)):
            # the plain call (what the refinement loop issues, refine.py:552): the whole render as one autograd node
            img, _ = self.render_channels(camera, bg, sh_deg=sh_deg, depth_channels=0)
            return img.transpose(0, 1).transpose(1, 2)                        # :1298
        settings, _view, campos = self._settings(camera, bg, sh_deg)
        positions = self.points if positions is None else positions
        shs = splat_colors = None
        if point_colors is None:
            if compute_color_in_rasterizer:
                shs = self.sh_coordinates                                     # :1208
            elif sh_rotations is None:
                splat_colors = self.get_points_rgb(positions=positions, camera_centers=campos, sh_levels=sh_deg + 1)
            else:                                                             # :1200-1205
                dirs = (torch.nn.functional.normalize(positions - campos.view(1, 3), dim=-1).unsqueeze(1) @ sh_rotations)[..., 0, :]
                splat_colors = self.get_points_rgb(positions=positions, camera_centers=None, directions=dirs, sh_levels=sh_deg + 1)
        else:
            splat_colors = point_colors                                       # :1211
        splat_opacities = self.strengths.view(-1, 1)
        quaternions = self.quaternions if quaternions is None else quaternions
        scales = self._scales_for_render(use_solid_surface, use_same_scale_in_all_directions)
        screenspace_points = torch.zeros(self.n_points, 3, dtype=positions.dtype, requires_grad=True, device=dev)
        if return_2d_radii:
            screenspace_points.retain_grad()
        cov3D = None
        if not compute_covariance_in_rasterizer:                              # :1237-1260
            cov3D = producers.covariance_3d(scales, quaternions)
            scales = quaternions = None
        rendered_image, radii = GaussianRasterizer(settings)(means3D=positions, means2D=screenspace_points, shs=shs,
                                                            colors_precomp=splat_colors, opacities=splat_opacities,
                                                            scales=scales, rotations=quaternions, cov3D_precomp=cov3D)
        image = rendered_image.transpose(0, 1).transpose(1, 2)                # :1298
        if not (return_2d_radii or return_opacities or return_colors):
            return image
        outputs = {"image": image, "radii": radii, "viewspace_points": screenspace_points}
        if return_opacities:
            outputs["opacities"] = splat_opacities
        if return_colors:
            outputs["colors"] = splat_colors
        return outputs

    def view_depth_colors(self, camera: NerfCamera, positions=None):
        """refine.py:603-605: view-space z of every Gaussian, expanded to three channels."""
        _cam, view, _proj, _campos = camera.on_device(self.device)
        positions = self.points if positions is None else positions
        return (positions @ view[:3, 2:3] + view[3, 2]).expand(-1, 3)

    def render_channels(self, camera: NerfCamera, bg: torch.Tensor, sh_deg: Optional[int] = None, depth_channels: int = 1):
        """-> (image [3 + depth_channels, H, W], radii): SH colours (+ view-space depth as `depth_channels` = 0, 1 or 3 more
        colour channels; bg has one entry per channel) rendered through ONE autograd node (_RenderMeshBound) -- the
        refinement loop's render.  Values and gradients are those of render_image_gaussian_rasterizer / the composition of
        producers.points_rgb_depth, torch.sigmoid and GaussianRasterizer (tests/test_gpu_harness.py)."""
        cfg = self._channels_cfg(camera, bg, sh_deg, depth_channels, torch.is_grad_enabled())
        return _RenderMeshBound.apply(self._points, self._scales, self._quaternions, self.all_densities, self._sh_coordinates_dc,
                                      self._sh_coordinates_rest, self._delta_t if self._loose_bind else None,
                                      self._delta_r if self._loose_bind else None, cfg)

    def _channels_cfg(self, camera: NerfCamera, bg: torch.Tensor, sh_deg: Optional[int], depth_channels: int, grad: bool):
        """What _RenderMeshBound needs besides the parameters (render_channels, rgbd_step)."""
        if depth_channels not in (0, 1, 3):
            raise ValueError("depth_channels must be 0, 1 or 3")
        sh_deg = self.sh_levels - 1 if sh_deg is None else int(sh_deg)
        settings, _view, _campos = self._settings(camera, bg, 0)
        producers._check_faces(self._surface_mesh_faces, int(self._points.shape[0]))
        if self._surface_mesh_faces.dtype != torch.int64 or not self._surface_mesh_faces.is_contiguous():
            # (the one-node render hands the buffer to the kernels as a raw pointer: a strided or int32 replacement of the
            # registered buffer would be read with the wrong layout)
            self._surface_mesh_faces = self._surface_mesh_faces.long().contiguous()
        dev = self.device
        cfg = {"settings": settings, "faces": self._surface_mesh_faces, "bary": self._bary_rows(), "depth_channels": int(depth_channels),
               "thickness": self._thickness(),
               "lo": float("-inf") if self.min_gaussian_scale is None else float(self.min_gaussian_scale),
               "hi": float("inf") if self.max_gaussian_scale is None else float(self.max_gaussian_scale), "sh_levels": sh_deg + 1,
               "sink": getattr(self, "grad_sink", None),
               "grad": bool(grad),   # (Function.forward cannot tell whether the caller runs under no_grad: see rasterizer._CALL)
               "params": (self._points, self._scales, self._quaternions, self.all_densities, self._sh_coordinates_dc,
                          self._sh_coordinates_rest, self._delta_t if self._loose_bind else None,
                          self._delta_r if self._loose_bind else None)}
        if settings.campos.device != dev or settings.campos.dtype != torch.float32:
            raise RuntimeError("camera matrices must be float32 tensors on the model's device")
        if cfg["sink"] is not None and not (hasattr(cfg["sink"], "grad_views") and hasattr(cfg["sink"], "written") and hasattr(cfg["sink"], "accepts")):
            raise TypeError("grad_sink must provide grad_views(), accepts(p) and written(params) (gaustar_amd.dist.ShardedAdam)")
        return cfg

    def rgbd_step(self, camera: NerfCamera, bg: torch.Tensor, gt_rgb: torch.Tensor, gt_depth: torch.Tensor, max_depth: float,
                  dssim_factor: float = 0.2, depth_factor: float = 1.0, mask_factor: float = 1.0, grad_scale: Optional[torch.Tensor] = None,
                  sh_deg: Optional[int] = None):
        """One refinement iteration's render + image losses + backward WITHOUT an autograd graph: the 4-channel render of
        render_channels(depth_channels=1), losses.rgb_depth_loss on it and both backward passes, run back to back through the very
        functions the autograd path runs (the two Functions' forward / backward bodies, called directly); the parameters' `.grad`
        are set (added to, if present) exactly as `loss.backward()` would.  -> (loss, image, radii), all detached.
        For loops whose only loss is this one: what it takes out is host time -- two Function.apply, the engine's thread
        hand-off and graph walk, eight AccumulateGrad nodes -- about a third of an iteration's Python at config-C size, which
        is what bounds the loop on a slow host (tools/window_phases.py).  grad_scale: a device scalar d(total)/d(loss), default 1."""
        from . import losses as _losses
        with torch.no_grad():
            cfg = self._channels_cfg(camera, bg, sh_deg, 1, True)
            params = (self._points, self._scales, self._quaternions, self.all_densities, self._sh_coordinates_dc, self._sh_coordinates_rest,
                      self._delta_t if self._loose_bind else None, self._delta_r if self._loose_bind else None)
            rctx = _PlainCtx(tuple(p is not None and p.requires_grad for p in params) + (False,))
            image, radii = _RenderMeshBound.forward(rctx, *params, cfg)
            lctx = _PlainCtx((True,) + (False,) * 7)
            loss, _parts = _losses._RGBDepthLoss.forward(lctx, image, gt_rgb, gt_depth, float(dssim_factor), None, float(max_depth),
                                                        float(depth_factor), float(mask_factor))
            if grad_scale is None:
                one = getattr(self, "_one_cache", None)
                if one is None or one.device != image.device:
                    one = self._one_cache = torch.ones((), device=image.device)
                grad_scale = one
            d_image = _losses._RGBDepthLoss.backward(lctx, grad_scale, None)[0]
            grads = _RenderMeshBound.backward(rctx, d_image, None)
            for p, g in zip(params, grads):
                if p is None or g is None or not p.requires_grad:
                    continue
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
        return loss, image, radii

    def _bary_rows(self):
        b = getattr(self, "_bary_rows_cache", None)
        if b is None or b.device != self.device:
            b = self.surface_triangle_bary_coords[..., 0].detach().to(torch.float32).contiguous()
            self._bary_rows_cache = b
        return b

    def render_rgb_depth(self, camera: NerfCamera, bg_color=None, max_depth: float = 10.0, sh_deg: Optional[int] = None):
        """The two renders of a refinement iteration (refine.py:552 RGB, :607 depth-as-colour with bg = max_depth) as ONE
        4-channel pass (RGB + one depth channel) through one autograd node (render_channels):
        -> (rgb [H,W,3], depth [H,W])."""
        dev = self.device
        bg_rgb = torch.zeros(3, device=dev) if bg_color is None else torch.as_tensor(bg_color, dtype=torch.float32, device=dev)
        bg4 = torch.cat([bg_rgb, torch.full((1,), float(max_depth), device=dev)])   # RGB + one depth channel
        img, _ = self.render_channels(camera, bg4, sh_deg=sh_deg, depth_channels=1)
        return img[:3].permute(1, 2, 0), img[3]

"""Synthetic scenes for the rasterizer hot path: cameras, meshes and mesh-bound
surface Gaussians, built exactly the way the reference's caller builds the
rasterizer arguments -- so tests and bench.py exercise the path with the inputs
`SuGaR.render_image_gaussian_rasterizer` would hand it.

Everything here is deterministic numpy (float32 outputs); no torch, no GPU.

Reference formulas restated (paths relative to /root/reference):
  * view / projection matrices: gaustar_utils/graphics_utils.py:38-85 and
    gaustar_scene/sugar_model.py:1129-1163 (matrices are handed over TRANSPOSED,
    `projmatrix` is the full view*proj product).
  * 6 Gaussians per face at fixed barycentric coordinates, in-plane radius
    min-edge / (4 + 2*sqrt(3)): gaustar_scene/sugar_model.py:213-226, :355-357.
  * scales = [thickness, s, s], rotation columns = [normal, R1, R2], quaternion
    (w,x,y,z) normalised: gaustar_scene/sugar_model.py:457-508.
  * colours from SH: clamp_min(eval_sh + 0.5, 0): gaustar_scene/sugar_model.py:714-716,
    gaustar_utils/spherical_harmonics.py:117-172.
The configs A-D are SURVEY.md section 8(d)'s.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

F32 = np.float32

# gaustar_scene/sugar_model.py:217-224
BARY6 = np.array(
    [[2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3],
     [1 / 6, 5 / 12, 5 / 12], [5 / 12, 1 / 6, 5 / 12], [5 / 12, 5 / 12, 1 / 6]], dtype=F32)
CIRCLE_RADIUS6 = 1.0 / (4.0 + 2.0 * math.sqrt(3.0))  # sugar_model.py:216

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


# --------------------------------------------------------------------------- cameras
@dataclass
class Camera:
    """What `GaussianRasterizationSettings` needs from one camera."""
    W: int
    H: int
    tanfovx: float
    tanfovy: float
    viewmatrix: np.ndarray   # [4,4] float32, TRANSPOSED world->view (row-vector convention)
    projmatrix: np.ndarray   # [4,4] float32, TRANSPOSED full projection (view * proj)
    campos: np.ndarray       # [3]
    name: str = ""           # img_name of cameras.json
    uid: int = 0


def focal2fov(focal: float, pixels: float) -> float:
    return 2.0 * math.atan(pixels / (2.0 * focal))


def get_world2view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """graphics_utils.py:38-51 -- R is the camera-to-world rotation ("stored transposed")."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return np.float32(Rt)


def get_projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> np.ndarray:
    """graphics_utils.py:66-85 (float32 like the torch.zeros(4,4) it fills)."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = np.zeros((4, 4), dtype=F32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_from_RT(R: np.ndarray, T: np.ndarray, W: int, H: int, fovx: float, fovy: float, znear: float = 0.01,
                   zfar: float = 100.0) -> Camera:
    """The matrices a GSCamera carries (gaustar_scene/cameras.py:206-220) from a 3DGS-convention pose: R = camera-to-world
    rotation ("stored transposed"), T = world-to-camera translation.  world_view = getWorld2View2(R, T)^T,
    full_proj = world_view @ getProjectionMatrix(...)^T, camera centre = inverse(world_view)[3, :3]."""
    view_t = get_world2view(np.asarray(R, dtype=np.float64), np.asarray(T, dtype=np.float64)).transpose()
    proj_t = get_projection_matrix(znear, zfar, fovx, fovy).transpose()
    full_t = (view_t @ proj_t).astype(F32)
    campos = np.linalg.inv(view_t.astype(np.float64))[3, :3]
    return Camera(W=int(W), H=int(H), tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5),
                  viewmatrix=np.ascontiguousarray(view_t, dtype=F32), projmatrix=np.ascontiguousarray(full_t),
                  campos=campos.astype(F32))


def look_at_camera(eye, target, W: int, H: int, focal_px: Optional[float] = None, fovx: Optional[float] = None,
                   up=(0.0, 1.0, 0.0), znear: float = 1e-4, zfar: float = 100.0) -> Camera:
    """COLMAP-style camera (x right, y down, z forward) looking from `eye` at `target`;
    matrices assembled as sugar_model.py:1138-1163 does (znear/zfar are pytorch3d's
    defaults there)."""
    eye = np.asarray(eye, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    z = target - eye
    z /= np.linalg.norm(z)
    upv = np.asarray(up, dtype=np.float64)
    x = np.cross(z, upv)          # right-handed with y pointing down
    if np.linalg.norm(x) < 1e-8:
        x = np.cross(z, np.array([1.0, 0.0, 0.0]))
    x /= np.linalg.norm(x)
    y = np.cross(z, x)            # down
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, eye
    w2c = np.linalg.inv(c2w)
    R = np.transpose(w2c[:3, :3])
    T = w2c[:3, 3]
    if focal_px is None:
        focal_px = W / (2.0 * math.tan(fovx / 2.0))
    fovx_ = focal2fov(focal_px, W)
    fovy_ = focal2fov(focal_px, H)
    view_t = get_world2view(R, T).transpose()                      # TRANSPOSED, float32
    proj_t = get_projection_matrix(znear, zfar, fovx_, fovy_).transpose()
    full_t = (view_t @ proj_t).astype(F32)
    return Camera(W=W, H=H, tanfovx=math.tan(fovx_ * 0.5), tanfovy=math.tan(fovy_ * 0.5),
                  viewmatrix=np.ascontiguousarray(view_t, dtype=F32), projmatrix=np.ascontiguousarray(full_t),
                  campos=eye.astype(F32))


def ring_cameras(n_rings: int = 5, n_azim: int = 32, W: int = 1920, H: int = 1080, focal_px: float = 1200.0,
                 center=(0.0, 1.2, 0.0), r_min: float = 3.0, r_max: float = 4.0,
                 elev_min: float = -30.0, elev_max: float = 45.0) -> list:
    """Config C's 160-camera rig: 5 rings x 32 azimuths around the subject."""
    cams = []
    c = np.asarray(center, dtype=np.float64)
    for ri in range(n_rings):
        elev = math.radians(elev_min + (elev_max - elev_min) * ri / max(1, n_rings - 1))
        for ai in range(n_azim):
            az = 2.0 * math.pi * (ai + 0.5 * (ri % 2)) / n_azim
            rad = r_min + (r_max - r_min) * ((ai * 7 + ri * 3) % 11) / 10.0
            eye = c + rad * np.array([math.cos(elev) * math.sin(az), math.sin(elev), math.cos(elev) * math.cos(az)])
            cams.append(look_at_camera(eye, c, W, H, focal_px=focal_px))
    return cams


# --------------------------------------------------------------------------- meshes
def uv_sphere(n_lon: int, n_lat: int, radius: float = 1.0, center=(0.0, 0.0, 0.0)):
    """UV sphere with 2*n_lon*(n_lat-1) triangles (n_lat latitude bands, triangle fans at the poles)."""
    verts = [[0.0, 1.0, 0.0]]
    for i in range(1, n_lat):
        th = math.pi * i / n_lat
        for j in range(n_lon):
            ph = 2.0 * math.pi * j / n_lon
            verts.append([math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)])
    verts.append([0.0, -1.0, 0.0])
    verts = np.asarray(verts, dtype=np.float64)
    j = np.arange(n_lon)
    jn = (j + 1) % n_lon
    faces = [np.stack([np.zeros_like(j), 1 + jn, 1 + j], axis=1)]
    for i in range(n_lat - 2):
        a = 1 + i * n_lon + j
        b = 1 + i * n_lon + jn
        c = 1 + (i + 1) * n_lon + j
        d = 1 + (i + 1) * n_lon + jn
        faces.append(np.stack([a, b, d], axis=1))
        faces.append(np.stack([a, d, c], axis=1))
    last = len(verts) - 1
    base = 1 + (n_lat - 2) * n_lon
    faces.append(np.stack([np.full_like(j, last), base + j, base + jn], axis=1))
    faces = np.concatenate(faces, axis=0).astype(np.int64)
    verts = verts * radius + np.asarray(center, dtype=np.float64)
    return verts.astype(F32), faces


def icosphere(level: int, radius: float = 1.0, center=(0.0, 0.0, 0.0)):
    """Icosahedron subdivided `level` times: 20 * 4**level faces."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        e_sorted = np.sort(e, axis=1)
        uniq, inv = np.unique(e_sorted, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        nv = len(v)
        v = np.concatenate([v, mid], axis=0)
        nf = len(f)
        m01, m12, m20 = nv + inv[:nf], nv + inv[nf:2 * nf], nv + inv[2 * nf:]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1),
                            np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)], axis=0)
    v = v * radius + np.asarray(center, dtype=np.float64)
    return v.astype(F32), f


# --------------------------------------------------------------------------- Gaussians
def matrix_to_quaternion(R: np.ndarray) -> np.ndarray:
    """Rotation matrices [N,3,3] -> quaternions (w,x,y,z).  Branch/sign choice is irrelevant
    downstream (R(q) == R(-q) and the caller normalises, sugar_model.py:508)."""
    m00, m01, m02 = R[:, 0, 0], R[:, 0, 1], R[:, 0, 2]
    m10, m11, m12 = R[:, 1, 0], R[:, 1, 1], R[:, 1, 2]
    m20, m21, m22 = R[:, 2, 0], R[:, 2, 1], R[:, 2, 2]
    q_abs = np.sqrt(np.maximum(0.0, np.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22,
                                               1 - m00 + m11 - m22, 1 - m00 - m11 + m22], axis=1)))
    cand = np.stack([
        np.stack([q_abs[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], axis=1),
        np.stack([m21 - m12, q_abs[:, 1] ** 2, m10 + m01, m02 + m20], axis=1),
        np.stack([m02 - m20, m10 + m01, q_abs[:, 2] ** 2, m12 + m21], axis=1),
        np.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[:, 3] ** 2], axis=1)], axis=1)
    cand = cand / (2.0 * np.maximum(q_abs[:, :, None], 0.1))
    best = np.argmax(q_abs, axis=1)
    return cand[np.arange(len(R)), best]


def eval_sh_rgb(deg: int, sh: np.ndarray, dirs: np.ndarray) -> np.ndarray:
    """clamp_min(eval_sh(deg, sh, dirs) + 0.5, 0) with sh [N,M,3], unit dirs [N,3]
    (spherical_harmonics.py:134-160; sugar_model.py:714-716)."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return np.maximum(res + 0.5, 0.0).astype(F32)


@dataclass
class GaussianSet:
    """Arguments of `GaussianRasterizer.forward` for one scene."""
    means3D: np.ndarray                      # [P,3]
    opacities: np.ndarray                    # [P,1]
    scales: Optional[np.ndarray] = None      # [P,3]
    rotations: Optional[np.ndarray] = None   # [P,4] (w,x,y,z), normalised by the caller
    colors_precomp: Optional[np.ndarray] = None  # [P,3]
    shs: Optional[np.ndarray] = None         # [P,M,3]
    sh_degree: int = 0
    cov3D_precomp: Optional[np.ndarray] = None
    meta: dict = field(default_factory=dict)

    @property
    def P(self) -> int:
        return int(self.means3D.shape[0])


def mesh_bound_gaussians(verts: np.ndarray, faces: np.ndarray, rng: np.random.Generator, thickness: float,
                         opacity_range=(0.8, 0.99), scale_clamp=(0.1, 5.0)) -> GaussianSet:
    """Surface Gaussians bound to a triangle mesh the way SuGaR binds them (6 per face)."""
    fv = verts[faces].astype(np.float64)                       # [F,3,3]
    means = np.einsum("fvc,gv->fgc", fv, BARY6.astype(np.float64)).reshape(-1, 3)   # sugar_model.py:422-431
    edges = np.linalg.norm(fv - fv[:, [1, 2, 0]], axis=-1)     # [F,3]
    s = np.maximum(edges.min(axis=1) * CIRCLE_RADIUS6, 1e-7)   # sugar_model.py:357-358
    mean_edge = edges.mean()
    s = np.clip(s, scale_clamp[0] * mean_edge * CIRCLE_RADIUS6, scale_clamp[1] * mean_edge)   # refine.py:305-311
    plane = np.repeat(s[:, None], 6, axis=1).reshape(-1)
    scales = np.stack([np.full_like(plane, thickness), plane, plane], axis=1)        # sugar_model.py:472-475
    n = np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    r1 = fv[:, 0] - fv[:, 1]
    r1 /= np.linalg.norm(r1, axis=1, keepdims=True)
    r2 = np.cross(n, r1)
    r2 /= np.linalg.norm(r2, axis=1, keepdims=True)
    R = np.stack([n, r1, r2], axis=-1)                         # columns [normal, R1, R2]; complex number = (1,0)
    q = matrix_to_quaternion(R)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    quats = np.repeat(q[:, None], 6, axis=1).reshape(-1, 4)
    P = means.shape[0]
    opac = rng.uniform(opacity_range[0], opacity_range[1], size=(P, 1))
    cols = rng.uniform(0.0, 1.0, size=(P, 3))
    return GaussianSet(means3D=means.astype(F32), opacities=opac.astype(F32), scales=scales.astype(F32),
                       rotations=quats.astype(F32), colors_precomp=cols.astype(F32),
                       meta=dict(n_faces=int(len(faces)), n_verts=int(len(verts))))


def random_gaussians(P: int, rng: np.random.Generator, sh_degree: int = 0, with_sh: bool = False,
                     box=((-1.5, 1.5), (-1.0, 1.0), (-0.5, 0.5)), scale_range=(0.01, 0.1)) -> GaussianSet:
    """Config A-style free Gaussians."""
    means = np.stack([rng.uniform(lo, hi, size=P) for lo, hi in box], axis=1)
    scales = np.exp(rng.uniform(math.log(scale_range[0]), math.log(scale_range[1]), size=(P, 3)))
    q = rng.normal(size=(P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(0.05, 0.95, size=(P, 1))
    M = (sh_degree + 1) ** 2
    sh = rng.uniform(-0.3, 0.3, size=(P, M, 3))
    sh[:, 0] = rng.uniform(-1.5, 1.5, size=(P, 3))
    gs = GaussianSet(means3D=means.astype(F32), opacities=opac.astype(F32), scales=scales.astype(F32),
                     rotations=q.astype(F32), sh_degree=sh_degree)
    if with_sh:
        gs.shs = sh.astype(F32)
    else:
        gs.colors_precomp = np.maximum(SH_C0 * sh[:, 0] + 0.5, 0.0).astype(F32)
    return gs


# --------------------------------------------------------------------------- configs
SUBJECT_CENTER = (0.0, 1.2, 0.0)
SUBJECT_RADIUS = 0.9
FOCAL_1080P = 1200.0      # sphere spans ~70 % of the image height from 3 m


def _extent_thickness(cam_radius: float = 3.5) -> float:
    # surface_mesh_thickness = camera spatial extent / 1e6  (sugar_model.py:179-180)
    return cam_radius / 1_000_000.0


def config_A(seed: int = 0):
    """10k random Gaussians, 1 cam @512x512, SH deg 0."""
    rng = np.random.default_rng(seed)
    gs = random_gaussians(10_000, rng, sh_degree=0, with_sh=False)
    cam = look_at_camera((0.0, 0.0, -4.0), (0.0, 0.0, 0.0), 512, 512, fovx=0.9, znear=0.01)
    return gs, cam, np.array([0.0, 1.0, 0.0], dtype=F32)


def config_B(seed: int = 0):
    """200 400 mesh-bound Gaussians (UV sphere 167x101), 1 cam @1080p."""
    rng = np.random.default_rng(seed)
    v, f = uv_sphere(167, 101, SUBJECT_RADIUS, SUBJECT_CENTER)
    gs = mesh_bound_gaussians(v, f, rng, _extent_thickness())
    cam = look_at_camera((0.0, 1.2, 3.0), SUBJECT_CENTER, 1920, 1080, focal_px=FOCAL_1080P)
    return gs, cam, np.array([0.0, 1.0, 0.0], dtype=F32)


def config_C(seed: int = 0, level: int = 6):
    """491 520 mesh-bound Gaussians (icosphere level 6), 160-camera rig @1080p."""
    rng = np.random.default_rng(seed)
    v, f = icosphere(level, SUBJECT_RADIUS, SUBJECT_CENTER)
    gs = mesh_bound_gaussians(v, f, rng, _extent_thickness())
    cams = ring_cameras()
    return gs, cams, np.array([0.0, 1.0, 0.0], dtype=F32)


def config_D(seed: int = 0):
    """1 001 232 mesh-bound Gaussians (UV sphere 409x205), SH deg 3 in-kernel + depth-as-colour pass."""
    rng = np.random.default_rng(seed)
    v, f = uv_sphere(409, 205, SUBJECT_RADIUS, SUBJECT_CENTER)
    gs = mesh_bound_gaussians(v, f, rng, _extent_thickness())
    P = gs.P
    sh = rng.uniform(-0.3, 0.3, size=(P, 16, 3))
    sh[:, 0] = rng.uniform(-1.5, 1.5, size=(P, 3))
    gs.shs = sh.astype(F32)
    gs.colors_precomp = None
    gs.sh_degree = 3
    cam = look_at_camera((0.0, 1.2, 3.0), SUBJECT_CENTER, 1920, 1080, focal_px=FOCAL_1080P)
    return gs, cam, np.array([0.0, 1.0, 0.0], dtype=F32)


def view_depth_colors(gs: GaussianSet, cam: Camera) -> np.ndarray:
    """Depth-as-colour pass of refine.py:603-607: colours = view-space z expanded to 3 channels."""
    z = gs.means3D @ cam.viewmatrix[:3, 2] + cam.viewmatrix[3, 2]
    return np.repeat(z[:, None], 3, axis=1).astype(F32)

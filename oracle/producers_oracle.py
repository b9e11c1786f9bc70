"""CPU restatement of the producers of rasterizer inputs (TEST INFRASTRUCTURE ONLY -- never imported by gaustar_amd/).

points_rgb follows SuGaR.get_points_rgb (gaustar_scene/sugar_model.py:698-716) over eval_sh
(gaustar_utils/spherical_harmonics.py:117-172), in plain PyTorch; gradients come from autograd.
Pinned by tests/golden/producers_kat.npz, produced by tests/golden/make_producers_golden.py with the reference's
own eval_sh."""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


C4 = [2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431,
      -0.6690465435572892, 0.47308734787878004, -1.7701307697799304, 0.6258357354491761]


def eval_sh(deg, sh, dirs):                                          # spherical_harmonics.py:117-172 (deg <= 4)
    result = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + C2[0] * xy * sh[..., 4] + C2[1] * yz * sh[..., 5] + C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] +
                      C2[3] * xz * sh[..., 7] + C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + C3[0] * y * (3 * xx - yy) * sh[..., 9] + C3[1] * xy * z * sh[..., 10] +
                          C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12] +
                          C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + C3[5] * z * (xx - yy) * sh[..., 14] +
                          C3[6] * x * (xx - 3 * yy) * sh[..., 15])
                if deg > 3:                                          # :162-171
                    result = (result + C4[0] * xy * (xx - yy) * sh[..., 16] + C4[1] * yz * (3 * xx - yy) * sh[..., 17] +
                              C4[2] * xy * (7 * zz - 1) * sh[..., 18] + C4[3] * yz * (7 * zz - 3) * sh[..., 19] +
                              C4[4] * (zz * (35 * zz - 30) + 3) * sh[..., 20] + C4[5] * xz * (7 * zz - 3) * sh[..., 21] +
                              C4[6] * (xx - yy) * (7 * zz - 1) * sh[..., 22] + C4[7] * xz * (xx - 3 * yy) * sh[..., 23] +
                              C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy)) * sh[..., 24])
    return result


def points_rgb(positions, camera_centers, sh_coordinates, sh_levels):
    dirs = torch.nn.functional.normalize(positions - camera_centers, dim=-1)          # sugar_model.py:700
    sh = sh_coordinates[:, :sh_levels ** 2]                                           # :711
    shs_view = sh.transpose(-1, -2).view(-1, 3, sh_levels ** 2)                       # :713
    return torch.clamp_min(eval_sh(sh_levels - 1, shs_view, dirs) + 0.5, 0.0).view(-1, 3)   # :714-716


# --------------------------------------------------------------------------------------------------------------
# Mesh-bound Gaussians: SuGaR.points / .scaling / .quaternions (gaustar_scene/sugar_model.py:417-435, :457-476,
# :478-508).  Two third-party pieces are absent from /root/reference and restated from their published algorithms
# (pytorch3d 0.7.4, pinned in the reference's environment.yml:161):
#   * Meshes.faces_normals_list(): n = (v1 - v0) x (v2 - v0), divided by max(|n|, 1e-6)
#     (pytorch3d/csrc/face_areas_normals);
#   * transforms.quaternion_to_matrix / matrix_to_quaternion (pytorch3d/transforms/rotation_conversions.py):
#     real-first quaternions; matrix_to_quaternion picks the best-conditioned of the four candidates.
# sugar_model.py itself cannot be imported here (open3d, pytorch3d), so this part of the oracle is pinned by
# closed-form properties in tests/ (the rotation's first axis is the face normal, means lie in the face plane,
# R(q) is orthonormal, finite differences) rather than by reference outputs: PARITY UNPINNED for these functions.
def face_normals(verts, faces):
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    n = torch.cross(v1 - v0, v2 - v0, dim=-1)
    return n / n.norm(dim=-1, keepdim=True).clamp(min=1e-6)


def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x):
    ret = torch.zeros_like(x)
    m = x > 0
    ret[m] = torch.sqrt(x[m])
    return ret


def matrix_to_quaternion(matrix):
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22,
                                             1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    return quat_candidates[torch.nn.functional.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :].reshape(batch_dim + (4,))


def mesh_bound_gaussians(verts, faces, bary, raw_scales, raw_complex, thickness, min_scale=None, max_scale=None,
                         delta_t=None, delta_r=None):
    """-> (points [N,3], scaling [N,3], quaternions [N,4]), N = F * G, Gaussian n = f * G + g.
    verts [V,3], faces [F,3] long, bary [G,3], raw_scales [N,2] (log), raw_complex [N,2]."""
    F_, G = faces.shape[0], bary.shape[0]
    fv = verts[faces]                                                        # sugar_model.py:425
    points = (fv[:, None] * bary[None, :, :, None]).sum(dim=-2).reshape(F_ * G, 3)   # :428-429
    if delta_t is not None:
        points = points + delta_t                                            # :432
    plane = torch.exp(raw_scales)                                            # :461 (scale_activation = exp)
    if max_scale is not None:
        plane = torch.clamp_max(plane, max_scale)                            # :463
    if min_scale is not None:
        plane = torch.clamp_min(plane, min_scale)                            # :465
    scaling = torch.cat([thickness * torch.ones(len(raw_scales), 1, dtype=plane.dtype, device=plane.device), plane], dim=-1)   # :472-475
    nf = torch.nn.functional.normalize
    R_0 = nf(face_normals(verts, faces), dim=-1)                             # :483
    base_R_1 = nf(fv[:, 0] - fv[:, 1], dim=-1)                               # :487
    base_R_2 = nf(torch.cross(R_0, base_R_1, dim=-1))                        # :490
    cplx = nf(raw_complex, dim=-1).view(F_, G, 2)                            # :493
    R_1 = cplx[..., 0:1] * base_R_1[:, None] + cplx[..., 1:2] * base_R_2[:, None]    # :494
    R_2 = -cplx[..., 1:2] * base_R_1[:, None] + cplx[..., 0:1] * base_R_2[:, None]   # :495
    R = torch.cat([R_0[:, None, ..., None].expand(-1, G, -1, -1).clone(), R_1[..., None], R_2[..., None]],
                  dim=-1).view(-1, 3, 3)                                     # :498-502
    if delta_r is not None:
        R = torch.bmm(quaternion_to_matrix(delta_r), R)                      # :504-505
    return points, scaling, nf(matrix_to_quaternion(R), dim=-1)              # :506-508

"""Independent float64 restatement of the mesh-bound Gaussian frames (sugar_model.py:417-435, :457-476, :478-508) with
numpy + scipy.spatial.transform.Rotation -- a CROSS-CHECK of oracle/producers_oracle.py and of the HIP producer, NOT a
pin: pytorch3d (whose matrix_to_quaternion / face normals the reference calls) is absent from this image, so no output
of the reference itself exists for these functions.  What it shares with the oracle is the reading of sugar_model.py;
what it does not share is any code: different library, different precision, different matrix->quaternion algorithm."""
import numpy as np


def mesh_frames_f64(verts, faces, bary, raw_scales, raw_complex, thickness, delta_t=None, delta_r=None):
    """-> points [N,3], scaling [N,3], R [N,3,3] (float64), N = F*G."""
    from scipy.spatial.transform import Rotation
    v = np.asarray(verts, np.float64); f = np.asarray(faces, np.int64); b = np.asarray(bary, np.float64)
    F, G = f.shape[0], b.shape[0]
    fv = v[f]                                                         # [F,3,3]
    pts = np.einsum("gk,fkd->fgd", b, fv).reshape(F * G, 3)
    if delta_t is not None:
        pts = pts + np.asarray(delta_t, np.float64)
    scaling = np.concatenate([np.full((F * G, 1), float(thickness)), np.exp(np.asarray(raw_scales, np.float64))], 1)
    unit = lambda x: x / np.linalg.norm(x, axis=-1, keepdims=True)
    n = unit(np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0]))      # face normal
    e1 = unit(fv[:, 0] - fv[:, 1])
    e2 = unit(np.cross(n, e1))
    c = unit(np.asarray(raw_complex, np.float64)).reshape(F, G, 2)
    r1 = c[..., 0:1] * e1[:, None] + c[..., 1:2] * e2[:, None]
    r2 = -c[..., 1:2] * e1[:, None] + c[..., 0:1] * e2[:, None]
    R = np.stack([np.broadcast_to(n[:, None], r1.shape), r1, r2], axis=-1).reshape(F * G, 3, 3)   # columns n, r1, r2
    if delta_r is not None:
        q = np.asarray(delta_r, np.float64)                           # (w, x, y, z) -> scipy's (x, y, z, w)
        R = Rotation.from_quat(q[:, [1, 2, 3, 0]]).as_matrix() @ R
    return pts, scaling, R


def quat_wxyz_to_matrix(q):
    from scipy.spatial.transform import Rotation
    q = np.asarray(q, np.float64)
    return Rotation.from_quat(q[:, [1, 2, 3, 0]]).as_matrix()


def quats_from_matrices_scipy(R):
    """Rotation.from_matrix -> (w, x, y, z), an algorithm unrelated to pytorch3d's best-conditioned-candidate rule."""
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R).as_quat()
    return q[:, [3, 0, 1, 2]]

"""The synthetic harness (gaustar_amd/scene.py) against (a) known-answer vectors produced by
importing the reference's own Python helpers (tests/golden/utils_kat.npz, make_utils_golden.py) and
(b) closed-form properties of SuGaR's mesh binding.  CPU only."""
import os

import numpy as np

from conftest import GOLDEN_DIR
from gaustar_amd import scene

KAT = np.load(os.path.join(GOLDEN_DIR, "utils_kat.npz"))


def test_projection_and_view_matrices_match_reference_helpers():
    for args, ref in zip(KAT["proj_args"], KAT["proj_out"]):
        np.testing.assert_allclose(scene.get_projection_matrix(*args), ref, rtol=1e-6, atol=1e-7)
    for (f, p), ref in zip(KAT["focal_in"], KAT["focal2fov_out"]):
        assert abs(scene.focal2fov(f, p) - ref) < 1e-12
    np.testing.assert_array_equal(scene.get_world2view(KAT["w2v_R"], KAT["w2v_t"]), KAT["w2v_out"])


def test_sh_colours_match_reference_eval_sh():
    for deg in range(4):
        m = (deg + 1) ** 2
        got = scene.eval_sh_rgb(deg, KAT["sh_coeffs"][:, :m].astype(np.float64), KAT["sh_dirs"].astype(np.float64))
        np.testing.assert_allclose(got, KAT[f"sh_rgb_deg{deg}"], rtol=0, atol=2e-6)


def test_camera_conventions():
    cam = scene.look_at_camera((0.3, 1.0, 3.0), (0.0, 1.2, 0.0), 640, 360, focal_px=400.0)
    V = cam.viewmatrix            # transposed world->view: p_view = [p,1] @ V
    c = np.append(cam.campos, 1.0) @ V
    np.testing.assert_allclose(c[:3], 0, atol=1e-6)                    # camera centre maps to the origin
    tgt = np.array([0.0, 1.2, 0.0, 1.0]) @ V
    assert tgt[2] > 0 and abs(tgt[0]) < 1e-5 and abs(tgt[1]) < 1e-5    # target on the +z axis
    up = np.array([0.0, 2.2, 0.0, 1.0]) @ V
    assert up[1] < tgt[1]                                              # y points down in view space
    h = np.array([0.0, 1.2, 0.0, 1.0]) @ cam.projmatrix                # full projection: target -> NDC (0,0)
    np.testing.assert_allclose(h[:2] / h[3], 0, atol=1e-5)
    assert abs(cam.tanfovx - 320 / 400) < 1e-9 and abs(cam.tanfovy - 180 / 400) < 1e-9
    assert len(scene.ring_cameras()) == 160


def test_mesh_sizes_of_the_configs():
    v, f = scene.uv_sphere(167, 101)
    assert len(f) == 33_400 and 6 * len(f) == 200_400
    v, f = scene.icosphere(3)
    assert len(f) == 20 * 4 ** 3 and len(v) == 10 * 4 ** 3 + 2
    assert 20 * 4 ** 6 * 6 == 491_520
    assert 2 * 409 * 204 * 6 == 1_001_232
    # closed, consistently wound: every edge appears once in each direction
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    fw = set(map(tuple, e))
    assert all((b, a) in fw for a, b in fw)


def test_mesh_binding_properties():
    rng = np.random.default_rng(0)
    v, f = scene.icosphere(2, 0.9, (0, 1.2, 0))
    gs = scene.mesh_bound_gaussians(v, f, rng, 3.5e-6)
    assert gs.P == 6 * len(f)
    fv = v[f].astype(np.float64)
    n = np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n6 = np.repeat(n, 6, axis=0)
    # barycentric means lie in the face plane
    d = np.einsum("pc,pc->p", gs.means3D.astype(np.float64) - np.repeat(fv[:, 0], 6, axis=0), n6)
    assert np.abs(d).max() < 1e-6
    # first column of R(q) (the thin axis) is the face normal; q is unit length
    q = gs.rotations.astype(np.float64)
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)
    r, x, y, z = q.T
    col0 = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + r * z), 2 * (x * z - r * y)], axis=1)
    np.testing.assert_allclose(col0, n6, atol=1e-5)
    assert np.allclose(gs.scales[:, 0], 3.5e-6) and (gs.scales[:, 1] == gs.scales[:, 2]).all()
    assert gs.opacities.min() >= 0.8 and gs.opacities.max() <= 0.99


def test_mesh_bound_producer_oracle_against_scipy_float64():
    """Cross-check (not a pin, see tests/crosscheck.py): oracle/producers_oracle.mesh_bound_gaussians -- torch float32 with the
    restated pytorch3d matrix_to_quaternion -- against an independent float64 numpy + scipy construction: same points,
    same scaling, and the same ROTATION (quaternions compared through R(q) and, up to sign, against scipy's from_matrix)."""
    import torch
    import crosscheck
    from gaustar_amd import harness, scene
    from oracle import producers_oracle as po
    rng = np.random.default_rng(11)
    v, f = scene.icosphere(2, 0.9, (0.0, 1.2, 0.0))
    G = 6
    N = len(f) * G
    bary = np.asarray(harness.BARY_COORDS[G], np.float64)
    raw_scales = rng.normal(-5.0, 0.3, (N, 2)); raw_complex = rng.normal(size=(N, 2))
    delta_t = 1e-3 * rng.normal(size=(N, 3)); delta_r = np.array([1.0, 0, 0, 0]) + 0.05 * rng.normal(size=(N, 4))
    for dt, dr in ((None, None), (delta_t, delta_r)):
        t = lambda x: None if x is None else torch.from_numpy(np.asarray(x, np.float32))
        pts, scl, quat = po.mesh_bound_gaussians(t(v), torch.from_numpy(f).long(), t(bary), t(raw_scales), t(raw_complex), 3e-6, None, None,
                                                 t(dt), t(dr))
        p64, s64, R64 = crosscheck.mesh_frames_f64(v, f, bary, raw_scales, raw_complex, 3e-6, dt, dr)
        np.testing.assert_allclose(pts.numpy(), p64, atol=2e-6)
        np.testing.assert_allclose(scl.numpy(), s64, rtol=2e-6)
        q = quat.numpy().astype(np.float64)
        np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)
        np.testing.assert_allclose(crosscheck.quat_wxyz_to_matrix(q), R64, atol=5e-6)
        qs = crosscheck.quats_from_matrices_scipy(R64)
        sign = np.sign(np.sum(q * qs, axis=1, keepdims=True))
        np.testing.assert_allclose(q, sign * qs, atol=5e-6)

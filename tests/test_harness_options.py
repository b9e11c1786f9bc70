"""CPU: the less-travelled options of the render harness (gaustar_amd/harness.py, counterpart of
SuGaR.render_image_gaussian_rasterizer, sugar_model.py:1065-1311): colours from given directions (pinned by the reference's
own eval_sh), `overwrite_extr` cameras, and the 3-D covariance handed over instead of scales + quaternions."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def test_colours_from_given_directions_against_the_reference_eval_sh():
    """producers.points_rgb_from_directions == clamp_min(eval_sh(...) + 0.5, 0) of the reference on non-unit directions
    (tests/golden/harness_options_kat.npz, made with /root/reference/gaustar_utils/spherical_harmonics.py), values and the
    gradients w.r.t. directions and coefficients, sh_levels 1..5."""
    from gaustar_amd import producers
    k = np.load(os.path.join(HERE, "golden", "harness_options_kat.npz"))
    for lv in (1, 2, 3, 4, 5):
        dirs = torch.from_numpy(k[f"dirs_{lv}"]).requires_grad_(True)
        sh = torch.from_numpy(k[f"sh_{lv}"]).requires_grad_(True)
        col = producers.points_rgb_from_directions(dirs, sh, lv)
        ref = torch.from_numpy(k[f"colors_{lv}"])
        scale = float(ref.abs().max()) + 1.0
        assert float((col - ref).abs().max()) <= 2e-6 * scale, lv
        (col * torch.from_numpy(k[f"w_{lv}"])).sum().backward()
        gd = dirs.grad if dirs.grad is not None else torch.zeros_like(dirs)
        for got, want in ((gd, k[f"ddirs_{lv}"]), (sh.grad, k[f"dsh_{lv}"])):
            want = torch.from_numpy(want)
            assert float((got - want).abs().max()) <= 5e-6 * (float(want.abs().max()) + 1.0), lv


def test_get_points_rgb_argument_rules():
    """sugar_model.py:698-703: camera_centers wins over directions; neither raises ValueError."""
    from gaustar_amd import harness, scene
    v, f = scene.icosphere(1, 1.0)
    m = harness.SurfaceGaussians(torch.from_numpy(v).float(), torch.from_numpy(f).long(), 1, sh_levels=2)
    with pytest.raises(ValueError, match="camera_centers or directions"):
        m.get_points_rgb()
    d = torch.nn.functional.normalize(torch.randn(m.n_points, 3), dim=-1)
    c = m.get_points_rgb(directions=d, sh_levels=2)
    assert tuple(c.shape) == (m.n_points, 3) and float(c.min()) >= 0.0


def test_overwrite_extr_camera_equals_the_camera_it_was_taken_from():
    """NerfCamera.with_extrinsic(world-to-camera) rebuilds view / full-projection matrices and the camera centre of the camera
    whose extrinsic it is given (sugar_model.py:1119-1127, :1141-1150), keeps the intrinsics, and moves with a new pose."""
    from gaustar_amd import harness, scene
    cam = scene.look_at_camera((0.5, 1.6, 3.0), scene.SUBJECT_CENTER, 320, 240, focal_px=260.0)
    nc = harness.nerf_camera_from_scene(cam)
    E = np.asarray(cam.viewmatrix, np.float64).T                 # world-to-camera, column-vector convention
    a, b = nc.rasterizer_camera(), nc.with_extrinsic(E).rasterizer_camera()
    for x, y in ((a.viewmatrix, b.viewmatrix), (a.projmatrix, b.projmatrix), (a.campos, b.campos)):
        assert np.abs(np.asarray(x) - np.asarray(y)).max() < 1e-6
    assert (a.W, a.H, a.tanfovx, a.tanfovy) == (b.W, b.H, b.tanfovx, b.tanfovy)
    cam2 = scene.look_at_camera((-1.0, 0.9, 2.5), scene.SUBJECT_CENTER, 320, 240, focal_px=260.0)
    E2 = torch.from_numpy(np.asarray(cam2.viewmatrix, np.float32).T.copy())     # a tensor, as refined_mesh.py:353 passes it
    c = nc.with_extrinsic(E2).rasterizer_camera()
    assert np.abs(np.asarray(c.viewmatrix) - np.asarray(cam2.viewmatrix)).max() < 1e-6
    assert np.abs(np.asarray(c.campos) - np.asarray(cam2.campos)).max() < 1e-5


def test_covariance_3d_is_R_S2_Rt():
    """producers.covariance_3d / quaternion_to_matrix (pytorch3d's published algorithm, real part first, any norm) against
    scipy's rotation and the closed form; the six entries in the rasterizer's order {xx, xy, xz, yy, yz, zz}."""
    from scipy.spatial.transform import Rotation
    from gaustar_amd import producers
    rng = np.random.default_rng(3)
    q = rng.normal(size=(64, 4)) * rng.uniform(0.3, 2.0, size=(64, 1))            # un-normalised on purpose
    s = rng.uniform(0.01, 0.5, size=(64, 3))
    R = producers.quaternion_to_matrix(torch.from_numpy(q)).numpy()
    R_ref = Rotation.from_quat(np.concatenate([q[:, 1:], q[:, :1]], axis=1)).as_matrix()   # scipy: scalar last, normalises
    assert np.abs(R - R_ref).max() < 1e-12
    c = producers.covariance_3d(torch.from_numpy(s), torch.from_numpy(q)).numpy()
    S = R_ref @ (s[:, :, None] ** 2 * np.eye(3)) @ R_ref.transpose(0, 2, 1)
    want = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)
    assert np.abs(c - want).max() < 1e-12


def test_readers_outside_the_fused_render_fence_lazily_gathered_parameters():
    """with dist.ShardedAdam(gather_first=...) step() leaves the SH / density gathers in flight; `strengths`,
    `sh_coordinates`, `get_points_rgb` and `state_dict()` must fence the parameters they read (wait_params), as the fused
    render does for its producers."""
    from gaustar_amd import harness, scene
    v, f = scene.icosphere(1, 1.0)
    m = harness.SurfaceGaussians(torch.from_numpy(v).float(), torch.from_numpy(f).long(), 1, sh_levels=2)

    class Sink:
        def __init__(self):
            self.calls = []

        def wait_params(self, params=None):
            self.calls.append(None if params is None else {id(p) for p in params})

    m.grad_sink = s = Sink()
    _ = m.strengths
    assert s.calls[-1] == {id(m.all_densities)}
    _ = m.sh_coordinates
    assert s.calls[-1] == {id(m._sh_coordinates_dc), id(m._sh_coordinates_rest)}
    n = len(s.calls)
    d = torch.nn.functional.normalize(torch.randn(m.n_points, 3), dim=-1)
    m.get_points_rgb(directions=d, sh_levels=2)
    assert len(s.calls) > n and s.calls[-1] == {id(m._sh_coordinates_dc), id(m._sh_coordinates_rest)}
    m.state_dict()
    assert s.calls[-1] is None                                   # everything
    m.grad_sink = None
    _ = m.strengths                                              # no sink: nothing to call

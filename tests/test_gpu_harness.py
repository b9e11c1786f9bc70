"""GPU: the render-harness counterpart (gaustar_amd/harness.py) against the oracle pieces:
camera matrices (sugar_model.py:1129-1163), properties via the producer oracle, image via the rasterizer oracle."""
import os

import numpy as np
import pytest
import torch

import parity

pytestmark = pytest.mark.gpu


def _model(level=2, loose=False, seed=0):
    from gaustar_amd import harness, scene
    v, f = scene.icosphere(level, radius=0.9, center=(0.0, 1.2, 0.0))
    g = torch.Generator().manual_seed(seed)
    m = harness.SurfaceGaussians(torch.from_numpy(v).float().cuda(), torch.from_numpy(f).long().cuda(), sh_levels=4,
                                 surface_mesh_thickness=3e-6, loose_bind=loose)
    with torch.no_grad():
        m._quaternions.copy_(torch.randn(m.n_points, 2, generator=g))
        m._scales.add_(0.3 * torch.randn(m.n_points, 2, generator=g).cuda())
        m._sh_coordinates_dc.copy_(torch.rand(m.n_points, 1, 3, generator=g) * 2 - 1)
        m._sh_coordinates_rest.copy_(0.2 * (torch.rand(m.n_points, 15, 3, generator=g) - 0.5))
        if loose:
            m._delta_t.copy_(0.005 * torch.randn(m.n_points, 3, generator=g))
            m._delta_r.copy_(torch.tensor([1.0, 0, 0, 0]) + 0.2 * torch.randn(m.n_points, 4, generator=g))
    return m


def test_camera_matrices_follow_the_nerf_to_colmap_recipe(hip_lib):
    from gaustar_amd import harness, scene
    ref = scene.look_at_camera((0.7, 1.9, 2.8), (0.0, 1.2, 0.0), 320, 200, focal_px=250.0)     # COLMAP-axes construction
    cam = harness.nerf_camera_from_scene(ref).rasterizer_camera()
    np.testing.assert_allclose(cam.viewmatrix, ref.viewmatrix, atol=2e-6)
    np.testing.assert_allclose(cam.projmatrix, ref.projmatrix, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(cam.campos, ref.campos, atol=2e-6)
    assert abs(cam.tanfovx - ref.tanfovx) < 1e-6 and abs(cam.tanfovy - ref.tanfovy) < 1e-6
    off = harness.NerfCamera(c2w=harness.nerf_camera_from_scene(ref).c2w, fx=250.0, fy=250.0, width=320, height=200,
                             principal_ndc=(0.05, -0.02)).rasterizer_camera()
    # the principal point only enters rows 2 of the transposed projection (sugar_model.py:1159-1160)
    d = off.projmatrix - ref.projmatrix
    p = np.zeros((4, 4)); p[2, 0], p[2, 1] = -0.05, 0.02
    np.testing.assert_allclose(d, (ref.viewmatrix.astype(np.float64) @ p).astype(np.float32), atol=1e-6)


@pytest.mark.parametrize("loose,in_rasterizer", [(False, False), (True, True)])
def test_render_matches_oracle_composition(loose, in_rasterizer, hip_lib):
    from gaustar_amd import harness, scene
    from oracle import producers_oracle as po
    m = _model(2, loose)
    ncam = harness.nerf_camera_from_scene(scene.look_at_camera((0.5, 1.6, 2.6), (0.0, 1.2, 0.0), 200, 160, focal_px=170.0))
    bg = [0.0, 1.0, 0.0]
    out = m.render_image_gaussian_rasterizer(camera=ncam, bg_color=bg, sh_deg=3, compute_color_in_rasterizer=in_rasterizer,
                                             return_2d_radii=True, return_opacities=True, return_colors=not in_rasterizer)
    img = out["image"]
    assert tuple(img.shape) == (160, 200, 3) and out["radii"].shape[0] == m.n_points
    loss = (img * torch.linspace(0.5, 1.5, 3, device=img.device)).sum()
    loss.backward()
    assert m._points.grad is not None and m._scales.grad is not None and m._quaternions.grad is not None
    assert out["viewspace_points"].grad is not None and torch.isfinite(m._points.grad).all()

    # the same composition from the oracles (CPU)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    pts, scl, quat = po.mesh_bound_gaussians(sd["_points"], sd["_surface_mesh_faces"], sd["surface_triangle_bary_coords"][..., 0],
                                             sd["_scales"], sd["_quaternions"], float(sd["surface_mesh_thickness"]), None, None,
                                             sd.get("_delta_t"), sd.get("_delta_r"))
    cam = ncam.rasterizer_camera()
    sh = torch.cat([sd["_sh_coordinates_dc"], sd["_sh_coordinates_rest"]], 1)
    kw = dict(means3D=pts.numpy(), opacities=torch.sigmoid(sd["all_densities"]).numpy(), view=cam.viewmatrix, proj=cam.projmatrix,
              campos=cam.campos, W=cam.W, H=cam.H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=np.array(bg, np.float32),
              scales=scl.numpy(), rotations=quat.numpy(), cov3D_precomp=None, scale_modifier=1.0)
    if in_rasterizer:
        kw.update(shs=sh.numpy(), colors_precomp=None, sh_degree=3)
    else:
        kw.update(shs=None, colors_precomp=po.points_rgb(pts, torch.from_numpy(cam.campos)[None], sh, 4).numpy(), sh_degree=0)
    st, _ = parity.run_oracle(kw)
    # the two compositions feed the rasterizer inputs that differ by rounding (HIP producers vs the torch restatement):
    # a (pixel, Gaussian) pair next to a blend threshold may flip, hence the threshold-flip allowance of parity.py here
    parity.check_image(img.detach().permute(2, 0, 1).cpu().numpy(), st["color"], "harness image vs oracle composition", tol=2e-4,
                       max_outlier_frac=2e-4)


def test_fused_rgb_depth_equals_the_two_reference_style_renders(hip_lib):
    from gaustar_amd import harness, scene
    m = _model(3, False, seed=3)
    ncam = harness.nerf_camera_from_scene(scene.look_at_camera((-0.6, 1.0, 2.7), (0.0, 1.2, 0.0), 256, 192, focal_px=200.0))
    with torch.no_grad():
        rgb, depth = m.render_rgb_depth(camera=ncam, bg_color=[0.0, 1.0, 0.0], max_depth=10.0, sh_deg=3)
        a = m.render_image_gaussian_rasterizer(camera=ncam, bg_color=[0.0, 1.0, 0.0], sh_deg=3)
        b = m.render_image_gaussian_rasterizer(camera=ncam, bg_color=[10.0, 10.0, 10.0], sh_deg=0,
                                               point_colors=m.view_depth_colors(ncam))[..., 0]     # refine.py:603-616
    # RGB: same colours into the same walk -> bit-identical.  Depth: the fused producer forms z = (p, 1) . column 2 of the
    # view matrix in one kernel, view_depth_colors with a torch matmul: the inputs of the blend differ in the last bit
    assert torch.equal(rgb, a)
    assert torch.allclose(depth, b, rtol=2e-6, atol=2e-6)


def test_checkpoint_round_trip_through_the_model(tmp_path, hip_lib):
    from gaustar_amd import formats, harness
    m = _model(1, True, seed=5)
    sd = m.state_dict()
    path = os.path.join(tmp_path, "7000.pt")
    formats.save_sugar_checkpoint(path, sd["_points"], sd["_surface_mesh_faces"], sd["_scales"], sd["_quaternions"],
                                  sd["all_densities"], torch.cat([sd["_sh_coordinates_dc"], sd["_sh_coordinates_rest"]], 1),
                                  float(sd["surface_mesh_thickness"]), sd["_delta_t"], sd["_delta_r"])
    m2 = harness.SurfaceGaussians.from_checkpoint(formats.load_sugar_checkpoint(path), "cuda")
    for k, v in sd.items():
        assert torch.equal(m2.state_dict()[k].cpu(), v.cpu()), k
    assert torch.equal(m2.points, m.points) and torch.equal(m2.quaternions, m.quaternions)
    # a reference-style state dict loads by name
    m3 = harness.SurfaceGaussians(sd["_points"], sd["_surface_mesh_faces"], loose_bind=True, surface_mesh_thickness=1.0)
    m3.load_state_dict(sd)
    assert torch.equal(m3.scaling, m.scaling)


@pytest.mark.parametrize("loose,depth_channels", [(False, 1), (True, 1), (True, 0), (False, 3)])
def test_one_node_render_equals_the_composition_of_nodes(loose, depth_channels, hip_lib):
    """SurfaceGaussians.render_channels (ONE autograd node: parameters in, image out) against the same render composed of
    autograd nodes the way the reference's harness composes it (properties -> SH colours -> sigmoid -> rasterizer): the
    image is bit-identical (same kernels on the same numbers), every parameter's gradient agrees to the run-to-run noise
    of the float atomics in the two blends' and the mesh producer's backward."""
    from gaustar_amd import GaussianRasterizer, harness, producers, scene
    m = _model(3, loose, seed=9)
    ncam = harness.nerf_camera_from_scene(scene.look_at_camera((0.7, 1.7, 2.6), (0.0, 1.2, 0.0), 320, 240, focal_px=260.0))
    C = 3 + depth_channels
    bg = torch.tensor([0.0, 1.0, 0.0] + [10.0] * depth_channels, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    d_img = torch.randn(C, 240, 320, device="cuda", generator=g)

    def grads():
        out = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        m.zero_grad(set_to_none=True)
        return out

    img, radii = m.render_channels(ncam, bg, sh_deg=3, depth_channels=depth_channels)
    (img * d_img).sum().backward()
    got = grads()

    settings, view, campos = m._settings(ncam, bg, 0)
    pts = m.points
    if depth_channels:
        col = producers.points_rgb_depth(pts, campos, m.sh_coordinates, 4, view, depth_channels=depth_channels)
    else:
        col = producers.points_rgb(pts, campos, m.sh_coordinates, 4)
    img_ref, radii_ref = GaussianRasterizer(settings)(means3D=pts, means2D=torch.zeros_like(pts), opacities=m.strengths,
                                                      colors_precomp=col, scales=m.scaling, rotations=m.quaternions)
    (img_ref * d_img).sum().backward()
    want = grads()
    assert torch.equal(img, img_ref) and torch.equal(radii, radii_ref)
    assert set(got) == set(want) and len(got) == (8 if loose else 6)
    for n in want:
        scale = float(want[n].abs().max())
        assert scale > 0, n
        assert float((got[n] - want[n]).abs().max()) <= 2e-5 * scale, (n, float((got[n] - want[n]).abs().max()), scale)
    # forward only: no graph, same image
    with torch.no_grad():
        img2, _ = m.render_channels(ncam, bg, sh_deg=3, depth_channels=depth_channels)
    assert torch.equal(img2, img_ref)


def test_thickness_is_read_back_once(hip_lib):
    """surface_mesh_thickness lives in a device buffer (state-dict compatibility); the render path must not read it back
    per call (a device-to-host copy drains the queue: one pipeline bubble per iteration)."""
    m = _model(2, False, seed=1)
    a = m._thickness()
    assert a == pytest.approx(3e-6) and m._thickness() is m._thickness_cache[2]
    with torch.no_grad():
        m.surface_mesh_thickness.fill_(5e-6)          # in-place change: version bump -> re-read
    assert m._thickness() == pytest.approx(5e-6)


def test_one_node_render_with_constant_colours_only(hip_lib):
    """sh_levels = 1: `_sh_coordinates_rest` is an empty [N,0,3] parameter -- the one-node render reads the dc array alone and
    returns an empty gradient for the empty parameter."""
    from gaustar_amd import harness, scene
    v, f = scene.icosphere(2, radius=0.9, center=(0.0, 1.2, 0.0))
    m = harness.SurfaceGaussians(torch.from_numpy(v).float().cuda(), torch.from_numpy(f).long().cuda(), sh_levels=1)
    with torch.no_grad():
        m._sh_coordinates_dc.copy_(torch.rand(m.n_points, 1, 3, generator=torch.Generator().manual_seed(2)) * 2 - 1)
    ncam = harness.nerf_camera_from_scene(scene.look_at_camera((0.7, 1.7, 2.6), (0.0, 1.2, 0.0), 160, 120, focal_px=130.0))
    bg = torch.tensor([0.0, 1.0, 0.0, 10.0], device="cuda")
    img, _ = m.render_channels(ncam, bg, depth_channels=1)
    img.sum().backward()
    assert tuple(m._sh_coordinates_rest.shape) == (m.n_points, 0, 3)
    assert m._sh_coordinates_rest.grad is None or m._sh_coordinates_rest.grad.numel() == 0
    assert float(m._sh_coordinates_dc.grad.abs().max()) > 0 and torch.isfinite(m._points.grad).all()
    ref = m.render_image_gaussian_rasterizer(ncam, bg_color=[0.0, 1.0, 0.0], return_opacities=True)["image"]   # (the composition of nodes)
    assert torch.equal(img[:3].permute(1, 2, 0), ref)
    assert torch.equal(m.render_image_gaussian_rasterizer(ncam, bg_color=[0.0, 1.0, 0.0]), ref)                  # (the plain call: one node)


# ------------------------------------------------------------------ the less-travelled arguments (sugar_model.py:1065-1311)
def test_overwrite_extr_renders_the_other_camera(hip_lib):
    """render_image_gaussian_rasterizer(camera A, overwrite_extr = world-to-camera of B) == render(camera B)
    (sugar_model.py:1119-1127, :1141-1147; refined_mesh.py:353 renders from a re-posed camera this way)."""
    from gaustar_amd import harness, scene
    m = _model(2, False, seed=5)
    a = scene.look_at_camera((0.5, 1.6, 2.6), (0.0, 1.2, 0.0), 200, 160, focal_px=170.0)
    b = scene.look_at_camera((-0.9, 0.8, 2.4), (0.0, 1.2, 0.0), 200, 160, focal_px=170.0)
    na, nb = harness.nerf_camera_from_scene(a), harness.nerf_camera_from_scene(b)
    E = torch.from_numpy(np.asarray(b.viewmatrix, np.float32).T.copy()).cuda()
    with torch.no_grad():
        want = m.render_image_gaussian_rasterizer(camera=nb, bg_color=[0.1, 0.2, 0.3], sh_deg=3)
        got = m.render_image_gaussian_rasterizer(camera=na, bg_color=[0.1, 0.2, 0.3], sh_deg=3, overwrite_extr=E)
        other = m.render_image_gaussian_rasterizer(camera=na, bg_color=[0.1, 0.2, 0.3], sh_deg=3)
    # (the pose goes through two more float64 inversions: matrices equal to 1e-6, a threshold flip allowed as elsewhere)
    parity.check_image(got.permute(2, 0, 1).cpu().numpy(), want.permute(2, 0, 1).cpu().numpy(), "overwrite_extr vs camera B", tol=2e-4,
                       max_outlier_frac=2e-4)
    assert float((got - other).abs().max()) > 0.05          # and it is not camera A's image


def test_sh_rotations_turn_the_view_directions(hip_lib):
    """sh_rotations = identity gives the plain render's colours; a random rotation per Gaussian gives
    clamp_min(eval_sh(normalize(p - c) @ R) + 0.5, 0) (sugar_model.py:1200-1205) -- checked against the producer oracle's
    eval_sh restatement (pinned by the reference's own) -- and gradients reach the SH coefficients and the positions."""
    from gaustar_amd import harness, scene
    from oracle import producers_oracle as po
    m = _model(2, True, seed=6)
    ncam = harness.nerf_camera_from_scene(scene.look_at_camera((0.4, 1.5, 2.7), (0.0, 1.2, 0.0), 160, 120, focal_px=140.0))
    P = m.n_points
    eye = torch.eye(3, device="cuda").expand(P, 3, 3).contiguous()
    with torch.no_grad():
        plain = m.render_image_gaussian_rasterizer(camera=ncam, sh_deg=3, return_colors=True)
        ident = m.render_image_gaussian_rasterizer(camera=ncam, sh_deg=3, sh_rotations=eye, return_colors=True)
    assert torch.allclose(plain["colors"], ident["colors"], rtol=1e-5, atol=2e-6)
    assert torch.allclose(plain["image"], ident["image"], rtol=1e-4, atol=1e-4)
    from scipy.spatial.transform import Rotation
    Rm = torch.from_numpy(Rotation.random(P, random_state=1).as_matrix().astype(np.float32)).cuda()
    out = m.render_image_gaussian_rasterizer(camera=ncam, sh_deg=3, sh_rotations=Rm, return_colors=True)
    _cam, _view, _proj, campos = ncam.on_device(m.device)
    pts = m.points.detach()
    dirs = (torch.nn.functional.normalize(pts - campos.view(1, 3), dim=-1).unsqueeze(1) @ Rm)[..., 0, :]
    sh = m.sh_coordinates.detach().cpu()
    want = torch.clamp_min(po.eval_sh(3, sh.transpose(-1, -2), dirs.cpu()) + 0.5, 0.0)
    assert torch.allclose(out["colors"].detach().cpu(), want, rtol=1e-5, atol=5e-6)
    out["image"].sum().backward()
    assert m._sh_coordinates_rest.grad is not None and float(m._sh_coordinates_rest.grad.abs().max()) > 0
    assert m._points.grad is not None and torch.isfinite(m._points.grad).all()


def test_covariance_handed_over_instead_of_scales_and_rotations(hip_lib):
    """compute_covariance_in_rasterizer=False (sugar_model.py:1237-1260): cov3D = R diag(s^2) R^T formed outside and passed
    as cov3D_precomp renders the same image as scales + quaternions, and gradients flow back through it to the mesh."""
    from gaustar_amd import harness, scene
    m = _model(2, False, seed=7)
    ncam = harness.nerf_camera_from_scene(scene.look_at_camera((0.5, 1.6, 2.6), (0.0, 1.2, 0.0), 200, 160, focal_px=170.0))
    inside = m.render_image_gaussian_rasterizer(camera=ncam, bg_color=[0.0, 1.0, 0.0], sh_deg=3, return_2d_radii=True)
    outside = m.render_image_gaussian_rasterizer(camera=ncam, bg_color=[0.0, 1.0, 0.0], sh_deg=3, compute_covariance_in_rasterizer=False,
                                                 return_2d_radii=True)
    assert int((inside["radii"] != outside["radii"]).sum()) <= 2
    parity.check_image(outside["image"].detach().permute(2, 0, 1).cpu().numpy(), inside["image"].detach().permute(2, 0, 1).cpu().numpy(),
                       "cov3D outside vs inside the rasterizer", tol=2e-4, max_outlier_frac=2e-4)
    g = {}
    for name, flag in (("in", True), ("out", False)):   # (rendered again: the two renders above share the cached geometry node)
        for p in m.parameters():
            p.grad = None
        m._geom_cache = None
        img = m.render_image_gaussian_rasterizer(camera=ncam, bg_color=[0.0, 1.0, 0.0], sh_deg=3, compute_covariance_in_rasterizer=flag,
                                                 return_2d_radii=True)["image"]
        (img * torch.linspace(0.5, 1.5, 3, device="cuda")).sum().backward()
        g[name] = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    assert set(g["in"]) == set(g["out"])
    for k in g["in"]:
        a, b = g["out"][k], g["in"][k]
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-7, k

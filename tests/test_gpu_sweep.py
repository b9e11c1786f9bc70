"""GPU: forward-only sweeps (gaustar_amd/sweep.py) and rendering from the wire formats (gaustar_amd/formats.py)."""
import os

import numpy as np
import pytest
import torch

import parity

pytestmark = pytest.mark.gpu


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def test_sweep_equals_individual_renders(hip_lib):
    """One 6-channel no-grad forward per camera == the RGB render and the depth-as-colour render of
    refined_mesh.py:733-760 done separately (bit for bit), and the per-view table comes back in camera order."""
    from gaustar_amd import scene, sweep
    rng = np.random.default_rng(2)
    verts, faces = scene.uv_sphere(20, 12, radius=0.9, center=(0, 1.2, 0))
    gs = scene.mesh_bound_gaussians(verts, faces, rng, thickness=3e-6)
    cams = scene.ring_cameras(2, 3, 200, 150, focal_px=130.0)
    fs = sweep.ForwardSweep(_t(gs.means3D), _t(gs.opacities), _t(gs.scales), _t(gs.rotations), rgb=_t(gs.colors_precomp))
    table = fs.sweep(cams, lambda i, cam, rgb, depth: torch.stack([rgb.mean(), depth.min(), torch.tensor(float(i), device=rgb.device)]))
    assert tuple(table.shape) == (6, 3) and table[:, 2].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
    for i in (0, 4):
        cam = cams[i]
        rgb, depth = fs.render_rgb_depth(cam)
        kw = dict(means3D=gs.means3D, opacities=gs.opacities, view=cam.viewmatrix, proj=cam.projmatrix, campos=cam.campos,
                  W=cam.W, H=cam.H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=np.array([0, 1, 0], np.float32), shs=None,
                  colors_precomp=gs.colors_precomp, scales=gs.scales, rotations=gs.rotations, cov3D_precomp=None, sh_degree=0)
        a = parity.run_hip(kw)
        kw_d = dict(kw, bg=np.full(3, 10.0, np.float32), colors_precomp=scene.view_depth_colors(gs, cam))
        b = parity.run_hip(kw_d)
        assert np.array_equal(rgb.permute(2, 0, 1).cpu().numpy(), a["color"])
        np.testing.assert_allclose(depth.cpu().numpy(), b["color"][0], rtol=1e-6, atol=1e-6)   # depth colours: matmul vs numpy
        assert np.array_equal(fs.render_depth(cam).cpu().numpy(), depth.cpu().numpy())
        assert abs(table[i, 0].item() - rgb.mean().item()) < 1e-7


def test_render_from_ply_and_cameras_json(tmp_path, hip_lib):
    """A 3DGS point cloud written to / read from PLY and cameras from cameras.json render exactly like the in-memory
    originals (vanilla caller path, gaussian_renderer/__init__.py:36-93: in-kernel SH)."""
    from gaustar_amd import formats, scene
    rng = np.random.default_rng(4)
    gs = scene.random_gaussians(2000, rng, sh_degree=3, with_sh=True, scale_range=(0.02, 0.08))
    logit = lambda p: np.log(p / (1 - p))
    cloud = formats.GaussianCloud(xyz=gs.means3D, features_dc=gs.shs[:, :1], features_rest=gs.shs[:, 1:],
                                  opacity=logit(gs.opacities.astype(np.float64)).astype(np.float32), scaling=np.log(gs.scales),
                                  rotation=(gs.rotations * 1.7).astype(np.float32))
    ply = os.path.join(tmp_path, "point_cloud.ply")
    formats.save_ply(ply, cloud)
    cams = scene.ring_cameras(1, 2, 160, 120, focal_px=110.0, center=(0.0, 0.0, 0.0))
    cj = os.path.join(tmp_path, "cameras.json")
    formats.save_cameras_json(cams, cj)
    ri = formats.load_ply(ply).rasterizer_inputs()
    cam = formats.load_cameras_json(cj, znear=1e-4, zfar=100.0)[1]
    bg = np.array([0.2, 0.2, 0.2], np.float32)
    mk = lambda c, d: dict(means3D=d["means3D"], opacities=d["opacities"], view=c.viewmatrix, proj=c.projmatrix, campos=c.campos,
                           W=c.W, H=c.H, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg, shs=d["shs"], colors_precomp=None,
                           scales=d["scales"], rotations=d["rotations"], cov3D_precomp=None, sh_degree=3)
    a = parity.run_hip(mk(cam, ri))
    b = parity.run_hip(mk(cams[1], dict(means3D=gs.means3D, opacities=gs.opacities, shs=gs.shs, scales=gs.scales, rotations=gs.rotations)))
    assert (a["radii"] != b["radii"]).sum() <= 2
    parity.check_image(a["color"], b["color"], "ply + cameras.json vs in-memory", tol=2e-4)


def test_three_pass_sweep_against_reference_build(hip_lib):
    """The per-camera renders of detect_topo_err / forward_rendering_and_mesh_update (gaustar_trainers/refined_mesh.py:733-775)
    at config-C size, one rig camera: (1) RGB with IN-KERNEL SH of degree 2 (compute_color_in_rasterizer=True,
    sugar_model.py:1207-1209), (2) depth-as-colour with bg = max depth, (3) the same with `use_solid_surface` scales
    (sugar_model.py:1230-1232) -- each run through the reference's own kernels (oracle/_ref/libgsr_ref.so) -- against
    sweep.ForwardSweep: passes 1 + 2 as ONE 4-channel forward with colours from the fused SH producer, pass 3 as
    render_depth(cam, scales=...)."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/libgsr_ref.so did not travel to this box")
    from gaustar_amd import scene, sweep
    gs, cams, _bg = scene.config_C()
    cam = cams[37]
    rng = np.random.default_rng(9)
    sh = np.concatenate([rng.uniform(-1.5, 1.5, (gs.P, 1, 3)), rng.uniform(-0.3, 0.3, (gs.P, 8, 3))], 1).astype(np.float32)
    max_depth = 10.0
    fs = sweep.ForwardSweep(_t(gs.means3D), _t(gs.opacities), _t(gs.scales), _t(gs.rotations), sh=_t(sh), sh_levels=3,
                            max_depth=max_depth)
    rgb, depth = fs.render_rgb_depth(cam)
    solid = _t(gs.scales).clone()
    solid[..., 1:] = torch.maximum(solid[..., 1:].mean(), solid[..., 1:])            # sugar_model.py:1230-1232
    sdepth = fs.render_depth(cam, scales=solid)

    common = dict(means3D=gs.means3D, opacities=gs.opacities, view=cam.viewmatrix, proj=cam.projmatrix, campos=cam.campos, W=cam.W,
                  H=cam.H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, rotations=gs.rotations)
    rr = ref.RefRasterizer()
    c1, _, _ = rr.forward(bg=np.array([0.0, 1.0, 0.0], np.float32), shs=sh, sh_degree=2, scales=gs.scales, **common)
    dcol = scene.view_depth_colors(gs, cam)
    c2, _, _ = rr.forward(bg=np.full(3, max_depth, np.float32), colors_precomp=dcol, scales=gs.scales, **common)
    c3, _, _ = rr.forward(bg=np.full(3, max_depth, np.float32), colors_precomp=dcol, scales=solid.cpu().numpy(), **common)
    full = dict(max_outlier_frac=parity.FULL_IMG_OUTLIERS)
    parity.check_image(rgb.permute(2, 0, 1).cpu().numpy(), c1.cpu().numpy(), "sweep pass 1: RGB, in-kernel SH deg 2", **full)
    parity.check_image(depth.cpu().numpy(), c2[0].cpu().numpy(), "sweep pass 2: depth", **full)
    parity.check_image(sdepth.cpu().numpy(), c3[0].cpu().numpy(), "sweep pass 3: solid-surface depth", **full)
    assert float((sdepth < max_depth - 1e-3).float().mean()) >= float((depth < max_depth - 1e-3).float().mean())   # larger splats cover more

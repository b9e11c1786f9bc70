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

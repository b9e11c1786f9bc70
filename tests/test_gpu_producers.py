"""GPU: fused producers of rasterizer inputs against vectors made with the reference's own eval_sh
(tests/golden/producers_kat.npz) and, at config-D size, against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
KAT = os.path.join(ROOT, "tests", "golden", "producers_kat.npz")


@pytest.mark.parametrize("lv", [1, 2, 3, 4])
def test_points_rgb_matches_reference_vectors(lv, hip_lib):
    from gaustar_amd import producers
    z = np.load(KAT)
    k = f"l{lv}"
    pos = torch.from_numpy(z[f"{k}_pos"]).cuda().requires_grad_(True)
    sh = torch.from_numpy(z[f"{k}_sh"]).cuda().requires_grad_(True)
    col = producers.points_rgb(pos, torch.from_numpy(z[f"{k}_cam"]).cuda(), sh, lv)
    col.backward(torch.from_numpy(z[f"{k}_dL"]).cuda())
    np.testing.assert_allclose(col.detach().cpu().numpy(), z[f"{k}_colors"], rtol=1e-5, atol=2e-6)
    assert ((col.detach().cpu().numpy() == 0) == (z[f"{k}_colors"] == 0)).mean() > 0.999   # same clamp decisions
    np.testing.assert_allclose(sh.grad.cpu().numpy(), z[f"{k}_dsh"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(pos.grad.cpu().numpy(), z[f"{k}_dpos"], rtol=1e-4, atol=2e-5)


def test_points_rgb_feeds_the_rasterizer_like_in_kernel_sh(hip_lib):
    """colors_precomp = points_rgb(...) must render exactly what shs=... renders (same arithmetic, shared device
    functions), image and gradients -- this is what lets the 6-channel multi-target path take SH colours."""
    import parity
    from gaustar_amd import producers, scene
    rng = np.random.default_rng(9)
    gs = scene.random_gaussians(3000, rng, sh_degree=3, with_sh=True, scale_range=(0.02, 0.1))
    cam = scene.look_at_camera((0.2, 0.1, -4.0), (0, 0, 0), 160, 120, fovx=0.9, znear=0.01)
    kw = dict(means3D=gs.means3D, opacities=gs.opacities, view=cam.viewmatrix, proj=cam.projmatrix, campos=cam.campos,
              W=cam.W, H=cam.H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=np.array([0.1, 0.3, 0.2], np.float32),
              shs=gs.shs, colors_precomp=None, scales=gs.scales, rotations=gs.rotations, cov3D_precomp=None, sh_degree=3)
    dpix = rng.normal(size=(3, cam.H, cam.W)).astype(np.float32)
    ref = parity.run_hip(kw, dpix)

    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    t = lambda x, g=False: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev).requires_grad_(g)
    means, sh, op, sc, ro = t(gs.means3D, True), t(gs.shs, True), t(gs.opacities, True), t(gs.scales, True), t(gs.rotations, True)
    cols = producers.points_rgb(means, t(cam.campos), sh, 4)
    s = GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, t(kw["bg"]), 1.0, t(cam.viewmatrix),
                                      t(cam.projmatrix), 0, t(cam.campos), False, False)
    img, radii = GaussianRasterizer(s)(means3D=means, means2D=torch.zeros(gs.P, 3, device=dev), opacities=op,
                                       colors_precomp=cols, scales=sc, rotations=ro)
    img.backward(t(dpix))
    assert np.array_equal(img.detach().cpu().numpy(), ref["color"])
    parity.check_grad(sh.grad.cpu().numpy(), ref["dL_dsh"], "dL_dsh via producer")
    parity.check_grad(means.grad.cpu().numpy(), ref["dL_dmeans3D"], "dL_dmeans3D via producer")


def test_points_rgb_config_d_size_against_oracle(hip_lib):
    from gaustar_amd import producers
    from oracle import producers_oracle
    P = 1_001_232
    g = torch.Generator().manual_seed(1)
    pos = (torch.rand(P, 3, generator=g) * 2 - 1)
    sh = torch.rand(P, 16, 3, generator=g) * 0.6 - 0.3
    sh[:, 0] = torch.rand(P, 3, generator=g) * 3 - 1.5
    cam = torch.tensor([0.0, 1.2, -3.0])
    dL = torch.randn(P, 3, generator=g)
    p1, s1 = pos.clone().requires_grad_(True), sh.clone().requires_grad_(True)
    c_ref = producers_oracle.points_rgb(p1, cam[None], s1, 4)
    c_ref.backward(dL)
    p2, s2 = pos.cuda().requires_grad_(True), sh.cuda().requires_grad_(True)
    c = producers.points_rgb(p2, cam.cuda(), s2, 4)
    c.backward(dL.cuda())
    np.testing.assert_allclose(c.detach().cpu().numpy(), c_ref.detach().numpy(), rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(s2.grad.cpu().numpy(), s1.grad.numpy(), rtol=1e-5, atol=3e-6)
    err = np.abs(p2.grad.cpu().numpy() - p1.grad.numpy()).max() / np.abs(p1.grad.numpy()).max()
    assert err < 1e-4, err


def test_validation(hip_lib):
    from gaustar_amd import producers
    pos = torch.rand(10, 3).cuda()
    with pytest.raises(RuntimeError, match="sh_levels"):
        producers.points_rgb(pos, torch.zeros(3).cuda(), torch.rand(10, 4, 3).cuda(), 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        producers.points_rgb(pos.cpu(), torch.zeros(3), torch.rand(10, 4, 3), 2)

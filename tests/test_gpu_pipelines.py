"""gaustar_amd.pipelines: V independent view pipelines (host thread + HIP stream each) give, view for view, what one view
at a time gives -- images bit-identical (the forward is deterministic), gradients equal up to the order of the backward's
float atomics -- and a worker's exception reaches the caller."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene():
    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, scene
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    v, f = scene.icosphere(3, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
    gs = scene.mesh_bound_gaussians(v, f, rng, 3.5e-6)
    cams = scene.ring_cameras(2, 6, 320, 240, focal_px=260.0)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    params = dict(means3D=t(gs.means3D), opacities=t(gs.opacities), colors=t(gs.colors_precomp), scales=t(gs.scales), rotations=t(gs.rotations))
    for p in params.values():
        p.requires_grad_(True)
    bg = t(np.array([0.0, 1.0, 0.0], np.float32))
    rasters = [GaussianRasterizer(GaussianRasterizationSettings(c.H, c.W, c.tanfovx, c.tanfovy, bg, 1.0, t(c.viewmatrix), t(c.projmatrix),
                                                                0, t(c.campos), False, False)) for c in cams]
    dpix = torch.randn(3, 240, 320, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    return dev, params, rasters, dpix


def _view(ps, r, dpix):
    for p in ps.values():
        p.grad = None
    m2 = torch.zeros(ps["means3D"].shape[0], 3, device=dpix.device, requires_grad=True)
    img, radii = r(means3D=ps["means3D"], means2D=m2, opacities=ps["opacities"], colors_precomp=ps["colors"], scales=ps["scales"],
                   rotations=ps["rotations"])
    img.backward(dpix)
    return img.detach().clone(), {k: p.grad.detach().clone() for k, p in ps.items()}


@pytest.mark.parametrize("V", [2, 3])
def test_pipelines_equal_serial_renders(V):
    import parity
    from gaustar_amd import pipelines
    dev, params, rasters, dpix = _scene()
    serial = [_view(params, r, dpix) for r in rasters]
    leaves = pipelines.clone_leaves(params, V)
    got = [None] * len(rasters)

    def work(t, i):
        got[i] = _view(leaves[t], rasters[i], dpix)
    for _ in range(3):   # several rounds: streams and allocator blocks get reused across pipelines
        pipelines.ViewPipelines(V, dev).run(work, list(range(len(rasters))))
    torch.cuda.synchronize()
    for i, ((img_s, g_s), (img_p, g_p)) in enumerate(zip(serial, got)):
        assert torch.equal(img_s, img_p), f"view {i}: image differs"
        for k in g_s:
            parity.check_grad(g_p[k].cpu().numpy(), g_s[k].cpu().numpy(), f"pipelines V={V} view {i} {k}", tol=1e-5, small_tol=None)


def test_worker_exception_reaches_the_caller():
    from gaustar_amd import pipelines
    dev = torch.device("cuda:0")

    def work(t, i):
        if i == 3:
            raise ValueError("boom")
    with pytest.raises(ValueError, match="boom"):
        pipelines.ViewPipelines(2, dev).run(work, list(range(6)))
    seen = []
    pipelines.ViewPipelines(1, dev).run(lambda t, i: seen.append((t, i)), [7, 8])
    assert seen == [(0, 7), (0, 8)]

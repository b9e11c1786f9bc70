"""GPU: fused producers of rasterizer inputs against vectors made with the reference's own eval_sh
(tests/golden/producers_kat.npz) and, at config-D size, against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
KAT = os.path.join(ROOT, "tests", "golden", "producers_kat.npz")


@pytest.mark.parametrize("lv", [1, 2, 3, 4, 5])   # 5 = degree 4, the highest eval_sh accepts
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


# ------------------------------------------------------------------ mesh-bound Gaussians
BARY6 = [[2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3], [1 / 6, 5 / 12, 5 / 12], [5 / 12, 1 / 6, 5 / 12],
         [5 / 12, 5 / 12, 1 / 6]]   # sugar_model.py:217-226


def _mesh_case(level, loose, seed, clamp):
    from gaustar_amd import scene
    v, f = scene.icosphere(level, radius=0.9, center=(0.0, 1.2, 0.0))
    g = torch.Generator().manual_seed(seed)
    v = torch.from_numpy(v).float() + 0.01 * torch.randn(v.shape[0], 3, generator=g)
    f = torch.from_numpy(f).long()
    N = f.shape[0] * 6
    d = dict(verts=v, faces=f, bary=torch.tensor(BARY6), raw_scales=torch.randn(N, 2, generator=g) * 0.4 - 4.0,
             raw_complex=torch.randn(N, 2, generator=g), thickness=3e-6,
             min_scale=(0.012 if clamp else None), max_scale=(0.03 if clamp else None),
             delta_t=(0.01 * torch.randn(N, 3, generator=g) if loose else None),
             delta_r=(torch.randn(N, 4, generator=g) * 0.3 + torch.tensor([1.0, 0, 0, 0]) if loose else None))
    w = dict(points=torch.randn(N, 3, generator=g), scaling=torch.randn(N, 3, generator=g), quats=torch.randn(N, 4, generator=g))
    return d, w


def _quat_R(q):
    from oracle import producers_oracle
    return producers_oracle.quaternion_to_matrix(q)


@pytest.mark.parametrize("loose,clamp,level", [(False, False, 2), (True, True, 2), (True, False, 4)])
def test_mesh_bound_gaussians_against_oracle(loose, clamp, level, hip_lib):
    """Forward (quaternions compared through R(q): q and -q are the same rotation) and every gradient against
    autograd of the torch restatement of sugar_model.py:417-508, with a loss that only depends on R(q) -- the
    way the rasterizer consumes the quaternion."""
    from gaustar_amd import producers
    from oracle import producers_oracle
    d, w = _mesh_case(level, loose, 100 + level, clamp)
    names = ["verts", "raw_scales", "raw_complex"] + (["delta_t", "delta_r"] if loose else [])

    def run(fn, dev):
        args = {k: (v.clone().to(dev).requires_grad_(k in names) if torch.is_tensor(v) else v) for k, v in d.items()}
        p, s, q = fn(args["verts"], args["faces"], args["bary"], args["raw_scales"], args["raw_complex"], args["thickness"],
                     args["min_scale"], args["max_scale"], args["delta_t"], args["delta_r"])
        W3 = torch.randn(3, 3, generator=torch.Generator().manual_seed(1)).to(dev)
        # the rasterizer sees q only through R(q) (polynomial in q, no normalisation inside)
        r, i, j, k = q.unbind(-1)
        R = torch.stack((1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r), 2 * (i * j + k * r),
                         1 - 2 * (i * i + k * k), 2 * (j * k - i * r), 2 * (i * k - j * r), 2 * (j * k + i * r),
                         1 - 2 * (i * i + j * j)), -1).reshape(-1, 3, 3)
        loss = (p * w["points"].to(dev)).sum() + (s * w["scaling"].to(dev)).sum() + ((R @ W3) * w["quats"].to(dev)[:, :3, None]).sum()
        loss.backward()
        return p.detach().cpu(), s.detach().cpu(), q.detach().cpu(), {k: args[k].grad.cpu() for k in names}

    p0, s0, q0, g0 = run(producers_oracle.mesh_bound_gaussians, "cpu")
    p1, s1, q1, g1 = run(producers.mesh_bound_gaussians, "cuda")
    np.testing.assert_allclose(p1.numpy(), p0.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(s1.numpy(), s0.numpy(), rtol=2e-6, atol=0)
    assert np.abs(q1.norm(dim=-1).numpy() - 1).max() < 1e-6
    np.testing.assert_allclose(_quat_R(q1).numpy(), _quat_R(q0).numpy(), rtol=0, atol=3e-6)
    assert (np.abs((q1 * q0).sum(-1).numpy()) > 1 - 1e-5).all()
    for k in names:
        a, b = g1[k].numpy().astype(np.float64), g0[k].numpy().astype(np.float64)
        err = np.abs(a - b).max() / np.abs(b).max()
        assert err < 2e-4, f"{k}: normalised max error {err:.3e}"


def test_mesh_bound_closed_form_properties(hip_lib):
    """What SURVEY.md 8c asks of this producer: the rotation's first axis is the face normal, the other two span
    the face plane, means lie in the face plane (strict binding), thickness is the first scale."""
    from gaustar_amd import producers
    d, _ = _mesh_case(3, False, 7, False)
    dev = "cuda"
    p, s, q = producers.mesh_bound_gaussians(d["verts"].to(dev), d["faces"].to(dev), d["bary"].to(dev), d["raw_scales"].to(dev),
                                             d["raw_complex"].to(dev), d["thickness"])
    R = _quat_R(q.cpu())
    fv = d["verts"][d["faces"]]
    n = torch.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=-1)
    n = (n / n.norm(dim=-1, keepdim=True)).repeat_interleave(6, 0)
    assert (R[:, :, 0] - n).abs().max() < 1e-5
    assert (R[:, :, 1] * n).sum(-1).abs().max() < 1e-5 and (R[:, :, 2] * n).sum(-1).abs().max() < 1e-5
    assert (R.transpose(1, 2) @ R - torch.eye(3)).abs().max() < 1e-5 and (torch.det(R) - 1).abs().max() < 1e-5
    assert (((p.cpu() - fv[:, 0].repeat_interleave(6, 0)) * n).sum(-1)).abs().max() < 1e-6
    assert (s[:, 0].cpu() == torch.tensor(3e-6)).all() and (s[:, 1:].cpu() > 0).all()


@pytest.mark.parametrize("lv", [1, 4])
def test_points_rgb_depth_equals_rgb_plus_view_depth(lv, hip_lib):
    """points_rgb_depth [P,6] = {points_rgb (pinned above by the reference's eval_sh), z_view x 3} with
    z_view = the reference's `transform_points(points)[..., 2:]` (refine.py:603), stated here in float64 from the same
    view matrix the rasterizer gets; gradients: dL_dsh from the rgb half, dL_dpos = rgb part + (g3 + g4 + g5) * column 2."""
    from gaustar_amd import producers, scene
    z = np.load(KAT)
    k = f"l{lv}"
    cam = scene.look_at_camera((0.3, -0.2, -4.0), (0.1, 0.0, 0.0), 64, 48, fovx=0.7, znear=0.01)
    view = torch.from_numpy(np.ascontiguousarray(cam.viewmatrix, dtype=np.float32)).cuda()
    campos = torch.from_numpy(z[f"{k}_cam"]).cuda()
    rng = np.random.default_rng(lv)
    P = z[f"{k}_pos"].shape[0]
    g6 = torch.from_numpy(np.concatenate([z[f"{k}_dL"], rng.normal(size=(P, 3)).astype(np.float32)], 1)).cuda()
    pos = torch.from_numpy(z[f"{k}_pos"]).cuda().requires_grad_(True)
    sh = torch.from_numpy(z[f"{k}_sh"]).cuda().requires_grad_(True)
    out = producers.points_rgb_depth(pos, campos, sh, lv, view)
    assert tuple(out.shape) == (P, 6)
    out.backward(g6)
    o = out.detach().cpu().numpy()
    np.testing.assert_allclose(o[:, :3], z[f"{k}_colors"], rtol=1e-5, atol=2e-6)
    V = cam.viewmatrix.astype(np.float64)
    zv = z[f"{k}_pos"].astype(np.float64) @ V[:3, 2] + V[3, 2]
    for c in (3, 4, 5):
        np.testing.assert_allclose(o[:, c], zv, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(sh.grad.cpu().numpy(), z[f"{k}_dsh"], rtol=1e-5, atol=2e-6)
    dpos = z[f"{k}_dpos"].astype(np.float64) + g6[:, 3:].sum(1, keepdim=True).cpu().numpy().astype(np.float64) * V[:3, 2][None]
    np.testing.assert_allclose(pos.grad.cpu().numpy(), dpos, rtol=1e-4, atol=3e-5)
    with pytest.raises(RuntimeError, match=r"\(4, 4\)"):
        producers.points_rgb_depth(pos, campos, sh, lv, view[:3])


@pytest.mark.parametrize("G", [1, 3, 8, 10])
def test_mesh_bound_gaussians_other_counts_per_face(G, hip_lib):
    """GauSTAR binds 1, 3, 4 or 6 Gaussians to a triangle (sugar_model.py:217-226); up to 8 take the lane-per-Gaussian
    kernels, more than 8 the per-face loop.  Forward and vertex / parameter gradients against the torch restatement."""
    from gaustar_amd import producers, scene
    from oracle import producers_oracle
    g = torch.Generator().manual_seed(40 + G)
    v, f = scene.icosphere(2, radius=0.9, center=(0.0, 1.2, 0.0))
    v = torch.from_numpy(v).float() + 0.01 * torch.randn(v.shape, generator=g)
    f = torch.from_numpy(f).long()
    N = f.shape[0] * G
    bary = torch.rand(G, 3, generator=g) + 0.1
    bary = bary / bary.sum(-1, keepdim=True)
    rs, rc = torch.randn(N, 2, generator=g) * 0.4 - 4.0, torch.randn(N, 2, generator=g)
    dt, dr = 0.01 * torch.randn(N, 3, generator=g), torch.randn(N, 4, generator=g) * 0.3 + torch.tensor([1.0, 0, 0, 0])
    wp, wq = torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
    W3 = torch.randn(3, 3, generator=g)

    def run(fn, dev):
        a = [t.clone().to(dev).requires_grad_(True) for t in (v, rs, rc, dt, dr)]
        p, s, q = fn(a[0], f.to(dev), bary.to(dev), a[1], a[2], 3e-6, None, None, a[3], a[4])
        r, i, j, k = q.unbind(-1)
        R = torch.stack((1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r), 2 * (i * j + k * r),
                         1 - 2 * (i * i + k * k), 2 * (j * k - i * r), 2 * (i * k - j * r), 2 * (j * k + i * r),
                         1 - 2 * (i * i + j * j)), -1).reshape(-1, 3, 3)
        ((p * wp.to(dev)).sum() + (s ** 2).sum() + ((R @ W3.to(dev)) * wq.to(dev)[:, :, None]).sum()).backward()
        return p.detach().cpu(), s.detach().cpu(), [t.grad.cpu() for t in a]

    p0, s0, g0 = run(producers_oracle.mesh_bound_gaussians, "cpu")
    p1, s1, g1 = run(producers.mesh_bound_gaussians, "cuda")
    np.testing.assert_allclose(p1.numpy(), p0.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(s1.numpy(), s0.numpy(), rtol=2e-6, atol=0)
    for name, a, b in zip(("verts", "raw_scales", "raw_complex", "delta_t", "delta_r"), g1, g0):
        err = np.abs(a.numpy().astype(np.float64) - b.numpy()).max() / max(np.abs(b.numpy()).max(), 1e-30)
        assert err < 2e-4, f"G={G} {name}: normalised max error {err:.3e}"


def test_hip_mesh_producer_against_scipy_float64(hip_lib):
    """The fused HIP producer (gsr_mesh_gaussians) against the independent float64 numpy + scipy construction of
    tests/crosscheck.py -- a cross-check, not a pin: pytorch3d, whose conventions the reference calls, is absent here."""
    import crosscheck
    from gaustar_amd import harness, producers, scene
    rng = np.random.default_rng(12)
    v, f = scene.icosphere(3, 0.9, (0.0, 1.2, 0.0))
    G = 6
    N = len(f) * G
    bary = np.asarray(harness.BARY_COORDS[G], np.float64)
    raw_scales = rng.normal(-5.0, 0.3, (N, 2)); raw_complex = rng.normal(size=(N, 2))
    delta_t = 1e-3 * rng.normal(size=(N, 3)); delta_r = np.array([1.0, 0, 0, 0]) + 0.05 * rng.normal(size=(N, 4))
    dev = torch.device("cuda:0")
    t = lambda x: None if x is None else torch.from_numpy(np.asarray(x, np.float32)).to(dev)
    for dt, dr in ((None, None), (delta_t, delta_r)):
        pts, scl, quat = producers.mesh_bound_gaussians(t(v), torch.from_numpy(f).long().to(dev), t(bary), t(raw_scales), t(raw_complex),
                                                        3e-6, None, None, t(dt), t(dr))
        p64, s64, R64 = crosscheck.mesh_frames_f64(v, f, bary, raw_scales, raw_complex, 3e-6, dt, dr)
        np.testing.assert_allclose(pts.cpu().numpy(), p64, atol=2e-6)
        np.testing.assert_allclose(scl.cpu().numpy(), s64, rtol=2e-6)
        q = quat.cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)
        np.testing.assert_allclose(crosscheck.quat_wxyz_to_matrix(q), R64, atol=5e-6)


@pytest.mark.parametrize("lv,n_rest,view_on,dens_on", [(1, 0, False, False), (3, 8, True, True), (4, 15, True, False), (2, 15, False, True),
                                                       (5, 24, True, True)])
def test_points_colors_split_is_bit_identical_to_the_one_array_path(lv, n_rest, view_on, dens_on, hip_lib):
    """producers.points_colors_split reads `_sh_coordinates_dc` / `_sh_coordinates_rest` where SuGaR keeps them and takes
    sigmoid(all_densities) along: same bits as points_rgb / points_rgb_depth on torch.cat of the two + torch.sigmoid,
    forward and backward (ragged P: the last wave's rows end inside a 64-row block)."""
    from gaustar_amd import producers
    g = torch.Generator(device="cuda").manual_seed(11 * lv + n_rest)
    P = 64 * 37 + 29
    pos = (torch.randn(P, 3, device="cuda", generator=g) * 0.5 + torch.tensor([0.0, 1.2, 0.0], device="cuda")).requires_grad_(True)
    dc = (torch.rand(P, 1, 3, device="cuda", generator=g) * 2 - 1).requires_grad_(True)
    rest = (0.3 * torch.randn(P, n_rest, 3, device="cuda", generator=g)).requires_grad_(True)
    dens = (2.0 * torch.randn(P, 1, device="cuda", generator=g)).requires_grad_(True)
    campos = torch.tensor([0.4, 1.5, 3.0], device="cuda")
    view = torch.eye(4, device="cuda"); view[:3, :3] = torch.linalg.qr(torch.randn(3, 3, device="cuda", generator=g))[0]; view[3, :3] = torch.tensor([0.1, -1.0, 3.0])
    C = 3 + (1 if view_on else 0)
    d_col = torch.randn(P, C, device="cuda", generator=g)
    d_op = torch.randn(P, 1, device="cuda", generator=g)

    out = producers.points_colors_split(pos, campos, dc, rest, lv, view if view_on else None, 1, dens if dens_on else None)
    col, op = out if dens_on else (out, None)
    loss = (col * d_col).sum() + ((op * d_op).sum() if dens_on else 0.0)
    loss.backward()
    got = [t.grad.clone() for t in (pos, dc, rest)] + ([dens.grad.clone()] if dens_on else [])
    for t in (pos, dc, rest, dens):
        t.grad = None

    sh = torch.cat([dc, rest], 1)
    col_ref = producers.points_rgb_depth(pos, campos, sh, lv, view, depth_channels=1) if view_on else producers.points_rgb(pos, campos, sh, lv)
    op_ref = torch.sigmoid(dens) if dens_on else None
    loss = (col_ref * d_col).sum() + ((op_ref * d_op).sum() if dens_on else 0.0)
    loss.backward()
    want = [t.grad for t in (pos, dc, rest)] + ([dens.grad] if dens_on else [])
    assert torch.equal(col, col_ref)
    if dens_on:
        assert torch.equal(op, op_ref)
    for a, b, n in zip(got, want, ("positions", "sh_dc", "sh_rest", "densities")):
        assert torch.equal(a, b), n


def test_points_colors_split_validation(hip_lib):
    from gaustar_amd import producers
    pos = torch.zeros(8, 3, device="cuda"); dc = torch.zeros(8, 1, 3, device="cuda"); rest = torch.zeros(8, 3, 3, device="cuda")
    cam = torch.zeros(3, device="cuda")
    with pytest.raises(RuntimeError, match="sh_levels"):
        producers.points_colors_split(pos, cam, dc, rest, 3)                       # 9 coefficients needed, 4 given
    with pytest.raises(RuntimeError, match="sh_dc must be"):
        producers.points_colors_split(pos, cam, dc[:, 0], rest, 2)
    with pytest.raises(RuntimeError, match="densities"):
        producers.points_colors_split(pos, cam, dc, rest, 2, None, 1, torch.zeros(7, 1, device="cuda"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        producers.points_colors_split(pos.cpu(), cam, dc, rest, 2)


def test_mesh_producer_backward_twice_over_one_graph(hip_lib):
    """ABI 14: the forward's launch clears the vertex-gradient accumulator of the backward to come (no fill in front of it); the
    cleared buffer is good for ONE backward -- a second pass over a retained graph takes a fresh one and gives the same gradient."""
    from gaustar_amd import producers, scene
    dev = torch.device("cuda:0")
    v, f = scene.icosphere(2, 1.0, (0.0, 0.0, 0.0))
    verts = torch.from_numpy(v).float().to(dev).requires_grad_(True)
    faces = torch.from_numpy(f).long().to(dev)
    G = 3
    gen = torch.Generator(device=dev).manual_seed(3)
    bary = torch.rand(G, 3, device=dev, generator=gen); bary = bary / bary.sum(1, keepdim=True)
    N = faces.size(0) * G
    rs = torch.randn(N, 2, device=dev, generator=gen).requires_grad_(True)
    rc = torch.randn(N, 2, device=dev, generator=gen).requires_grad_(True)
    pts, sc, qu = producers.mesh_bound_gaussians(verts, faces, bary, rs, rc, 1e-3)
    wp, ws, wq = (torch.randn(t.shape, device=dev, generator=gen) for t in (pts, sc, qu))
    loss = (pts * wp).sum() + (sc * ws).sum() + (qu * wq).sum()
    g1 = torch.autograd.grad(loss, (verts, rs, rc), retain_graph=True)
    g2 = torch.autograd.grad(loss, (verts, rs, rc))
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7) and float(a.abs().max()) > 0
    # ... and a fresh graph (its own pre-cleared accumulator) agrees with both
    pts, sc, qu = producers.mesh_bound_gaussians(verts, faces, bary, rs, rc, 1e-3)
    g3 = torch.autograd.grad((pts * wp).sum() + (sc * ws).sum() + (qu * wq).sum(), (verts, rs, rc))
    for a, b in zip(g1, g3):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)

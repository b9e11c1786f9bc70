"""Parity tests proper (-m gpu): the HIP path, called through the public Python API -> C ABI,
against (1) the committed golden vectors of the reference, (2) the C oracle run on the host CPU on
seeded inputs, (3) -- when oracle/_ref travelled to the box -- the reference build itself at
BASELINE.json's full sizes, plus (4) size-independent properties at full size and the edge cases.
Tolerances: parity.py (1e-4, fp32)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import parity
from conftest import ROOT, golden_names

pytestmark = pytest.mark.gpu


def _kw(gs, cam, bg, sm=1.0):
    return dict(means3D=gs.means3D, opacities=gs.opacities, view=cam.viewmatrix, proj=cam.projmatrix,
                campos=cam.campos, W=cam.W, H=cam.H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, shs=gs.shs,
                colors_precomp=gs.colors_precomp, scales=gs.scales, rotations=gs.rotations,
                cov3D_precomp=gs.cov3D_precomp, sh_degree=gs.sh_degree, scale_modifier=sm)


@pytest.fixture(scope="module", autouse=True)
def _native_loaded(hip_lib):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return hip_lib


# ------------------------------------------------------------------ (1) golden vectors of the reference
@pytest.mark.parametrize("name", golden_names())
def test_hip_matches_reference_golden(name):
    kw, d = parity.load_golden(name)
    hip = parity.run_hip(kw, d["in_dL_dpix"])
    ref_grads = {k: d["grad_" + k] for k in parity.GRAD_KEYS}
    parity.compare_hip_to(hip, d["out_color"], d["out_radii"], ref_grads, what=name)


@pytest.mark.parametrize("name", ["edge_cases", "mesh_sphere", "sh3_random"])
def test_hip_internal_state_matches_reference(name, hip_lib):
    """Stage-level parity: projected means, conics, depths, SH colours, per-pixel transmittance, and
    the depth-sorted per-tile lists (compared as lists of contributing Gaussians per tile, since this
    library bins a subset of the reference's instances -- only ones that can reach alpha >= 1/255)."""
    import torch
    from gaustar_amd import rasterizer as R
    kw, d = parity.load_golden(name)
    dev = torch.device("cuda:0")
    t = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    e = torch.Tensor([])
    P, W, H = len(kw["means3D"]), kw["W"], kw["H"]
    opt = lambda k: t(kw[k]) if kw.get(k) is not None else e
    out = R.rasterize_gaussians_native(t(kw["bg"]), t(kw["means3D"]), opt("colors_precomp"), t(kw["opacities"]),
                                       opt("scales"), opt("rotations"), kw["scale_modifier"], opt("cov3D_precomp"),
                                       t(kw["view"]), t(kw["proj"]), kw["tanfovx"], kw["tanfovy"], H, W, opt("shs"),
                                       kw["sh_degree"], t(kw["campos"]), False, False,
                                       use_plan=False)   # (the checks below read the exact path's compact lists and unit numbering)
    Rn, color, radii, geom, binning, img, maxc, nseg = out
    assert 0 < Rn <= int(d["out_num_rendered"]) and 0 < maxc <= Rn
    T = ((W + 15) // 16) * ((H + 15) // 16)
    m2 = torch.zeros(P, 2, device=dev); co = torch.zeros(P, 4, device=dev); dp = torch.zeros(P, device=dev)
    rgb = torch.zeros(P, 3, device=dev)
    rng_ = torch.zeros(T, 2, dtype=torch.int32, device=dev); pl = torch.zeros(max(Rn, 1), dtype=torch.int32, device=dev)
    fT = torch.zeros(H, W, device=dev); nc = torch.zeros(H, W, dtype=torch.int32, device=dev)
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    rc = hip_lib.gsr_debug_export(P, Rn, nseg, W, H, p(geom), p(binning), p(img), p(m2), p(co), p(dp),
                                  p(rgb) if kw["shs"] is not None else None, p(rng_), p(pl), p(fT), p(nc), None)
    assert rc == 0, hip_lib.gsr_last_error()
    torch.cuda.synchronize()
    vis = d["state_visible"] & (d["in_opacities"].reshape(-1) * 255.0 >= 1.0)
    np.testing.assert_allclose(m2.cpu().numpy()[vis], d["state_means2D"][vis], rtol=0, atol=2e-4)
    # depth is the sort key: bit-identical to the reference build (the blend order of near-coplanar surface
    # splats depends on single ulps of it)
    assert np.array_equal(dp.cpu().numpy()[vis].view(np.uint32), d["state_depths"][vis].view(np.uint32))
    b = d["state_conic_opacity"][vis]
    np.testing.assert_allclose(co.cpu().numpy()[vis], b, rtol=2e-4, atol=1e-6 * max(1.0, float(np.abs(b).max())))
    if kw["shs"] is not None:
        np.testing.assert_allclose(rgb.cpu().numpy()[vis], d["state_rgb"][vis], rtol=0, atol=2e-6)
    parity.check_image(fT.cpu().numpy(), d["state_final_T"], name + " final_T")
    # sorted lists: ours must be a subsequence of the reference's per tile, in the same order
    ours_r, ours_l = rng_.cpu().numpy().astype(np.int64), pl.cpu().numpy().astype(np.int64)
    ref_r, ref_l = d["state_ranges"].astype(np.int64), d["state_point_list"].astype(np.int64)
    assert ours_r[-1, 1] == Rn and (ours_r[1:, 0] == ours_r[:-1, 1]).all()
    depth = d["state_depths"]
    for tile in range(T):
        a = ours_l[ours_r[tile, 0]:ours_r[tile, 1]]
        bb = ref_l[ref_r[tile, 0]:ref_r[tile, 1]]
        assert (np.diff(depth[a]) >= 0).all(), f"tile {tile}: list not depth sorted"
        it = iter(bb.tolist())
        assert all(x in it for x in a.tolist()), f"tile {tile}: not an ordered subset of the reference list"


# ------------------------------------------------------------------ (2) C oracle on the host, seeded inputs
@pytest.mark.parametrize("seed,P,W,H,deg", [(1, 4000, 256, 192, 0), (2, 3000, 200, 200, 3), (3, 6000, 333, 127, 0)])
def test_hip_matches_oracle_seeded(seed, P, W, H, deg):
    from gaustar_amd import scene
    rng = np.random.default_rng(seed)
    gs = scene.random_gaussians(P, rng, sh_degree=deg, with_sh=deg > 0, scale_range=(0.01, 0.15))
    cam = scene.look_at_camera((0.3, -0.2, -4.0), (0, 0, 0), W, H, fovx=0.9, znear=0.01)
    kw = _kw(gs, cam, np.array([0.2, 0.4, 0.6], np.float32))
    dpix = rng.normal(size=(3, H, W)).astype(np.float32)
    st, g = parity.run_oracle(kw, dpix)
    hip = parity.run_hip(kw, dpix)
    # random scenes: the threshold-flip allowance of parity.py applies (not needed so far; reported in the log)
    parity.compare_hip_to(hip, st["color"], st["radii"], g, what=f"seed{seed}", kw=kw, max_radii_flips=1, strict=False)


def test_config_A_against_oracle():
    """BASELINE.json configs[0]: 10k random Gaussians, 1 cam @512x512, SH deg 0."""
    from gaustar_amd import scene
    gs, cam, bg = scene.config_A()
    kw = _kw(gs, cam, bg)
    dpix = np.random.default_rng(0).normal(size=(3, cam.H, cam.W)).astype(np.float32)
    st, g = parity.run_oracle(kw, dpix)
    hip = parity.run_hip(kw, dpix)
    parity.compare_hip_to(hip, st["color"], st["radii"], g, what="config A", kw=kw, max_radii_flips=1)


# ------------------------------------------------------------------ (3) the reference build itself, full sizes
def _ref_or_skip():
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/libgsr_ref.so did not travel to this box")
    return ref


@pytest.mark.parametrize("cfg", ["B", "C", "C:3", "C:90", "C:141", "D", "D_depth"])
def test_full_size_configs_against_reference_build(cfg):
    """BASELINE.json configs[1..3] at full size: 200k / 491k / 1M mesh-bound Gaussians @1080p (config C: views from four
    of the rig's five rings)."""
    ref = _ref_or_skip()
    from gaustar_amd import scene
    if cfg == "B":
        gs, cam, bg = scene.config_B()
    elif cfg.startswith("C"):
        gs, cams, bg = scene.config_C()
        cam = cams[int(cfg.split(":")[1]) if ":" in cfg else 37]
    else:
        gs, cam, bg = scene.config_D()
        if cfg == "D_depth":          # refine.py:603-607 second pass: depth as colour, bg = 10
            gs.shs, gs.sh_degree = None, 0
            gs.colors_precomp = scene.view_depth_colors(gs, cam)
            bg = np.array([10.0, 10.0, 10.0], np.float32)
    kw = _kw(gs, cam, bg)
    dpix = np.random.default_rng(5).normal(size=(3, cam.H, cam.W)).astype(np.float32)
    rr = ref.RefRasterizer()
    color, radii, R = rr.forward(**kw)
    g = rr.backward(dpix)
    hip = parity.run_hip(kw, dpix)
    radii = radii.cpu().numpy()
    # full size: a few dozen threshold-flip elements of millions (parity.FULL_*_OUTLIERS); radii may differ only by
    # ceil(3 sigma) flips at an integer boundary
    parity.compare_hip_to(hip, color.cpu().numpy(), radii, {k: v.cpu().numpy() for k, v in g.items()}, what=cfg, kw=kw,
                          max_radii_flips=max(2, gs.P // 100_000), strict=False, img_outliers=parity.FULL_IMG_OUTLIERS,
                          grad_outliers=parity.FULL_GRAD_OUTLIERS)


def test_mark_visible_matches_reference_build():
    ref = _ref_or_skip()
    import torch
    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer
    kw, d = parity.load_golden("edge_cases")
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    s = GaussianRasterizationSettings(kw["H"], kw["W"], kw["tanfovx"], kw["tanfovy"], t(kw["bg"]), 1.0, t(kw["view"]),
                                      t(kw["proj"]), 0, t(kw["campos"]), False, False)
    ours = GaussianRasterizer(s).markVisible(t(kw["means3D"]))
    theirs = ref.mark_visible(kw["means3D"], kw["view"], kw["proj"])
    assert ours.dtype == torch.bool and torch.equal(ours, theirs) and 0 < int(ours.sum()) < len(ours)


# ------------------------------------------------------------------ (4) size-independent properties, full size
def test_full_size_properties():
    """Config C geometry @1080p: determinism of the forward, background linearity
    (C(bg1) - C(bg2) = T_final * (bg1 - bg2)), linearity of the backward in dL_dpix."""
    from gaustar_amd import scene
    gs, cams, bg = scene.config_C()
    cam = cams[5]
    kw = _kw(gs, cam, bg)
    rng = np.random.default_rng(0)
    dpix = rng.normal(size=(3, cam.H, cam.W)).astype(np.float32)
    a = parity.run_hip(kw, dpix)
    b = parity.run_hip(kw, dpix)
    assert np.array_equal(a["color"], b["color"]) and np.array_equal(a["radii"], b["radii"])
    for k in parity.GRAD_KEYS:
        parity.check_grad(a[k], b[k], "rerun " + k, tol=2e-6, max_outlier_frac=0.0)
    kw2 = dict(kw, bg=np.array([0.7, 0.1, 0.4], np.float32))
    c = parity.run_hip(kw2)
    dbg = (kw2["bg"] - kw["bg"]).reshape(3, 1, 1)
    Tf = (c["color"] - a["color"])[0] / dbg[0]
    assert Tf.min() > -1e-5 and Tf.max() < 1 + 1e-5
    np.testing.assert_allclose(c["color"] - a["color"], Tf[None] * dbg, atol=3e-6)
    assert (Tf > 0.999).mean() > 0.3 and (Tf < 0.01).mean() > 0.1       # empty background and opaque subject
    h = parity.run_hip(kw, 2.0 * dpix)
    for k in parity.GRAD_KEYS:
        parity.check_grad(h[k], 2.0 * a[k], "linearity " + k, tol=1e-5, max_outlier_frac=0.0)


# ------------------------------------------------------------------ edge cases and ABI variants
def test_empty_and_all_culled():
    import torch
    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    kw, _ = parity.load_golden("colors_rgb")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    s = GaussianRasterizationSettings(40, 56, kw["tanfovx"], kw["tanfovy"], t(kw["bg"]), 1.0, t(kw["view"]),
                                      t(kw["proj"]), 0, t(kw["campos"]), False, True)
    r = GaussianRasterizer(s)
    z = lambda *sh: torch.zeros(*sh, device=dev, requires_grad=True)
    # P == 0: zero image (NOT background), like rasterize_points.cu:68-81
    color, radii = r(z(0, 3), z(0, 3), z(0, 1), colors_precomp=z(0, 3), scales=z(0, 3), rotations=z(0, 4))
    assert color.shape == (3, 40, 56) and not color.any() and radii.numel() == 0
    # everything behind the camera: background everywhere, zero gradients
    m = torch.tensor(np.tile([[0.0, 0.0, -10.0]], (50, 1)), dtype=torch.float32, device=dev, requires_grad=True)
    o, c = torch.full((50, 1), 0.5, device=dev, requires_grad=True), torch.rand(50, 3, device=dev, requires_grad=True)
    sc, q = torch.full((50, 3), 0.1, device=dev, requires_grad=True), torch.tensor([[1.0, 0, 0, 0]] * 50, device=dev, requires_grad=True)
    color, radii = r(m, z(50, 3), o, colors_precomp=c, scales=sc, rotations=q)
    assert not radii.any()
    np.testing.assert_allclose(color.detach().cpu().numpy(), np.broadcast_to(kw["bg"].reshape(3, 1, 1), (3, 40, 56)))
    color.sum().backward()
    for x in (m, o, c, sc, q):
        assert x.grad is not None and not x.grad.any()


def test_callback_forward_equals_staged(hip_lib):
    """gsr_forward (allocator callbacks, the reference's Rasterizer::forward shape) == stage1 + stage2."""
    import torch
    from gaustar_amd import _lib
    kw, d = parity.load_golden("mesh_sphere")
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    P, W, H = len(kw["means3D"]), kw["W"], kw["H"]
    keep = []

    def alloc(ctx, n):
        buf = torch.empty(max(int(n), 1), dtype=torch.uint8, device=dev)
        keep.append(buf)
        return buf.data_ptr()

    cb = _lib.ALLOC_FN(alloc)
    ten = {k: t(kw[k]) for k in ("bg", "means3D", "colors_precomp", "opacities", "scales", "rotations", "view", "proj", "campos")}
    p = lambda k: ctypes.c_void_p(ten[k].data_ptr())
    out, radii, R = torch.empty(3, H, W, device=dev), torch.empty(P, dtype=torch.int32, device=dev), ctypes.c_int(0)
    rc = hip_lib.gsr_forward(cb, cb, cb, None, P, 0, 0, p("bg"), W, H, p("means3D"), None, p("colors_precomp"),
                             p("opacities"), p("scales"), 1.0, p("rotations"), None, p("view"), p("proj"), p("campos"),
                             kw["tanfovx"], kw["tanfovy"], 0, ctypes.c_void_p(out.data_ptr()),
                             ctypes.c_void_p(radii.data_ptr()), ctypes.byref(R), None)
    assert rc == 0, hip_lib.gsr_last_error()
    torch.cuda.synchronize()
    assert len(keep) == 3 and 0 < R.value <= int(d["out_num_rendered"])
    staged = parity.run_hip(kw)
    assert np.array_equal(out.cpu().numpy(), staged["color"]) and np.array_equal(radii.cpu().numpy(), staged["radii"])


def test_sort_fallback_path_gives_identical_images():
    """Tiles longer than the LDS sort capacity take the global-memory network; force it with a tiny cap."""
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import parity;"
            "kw, d = parity.load_golden('colors_rgb'); h = parity.run_hip(kw, d['in_dL_dpix']);"
            "parity.compare_hip_to(h, d['out_color'], d['out_radii'], {k: d['grad_' + k] for k in parity.GRAD_KEYS}, 'fallback');"
            "print('FALLBACK_OK')") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, GSR_DEBUG_SORT_CAP="64")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "FALLBACK_OK" in r.stdout, r.stdout + r.stderr


def test_long_tile_lists_use_the_big_lds_sort():
    """> 2048 instances in one tile: 3000 splats stacked on the same pixel block."""
    from gaustar_amd import scene
    rng = np.random.default_rng(4)
    gs = scene.random_gaussians(3000, rng, scale_range=(0.02, 0.05), box=((-0.05, 0.05), (-0.05, 0.05), (-0.5, 0.5)))
    gs.opacities[:] = rng.uniform(0.004, 0.02, (3000, 1)).astype(np.float32)
    cam = scene.look_at_camera((0, 0, -4.0), (0, 0, 0), 64, 64, fovx=0.5, znear=0.01)
    kw = _kw(gs, cam, np.zeros(3, np.float32))
    dpix = rng.normal(size=(3, 64, 64)).astype(np.float32)
    st, g = parity.run_oracle(kw, dpix)
    assert (st["ranges"][:, 1] - st["ranges"][:, 0]).max() > 2048
    hip = parity.run_hip(kw, dpix)
    parity.compare_hip_to(hip, st["color"], st["radii"], g, what="long lists")


# ------------------------------------------------------------------ robustness: sizes, streams, extremes
def _compare_with_ref_or_oracle(kw, dpix, what, radii_slack=1, full_size=False):
    from oracle import ref
    hip = parity.run_hip(kw, dpix)
    if ref.available():
        rr = ref.RefRasterizer()
        color, radii, _ = rr.forward(**kw)
        g = {k: v.cpu().numpy() for k, v in rr.backward(dpix).items()}
        color, radii = color.cpu().numpy(), radii.cpu().numpy()
    else:
        st, g = parity.run_oracle(kw, dpix)
        color, radii = st["color"], st["radii"]
    loose = dict(strict=False, img_outliers=parity.FULL_IMG_OUTLIERS, grad_outliers=parity.FULL_GRAD_OUTLIERS) if full_size else {}
    parity.compare_hip_to(hip, color, radii, g, what=what, kw=kw, max_radii_flips=radii_slack, **loose)


def test_4k_image_many_tiles():
    """3840x2160 = 32 400 tiles: exercises the wide tile-scan variant and 16-bit tile rects."""
    from gaustar_amd import scene
    rng = np.random.default_rng(21)
    gs = scene.random_gaussians(60_000, rng, scale_range=(0.004, 0.05))
    cam = scene.look_at_camera((0.2, 0.1, -4.0), (0, 0, 0), 3840, 2160, fovx=0.9, znear=0.01)
    kw = _kw(gs, cam, np.array([0.1, 0.2, 0.3], np.float32))
    dpix = rng.normal(size=(3, cam.H, cam.W)).astype(np.float32)
    _compare_with_ref_or_oracle(kw, dpix, "4k", full_size=True)


@pytest.mark.parametrize("W,H", [(1, 1), (17, 33), (16, 16), (250, 9)])
def test_tiny_and_odd_image_sizes(W, H):
    from gaustar_amd import scene
    rng = np.random.default_rng(W * 100 + H)
    gs = scene.random_gaussians(200, rng, scale_range=(0.05, 0.4))
    cam = scene.look_at_camera((0, 0, -4.0), (0, 0, 0), W, H, fovx=0.8, znear=0.01)
    kw = _kw(gs, cam, np.array([0.9, 0.1, 0.5], np.float32))
    dpix = rng.normal(size=(3, H, W)).astype(np.float32)
    _compare_with_ref_or_oracle(kw, dpix, f"{W}x{H}")


def test_screen_filling_splats_and_giant_tile_lists():
    """Splats that cover every tile (rect = whole grid) and > 16 384 instances per tile (the
    global-memory sort network), at a size the reference build finishes quickly."""
    from gaustar_amd import scene
    rng = np.random.default_rng(33)
    gs = scene.random_gaussians(18_000, rng, scale_range=(0.6, 1.5), box=((-0.3, 0.3), (-0.3, 0.3), (-0.5, 0.5)))
    gs.opacities[:] = rng.uniform(0.004, 0.012, (gs.P, 1)).astype(np.float32)   # keep every pixel unsaturated: deep lists
    cam = scene.look_at_camera((0, 0, -4.0), (0, 0, 0), 96, 64, fovx=0.6, znear=0.01)
    kw = _kw(gs, cam, np.zeros(3, np.float32))
    dpix = rng.normal(size=(3, cam.H, cam.W)).astype(np.float32)
    _compare_with_ref_or_oracle(kw, dpix, "giant lists")


def test_non_default_stream_and_degree_validation():
    import torch
    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    kw, d = parity.load_golden("sh2_random")
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    mk = lambda deg: GaussianRasterizationSettings(kw["H"], kw["W"], kw["tanfovx"], kw["tanfovy"], t(kw["bg"]), 1.0,
                                                   t(kw["view"]), t(kw["proj"]), deg, t(kw["campos"]), False, False)
    args = dict(means3D=t(kw["means3D"]), means2D=torch.zeros(len(kw["means3D"]), 3, device=dev),
                opacities=t(kw["opacities"]), shs=t(kw["shs"]), scales=t(kw["scales"]), rotations=t(kw["rotations"]))
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        color, radii = GaussianRasterizer(mk(2))(**args)
    side.synchronize()
    parity.check_image(color.cpu().numpy(), d["out_color"], "side stream")
    with pytest.raises(_lib.GsrError, match="sh degree"):
        GaussianRasterizer(mk(3))(**args)          # 9 coefficients cannot hold degree 3
    with pytest.raises(_lib.GsrError, match="sh degree"):
        GaussianRasterizer(mk(4))(**args)


def test_lists_spanning_many_chunks():
    """Tile lists of more than 1 000 entries: the blend kernels stage them in LDS several chunks at a time
    (gsr_blend_fwd.hip), pixels terminate inside different chunks (opaque case) or never (translucent case); images,
    internal state and gradients must still match the oracle."""
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import parity
from gaustar_amd import scene
for seed, opac in ((1, (0.6, 0.99)), (2, (0.02, 0.1))):
    rng = np.random.default_rng(seed)
    gs = scene.random_gaussians(6000, rng, scale_range=(0.03, 0.12), box=((-0.4, 0.4), (-0.3, 0.3), (-0.5, 0.5)))
    gs.opacities[:] = rng.uniform(*opac, (gs.P, 1)).astype(np.float32)
    cam = scene.look_at_camera((0.1, 0.0, -4.0), (0, 0, 0), 96, 80, fovx=0.5, znear=0.01)
    kw = dict(means3D=gs.means3D, opacities=gs.opacities, view=cam.viewmatrix, proj=cam.projmatrix, campos=cam.campos, W=cam.W, H=cam.H,
              tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=np.array([0.3, 0.1, 0.6], np.float32), shs=None, colors_precomp=gs.colors_precomp,
              scales=gs.scales, rotations=gs.rotations, cov3D_precomp=None, sh_degree=0, scale_modifier=1.0)
    dpix = rng.normal(size=(3, cam.H, cam.W)).astype(np.float32)
    st, g = parity.run_oracle(kw, dpix)
    assert (st["ranges"][:, 1] - st["ranges"][:, 0]).max() > 1000
    hip = parity.run_hip(kw, dpix)
    parity.compare_hip_to(hip, st["color"], st["radii"], g, what="long lists seed %d" % seed)
print("LONG_PATH_OK")
'''
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "LONG_PATH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_lists_blended_in_parts():
    """Lists above PART_FROM = 1 024 entries in a view that splits (GSR_SPLIT_FROM=0 makes every such view split; by default
    only views with a list above 1 792 entries do): the forward blends them in independent parts of one chunk -- part
    workers inside the blend's own launch -- and the part that finishes LAST (a ticket per tile, no waiting, no dispatch-order
    assumption) folds the parts together, pixels that may terminate inside a part walking their unit exactly
    (gsr_blend_fwd.hip).  Translucent stacks (nobody terminates), opaque ones (everybody does, in different parts), a mix, a
    list above 16 384 entries (four merge passes of the pre-sort), and a view whose longest list stays below 2 048 (the
    pre-sort writes the ids itself); against the oracle, and the no-grad forward must stay bit-identical to the
    differentiable one."""
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import parity
from gaustar_amd import scene, GaussianRasterizationSettings, GaussianRasterizer
for seed, opac, P, half in ((1, (0.004, 0.02), 9000, 0.12), (2, (0.3, 0.99), 9000, 0.12), (3, (0.01, 0.6), 12000, 0.12),
                            (4, (0.004, 0.01), 20000, 0.02),      # (one list above 16 384 entries -- four merge passes)
                            (5, (0.01, 0.6), 1700, 0.12)):        # (longest list between 1 024 and 2 048: no merge pass at all)
    rng = np.random.default_rng(seed)
    gs = scene.random_gaussians(P, rng, scale_range=(0.02, 0.06), box=((-half, half), (-half, half), (-0.5, 0.5)))
    gs.opacities[:] = rng.uniform(*opac, (gs.P, 1)).astype(np.float32)
    cam = scene.look_at_camera((0.05, 0.0, -4.0), (0, 0, 0), 80, 64, fovx=0.5, znear=0.01)
    kw = dict(means3D=gs.means3D, opacities=gs.opacities, view=cam.viewmatrix, proj=cam.projmatrix, campos=cam.campos, W=cam.W, H=cam.H,
              tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=np.array([0.3, 0.1, 0.6], np.float32), shs=None, colors_precomp=gs.colors_precomp,
              scales=gs.scales, rotations=gs.rotations, cov3D_precomp=None, sh_degree=0, scale_modifier=1.0)
    dpix = rng.normal(size=(3, cam.H, cam.W)).astype(np.float32)
    st, g = parity.run_oracle(kw, dpix)
    longest = int((st["ranges"][:, 1] - st["ranges"][:, 0]).max())
    assert (1024 + 256 < longest <= 2048) if seed == 5 else longest > (16384 if seed == 4 else 2048 + 512), longest
    hip = parity.run_hip(kw, dpix)
    parity.compare_hip_to(hip, st["color"], st["radii"], g, what="parts seed %d (longest list %d)" % (seed, longest))
    # no-grad forward == differentiable forward, bit for bit
    dev = torch.device("cuda:0")
    t = lambda x, rg=False: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev).requires_grad_(rg)
    s = GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, t(kw["bg"]), 1.0, t(cam.viewmatrix), t(cam.projmatrix), 0,
                                      t(cam.campos), False, False)
    args = dict(means2D=torch.zeros(gs.P, 3, device=dev), colors_precomp=t(gs.colors_precomp), scales=t(gs.scales), rotations=t(gs.rotations))
    img_g, _ = GaussianRasterizer(s)(means3D=t(gs.means3D, True), opacities=t(gs.opacities, True), **args)
    with torch.no_grad():
        img_n, _ = GaussianRasterizer(s)(means3D=t(gs.means3D), opacities=t(gs.opacities), **args)
    assert torch.equal(img_g.detach(), img_n), "forward-only render differs"
print("PARTS_OK")
'''
    env = dict(os.environ, GSR_SPLIT_FROM="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PARTS_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-4000:]


def test_hostile_inputs_neither_hang_nor_fault():
    """NaN / Inf / huge / degenerate parameters (tests/devtools/fuzz_inputs.py) must come back from forward + backward:
    no hang, no device fault.  (Values are garbage in the reference too and are not compared.)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "devtools", "fuzz_inputs.py")], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "FUZZ_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_forward_only_mode_skips_snapshots_but_not_pixels():
    """Under torch.no_grad() (or when no input requires grad) stage 2 gets the negated segment count and writes no
    per-segment snapshots; image and radii must be bit-identical to the differentiable call, and the C ABI refuses a
    backward on such a forward."""
    import ctypes
    import torch
    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, scene
    from gaustar_amd import rasterizer as R
    rng = np.random.default_rng(12)
    gs = scene.random_gaussians(5000, rng, scale_range=(0.03, 0.1), box=((-0.4, 0.4), (-0.3, 0.3), (-0.5, 0.5)))
    gs.opacities[:] = rng.uniform(0.02, 0.2, (gs.P, 1)).astype(np.float32)      # lists span several segments
    cam = scene.look_at_camera((0, 0, -4.0), (0, 0, 0), 128, 96, fovx=0.5, znear=0.01)
    dev = torch.device("cuda:0")
    t = lambda x, g=False: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev).requires_grad_(g)
    s = GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, t(np.array([0.1, 0.2, 0.3])), 1.0, t(cam.viewmatrix),
                                      t(cam.projmatrix), 0, t(cam.campos), False, False)
    args = dict(means2D=torch.zeros(gs.P, 3, device=dev), colors_precomp=t(gs.colors_precomp), scales=t(gs.scales), rotations=t(gs.rotations))
    img_g, rad_g = GaussianRasterizer(s)(means3D=t(gs.means3D, True), opacities=t(gs.opacities, True), **args)
    with torch.no_grad():
        img_n, rad_n = GaussianRasterizer(s)(means3D=t(gs.means3D), opacities=t(gs.opacities), **args)
    assert torch.equal(img_g.detach(), img_n) and torch.equal(rad_g, rad_n)
    e = torch.Tensor([])
    out = R.rasterize_gaussians_native(s.bg, t(gs.means3D), t(gs.colors_precomp), t(gs.opacities), t(gs.scales), t(gs.rotations), 1.0, e,
                                       s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, cam.H, cam.W, e, 0, s.campos, False, False,
                                       need_backward=False)
    assert out[0] > 0 and out[7] == 0 and torch.equal(out[1], img_n)
    with pytest.raises(_lib.GsrError, match="forward-only"):
        R.rasterize_gaussians_backward_native(s.bg, t(gs.means3D), out[2], t(gs.colors_precomp), t(gs.scales), t(gs.rotations), 1.0, e,
                                              s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, torch.ones_like(img_n), e, 0, s.campos,
                                              out[3], out[0], out[4], out[5], False, num_segments=0)


def test_fused_forward_equals_two_stage_forward():
    """gsr_forward_fused (binning scratch sized in advance from earlier views) vs the reference's order of events
    (stage 1, allocate exactly, stage 2): identical images, radii and gradients -- with no hint (two-stage), with a
    sufficient hint (fused), and with a hint that turns out too small (fallback inside one call)."""
    from gaustar_amd import rasterizer as R, scene
    rng = np.random.default_rng(21)
    gs = scene.random_gaussians(6000, rng, scale_range=(0.02, 0.1))
    cam = scene.look_at_camera((0.2, 0.1, -4.0), (0, 0, 0), 320, 200, fovx=0.8, znear=0.01)
    kw = dict(means3D=gs.means3D, opacities=gs.opacities, view=cam.viewmatrix, proj=cam.projmatrix, campos=cam.campos,
              W=cam.W, H=cam.H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=np.array([0.3, 0.1, 0.6], np.float32), shs=None,
              colors_precomp=gs.colors_precomp, scales=gs.scales, rotations=gs.rotations, cov3D_precomp=None, sh_degree=0)
    dpix = rng.normal(size=(3, cam.H, cam.W)).astype(np.float32)
    saved = dict(R._BINNING_HINT)
    try:
        R._BINNING_HINT.clear()
        two_stage = parity.run_hip(kw, dpix)                  # no hint yet: stage 2 runs as a separate call
        hint = R._BINNING_HINT[0]
        assert hint > 0
        fused = parity.run_hip(kw, dpix)                      # hint covers the need: one call
        R._BINNING_HINT[0] = 4096                             # far too small: falls back after stage 1
        small = parity.run_hip(kw, dpix)
        assert R._BINNING_HINT[0] == hint                     # and the hint recovers
    finally:
        R._BINNING_HINT.clear()
        R._BINNING_HINT.update(saved)
    for other, what in ((fused, "fused"), (small, "undersized hint")):
        for k in ("color", "radii", "dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations"):
            if k in ("color", "radii"):
                assert np.array_equal(two_stage[k], other[k]), f"{what}: {k} differs from the two-stage forward"
            else:   # gradients are sums of float atomics: order-dependent in the last bits
                parity.check_grad(other[k], two_stage[k], f"{what} {k}")


def test_two_backward_passes_over_one_forward_agree():
    """The fused forward hands the backward an accumulation table it cleared on the side; that is good for ONE backward.
    A second one over the same graph (retain_graph=True) must fill its own and give the same gradients."""
    import torch
    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, scene
    rng = np.random.default_rng(33)
    gs = scene.random_gaussians(3000, rng, scale_range=(0.02, 0.1))
    cam = scene.look_at_camera((0.1, 0.0, -4.0), (0, 0, 0), 160, 120, fovx=0.8, znear=0.01)
    dev = torch.device("cuda:0")
    t = lambda x, g=False: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev).requires_grad_(g)
    s = GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, t(np.array([0.1, 0.2, 0.3])), 1.0, t(cam.viewmatrix),
                                      t(cam.projmatrix), 0, t(cam.campos), False, False)
    m3, op, sc, ro, co = t(gs.means3D, True), t(gs.opacities, True), t(gs.scales, True), t(gs.rotations, True), t(gs.colors_precomp, True)
    d = torch.randn(3, cam.H, cam.W, device=dev)
    grads = []
    for _ in range(2):   # the first render has no size hint (two-stage); the second takes the fused path
        img, _r = GaussianRasterizer(s)(means3D=m3, means2D=torch.zeros(gs.P, 3, device=dev), opacities=op, colors_precomp=co,
                                        scales=sc, rotations=ro)
        for rep in range(2):
            for p in (m3, op, sc, ro, co):
                p.grad = None
            img.backward(d, retain_graph=(rep == 0))
            grads.append([p.grad.clone().cpu().numpy() for p in (m3, op, sc, ro, co)])
    for other in grads[1:]:
        for a, b, name in zip(grads[0], other, ("means3D", "opacity", "scales", "rotations", "colors")):
            parity.check_grad(b, a, f"repeated backward: dL_d{name}")


def test_library_owned_counters_survive_size_changes_and_fallbacks():
    """The fused forward keeps the tile counters in a library-owned, self-cleaning block.  Alternate image sizes (other
    tile counts, other row strides in the same block), force the undersized-hint fallback in between (counts copied back
    to the image buffer, block re-filled) and compare every render with GSR-independent ground truth: the same view
    rendered through the plain two-stage entry points."""
    import torch
    from gaustar_amd import rasterizer as R, scene
    rng = np.random.default_rng(55)
    gs = scene.random_gaussians(5000, rng, scale_range=(0.02, 0.12))
    sizes = [(320, 200), (97, 61), (640, 360), (97, 61), (320, 200), (33, 17), (640, 360)]
    saved = dict(R._BINNING_HINT)
    try:
        for i, (W, H) in enumerate(sizes):
            cam = scene.look_at_camera((0.2, 0.1, -4.0), (0, 0, 0), W, H, fovx=0.8, znear=0.01)
            kw = dict(means3D=gs.means3D, opacities=gs.opacities, view=cam.viewmatrix, proj=cam.projmatrix, campos=cam.campos,
                      W=W, H=H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=np.array([0.3, 0.1, 0.6], np.float32), shs=None,
                      colors_precomp=gs.colors_precomp, scales=gs.scales, rotations=gs.rotations, cov3D_precomp=None, sh_degree=0)
            R._BINNING_HINT.clear()
            truth = parity.run_hip(kw)                      # no hint: stage 1 over the library's block, copy-back, stage 2
            R._BINNING_HINT[0] = 1 << 28
            fused = parity.run_hip(kw)                      # fused, self-cleaning
            if i % 2:
                R._BINNING_HINT[0] = 4096
                assert np.array_equal(parity.run_hip(kw)["color"], truth["color"])   # undersized hint
                R._BINNING_HINT[0] = 1 << 28
            again = parity.run_hip(kw)                      # the block must have come back clean
            for other in (fused, again):
                assert np.array_equal(other["color"], truth["color"]) and np.array_equal(other["radii"], truth["radii"]), (W, H)
    finally:
        R._BINNING_HINT.clear()
        R._BINNING_HINT.update(saved)


def test_interleaved_streams_keep_their_own_counters():
    """The fused forward's library-owned tile counters are per (device, stream): renders issued alternately on two side
    streams (different views, different image sizes) must each match the same view rendered on the default stream."""
    import torch
    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, scene
    rng = np.random.default_rng(77)
    gs = scene.random_gaussians(4000, rng, scale_range=(0.02, 0.1))
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
    m3, op, sc, ro, co = t(gs.means3D), t(gs.opacities), t(gs.scales), t(gs.rotations), t(gs.colors_precomp)
    views = [scene.look_at_camera((0.3 * i - 0.4, 0.1, -4.0), (0, 0, 0), 160 + 48 * (i % 2), 120 + 16 * (i % 3), fovx=0.8, znear=0.01)
             for i in range(4)]

    def render(cam):
        s = GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, t(np.array([0.1, 0.2, 0.3])), 1.0, t(cam.viewmatrix),
                                          t(cam.projmatrix), 0, t(cam.campos), False, False)
        with torch.no_grad():
            return GaussianRasterizer(s)(means3D=m3, means2D=m3, opacities=op, colors_precomp=co, scales=sc, rotations=ro)[0]

    for cam in views:          # warm the size hint so that every later call takes the fused path
        render(cam)
    truth = [render(cam) for cam in views]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [None] * len(views)
    for rep in range(3):
        for i, cam in enumerate(views):
            with torch.cuda.stream(streams[i % 2]):
                got[i] = render(cam)
    torch.cuda.synchronize()
    for a, b in zip(got, truth):
        assert torch.equal(a, b)

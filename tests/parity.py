"""Shared helpers of the parity tests: run the HIP path through its public Python API, load golden
fixtures, and compare with the tolerances BASELINE.json's north_star states (1e-4, fp32).

Tolerance model (written once, used by every parity test):
  * integer outputs (radii, visibility, sorted lists) must match EXACTLY;
  * images: |a-b| <= 1e-4 * max(1, |b|) element-wise;
  * gradients: |a-b| <= 1e-4 * max|b| + 1e-4 * |b| (normalised to the tensor's largest entry --
    the reference itself is only reproducible to float-atomic summation order); entries far below the
    tensor's largest one are additionally held to a 10x tighter absolute bound (SMALL_TOL), so that they are
    not "within tolerance" merely by being small;
  * threshold flips: alpha >= 1/255 and T >= 1e-4 are hard cuts evaluated on values that differ by
    an ulp between exp implementations (device v_exp_f32 / ocml expf / glibc expf), so a (pixel,
    Gaussian) pair may flip in or out.  The golden vectors need NO such allowance and are checked with
    none (compare_hip_to(strict=True), the default).  Full-size renders get FULL_IMG_OUTLIERS /
    FULL_GRAD_OUTLIERS (a few dozen elements of millions, see below), randomly generated scenes
    (tests/devtools/random_parity_sweep.py, the seeded oracle cases) the wider 2e-4 / 1e-3 -- always
    capped by `outlier_cap` (a flipped pair moves a pixel by < 1/255).
  * radii: ceil(3 sigma) of a value that sits within rounding of an integer may flip by one between
    implementations of sqrt; assert_radii() accepts exactly that and nothing else.
"""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

IMG_TOL = 1e-4
GRAD_TOL = 1e-4
# Full-size renders (6 - 25 million image elements, millions of gradient entries) do show a handful of pairs on the other
# side of a blend threshold than in the reference build: measured on MI355X, 1 - 25 image elements per 1080p view (102 at
# 4K) up to 2.7e-3, and up to 30 of 1.5 - 3 million entries of a gradient tensor (the flipped pair's Gaussian) up to
# 5.3e-3 of the tensor's maximum.  Allowed there: 1e-5 of the image elements, 5e-5 of a gradient tensor's entries.
# Round 6: with the projection in the reference build's operation order (gsr_ref_order.h) the whole 160-camera rig shows at most 5
# image elements and 23 entries of a gradient tensor per view outside the tolerance, median 0 (profiles/r06_parity_report.txt):
# the budgets came down from 1e-5 / 5e-5 to 2e-6 of the image elements (12 of a 1080p image) and 2e-5 of a gradient tensor's entries.
FULL_IMG_OUTLIERS = 2e-6
FULL_GRAD_OUTLIERS = 2e-5
FULL_MIN_OUTLIERS = 4      # a fraction budget never means fewer than this many elements (a 60 000-entry tensor: 2e-5 would be one)
SMALL_FRAC = 1e-3    # "small" gradient entries: |ref| < SMALL_FRAC * max|ref| ...
SMALL_TOL = 1e-5     # ... must be within SMALL_TOL * max|ref| + GRAD_TOL * |ref|
REPORT = []          # (what, measured max normalised error) of every check of this process, for the test log


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    d = {k: z[k] for k in z.files}
    kw = {}
    for k, v in d.items():
        if k.startswith("in_") and k != "in_dL_dpix":
            kw[k[3:]] = None if (isinstance(v, np.ndarray) and v.size == 0 and k[3:] in
                                 ("shs", "colors_precomp", "scales", "rotations", "cov3D_precomp")) else v
    for k in ("W", "H", "sh_degree"):
        kw[k] = int(kw[k])
    for k in ("tanfovx", "tanfovy", "scale_modifier"):
        kw[k] = float(kw[k])
    return kw, d


def check_image(a, b, what="image", tol=IMG_TOL, max_outlier_frac=0.0, outlier_cap=8e-3):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    scale = np.maximum(1.0, np.abs(b))
    err = np.abs(a - b) / scale
    bad = err > tol
    allowed = max(int(np.ceil(max_outlier_frac * bad.size)), FULL_MIN_OUTLIERS) if max_outlier_frac > 0.0 else 0
    assert bad.sum() <= allowed, f"{what}: {bad.sum()} of {bad.size} elements differ by more than {tol} (max {err.max():.3e}; allowed {allowed})"
    if bad.any():
        assert err.max() <= outlier_cap, f"{what}: outlier {err.max():.3e} exceeds the threshold-flip cap {outlier_cap}"
    worst = float(err.max()) if err.size else 0.0
    REPORT.append((what, worst))
    print(f"[parity] {what}: max normalised error {worst:.3e} (tolerance {tol:.0e}, outliers {int(bad.sum())})")
    return worst


def check_grad(a, b, what="grad", tol=GRAD_TOL, max_outlier_frac=0.0, outlier_cap=0.05, small_tol=SMALL_TOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.size == 0:
        return 0.0
    assert np.isfinite(a).all(), f"{what}: non-finite values"
    ref = np.abs(b).max()
    if ref == 0.0:
        assert np.abs(a).max() <= 1e-12, f"{what}: expected all zeros, got max {np.abs(a).max():.3e}"
        return 0.0
    err = np.abs(a - b) / (ref + 1e-30) - tol * np.abs(b) / ref
    bad = err > tol
    allowed = max(int(np.ceil(max_outlier_frac * bad.size)), FULL_MIN_OUTLIERS) if max_outlier_frac > 0.0 else 0
    assert bad.sum() <= allowed, (f"{what}: {bad.sum()} of {bad.size} elements off by more than {tol} of max|ref| "
                                  f"(worst {err.max():.3e}, max|ref| {ref:.3e}; allowed {allowed})")
    if bad.any():
        assert err.max() <= outlier_cap, f"{what}: outlier {err.max():.3e} (normalised) exceeds cap {outlier_cap}"
    # entries far below the largest one: a bound 10x tighter in absolute terms (they would pass the test above with
    # any value up to tol * max|ref|)
    small = np.abs(b) < SMALL_FRAC * ref
    worst_small = 0.0
    if small_tol is not None and small.any():
        es = (np.abs(a - b)[small] - tol * np.abs(b)[small]) / ref
        worst_small = float(max(es.max(), 0.0))
        nbad = int((es > small_tol).sum())
        assert nbad <= max_outlier_frac * a.size, (f"{what}: {nbad} small entries (|ref| < {SMALL_FRAC:g} max) off by more than "
                                                   f"{small_tol:g} of max|ref| (worst {worst_small:.3e})")
    worst = float(max(err.max(), 0.0))
    REPORT.append((what, worst))
    print(f"[parity] {what}: max normalised error {worst:.3e} (tolerance {tol:.0e}), small entries {worst_small:.3e} "
          f"(tolerance {small_tol if small_tol is not None else float('nan'):.0e}), outliers {int(bad.sum())}")
    return worst


GRAD_KEYS = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drotations"]


def run_hip(kw, dL_dpix=None, device="cuda:0", debug=False):
    """Forward (+ backward) through gaustar_amd's public API, exactly as a GauSTAR caller would:
    GaussianRasterizationSettings + GaussianRasterizer, gradients via autograd."""
    import torch
    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device(device)

    def t(x, grad=False):
        if x is None:
            return None
        y = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
        return y.requires_grad_(grad)

    want_grad = dL_dpix is not None
    means3D = t(np.asarray(kw["means3D"]).reshape(-1, 3), want_grad)
    P = means3D.shape[0]
    opac = t(np.asarray(kw["opacities"]).reshape(-1, 1), want_grad)
    shs, cols = t(kw.get("shs"), want_grad), t(kw.get("colors_precomp"), want_grad)
    scales, rots, cov = t(kw.get("scales"), want_grad), t(kw.get("rotations"), want_grad), t(kw.get("cov3D_precomp"), want_grad)
    means2D = torch.zeros(P, 3, device=dev, requires_grad=want_grad)
    settings = GaussianRasterizationSettings(
        image_height=int(kw["H"]), image_width=int(kw["W"]), tanfovx=kw["tanfovx"], tanfovy=kw["tanfovy"],
        bg=t(kw["bg"]), scale_modifier=float(kw.get("scale_modifier", 1.0)),
        # handed over the way sugar_model.py:1149-1150 does: a transposed (non-contiguous) view
        viewmatrix=t(np.asarray(kw["view"]).reshape(4, 4).T.copy()).transpose(0, 1),
        projmatrix=t(np.asarray(kw["proj"]).reshape(4, 4)), sh_degree=int(kw.get("sh_degree", 0)),
        campos=t(np.asarray(kw["campos"]).reshape(1, 3)), prefiltered=False, debug=debug)
    rast = GaussianRasterizer(settings)
    color, radii = rast(means3D=means3D, means2D=means2D, opacities=opac, shs=shs, colors_precomp=cols,
                        scales=scales, rotations=rots, cov3D_precomp=cov)
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy())
    if want_grad:
        color.backward(t(dL_dpix))
        z = lambda x, shape: (x.grad.cpu().numpy() if x is not None and x.grad is not None else np.zeros(shape, np.float32))
        M = 0 if shs is None else shs.shape[1]
        out.update(dL_dmeans2D=z(means2D, (P, 3)), dL_dcolors=z(cols, (P, 3)), dL_dopacity=z(opac, (P, 1)),
                   dL_dmeans3D=z(means3D, (P, 3)), dL_dcov3D=z(cov, (P, 6)), dL_dsh=z(shs, (P, M, 3)),
                   dL_dscales=z(scales, (P, 3)), dL_drotations=z(rots, (P, 4)))
        out["_has"] = dict(dL_dcolors=cols is not None, dL_dcov3D=cov is not None, dL_dsh=shs is not None,
                           dL_dscales=scales is not None, dL_drotations=rots is not None)
    torch.cuda.synchronize()
    return out


def run_oracle(kw, dL_dpix=None):
    from oracle import oracle
    st = oracle.forward(kw["means3D"], kw["opacities"], kw["view"], kw["proj"], kw["campos"], kw["W"], kw["H"],
                        kw["tanfovx"], kw["tanfovy"], kw["bg"], shs=kw.get("shs"),
                        colors_precomp=kw.get("colors_precomp"), scales=kw.get("scales"),
                        rotations=kw.get("rotations"), cov3D_precomp=kw.get("cov3D_precomp"),
                        sh_degree=kw.get("sh_degree", 0), scale_modifier=kw.get("scale_modifier", 1.0))
    g = oracle.backward(st, dL_dpix) if dL_dpix is not None else None
    return st, g


def assert_radii(kw, hip_radii, ref_radii, what="", max_flips=0):
    """radii must be identical, except that at most `max_flips` entries may differ by exactly one where 3*sqrt(lambda_max)
    (forward.cu:229-232) sits within rounding of an integer -- a ceil() flip between sqrt implementations.  The value is
    recomputed in float64 from the oracle's conic of just those Gaussians."""
    hip_radii, ref_radii = np.asarray(hip_radii).reshape(-1), np.asarray(ref_radii).reshape(-1)
    idx = np.nonzero(hip_radii != ref_radii)[0]
    assert len(idx) <= max_flips, f"{what}: radii differ in {len(idx)} entries (allowed: {max_flips} ceil() flips)"
    if len(idx) == 0:
        return 0
    assert (np.abs(hip_radii[idx].astype(np.int64) - ref_radii[idx]) == 1).all(), f"{what}: radii differ by more than one: {idx[:8]}"
    from oracle import oracle
    sub = lambda k: None if kw.get(k) is None else np.asarray(kw[k])[idx]
    st = oracle.forward(np.asarray(kw["means3D"]).reshape(-1, 3)[idx], np.asarray(kw["opacities"]).reshape(-1)[idx], kw["view"],
                        kw["proj"], kw["campos"], kw["W"], kw["H"], kw["tanfovx"], kw["tanfovy"], kw["bg"], shs=sub("shs"),
                        colors_precomp=sub("colors_precomp"), scales=sub("scales"), rotations=sub("rotations"),
                        cov3D_precomp=sub("cov3D_precomp"), sh_degree=kw.get("sh_degree", 0),
                        scale_modifier=kw.get("scale_modifier", 1.0))
    co = st["conic_opacity"].astype(np.float64)
    det_c = co[:, 0] * co[:, 2] - co[:, 1] ** 2
    cx, cy, cz = co[:, 2] / det_c, -co[:, 1] / det_c, co[:, 0] / det_c        # cov2D = conic^-1
    mid = 0.5 * (cx + cz)
    lam = mid + np.sqrt(np.maximum(0.1, mid * mid - (cx * cz - cy * cy)))
    r = 3.0 * np.sqrt(lam)
    dist = np.abs(r - np.round(r))
    assert (dist <= 3e-5 * r).all(), f"{what}: radii mismatch away from a ceil() boundary: 3 sigma = {r[dist > 3e-5 * r][:5]}"
    print(f"[parity] {what}: {len(idx)} radii differ by one at a ceil() boundary (|3 sigma - integer| <= {dist.max():.1e})")
    return len(idx)


def compare_hip_to(hip, ref_color, ref_radii, ref_grads=None, what="", kw=None, max_radii_flips=0, strict=True,
                   img_outliers=2e-4, grad_outliers=1e-3):
    """hip = run_hip() result; ref_grads = dict with the GRAD_KEYS that apply.  strict: no element may exceed the
    tolerance (golden vectors, full-size configs); otherwise the threshold-flip allowance of the module docstring."""
    if kw is not None:
        assert_radii(kw, hip["radii"], ref_radii, what, max_radii_flips)
    else:
        assert np.array_equal(hip["radii"], ref_radii), f"{what}: radii differ in {int((hip['radii'] != ref_radii).sum())} entries"
    check_image(hip["color"], ref_color, f"{what} color", max_outlier_frac=0.0 if strict else img_outliers)
    if ref_grads is not None:
        has = hip["_has"]
        for k in GRAD_KEYS:
            if k in has and not has[k]:
                continue   # gradient w.r.t. an absent optional input
            check_grad(hip[k], np.asarray(ref_grads[k]).reshape(hip[k].shape), f"{what} {k}",
                       max_outlier_frac=0.0 if strict else grad_outliers)

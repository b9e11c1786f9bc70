"""Shared helpers of the parity tests: run the HIP path through its public Python API, load golden
fixtures, and compare with the tolerances BASELINE.json's north_star states (1e-4, fp32).

Tolerance model (written once, used by every parity test):
  * integer outputs (radii, visibility, sorted lists) must match EXACTLY;
  * images: |a-b| <= 1e-4 * max(1, |b|) element-wise;
  * gradients: |a-b| <= 1e-4 * max|b| + 1e-4 * |b| (normalised to the tensor's largest entry --
    the reference itself is only reproducible to float-atomic summation order);
  * threshold flips: alpha >= 1/255 and T >= 1e-4 are hard cuts evaluated on values that differ by
    an ulp between exp implementations (device v_exp_f32 / ocml expf / glibc expf), so a (pixel,
    Gaussian) pair may flip in or out.  Such outliers are allowed for at most `max_outlier_frac`
    of the elements and must stay below `outlier_cap` (a flipped pair moves a pixel by < 1/255).
"""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

IMG_TOL = 1e-4
GRAD_TOL = 1e-4


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


def check_image(a, b, what="image", tol=IMG_TOL, max_outlier_frac=2e-4, outlier_cap=8e-3):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    scale = np.maximum(1.0, np.abs(b))
    err = np.abs(a - b) / scale
    bad = err > tol
    frac = bad.mean() if bad.size else 0.0
    assert frac <= max_outlier_frac, f"{what}: {bad.sum()} of {bad.size} elements differ by more than {tol} (max {err.max():.3e})"
    if bad.any():
        assert err.max() <= outlier_cap, f"{what}: outlier {err.max():.3e} exceeds the threshold-flip cap {outlier_cap}"
    return float(err.max()) if err.size else 0.0


def check_grad(a, b, what="grad", tol=GRAD_TOL, max_outlier_frac=1e-3, outlier_cap=0.05):
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
    frac = bad.mean()
    assert frac <= max_outlier_frac, (f"{what}: {bad.sum()} of {bad.size} elements off by more than {tol} of max|ref| "
                                      f"(worst {err.max():.3e}, max|ref| {ref:.3e})")
    if bad.any():
        assert err.max() <= outlier_cap, f"{what}: outlier {err.max():.3e} (normalised) exceeds cap {outlier_cap}"
    return float(max(err.max(), 0.0))


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


def compare_hip_to(hip, ref_color, ref_radii, ref_grads=None, what=""):
    """hip = run_hip() result; ref_grads = dict with the GRAD_KEYS that apply."""
    assert np.array_equal(hip["radii"], ref_radii), f"{what}: radii differ in {int((hip['radii'] != ref_radii).sum())} entries"
    check_image(hip["color"], ref_color, f"{what} color")
    if ref_grads is not None:
        has = hip["_has"]
        for k in GRAD_KEYS:
            if k in has and not has[k]:
                continue   # gradient w.r.t. an absent optional input
            check_grad(hip[k], np.asarray(ref_grads[k]).reshape(hip[k].shape), f"{what} {k}")

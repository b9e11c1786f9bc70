"""Parity of the HIP rasterizer against the reference build over MANY views, compared on the device.

TEST INFRASTRUCTURE ONLY (tests/test_gpu_rig_parity.py, bench.py's extras leg -- never the product path).  Needs a GPU and
oracle/_ref/libgsr_ref.so (the reference's own kernels, oracle/build_ref.sh).  Tolerances are tests/parity.py's, restated on
torch tensors so that 160 full-size views take a minute instead of ten: images |a-b| <= 1e-4 max(1, |b|); gradients
|a-b| <= 1e-4 max|b| + 1e-4 |b|, entries below 1e-3 max|b| additionally within 1e-5 max|b|.  An element outside is a
THRESHOLD FLIP (a (pixel, Gaussian) pair on the other side of alpha >= 1/255 or T < 1e-4 than in the reference, because the
HIP path evaluates exp in the exp2 domain): they are COUNTED per tensor and per view, capped in number and in size."""
from __future__ import annotations

import numpy as np
import torch

IMG_TOL, GRAD_TOL, SMALL_FRAC, SMALL_TOL = 1e-4, 1e-4, 1e-3, 1e-5
IMG_CAP, GRAD_CAP = 8e-3, 0.05          # largest admissible flip (parity.check_image / check_grad outlier_cap)
GRAD_KEYS = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dscales", "dL_drotations")


def _img_flips(a, b):
    err = (a.double() - b.double()).abs() / b.double().abs().clamp_min(1.0)
    return int((err > IMG_TOL).sum()), float(err.max())


def _grad_flips(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    ref = float(b.abs().max())
    if ref == 0.0:
        return int((a.abs() > 1e-12).sum()), float(a.abs().max())
    d = (a - b).abs()
    err = d / ref - GRAD_TOL * b.abs() / ref
    bad = err > GRAD_TOL
    small = b.abs() < SMALL_FRAC * ref
    bad_small = small & ((d - GRAD_TOL * b.abs()) / ref > SMALL_TOL)
    return int((bad | bad_small).sum()), float(err.max().clamp_min(0.0))


def compare_views(gs, cams, bg, views, device="cuda:0", dpix_seed=5):
    """-> list of {"view", "flips": {tensor: count}, "worst": {tensor: largest normalised error}, "radii_diff"} for the given
    camera indices; gs / cams / bg as gaustar_amd.scene.config_C() returns them (colours precomputed)."""
    from oracle import ref
    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device(device)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
    base = dict(means3D=t(gs.means3D).reshape(-1, 3), opacities=t(gs.opacities).reshape(-1, 1), colors=t(gs.colors_precomp),
                scales=t(gs.scales), rotations=t(gs.rotations))
    P = base["means3D"].shape[0]
    bg_t = t(bg)
    g = torch.Generator(device="cpu").manual_seed(dpix_seed)
    dpix = torch.randn(3, cams[0].H, cams[0].W, generator=g).to(dev)
    rr = ref.RefRasterizer(str(dev))
    out = []
    for vi in views:
        c = cams[vi]
        vm, pm, cp = t(c.viewmatrix), t(c.projmatrix), t(c.campos)
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in base.items()}
        m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
        st = GaussianRasterizationSettings(image_height=c.H, image_width=c.W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg_t,
                                           scale_modifier=1.0, viewmatrix=vm, projmatrix=pm, sh_degree=0, campos=cp.reshape(1, 3),
                                           prefiltered=False, debug=False)
        img, radii = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                            colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
        img.backward(dpix)
        color_r, radii_r, _R = rr.forward(base["means3D"], base["opacities"], vm, pm, cp, c.W, c.H, c.tanfovx, c.tanfovy, bg_t,
                                          colors_precomp=base["colors"], scales=base["scales"], rotations=base["rotations"])
        gr = rr.backward(dpix)
        ours = {"dL_dmeans2D": m2.grad, "dL_dcolors": leaves["colors"].grad, "dL_dopacity": leaves["opacities"].grad,
                "dL_dmeans3D": leaves["means3D"].grad, "dL_dscales": leaves["scales"].grad, "dL_drotations": leaves["rotations"].grad}
        flips, worst = {}, {}
        flips["color"], worst["color"] = _img_flips(img.detach(), color_r)
        for k in GRAD_KEYS:
            flips[k], worst[k] = _grad_flips(ours[k], gr[k].reshape(ours[k].shape))
        out.append({"view": int(vi), "flips": flips, "worst": worst,
                    "radii_diff": int((radii.to(torch.int32) != radii_r).sum())})
    torch.cuda.synchronize(dev)
    return out


def summarise(rows):
    """min / median / max of the per-view flip totals (image + all gradient tensors) and of the largest flip."""
    tot = [sum(r["flips"].values()) for r in rows]
    per_tensor = {k: [r["flips"][k] for r in rows] for k in rows[0]["flips"]}
    return {"views": len(rows), "flips_per_view": {"min": int(min(tot)), "median": float(np.median(tot)), "max": int(max(tot))},
            "flips_per_view_by_tensor_max": {k: int(max(v)) for k, v in per_tensor.items()},
            "largest_image_flip": max(r["worst"]["color"] for r in rows),
            "largest_gradient_flip": max(max(v for k, v in r["worst"].items() if k != "color") for r in rows),
            "radii_diff_max": max(r["radii_diff"] for r in rows)}

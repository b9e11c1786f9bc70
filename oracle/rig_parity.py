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


def _img_flips(a, b, mask=False):
    err = (a.double() - b.double()).abs() / b.double().abs().clamp_min(1.0)
    if mask:
        return int((err > IMG_TOL).sum()), float(err.max()), err > IMG_TOL
    return int((err > IMG_TOL).sum()), float(err.max())


def _grad_flips(a, b, mask=False):
    shape = b.shape
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    ref = float(b.abs().max())
    if ref == 0.0:
        m = a.abs() > 1e-12
        return (int(m.sum()), float(a.abs().max()), m.reshape(shape)) if mask else (int(m.sum()), float(a.abs().max()))
    d = (a - b).abs()
    err = d / ref - GRAD_TOL * b.abs() / ref
    bad = err > GRAD_TOL
    small = b.abs() < SMALL_FRAC * ref
    bad_small = small & ((d - GRAD_TOL * b.abs()) / ref > SMALL_TOL)
    m = bad | bad_small
    if mask:
        return int(m.sum()), float(err.max().clamp_min(0.0)), m.reshape(shape)
    return int(m.sum()), float(err.max().clamp_min(0.0))


# ---------------------------------------------------------------------------------------------------------------------------
# WHICH decision flipped (VERDICT r5 task 2).  An element outside the tolerance is the trace of a (pixel, Gaussian) pair that
# the two rasterizers decided differently.  The reference's walk (DGR/cuda_rasterizer/forward.cu:330-352) is replayed here from
# the reference build's OWN per-Gaussian state (means2D, conic_opacity, ranges, point_list: ref.RefRasterizer.state()) for the
# pixels concerned, and every decision's MARGIN is measured:
#   alpha-cut   |255 alpha - 1|             (forward.cu:340-342 `alpha < 1.0f / 255.0f`; backward.cu:499-500)
#   T-cut       |T (1 - alpha) - 1e-4| / 1e-4  (forward.cu:343-348 `test_T < 0.0001f`: that pair is NOT blended and the pixel stops)
#   power-cut   |power|                      (forward.cu:333-335 `power > 0.0f`)
# A decision can only come out differently if its margin is of the order of the two evaluations' rounding difference (exp vs
# exp2-domain: ~1e-7 relative on alpha; a product of up to hundreds of (1 - alpha) factors: ~1e-6 on T), so a flagged pixel /
# Gaussian is attributed to the kind of the decision whose margin is below KIND_THR; "downstream" = no decision of the Gaussian's
# own pairs is that close, but one of its pixels holds such a decision EARLIER in its walk (every later pair of that pixel then
# sees T scaled by 1 - 1/255 or the walk one pair longer); "unexplained" = none found.
KIND_THR_ALPHA, KIND_THR_T, KIND_THR_POWER = 2e-5, 2e-4, 1e-6


def _walk(st, px, py, gx, m2=None, co=None):
    """The reference's forward walk of pixel (px, py), replayed in float32 over the reference's list of the pixel's tile, from
    the reference's own per-Gaussian state or (m2, co given) from another implementation's means2D / conic_opacity.  -> arrays
    over the list (ids, power, raw alpha, validity, test_T) and the stop position (len(list) if the pixel never stops)."""
    tile = (py // 16) * gx + (px // 16)
    r0, r1 = int(st["ranges"][tile, 0]), int(st["ranges"][tile, 1])
    ids = st["point_list"][r0:r1].astype(np.int64)
    f = np.float32
    xy = (st["means2D"] if m2 is None else m2)[ids]
    cq = (st["conic_opacity"] if co is None else co)[ids]
    dx, dy = xy[:, 0] - f(px), xy[:, 1] - f(py)
    power = f(-0.5) * (cq[:, 0] * dx * dx + cq[:, 2] * dy * dy) - cq[:, 1] * dx * dy
    a_raw = cq[:, 3] * np.exp(power, dtype=np.float32)
    alpha = np.minimum(f(0.99), a_raw)
    valid = (power <= 0) & (alpha >= f(1.0 / 255.0))
    one_m = np.where(valid, f(1) - alpha, f(1)).astype(np.float32)
    T_before = np.concatenate([[f(1)], np.cumprod(one_m, dtype=np.float32)[:-1]]) if len(ids) else np.zeros(0, np.float32)
    test_T = T_before * (f(1) - alpha)
    stops = np.nonzero(valid & (test_T < f(1e-4)))[0]
    stop = int(stops[0]) if len(stops) else len(ids)
    return {"ids": ids, "power": power, "a_raw": a_raw, "valid": valid, "test_T": test_T, "stop": stop}


def _events(w, wh):
    """Decisions of one pixel's walk that can have come out differently on the two sides, as {list position: kind}:
      *_inputs  the replay from the reference's state (w) and the replay from the HIP path's state (wh: its means2D / conic differ
                from the reference build's in their last bits for part of the Gaussians) DECIDE differently -- certain;
      *_eval    both replays agree but the decision's margin is within what the two evaluations of exp / of the running product
                differ by (KIND_THR_*) -- the only way left."""
    ev = {}
    n = len(w["ids"])
    stop = min(w["stop"], n - 1)
    if n == 0:
        return ev
    if wh is not None:
        lim = min(max(w["stop"], wh["stop"]), n - 1) + 1
        for k in np.nonzero(w["valid"][:lim] != wh["valid"][:lim])[0]:
            ev[int(k)] = "power_cut_inputs" if (w["power"][k] <= 0) != (wh["power"][k] <= 0) else "alpha_cut_inputs"
        if w["stop"] != wh["stop"]:
            ev.setdefault(int(min(w["stop"], wh["stop"], n - 1)), "T_cut_inputs")
    pw, ar, tt, va = w["power"][:stop + 1], w["a_raw"][:stop + 1], w["test_T"][:stop + 1], w["valid"][:stop + 1]
    for k in np.nonzero((pw <= 0) & (np.abs(255.0 * ar.astype(np.float64) - 1.0) < KIND_THR_ALPHA))[0]:
        ev.setdefault(int(k), "alpha_cut_eval")
    for k in np.nonzero(va & (np.abs(tt.astype(np.float64) - 1e-4) / 1e-4 < KIND_THR_T))[0]:
        ev.setdefault(int(k), "T_cut_eval")
    for k in np.nonzero(np.abs(pw.astype(np.float64)) < KIND_THR_POWER)[0]:
        ev.setdefault(int(k), "power_cut_eval")
    return ev


def classify_flips(st, radii_ref, img_mask, grad_masks, W, H, hip_m2=None, hip_co=None, max_gaussians=600,
                   max_pixels_per_gaussian=2048):
    """st: RefRasterizer.state(); img_mask [3,H,W] bool; grad_masks {tensor: [P, k] bool}; hip_m2 / hip_co: the HIP path's
    per-Gaussian means2D [P,2] / conic_opacity [P,4] (gsr_debug_export).  Every flagged image pixel and every flagged Gaussian
    (a row of a gradient tensor with an element outside the tolerance) is attributed to a decision event of _events():
    a pixel to the first event of its walk; a Gaussian to an event at one of ITS OWN pairs ("own:<kind>") or, failing that, to an
    event elsewhere in the walk of one of its pixels ("same_pixel:<kind>": every other pair of that pixel sees T scaled by
    1 - alpha of the flipped pair, or accum_rec changed by it), else "unexplained"."""
    gx = (W + 15) // 16
    out = {"image_pixels": {}, "image_elements": {}, "gaussians": {}, "gradient_elements": {}}
    bump = lambda d, k, n=1: d.__setitem__(k, d.get(k, 0) + n)
    walks = {}

    def events_of(px, py):
        e = walks.get((px, py))
        if e is None:
            w = _walk(st, px, py, gx)
            wh = _walk(st, px, py, gx, hip_m2, hip_co) if hip_m2 is not None else None
            e = walks[(px, py)] = (w, _events(w, wh))
        return e
    for py, px in np.argwhere(img_mask.any(0)):
        _w, ev = events_of(int(px), int(py))
        k = ev[min(ev)] if ev else "unexplained"
        bump(out["image_pixels"], k)
        bump(out["image_elements"], k, int(img_mask[:, py, px].sum()))
    rows = None
    for m in grad_masks.values():
        r_ = m.reshape(m.shape[0], -1).any(1)
        rows = r_ if rows is None else (rows | r_)
    gids = np.nonzero(rows)[0] if rows is not None else np.zeros(0, np.int64)
    f = np.float32
    for g in gids[:max_gaussians]:
        x, y = st["means2D"][g]
        r = int(radii_ref[g])
        co = st["conic_opacity"][g]
        x0, x1 = max(0, int(np.floor(x - r))), min(W - 1, int(np.ceil(x + r)))
        y0, y1 = max(0, int(np.floor(y - r))), min(H - 1, int(np.ceil(y + r)))
        own, near = None, None
        if r > 0 and x1 >= x0 and y1 >= y0:
            xs, ys = np.meshgrid(np.arange(x0, x1 + 1), np.arange(y0, y1 + 1))
            dx, dy = f(x) - xs.astype(f), f(y) - ys.astype(f)
            power = f(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy
            a_raw = co[3] * np.exp(power, dtype=np.float32)
            cand = np.argwhere((a_raw >= f(0.9 / 255.0)) | (np.abs(power) < 1e-5))[:max_pixels_per_gaussian]
            for iy, ix in cand:
                w, ev = events_of(int(xs[iy, ix]), int(ys[iy, ix]))
                if not ev:
                    continue
                pos = np.nonzero(w["ids"] == g)[0]
                if len(pos) and int(pos[0]) in ev:
                    own = ev[int(pos[0])]
                    break
                if len(pos) and near is None:
                    near = ev[min(ev)]
        k = ("own:" + own) if own else ("same_pixel:" + near) if near else "unexplained"
        bump(out["gaussians"], k)
        for m in grad_masks.values():
            n = int(m[g].sum())
            if n:
                bump(out["gradient_elements"], k, n)
    out["gaussians_skipped"] = int(max(0, len(gids) - max_gaussians))
    return out


def hip_state(kw_t, c, P):
    """means2D [P,2] / conic_opacity [P,4] of the HIP path for one view (gsr_debug_export of an exact, unplanned forward)."""
    import ctypes
    from gaustar_amd import _lib, rasterizer as rz
    lib = _lib.load()
    e = torch.Tensor([])
    out = rz.rasterize_gaussians_native(kw_t["bg"], kw_t["means3D"], kw_t["colors"], kw_t["opacities"], kw_t["scales"], kw_t["rotations"],
                                        1.0, e, kw_t["view"], kw_t["proj"], c.tanfovx, c.tanfovy, c.H, c.W, e, 0, kw_t["campos"], False,
                                        False, use_plan=False)
    Rn, _col, _radii, geom, binning, img, _maxc, _U = out
    dev = kw_t["means3D"].device
    T = ((c.W + 15) // 16) * ((c.H + 15) // 16)
    m2 = torch.zeros(P, 2, device=dev); co = torch.zeros(P, 4, device=dev)
    rng_t = torch.zeros(T, 2, dtype=torch.int32, device=dev); pl_t = torch.zeros(max(Rn, 1), dtype=torch.int32, device=dev)
    fT = torch.zeros(c.H, c.W, device=dev); nc = torch.zeros(c.H, c.W, dtype=torch.int32, device=dev)
    pp = lambda x: ctypes.c_void_p(x.data_ptr())
    _lib.check(lib.gsr_debug_export(P, Rn, 1, c.W, c.H, pp(geom), pp(binning), pp(img), pp(m2), pp(co), None, None, pp(rng_t), pp(pl_t),
                                    pp(fT), pp(nc), None), "gsr_debug_export")
    torch.cuda.synchronize(dev)
    return m2.cpu().numpy(), co.cpu().numpy()


def compare_views(gs, cams, bg, views, device="cuda:0", dpix_seed=5, classify=False):
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
        row = {"view": int(vi), "flips": flips, "worst": worst, "radii_diff": int((radii.to(torch.int32) != radii_r).sum())}
        if not classify:
            flips["color"], worst["color"] = _img_flips(img.detach(), color_r)
            for k in GRAD_KEYS:
                flips[k], worst[k] = _grad_flips(ours[k], gr[k].reshape(ours[k].shape))
        else:
            flips["color"], worst["color"], im = _img_flips(img.detach(), color_r, mask=True)
            gm = {}
            for k in GRAD_KEYS:
                flips[k], worst[k], m_ = _grad_flips(ours[k], gr[k].reshape(ours[k].shape), mask=True)
                gm[k] = m_.cpu().numpy()
            torch.cuda.synchronize(dev)
            hm2, hco = hip_state(dict(bg=bg_t, means3D=base["means3D"], colors=base["colors"], opacities=base["opacities"],
                                      scales=base["scales"], rotations=base["rotations"], view=vm, proj=pm, campos=cp), c, P)
            st_ = rr.state()
            vis = radii_r.cpu().numpy() > 0
            row["input_bits"] = {"means2D_differ": int((hm2[vis].view(np.uint32) != st_["means2D"][vis].view(np.uint32)).any(1).sum()),
                                 "conic_differ": int((hco[vis, :3].view(np.uint32) != st_["conic_opacity"][vis, :3].view(np.uint32)).any(1).sum()),
                                 "visible": int(vis.sum())}
            row["kinds"] = classify_flips(st_, radii_r.cpu().numpy(), im.cpu().numpy(), gm, c.W, c.H, hm2, hco)
        out.append(row)
    torch.cuda.synchronize(dev)
    return out


def summarise(rows):
    """min / median / max of the per-view flip totals (image + all gradient tensors) and of the largest flip."""
    tot = [sum(r["flips"].values()) for r in rows]
    per_tensor = {k: [r["flips"][k] for r in rows] for k in rows[0]["flips"]}
    out = {"views": len(rows), "flips_per_view": {"min": int(min(tot)), "median": float(np.median(tot)), "max": int(max(tot))},
           "flips_per_view_by_tensor_max": {k: int(max(v)) for k, v in per_tensor.items()},
           "largest_image_flip": max(r["worst"]["color"] for r in rows),
           "largest_gradient_flip": max(max(v for k, v in r["worst"].items() if k != "color") for r in rows),
           "radii_diff_max": max(r["radii_diff"] for r in rows)}
    if all("kinds" in r for r in rows):
        agg = {}
        for r in rows:
            for group, d in r["kinds"].items():
                if isinstance(d, dict):
                    for k, n in d.items():
                        agg.setdefault(group, {})[k] = agg.setdefault(group, {}).get(k, 0) + n
        elems = {}
        for group in ("image_elements", "gradient_elements"):
            for k, n in agg.get(group, {}).items():
                elems[k] = elems.get(k, 0) + n
        tot_e = max(1, sum(elems.values()))
        if all("input_bits" in r for r in rows):
            out["input_bits_mean"] = {k: round(float(np.mean([r["input_bits"][k] for r in rows])), 1) for k in rows[0]["input_bits"]}
        out["flip_kinds"] = {"elements_by_kind": elems, "share_by_kind": {k: round(n / tot_e, 4) for k, n in elems.items()},
                             "per_view_mean_elements": {k: round(n / len(rows), 2) for k, n in elems.items()},
                             "detail": agg, "thresholds": {"alpha": KIND_THR_ALPHA, "T": KIND_THR_T, "power": KIND_THR_POWER},
                             "what": "flagged elements attributed to the decision that came out differently (oracle/rig_parity.py "
                                     "classify_flips): *_inputs = the two sides' means2D / conic bits differ and decide the pair "
                                     "differently, *_eval = same inputs, margin within the rounding of exp / of the product; own = at the "
                                     "Gaussian's own pair, same_pixel = elsewhere in the walk of one of its pixels"}
    return out

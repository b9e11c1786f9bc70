"""GPU integration: one GauSTAR refinement iteration (gaustar_trainers/refine.py:538-794) two ways on identical
parameters, camera and ground truth, and the windowed loop of config E (BASELINE.json configs[4]).

  fused              harness.SurfaceGaussians -> producers.points_rgb_depth -> ONE 4-channel render -> losses.rgb_depth_loss
  reference-shaped   oracle/producers_oracle (torch) -> TWO 3-channel renders through oracle/_ref/libgsr_ref.so (the
                     reference's own kernels: RGB with bg (0,1,0) at refine.py:552, depth-as-colour with bg = max depth at
                     :607) -> oracle/loss_oracle (torch l1 + ssim, masked depth L1), gradients by autograd

compared on the loss and on the gradient of every parameter the optimiser steps (sugar_optimizer.py:67-87):
_points, _scales, _quaternions, all_densities, _sh_coordinates_dc/_rest, _delta_t, _delta_r."""
import argparse
import os
import sys

import numpy as np
import pytest

import parity
from conftest import ROOT

pytestmark = pytest.mark.gpu

MAX_DEPTH = 10.0
PARAMS = ["_points", "_scales", "_quaternions", "all_densities", "_sh_coordinates_dc", "_sh_coordinates_rest", "_delta_t", "_delta_r"]


def _ref_render_fn():
    """libgsr_ref.so as an autograd op (the reference's _RasterizeGaussians, __init__.py:44-155, over its own kernels)."""
    import torch
    from oracle import ref

    class RefRender(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, colors, opac, scales, rots, cam, bg):
            rr = ref.RefRasterizer()
            color, _radii, _R = rr.forward(means3D.detach(), opac.detach(), cam.viewmatrix, cam.projmatrix, cam.campos, cam.W, cam.H,
                                           cam.tanfovx, cam.tanfovy, bg, colors_precomp=colors.detach(), scales=scales.detach(),
                                           rotations=rots.detach())
            ctx.rr = rr
            return color

        @staticmethod
        def backward(ctx, g):
            d = ctx.rr.backward(g.contiguous())
            return d["dL_dmeans3D"], d["dL_dcolors"], d["dL_dopacity"], d["dL_dscales"], d["dL_drotations"], None, None

    return RefRender.apply


def _one_iteration(level, W, H, cam_index, full_size):
    import torch
    from gaustar_amd import GaussianRasterizer, harness, losses, producers, scene
    from oracle import loss_oracle as lo
    from oracle import producers_oracle as po
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/libgsr_ref.so did not travel to this box")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    v, f = scene.icosphere(level, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
    verts, faces = torch.from_numpy(v).float().to(dev), torch.from_numpy(f).long().to(dev)
    model = harness.SurfaceGaussians(verts, faces, n_gaussians_per_surface_triangle=6, sh_levels=3, loose_bind=True).to(dev)
    N = model.n_points
    with torch.no_grad():   # a state in the middle of a refinement: coloured, slightly rotated / shifted off the faces
        model._sh_coordinates_dc.copy_((torch.rand(N, 1, 3, generator=g) * 2 - 1).to(dev))
        model._sh_coordinates_rest.copy_((0.15 * (torch.rand(N, 8, 3, generator=g) - 0.5)).to(dev))
        model._quaternions.add_((0.2 * torch.randn(N, 2, generator=g)).to(dev))
        model._scales.add_((0.1 * torch.randn(N, 2, generator=g)).to(dev))
        model.all_densities.add_((0.5 * torch.randn(N, 1, generator=g)).to(dev))
        edge = float((verts[faces[:, 0]] - verts[faces[:, 1]]).norm(dim=-1).mean())
        model._delta_t.add_((0.02 * edge * torch.randn(N, 3, generator=g)).to(dev))
        model._delta_r.add_((0.02 * torch.randn(N, 4, generator=g)).to(dev))
    cams = scene.ring_cameras(5, 32, W, H, focal_px=1200.0 * W / 1920.0)
    cam = cams[cam_index]
    ncam = harness.nerf_camera_from_scene(cam)
    rcam, view, proj, campos = ncam.on_device(dev)
    bg_rgb = torch.tensor([0.0, 1.0, 0.0], device=dev)

    RASTER_INPUTS = ["means3D", "colors_rgb", "colors_depth", "opacities", "scales", "rotations"]

    def fused_image(m, taps=None):
        bg4 = torch.cat([bg_rgb, torch.full((1,), MAX_DEPTH, device=dev)])
        settings, view_, campos_ = m._settings(ncam, bg4, 0)
        pts = m.points
        colors4 = producers.points_rgb_depth(pts, campos_, m.sh_coordinates, m.sh_levels, view_, depth_channels=1)
        ins = dict(means3D=pts, colors=colors4, opacities=m.strengths, scales=m.scaling, rotations=m.quaternions)
        if taps is not None:
            # the rasterizer's OWN input gradients: identity copies whose .grad receives nothing but the render's backward
            # (the producers' backward adds to the originals, not to these)
            ins = {k: v_ * 1.0 for k, v_ in ins.items()}
            for k, v_ in ins.items():
                v_.retain_grad()
            taps.update(ins)
        img, _ = GaussianRasterizer(settings)(means3D=ins["means3D"], means2D=torch.zeros_like(pts), opacities=ins["opacities"],
                                              colors_precomp=ins["colors"], scales=ins["scales"], rotations=ins["rotations"])
        return img

    # ground truth: a render of a perturbed copy (SURVEY.md 8d config E's synthetic-GT recipe)
    with torch.no_grad():
        gt_model = harness.SurfaceGaussians(verts, faces, 6, 3, loose_bind=True).to(dev)
        gt_model.load_state_dict(model.state_dict())
        gt_model._points.add_(0.3 * edge * torch.randn(verts.shape, generator=g).to(dev))
        gt_model._sh_coordinates_dc.add_(0.3 * torch.randn(N, 1, 3, generator=g).to(dev))
        gt = fused_image(gt_model)
        gt_rgb = gt[:3].clone()
        gt_depth = gt[3].clone()
        gt_depth[gt_depth >= MAX_DEPTH - 1e-3] = 2 * MAX_DEPTH     # real captures carry "far" values behind the subject

    # (i) fused
    taps_f = {}
    img = fused_image(model, taps_f)
    loss_f = losses.rgb_depth_loss(img, gt_rgb, gt_depth, MAX_DEPTH, 0.2, 1.0, 0.5)
    loss_f.backward()
    grads_f = {k: getattr(model, k).grad.detach().cpu().numpy() for k in PARAMS}
    cg = taps_f["colors"].grad
    rin_f = dict(means3D=taps_f["means3D"].grad, colors_rgb=cg[:, :3], colors_depth=cg[:, 3:4], opacities=taps_f["opacities"].grad,
                 scales=taps_f["scales"].grad, rotations=taps_f["rotations"].grad)
    rin_f = {k: v_.detach().cpu().numpy() for k, v_ in rin_f.items()}

    # (ii) reference-shaped, refine.py:552-660
    render = _ref_render_fn()

    def reference_shaped(nudge):
        P = {k: getattr(model, k).detach().clone().requires_grad_(True) for k in PARAMS}
        verts_in = P["_points"] * (1.0 + nudge)            # nudge = one float32 ulp: the sensitivity probe below
        pts, scl, quat = po.mesh_bound_gaussians(verts_in, faces, model.surface_triangle_bary_coords[..., 0], P["_scales"],
                                                 P["_quaternions"], float(model.surface_mesh_thickness), None, None, P["_delta_t"],
                                                 P["_delta_r"])
        sh = torch.cat([P["_sh_coordinates_dc"], P["_sh_coordinates_rest"]], 1)
        rgb = po.points_rgb(pts, campos[None], sh, 3)
        opac = torch.sigmoid(P["all_densities"])
        depth1 = pts @ view[:3, 2:3] + view[3, 2]                                                        # :603-605
        depth_col = depth1.expand(-1, 3)
        # identity copies per render: their .grad is what that render's backward alone returns for its inputs
        tap = lambda *ts: [t_ * 1.0 for t_ in ts]
        a = tap(pts, rgb, opac, scl, quat)
        b = tap(pts, depth1, opac, scl, quat)
        for t_ in a + b:
            t_.retain_grad()
        pred_rgb = render(a[0], a[1], a[2], a[3], a[4], rcam, bg_rgb)                                    # refine.py:552
        pred_depth = render(b[0], b[1].expand(-1, 3), b[2], b[3], b[4], rcam, torch.full((3,), MAX_DEPTH, device=dev))[0]   # :607, :616
        loss = lo.l1_dssim(pred_rgb[None], gt_rgb[None], 0.2)[0] + sum(lo.depth_mask_l1(pred_depth, gt_depth, MAX_DEPTH, 1.0, 0.5))
        loss.backward()
        # what autograd accumulates for inputs both renders share = the sum of the two backward passes
        rin = dict(means3D=a[0].grad + b[0].grad, colors_rgb=a[1].grad, colors_depth=b[1].grad, opacities=a[2].grad + b[2].grad,
                   scales=a[3].grad + b[3].grad, rotations=a[4].grad + b[4].grad)
        return float(loss), {k: P[k].grad.detach().cpu().numpy() for k in PARAMS}, {k: v_.detach().cpu().numpy() for k, v_ in rin.items()}

    loss_r, grads_r, rin_r = reference_shaped(0.0)

    # (iii) THIS library's rasterizer on the reference-shaped chain's own inputs (torch producers, torch losses): identical
    # Gaussians and camera on both sides, so the rasterizer-input gradients are held to the rasterizer's tolerances -- no
    # noise-floor allowance
    def hip_on_reference_inputs():
        P = {k: getattr(model, k).detach().clone() for k in PARAMS}
        with torch.no_grad():
            pts, scl, quat = po.mesh_bound_gaussians(P["_points"], faces, model.surface_triangle_bary_coords[..., 0], P["_scales"],
                                                     P["_quaternions"], float(model.surface_mesh_thickness), None, None, P["_delta_t"],
                                                     P["_delta_r"])
            rgb = po.points_rgb(pts, campos[None], torch.cat([P["_sh_coordinates_dc"], P["_sh_coordinates_rest"]], 1), 3)
            depth1 = pts @ view[:3, 2:3] + view[3, 2]
            opac = torch.sigmoid(P["all_densities"])
        ins = dict(means3D=pts, colors=torch.cat([rgb, depth1], 1), opacities=opac, scales=scl, rotations=quat)
        ins = {k: v_.clone().requires_grad_(True) for k, v_ in ins.items()}
        bg4 = torch.cat([bg_rgb, torch.full((1,), MAX_DEPTH, device=dev)])
        settings, _, _ = model._settings(ncam, bg4, 0)
        img4, _ = GaussianRasterizer(settings)(means3D=ins["means3D"], means2D=torch.zeros_like(pts), opacities=ins["opacities"],
                                               colors_precomp=ins["colors"], scales=ins["scales"], rotations=ins["rotations"])
        loss = lo.l1_dssim(img4[:3][None], gt_rgb[None], 0.2)[0] + sum(lo.depth_mask_l1(img4[3], gt_depth, MAX_DEPTH, 1.0, 0.5))
        loss.backward()
        cg_ = ins["colors"].grad
        out = dict(means3D=ins["means3D"].grad, colors_rgb=cg_[:, :3], colors_depth=cg_[:, 3:4], opacities=ins["opacities"].grad,
                   scales=ins["scales"].grad, rotations=ins["rotations"].grad)
        return float(loss), {k: v_.detach().cpu().numpy() for k, v_ in out.items()}

    loss_h, rin_h = hip_on_reference_inputs()
    assert abs(loss_h - loss_r) <= 2e-5 * max(1.0, abs(loss_r)), (loss_h, loss_r)
    # Noise floor of the comparison: the reference-shaped iteration against ITSELF with the vertices moved by one ulp.  The
    # losses are means over 2 M pixels (dL_dpix ~ 5e-7, smooth), so a Gaussian's gradient is a few pixel terms that nearly
    # cancel, and ONE (pixel, Gaussian) pair changing sides of the alpha >= 1/255 cut moves it by a sizeable fraction of the
    # tensor's maximum.  Rounding-level input differences flip such pairs in the reference itself; the fused path feeds the
    # blend inputs that differ from the torch chain's by rounding and is held to a small multiple of that floor.
    _, grads_n, rin_n = reference_shaped(2.0 ** -23)
    lf, lr = float(loss_f), loss_r
    print(f"[iteration] N={N} {W}x{H}: loss fused {lf:.7f} reference-shaped {lr:.7f}")
    assert abs(lf - lr) <= 2e-5 * max(1.0, abs(lr)), (lf, lr)
    if os.environ.get("GSR_DUMP_ITER"):
        np.savez(os.environ["GSR_DUMP_ITER"], **{"f_" + k: grads_f[k] for k in PARAMS}, **{"r_" + k: grads_r[k] for k in PARAMS})
    # The FUSED path's rasterizer-input gradients (before any producer's backward).  Its rasterizer inputs come out of the
    # fused producers and differ from the torch chain's by rounding; at config-C size that alone flips (pixel, Gaussian) pairs
    # across the alpha >= 1/255 cut (the floor below: the reference against itself one ulp away shows MORE flipped entries
    # than the fused path against the reference), so here the bound is ONE times the floor's count -- the factor 3 further
    # down is only for what the producers' chain rule makes of those entries -- and strict at the small size.
    for k in RASTER_INPUTS:
        ref = float(np.abs(rin_r[k]).max())
        fe = np.abs(rin_n[k].astype(np.float64) - rin_r[k])
        n_floor = int((fe > parity.GRAD_TOL * (ref + np.abs(rin_r[k]))).sum())
        worst_floor = float(fe.max() / max(ref, 1e-30))
        print(f"[iteration] rasterizer input {k}: reference vs itself + 1 ulp: {n_floor} of {fe.size} entries beyond tolerance, "
              f"worst {worst_floor:.2e} of max")
        parity.check_grad(rin_f[k], rin_r[k], f"iteration rasterizer-input {k} (fused producers)", small_tol=None,
                          max_outlier_frac=(n_floor + 16) / fe.size if full_size else 0.0, outlier_cap=max(0.05, 2 * worst_floor))
        # identical rasterizer inputs on both sides: strict at the small size.  At config-C size the few pairs per view
        # whose alpha lands on the other side of 1/255 (exp2f(x log2 e) against the reference's expf(x), one ulp apart --
        # the flips tests/parity.py allows 5e-5 of the entries for under a RANDOM image gradient) weigh more under this
        # loss's gradient (dL_dpix ~ 5e-7 and smooth: a Gaussian's gradient is a near-cancelling sum, the tensor's
        # maximum is small): measured 195 of 1 474 560 entries for means3D where the reference against itself one ulp
        # away shows 2 192.  Allowed: that tolerance or a QUARTER of the floor's count, whichever is larger.
        parity.check_grad(rin_h[k], rin_r[k], f"iteration rasterizer-input {k} (identical inputs)", small_tol=None,
                          max_outlier_frac=max(parity.FULL_GRAD_OUTLIERS, 0.25 * n_floor / fe.size) if full_size else 0.0,
                          outlier_cap=max(0.05, 1.25 * worst_floor))
    for k in PARAMS:
        ref = float(np.abs(grads_r[k]).max())
        tol = lambda x: parity.GRAD_TOL * (ref + np.abs(x))
        floor_err = np.abs(grads_n[k].astype(np.float64) - grads_r[k])
        floor_bad = floor_err > tol(grads_r[k])
        n_floor, worst_floor = int(floor_bad.sum()), float(floor_err.max() / ref)
        print(f"[iteration] {k}: reference vs itself + 1 ulp: {n_floor} of {floor_bad.size} entries beyond tolerance, worst {worst_floor:.2e} of max")
        parity.check_grad(grads_f[k], grads_r[k], f"iteration {k}", small_tol=None,
                          max_outlier_frac=(3 * n_floor + 16) / floor_bad.size, outlier_cap=max(0.05, 3 * worst_floor))


def test_fused_iteration_matches_reference_shaped_iteration_small():
    _one_iteration(level=3, W=320, H=240, cam_index=37, full_size=False)


def test_fused_iteration_matches_reference_shaped_iteration_config_c_size():
    """491 520 Gaussians (icosphere level 6, 6 per face), 1920x1080, a config-C rig camera."""
    _one_iteration(level=6, W=1920, H=1080, cam_index=37, full_size=True)


def test_windowed_refinement_loop(hip_lib):
    """Config E's shape at test size: 3 frames x 50 iterations, cameras in dist.shard_views order, parameters carried from
    frame to frame, gaustar_amd.optim.Adam stepping harness.SurfaceGaussians (tools/bench_window.py).  Every frame's loss
    must fall; the model's geometry cache must follow the optimiser's in-place updates."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_window
    r = bench_window.run(argparse.Namespace(frames=3, iters=50, level=3, width=320, height=240, cameras=16))
    assert r["gaussians"] == 20 * 4 ** 3 * 6 and len(r["frames"]) == 3
    for fr in r["frames"]:
        assert fr["loss_first"] == fr["loss_first"] and fr["loss_last"] == fr["loss_last"], "NaN loss"
        assert fr["loss_last"] < 0.95 * fr["loss_first"], r
    assert r["geometry_moved"] > 0.0, "optimiser steps did not reach the harness's cached geometry"
    print("[window]", r)


def test_refinement_iterations_reduce_the_loss(hip_lib):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_iteration
    r = bench_iteration.run(argparse.Namespace(steps=60, warmup=0, level=3, width=320, height=240))
    assert r["gaussians"] == 20 * 4 ** 3 * 6
    assert r["loss_first"] == r["loss_first"] and r["loss_last"] == r["loss_last"], "NaN loss"
    assert r["loss_last"] < 0.8 * r["loss_first"], r


def test_graph_free_step_equals_the_autograd_iteration(hip_lib):
    """harness.SurfaceGaussians.rgbd_step = render_channels(depth_channels=1) -> losses.rgb_depth_loss -> backward() without an
    autograd graph (the same Function bodies called directly): same loss bit for bit, same image, gradients equal up to the
    order of the backward's float atomics (an analytically zero gradient -- the in-plane rotation's here -- is nothing but that
    noise: absolute floor 1e-9); a second call ADDS to existing gradients like autograd does."""
    import torch
    from gaustar_amd import harness, losses, scene
    dev = torch.device("cuda:0")
    v, f = scene.icosphere(3, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
    model = harness.SurfaceGaussians(torch.from_numpy(v).float().to(dev), torch.from_numpy(f).long().to(dev), 6, 3).to(dev)
    g = torch.Generator(device=dev).manual_seed(4)
    with torch.no_grad():
        model._sh_coordinates_dc.copy_(torch.rand(model.n_points, 1, 3, device=dev, generator=g) * 2 - 1)
        model._sh_coordinates_rest.copy_(torch.randn(model._sh_coordinates_rest.shape, device=dev, generator=g) * 0.1)
    cam = harness.nerf_camera_from_scene(scene.ring_cameras(5, 32, 320, 240, focal_px=200.0)[37])
    bg4 = torch.tensor([0.0, 1.0, 0.0, 10.0], device=dev)
    gt_rgb = torch.rand(3, 240, 320, device=dev, generator=g)
    gt_d = torch.rand(240, 320, device=dev, generator=g) * 12.0
    params = [p for p in model.parameters() if p.requires_grad]
    img = model.render_channels(cam, bg4, depth_channels=1)[0]
    loss = losses.rgb_depth_loss(img, gt_rgb, gt_d, 10.0, 0.2, 1.0, 0.5)
    loss.backward()
    ref = [None if p.grad is None else p.grad.clone() for p in params]
    for p in params:
        p.grad = None
    loss2, img2, radii2 = model.rgbd_step(cam, bg4, gt_rgb, gt_d, 10.0, 0.2, 1.0, 0.5)
    assert not loss2.requires_grad and float(loss2) == float(loss) and torch.equal(img2, img.detach())
    n_with = 0
    for p, r in zip(params, ref):
        assert (p.grad is None) == (r is None)
        if r is not None and r.numel():
            n_with += 1
            assert torch.allclose(p.grad, r, rtol=2e-4, atol=1e-4 * float(r.abs().max()) + 1e-9), float((p.grad - r).abs().max())
    assert n_with >= 6
    model.rgbd_step(cam, bg4, gt_rgb, gt_d, 10.0, 0.2, 1.0, 0.5)            # gradients accumulate
    for p, r in zip(params, ref):
        if r is not None and r.numel():
            assert torch.allclose(p.grad, 2.0 * r, rtol=4e-4, atol=2e-4 * float(r.abs().max()) + 2e-9)
    # a device scalar as d(total)/d(loss)
    for p in params:
        p.grad = None
    model.rgbd_step(cam, bg4, gt_rgb, gt_d, 10.0, 0.2, 1.0, 0.5, grad_scale=torch.tensor(-0.5, device=dev))
    for p, r in zip(params, ref):
        if r is not None and r.numel():
            assert torch.allclose(p.grad, -0.5 * r, rtol=2e-4, atol=1e-4 * float(r.abs().max()) + 1e-9)

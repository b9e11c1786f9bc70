#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric: rasterizer views/s, forward+backward, 1920x1080,
491 520 mesh-bound surface Gaussians (config C), one view per GPU per step.

    python bench.py --gpus 1 --steps 40 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one view per rank: GaussianRasterizer forward + backward
through the public API (all inputs already resident in HBM).  For N > 1 that same forward + backward
sits inside one refinement step per rank on the reference loop's own parameters (producers, then the
gradient reduce-scatter / rank-sharded Adam / parameter all-gather over RCCL that stands where
sugar_optimizer.py:99-101 steps Adam) -- see build_refinement_workload.  Rank r renders camera
(step * N + r) mod 160 of the rig -- views shard, nothing else is exchanged.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      dominant kernel (the one with the largest summed time), algorithmic bytes per launch
                (SURVEY.md 8d per-unit figures x the units the launch processed) / its mean launch
                duration, measured with HIP events on the launch stream in a second, instrumented
                pass over the same K steps (gsr_profile_*), plus per-kernel means for every stage;
  cpu_baseline  the C oracle (oracle/gsr_oracle.c, a port of the reference algorithm -- the
                reference has no CPU path) timed on the host cores on a bounded sample: the first
                few views of the same workload at full size.  N = 1 / rank 0 only.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as tdist  # noqa: E402

from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, scene  # noqa: E402
from gaustar_amd import dist as gdist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
N_SIMD, CLOCK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMDs; a wave64 vector instruction occupies its SIMD for 4 cycles


def csrc_sha256():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gaustar_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def algorithmic_bytes(P, R, W, H, M=0):
    """B_alg = P*beta_P + R*beta_R + N_pix*40 + T*16 per view (fwd+bwd), and the
    share of each kernel of this library under the same per-unit figures."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    N = W * H
    beta_P = 464 if M == 0 else 494 + 48 * M
    total = P * beta_P + R * 160 + N * 40 + T * 16
    sh_in = 12 * M if M else 12
    per_kernel = {
        # reads 44 (+colours/SH) + writes 60 per Gaussian, + scan 8
        "preprocess_kernel": P * (44 + sh_in + 60 + (15 if M else 0) + 8),
        "tile_scan_kernel": T * 16,
        # key-emit reads 20 per Gaussian, key+value write 12 per instance
        "scatter_kernel": P * 20 + R * 12,
        # lists of up to 2 048 entries (every tile of this workload) are sorted INSIDE blend_fwd; this stage only times the
        # kernels for longer lists and carries no bytes of its own here
        "tile_sort_kernel": 0,
        # one logical sort pass (read + write) 24 + range detect 8 per instance, then list fetch 28 + colour 12 per
        # instance; 20 B written per pixel; ranges read
        "blend_fwd_kernel": R * 72 + N * 20 + T * 8,
        "zero_fill": P * (108 + 12 * M),
        # fetch 40 + one reduced 9-float flush 36 per instance; 20 B read per pixel; ranges read
        "blend_bwd_kernel": R * 76 + N * 20 + T * 8,
        # cov2D-bwd 56 in / 36 out + preprocess-bwd 92 in / 40 out (+SH)
        "geom_bwd_kernel": P * (56 + 36 + 92 + 40 + ((24 * M + 15) if M else 0)),
    }
    return total, per_kernel


def build_workload(device, rank):
    gs, cams, bg = scene.config_C()
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)
    params = dict(means3D=t(gs.means3D), opacities=t(gs.opacities), colors=t(gs.colors_precomp), scales=t(gs.scales),
                  rotations=t(gs.rotations))
    for p in params.values():
        p.requires_grad_(True)
    means2D = torch.zeros(gs.P, 3, device=device, requires_grad=True)
    bg_t = t(bg)
    rasters = []
    for cam in cams:   # camera tensors are built once and stay resident (the reference re-uploads per call)
        s = GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, bg_t, 1.0, t(cam.viewmatrix),
                                          t(cam.projmatrix), 0, t(cam.campos), False, False)
        rasters.append(GaussianRasterizer(s))
    W, H = cams[0].W, cams[0].H
    dpix = torch.randn(3, H, W, device=device, generator=torch.Generator(device=device).manual_seed(1234 + rank))
    return gs, cams, bg, params, means2D, rasters, dpix


# N > 1: what N view-parallel GPUs really exchange.  The step is one refinement step per rank on the reference loop's own
# parameters (harness.SurfaceGaussians: `_points` 3 floats per mesh vertex + 39 floats per Gaussian -- SH dc 3 + rest 24,
# density 1, scales 2, quaternions 2, delta_t 3, delta_r 4 -- = 77 MB at config C, SURVEY.md 8e / sugar_optimizer.py:67-87):
# mesh + SH producers -> the SAME rasterizer forward + backward as at N = 1 (same Gaussians, cameras, image gradient) ->
# producers' backward -> gradients reduce-scattered from autograd hooks as they become final -> Adam on this rank's
# 1/N of the parameters -> all-gather of the updated parameters (gaustar_amd.dist.ShardedAdam).  Learning rates are 0 (the
# reference constructs its Adam with lr = 0.0 and the trainer sets the rates, sugar_optimizer.py:87): the scene stays put,
# so every step and every rank renders the workload the N = 1 line is quoted on, while the optimiser moves all its bytes.
SH0 = 0.28209479177387814


def build_refinement_workload(device, rank, gs, cams, bg, overlap=True, solo=False, communicate=True):
    from gaustar_amd import harness
    v, f = scene.icosphere(6, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
    verts, faces = torch.from_numpy(v).float().to(device), torch.from_numpy(f).long().to(device)
    model = harness.SurfaceGaussians(verts, faces, 6, sh_levels=3, surface_mesh_thickness=scene._extent_thickness(),
                                     loose_bind=True).to(device)
    assert model.n_points == gs.P
    with torch.no_grad():   # config C's opacities and colours, expressed in the harness's parameters
        op = torch.from_numpy(gs.opacities).to(device).clamp(1e-4, 1 - 1e-4)
        model.all_densities.copy_(torch.log(op / (1 - op)).view(-1, 1))
        model._sh_coordinates_dc.copy_(((torch.from_numpy(gs.colors_precomp).to(device) - 0.5) / SH0).view(-1, 1, 3))
    ncams = [harness.nerf_camera_from_scene(c) for c in cams]
    groups = [{"params": [model._points], "lr": 0.0},
              {"params": [model._sh_coordinates_dc, model._sh_coordinates_rest], "lr": 0.0},
              {"params": [model._scales, model._quaternions, model.all_densities, model._delta_t, model._delta_r], "lr": 0.0}]
    opt = gdist.ShardedAdam(groups, ready_order=model.grad_ready_order(), eps=1e-15, overlap=overlap, run_at_world_size_1=solo,
                            communicate=communicate, gather_first=model.mesh_parameters())
    model.grad_sink = opt   # the render's backward writes the parameter gradients straight into the optimiser's flat buffer
    bg_t = torch.from_numpy(np.ascontiguousarray(bg, dtype=np.float32)).to(device)
    dpix = torch.randn(3, cams[0].H, cams[0].W, device=device, generator=torch.Generator(device=device).manual_seed(1234 + rank))
    return model, ncams, opt, bg_t, dpix


def refinement_step(step, rank, world, model, ncams, opt, bg_t, dpix):
    ncam = ncams[(step * world + rank) % len(ncams)]
    opt.zero_grad(set_to_none=True)
    # parameters in, image out: mesh producer -> SH colours + sigmoid -> rasterizer as one autograd node (harness.py)
    color, _radii = model.render_channels(ncam, bg_t, depth_channels=0)
    color.backward(dpix)
    opt.step()
    return color


def one_step(step, rank, world, params, means2D, rasters, dpix):
    r = rasters[(step * world + rank) % len(rasters)]
    for p in params.values():
        p.grad = None
    means2D.grad = None
    color, radii = r(means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                     colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
    color.backward(dpix)
    return color


def reference_caller_step(step, cams, bg_t, params, means2D, dpix):
    """One forward + backward the way the reference's mesh-bound caller drives the rasterizer (gaustar_scene/sugar_model.py:
    1149-1187): on EVERY call a fresh `torch.Tensor(getWorld2View(...)).transpose(0, 1).cuda()`, a fresh projection, their
    product by bmm, a fresh camera centre, a fresh GaussianRasterizationSettings / GaussianRasterizer -- nothing persists, so a
    camera is recognisable by the contents of its matrices only (gsr_camera_key)."""
    import math
    cam = cams[step % len(cams)]
    world_view_transform = torch.Tensor(cam._w2v).transpose(0, 1).cuda()
    proj_transform = torch.from_numpy(cam._proj).transpose(0, 1).cuda()
    full_proj_transform = (world_view_transform.unsqueeze(0).bmm(proj_transform.unsqueeze(0))).squeeze(0)
    camera_center = torch.Tensor(cam._center).cuda()
    rs = GaussianRasterizationSettings(image_height=int(cam.H), image_width=int(cam.W), tanfovx=np.float32(cam.tanfovx),
                                       tanfovy=np.float32(cam.tanfovy), bg=bg_t, scale_modifier=1., viewmatrix=world_view_transform,
                                       projmatrix=full_proj_transform, sh_degree=0, campos=camera_center, prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(raster_settings=rs)
    for p in params.values():
        p.grad = None
    means2D.grad = None
    color, _radii = rasterizer(means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                               colors_precomp=params["colors"], scales=params["scales"], rotations=params["rotations"])
    color.backward(dpix)
    return color


def prepare_reference_caller(cams):
    """(host-side inputs of reference_caller_step, made once: what getWorld2View / getProjectionMatrix return per call is a few
    microseconds of numpy that belong to the caller, not to the rasterizer)"""
    import math
    for cam in cams:
        cam._w2v = np.ascontiguousarray(np.asarray(cam.viewmatrix, np.float32).T)
        cam._proj = scene.get_projection_matrix(1e-4, 100.0, 2.0 * math.atan(cam.tanfovx), 2.0 * math.atan(cam.tanfovy))
        cam._center = np.asarray(cam.campos, np.float32)[None].copy()


def timed(fn, steps, world, device, prewarm=2, on_timed_start=None):
    """on_timed_start: called after the untimed first steps have drained, right before the K timed ones (the instrumented
    passes reset the per-kernel brackets there, so that launches_per_step counts the K timed steps only)."""
    import gc
    gc.collect(); gc.disable()   # (a generation-2 pass of Python's collector stops the host for tens of milliseconds; between timed regions, not inside)
    # (untimed: the collector pass above and whatever ran before this region -- another mode, the other config's workload --
    # leave the GPU idle for milliseconds; a 20-step region is 6 ms, so the first steps after an idle gap would be a third of it)
    for s in range(prewarm):
        fn(s)
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize(device)
    if on_timed_start is not None:
        on_timed_start()
    t0 = time.perf_counter()
    try:
        for s in range(steps):
            fn(s)
        torch.cuda.synchronize(device)
    finally:
        gc.enable()
    if world > 1:
        tdist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        tdist.all_reduce(tt, op=tdist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def cpu_baseline(gs, cams, bg, budget_s=20.0):
    """The oracle (CPU port of the reference algorithm) on the host cores, same workload, full size."""
    from oracle import oracle
    ncores = os.cpu_count() or 1
    oracle.set_threads(ncores)
    rng = np.random.default_rng(0)
    dpix = rng.normal(size=(3, cams[0].H, cams[0].W)).astype(np.float32)
    t_total, n = 0.0, 0
    while n < 8 and (n == 0 or t_total + t_total / n < budget_s):
        cam = cams[n]
        t0 = time.perf_counter()
        st = oracle.forward(gs.means3D, gs.opacities, cam.viewmatrix, cam.projmatrix, cam.campos, cam.W, cam.H,
                            cam.tanfovx, cam.tanfovy, bg, colors_precomp=gs.colors_precomp, scales=gs.scales,
                            rotations=gs.rotations)
        oracle.backward(st, dpix)
        t_total += time.perf_counter() - t0
        n += 1
    out = {"value": n / t_total, "unit": "views/s", "cores": ncores, "kind": "port",
           "sample": f"first {n} of the 160 views of the same workload (491520 Gaussians, 1920x1080, fwd+bwd), "
                     f"C restatement of the reference algorithm with OpenMP over {ncores} threads, {t_total:.1f} s"}
    try:   # SURVEY.md 8d's own form of the CPU figure: configs[0] (10k random Gaussians, 512x512), median of 7 runs
        ga, ca, bga = scene.config_A()
        dA = rng.normal(size=(3, ca.H, ca.W)).astype(np.float32)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            st = oracle.forward(ga.means3D, ga.opacities, ca.viewmatrix, ca.projmatrix, ca.campos, ca.W, ca.H, ca.tanfovx,
                                ca.tanfovy, bga, colors_precomp=ga.colors_precomp, scales=ga.scales, rotations=ga.rotations)
            oracle.backward(st, dA)
            ts.append(time.perf_counter() - t0)
        out["config_A"] = {"median_ms_per_view": round(float(np.median(ts)) * 1e3, 3), "runs": 7,
                           "what": "BASELINE configs[0]: 10k random Gaussians, one camera at 512x512, fwd+bwd, same C port and threads"}
    except Exception as ex:
        out["config_A"] = {"error": repr(ex)[:200]}
    return out


def issue_statistics(lib, rs, params, device):
    """Live (pixel, Gaussian) pairs of one view and the backward blend's pair trips, from the candidate words and
    n_contrib the forward leaves behind (gsr_debug_export / gsr_debug_export_masks): a pixel's live pairs are the set bits
    of its words below its last contributor (the words are tight: 1.006 bits per pair that passes the reference's test,
    tests/test_gpu_masks.py); a (unit, 8x8 block) pair makes one trip per list position ANY of its pixels replays."""
    from gaustar_amd import rasterizer as rz
    e = torch.Tensor([])
    P = int(params["means3D"].shape[0])
    W, H = rs.image_width, rs.image_height
    out = rz.rasterize_gaussians_native(rs.bg, params["means3D"].detach(), params["colors"].detach(), params["opacities"].detach(),
                                        params["scales"].detach(), params["rotations"].detach(), 1.0, e, rs.viewmatrix, rs.projmatrix,
                                        rs.tanfovx, rs.tanfovy, H, W, e, 0, rs.campos, False, False, need_backward=True,
                                        use_plan=False)   # (exact binning: compact unit numbering, true num_rendered)
    Rn, _color, _radii, geom, binning, img, _maxc, U = out
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    m2 = torch.zeros(P, 2, device=device); co = torch.zeros(P, 4, device=device)
    rng_ = torch.zeros(T, 2, dtype=torch.int32, device=device); pl = torch.zeros(max(Rn, 1), dtype=torch.int32, device=device)
    fT = torch.zeros(H, W, device=device); nc = torch.zeros(H, W, dtype=torch.int32, device=device)
    masks = torch.zeros(max(U, 1), 4, 64, dtype=torch.int64, device=device)
    torch.cuda.synchronize(device)
    p_ = lambda x: x.data_ptr()
    _lib.check(lib.gsr_debug_export(P, Rn, U, W, H, p_(geom), p_(binning), p_(img), p_(m2), p_(co), None, None, p_(rng_), p_(pl),
                                    p_(fT), p_(nc), None), "gsr_debug_export")
    _lib.check(lib.gsr_debug_export_masks(Rn, U, p_(binning), p_(masks), None), "gsr_debug_export_masks")
    torch.cuda.synchronize(device)
    ranges = rng_.cpu().numpy().astype(np.int64)
    n = ranges[:, 1] - ranges[:, 0]
    units = (n + 63) // 64
    unit0 = np.concatenate([[0], np.cumsum(units)])
    assert int(unit0[-1]) == U, (int(unit0[-1]), U)
    words = masks.cpu().numpy().view(np.uint64)                                  # [U, 4, 64]
    ncp = np.zeros((gy * 16, gx * 16), np.int64)
    ncp[:H, :W] = nc.cpu().numpy()
    nct = ncp.reshape(gy, 2, 8, gx, 2, 8).transpose(0, 3, 1, 4, 2, 5).reshape(T, 4, 64)   # [tile, block 2*by+bx, lane 8*y+x]
    tile_of = np.repeat(np.arange(T), units)
    s0 = 64 * (np.arange(U) - unit0[tile_of])
    lim = np.clip(nct[tile_of] - s0[:, None, None], 0, 64).astype(np.uint64)
    below = np.where(lim >= 64, np.uint64(0xffffffffffffffff), (np.uint64(1) << (lim & np.uint64(63))) - np.uint64(1))
    live_words = words & below
    live = int(np.bitwise_count(live_words).sum())
    kept = np.bitwise_or.reduce(live_words, axis=2)                              # [U, 4]: positions any pixel of the block replays
    trips = int(np.bitwise_count(kept).sum())
    # The backward blend (gsr_blend_bwd.hip, round 6) walks a block 4x4 sub-block by sub-block, four kept instances of the
    # sub-block per trip, the unit's kept instances in chunks of 32 (deepest first): trips = sum over (unit, block, chunk,
    # sub-block) of ceil(kept instances of the sub-block in the chunk / 4).
    lanes_ = np.arange(64)
    sub_of = ((lanes_ >> 5) << 1) | ((lanes_ >> 2) & 1)
    kept_sub = np.stack([np.bitwise_or.reduce(live_words[:, :, sub_of == q], axis=2) for q in range(4)], axis=2)   # [U, 4, 4]
    sub_pairs = int(np.bitwise_count(kept_sub).sum())
    pc = np.bitwise_count(kept)
    quad_trips = int(((np.bitwise_count(kept_sub).astype(np.int64) + 3) // 4)[pc <= 32].sum())
    for u_, b_ in zip(*np.nonzero(pc > 32)):          # (the few unit-blocks with a second chunk)
        k_ = int(kept[u_, b_])
        pos = [i for i in range(63, -1, -1) if (k_ >> i) & 1]                     # deepest first = highest position first
        for c0 in range(0, len(pos), 32):
            cm = 0
            for i in pos[c0:c0 + 32]:
                cm |= 1 << i
            quad_trips += sum((bin(int(kept_sub[u_, b_, q]) & cm).count("1") + 3) // 4 for q in range(4))
    # Snapshots (one float4 per pixel and unit, written by the forward when the pixel first consumes a candidate of the unit):
    # NEEDED by the backward = the pixel has a candidate in the unit and its last contributor lies in or behind it (exactly the
    # snapshots of units the pixel reached before its last contributor); WRITTEN beyond those: a pixel walks on behind its
    # last contributor until a candidate stops it (T (1 - alpha) < 1e-4) -- if that candidate opens a new unit, one more
    # snapshot is written that nobody reads (upper estimate: every pixel with a candidate in a later unit writes one).
    has = words != 0
    u_local = (np.arange(U) - unit0[tile_of])[:, None, None]
    last_unit = np.where(nct > 0, (nct - 1) // 64, -1)[tile_of]
    needed = int((has & (u_local > 0) & (u_local <= last_unit)).sum())
    later = has & (u_local > 0) & (u_local > last_unit)
    nz_tiles = np.nonzero(units > 0)[0]
    extra = int((np.add.reduceat(later.astype(np.int32), unit0[nz_tiles], axis=0) > 0).sum()) if len(nz_tiles) else 0
    snaps = {"needed_by_backward": needed, "written_upper_estimate": needed + extra,
             "needed_over_written": round(needed / max(needed + extra, 1), 4),
             "what": "per (pixel, unit) snapshots of view 0; written = needed + at most one per pixel whose walk went on into a later unit"}
    return {"live_pairs_per_view": live, "bwd_pair_trips_per_view": trips, "bwd_unit_blocks_with_work": int((kept != 0).sum()),
            "snapshots": snaps,
            "bwd_live_lanes_per_pair_trip": round(live / max(trips, 1), 2),
            "bwd_subblock_instances_per_view": sub_pairs, "bwd_trips_per_view": quad_trips,
            "bwd_live_lanes_per_trip": round(live / max(quad_trips, 1), 2),
            "bwd_what": "bwd_pair_trips = (kept instance, 8x8 block) pairs = the trips of rounds 1-5's uniform pair loop (one instance "
                        "on 64 pixels); bwd_trips = trips of round 6's kernel (four kept instances of a 4x4 sub-block on its 16 pixels)",
            "num_rendered": int(Rn), "units": int(U)}


def other_config(name, device, lib, steps=20, repeats=3):
    """BASELINE.json's other single-GPU configs through the same public API, outside the timed region: B (configs[1]: 200 400
    mesh-bound Gaussians, precomputed colours), D (configs[3]: 1 001 232 Gaussians, SH degree 3 evaluated in the kernels;
    refine.py:552's pass) and D_depth (the same geometry, depth as colour with bg = 10; refine.py:607's pass).  Per config:
    median ms per view over `repeats` regions of `steps` forward + backward passes, num_rendered, the algorithmic bytes of
    SURVEY.md 8(d) with the config's own beta_P (464 for M = 0, 494 + 48 M for in-kernel SH: 1 262 at M = 16), per-kernel
    HIP-event means with each kernel's share of those bytes, and the fraction of the 8 TB/s roofline for the path and for
    the kernel with the best and the largest figure."""
    from gaustar_amd import rasterizer as rz
    if name == "C_solid":
        # config C's geometry rendered the way sugar_model.py:1230-1232 does under use_solid_surface (refined_mesh.py:771, :1130):
        # the two in-plane scales raised to at least their mean.  The synthetic icosphere's triangles are all alike, so the
        # in-plane scales first get the spread trained surfels have (log-normal, sigma 1.0, seeded) -- then R >> P, the case SURVEY a7
        # warns about (sort + blend traffic grow with R).
        gs, cams_, bg = scene.config_C()
        cam = cams_[0]
        sc_ = np.array(gs.scales, dtype=np.float32, copy=True)
        sc_[:, 1:] *= np.exp(np.random.default_rng(7).normal(0.0, 1.0, size=(gs.P, 1))).astype(np.float32)
        sc_[:, 1:] = np.maximum(sc_[:, 1:].mean(), sc_[:, 1:])
        gs.scales = sc_
    else:
        gs, cam, bg = scene.config_B() if name == "B" else scene.config_D()
    t = lambda x, g=False: None if x is None else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device).requires_grad_(g)
    shs, cols, deg = gs.shs, gs.colors_precomp, gs.sh_degree
    if name == "D_depth":
        shs, cols, deg, bg = None, scene.view_depth_colors(gs, cam), 0, np.array([10.0, 10.0, 10.0], np.float32)
    M = 0 if shs is None else int(shs.shape[1])
    m3, op, sc, ro = t(gs.means3D, True), t(gs.opacities, True), t(gs.scales, True), t(gs.rotations, True)
    sh_t, col_t = t(shs, True), t(cols, True)
    m2 = torch.zeros(gs.P, 3, device=device, requires_grad=True)
    st = GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, t(bg), 1.0, t(cam.viewmatrix), t(cam.projmatrix), deg,
                                       t(cam.campos), False, False)
    r = GaussianRasterizer(st)
    dp = torch.randn(3, cam.H, cam.W, device=device, generator=torch.Generator(device=device).manual_seed(99))
    leaves = [x for x in (m3, op, sc, ro, sh_t, col_t, m2) if x is not None]

    def step(_s):
        for x in leaves:
            x.grad = None
        img, _ = r(means3D=m3, means2D=m2, opacities=op, shs=sh_t, colors_precomp=col_t, scales=sc, rotations=ro)
        img.backward(dp)
    for s_ in range(4):
        step(s_)
    dts = [timed(step, steps, 1, device) for _ in range(repeats)]
    ms = float(np.median(dts)) / steps * 1e3
    nst = lib.gsr_num_stages()
    names = [lib.gsr_stage_name(i).decode() for i in range(nst)]
    msv = (ctypes.c_float * nst)()
    cnt = (ctypes.c_int * nst)()
    lib.gsr_profile_enable(1)
    timed(step, steps, 1, device, on_timed_start=lambda: lib.gsr_profile_read(msv, cnt, 1))   # (brackets of the K timed steps only)
    _lib.check(lib.gsr_profile_read(msv, cnt, 1), "gsr_profile_read")
    lib.gsr_profile_enable(0)
    e = torch.Tensor([])
    Rn = rz.rasterize_gaussians_native(st.bg, m3.detach(), e if col_t is None else col_t.detach(), op.detach(), sc.detach(), ro.detach(),
                                       1.0, e, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, cam.H, cam.W,
                                       e if sh_t is None else sh_t.detach(), deg, st.campos, False, False, use_plan=False)[0]
    total_b, per_kernel_b = algorithmic_bytes(gs.P, Rn, cam.W, cam.H, M)
    kern = {}
    for i, nme in enumerate(names):
        if cnt[i]:
            lps, mean_ms = cnt[i] / steps, msv[i] / cnt[i]
            gbs = per_kernel_b.get(nme, 0) / lps / (mean_ms * 1e-3) / 1e9
            kern[nme] = {"ms_per_launch": round(mean_ms, 5), "launches_per_step": lps, "alg_bytes_per_step": int(per_kernel_b.get(nme, 0)),
                         "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    dom = max(kern, key=lambda k_: kern[k_]["ms_per_launch"] * kern[k_]["launches_per_step"])
    best = max(kern, key=lambda k_: kern[k_]["frac"])
    path_gbs = total_b / (ms * 1e-3) / 1e9
    out = {"gaussians": gs.P, "sh_coeffs_in_kernel": M, "beta_P": 464 if M == 0 else 494 + 48 * M, "image": [cam.W, cam.H],
           "ms_per_view": round(ms, 4), "views_per_s": round(1e3 / ms, 1),
           "ms_per_view_min_max": [round(min(dts) / steps * 1e3, 4), round(max(dts) / steps * 1e3, 4)], "steps": steps, "repeats": repeats,
           "num_rendered": int(Rn), "alg_bytes_per_view": int(total_b), "path_achieved_GBps": round(path_gbs, 1),
           "path_frac": round(path_gbs / HBM_PEAK_GBS, 4), "dominant_kernel": dom, "dominant_frac": kern[dom]["frac"],
           "best_kernel": best, "best_frac": kern[best]["frac"], "kernels": kern}
    del m3, op, sc, ro, sh_t, col_t, m2, r, dp, leaves
    torch.cuda.empty_cache()
    return out


def window_benchmark():
    """BASELINE.json configs[4]'s shape at config-C size on this GPU (tools/bench_window.py: 2 frames x 50 iterations,
    491 520 Gaussians, 1080p, 160 cameras; producers -> one 4-channel render -> losses -> backward -> Adam)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_window
    # (one short untimed pass first: kernel modules, the caching allocator and the binning-size hint warm up outside the figure)
    bench_window.run(argparse.Namespace(frames=1, iters=8, level=6, width=1920, height=1080, cameras=16))
    r = bench_window.run(argparse.Namespace(frames=2, iters=50, level=6, width=1920, height=1080, cameras=160))
    out = {"iterations_per_s": r["iterations_per_s"], "ms_per_iteration": r["ms_per_iteration"],
           "median_ms_per_iteration": r["median_ms_per_iteration"], "p90_ms_per_iteration": r["p90_ms_per_iteration"],
           "per_frame_ms_per_iteration": [f["ms_per_iteration"] for f in r["frames"]], "frames": 2, "iterations_per_frame": 50,
           "gaussians": r["gaussians"], "cameras": r["cameras"], "image": r["image"],
           "what": "tools/bench_window.py: refinement loop of config E's shape at config-C size, one GPU, outside the timed region; "
                   "100 iterations over 160 cameras: no camera is rendered twice, every view is its camera's first (exact binning)"}
    # the same 2 x 50 iterations with render + losses + backward as ONE graph-free call (harness.SurfaceGaussians.rgbd_step: the
    # same kernels; without autograd's per-iteration host work the loop is GPU-bound on slow hosts too)
    r1 = bench_window.run(argparse.Namespace(frames=2, iters=50, level=6, width=1920, height=1080, cameras=160, fused_step=True))
    out["fused_step"] = {"median_ms_per_iteration": r1["median_ms_per_iteration"], "p90_ms_per_iteration": r1["p90_ms_per_iteration"],
                         "ms_per_iteration": r1["ms_per_iteration"], "iterations_per_s": r1["iterations_per_s"],
                         "host_wait_ms_per_iteration": r1.get("host_wait_ms_per_iteration"),
                         "what": "the same loop through SurfaceGaussians.rgbd_step (no autograd graph: the two Functions' forward / "
                                 "backward bodies called back to back, .grad set directly); identical kernels and numbers"}
    out["host_wait_ms_per_iteration"] = r.get("host_wait_ms_per_iteration")
    # The reference refines 2 000 iterations per frame (train_seq.py:45: 12.5 views per camera); 480 per frame = 3 per camera
    # show what a loop that comes back to its cameras pays: the later views are binned by their camera's plan, those the moving
    # Gaussians have outgrown fall back (plan_stats).
    from gaustar_amd import rasterizer as rz
    before = dict(rz.PLAN_STATS)
    r2 = bench_window.run(argparse.Namespace(frames=2, iters=480, level=6, width=1920, height=1080, cameras=160))
    out["revisits"] = {"median_ms_per_iteration": r2["median_ms_per_iteration"], "ms_per_iteration": r2["ms_per_iteration"],
                       "iterations_per_s": r2["iterations_per_s"], "frames": 2, "iterations_per_frame": 480,
                       "plan_stats": {k: rz.PLAN_STATS[k] - before[k] for k in before},
                       "what": "the same loop, 480 iterations per frame (3 views per camera; ground-truth renders included in plan_stats)"}
    return out


def ref_gpu_baseline(gs, cams, bg, device, views=4):
    """The REFERENCE's own kernels built for gfx950 (oracle/_ref/libgsr_ref.so, oracle/build_ref.sh: hipify-perl of the
    sources under /root/reference + hipCUB), forward + backward on the first `views` config-C views, inputs resident."""
    from oracle import ref
    if not ref.available():
        return None
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)
    a = dict(means3D=t(gs.means3D), opacities=t(gs.opacities), colors_precomp=t(gs.colors_precomp), scales=t(gs.scales),
             rotations=t(gs.rotations))
    dpix = torch.randn(3, cams[0].H, cams[0].W, device=device)
    bg_t = t(bg)
    camt = [(t(c.viewmatrix), t(c.projmatrix), t(c.campos), c) for c in cams[:views]]

    def one(i):
        v, p, cp, c = camt[i]
        rr = ref.RefRasterizer(str(device))
        rr.forward(a["means3D"], a["opacities"], v, p, cp, c.W, c.H, c.tanfovx, c.tanfovy, bg_t, colors_precomp=a["colors_precomp"],
                   scales=a["scales"], rotations=a["rotations"])
        rr.backward(dpix)
    one(0)
    torch.cuda.synchronize(device); t0 = time.perf_counter()
    for i in range(views):
        one(i)
    torch.cuda.synchronize(device); dt = time.perf_counter() - t0
    return {"value": round(views / dt, 1), "unit": "views/s", "ms_per_view": round(dt / views * 1e3, 3), "views": views,
            "what": "the reference's kernels (diff-gaussian-rasterization) compiled for gfx950 by oracle/build_ref.sh, fwd+bwd, "
                    "same workload, one GPU, outside the timed region; includes the wrapper's per-call allocations and syncs"}


def parity_leg(gs, cams, bg, device, every=10):
    """Threshold-flip counts against the reference build (oracle/rig_parity.py, the checker of tests/test_gpu_rig_parity.py) on
    every `every`-th view of the rig: how many elements per view sit outside the 1e-4 tolerance, and how far."""
    from oracle import ref, rig_parity
    if not ref.available():
        return None
    s_ = rig_parity.summarise(rig_parity.compare_views(gs, cams, bg, range(0, len(cams), every), device=str(device), classify=True))
    s_["what"] = ("elements per view (image + six gradient tensors) outside tests/parity.py's tolerances against the reference's "
                  "kernels built for gfx950: pairs on the other side of alpha >= 1/255 or T < 1e-4 (exp2-domain alpha); outside the "
                  "timed region")
    return s_


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=160)   # one pass over the 160-camera rig
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the window / reference-GPU / issue-statistics legs (N = 1 only)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed region is run this many times back to back (modes interleaved); the MEDIAN is reported")
    ap.add_argument("--scale-step", action="store_true",
                    help="N = 1: time the refinement step that --gpus N > 1 times (producers + render + Adam) instead of the rasterizer "
                         "alone, so that value(N) / value(1) is the scaling of ONE thing (also GSR_BENCH_SCALE_STEP=1)")
    ap.add_argument("--views-in-flight", type=int, default=2,
                    help="N = 1: independent view pipelines (host thread + HIP stream each, gaustar_amd.pipelines); 1 = one view at a time")
    args = ap.parse_args()

    # GSR_BENCH_BACKEND=gloo lets the multi-rank path be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices; RCCL itself refuses two ranks on one GPU).  Default: nccl (= RCCL).
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and int(os.environ.get("WORLD_SIZE", "1")) > ndev:
        raise SystemExit(f"WORLD_SIZE={os.environ.get('WORLD_SIZE')} ranks but only {ndev} GPUs visible")
    local_env = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_env % max(ndev, 1))
    rank, world, local = gdist.init_from_env(backend)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = torch.device("cuda", (local % max(ndev, 1)) if world > 1 else 0)
    torch.cuda.set_device(device)
    # one process per GPU, pinned to a few cores of the GPU's NUMA node (see gaustar_amd.dist.bind_to_local_cpus);
    # ranks whose GPUs share a node take disjoint cores.  GSR_BENCH_NO_PIN=1 leaves placement to the OS.
    cpus_before = os.sched_getaffinity(0)
    pinned = []
    if os.environ.get("GSR_BENCH_NO_PIN", "0") != "1":
        node_of = [tuple(gdist._gpu_local_cpus(i)) for i in range(ndev)]
        same = [i for i in range(ndev) if node_of[i] == node_of[device.index]] if world > 1 else [device.index]
        pinned = gdist.bind_to_local_cpus(device.index, slot=same.index(device.index), slots=len(same))
    lib = _lib.load()

    gs, cams, bg, params, means2D, rasters, dpix = build_workload(device, rank)
    scale_step = world == 1 and (args.scale_step or os.environ.get("GSR_BENCH_SCALE_STEP", "0") == "1")
    R = max(1, args.repeats)
    med = lambda xs: float(np.median(xs))
    spread = lambda dts: {"repeats": len(dts), "ms_per_step_min": round(min(dts) / args.steps * 1e3, 4),
                          "ms_per_step_max": round(max(dts) / args.steps * 1e3, 4)}
    refine, same_step_1gpu_ms = None, None
    if world > 1:
        # the SAME step without any exchange, every rank on its own GPU at once: the single-GPU denominator of this step's
        # scaling (the N = 1 line of this script times the rasterizer alone, which is what the metric is quoted on)
        solo_w = build_refinement_workload(device, rank, gs, cams, bg, communicate=False)
        solo_step = lambda s: refinement_step(s, rank, world, *solo_w)
        for s in range(5):
            solo_step(s)
        n_solo = max(10, min(args.steps, 40))
        same_step_1gpu_ms = med([timed(solo_step, n_solo, world, device) for _ in range(3)]) / n_solo * 1e3
        solo_w[2].close(); del solo_w, solo_step
        refine = build_refinement_workload(device, rank, gs, cams, bg)
        step = lambda s: refinement_step(s, rank, world, *refine)
    elif scale_step:
        refine = build_refinement_workload(device, rank, gs, cams, bg)
        step = lambda s: refinement_step(s, 0, 1, *refine)
    else:
        step = lambda s: one_step(s, rank, world, params, means2D, rasters, dpix)
    raster_step = lambda s: one_step(s, rank, world, params, means2D, rasters, dpix)   # (the instrumented pass below)

    for s in range(args.warmup):
        step(s)
    # Planning pass (untimed): every camera the timed regions visit is rendered once more, so that each has a PLAN of its
    # buckets (include/gsr.h gsr_forward_planned: a camera's first view renders the exact way -- scan + scatter -- and leaves
    # the plan its later views are binned by).  The timed regions then measure the rig's steady state, which is what a
    # refinement run of thousands of iterations over 160 cameras spends its time in; the exact-binning figure of the same
    # steps is reported beside it (`exact_binning`).
    from gaustar_amd import rasterizer as rz_plan
    for s in range(args.steps):
        step(s)
    plan_stats0 = dict(rz_plan.PLAN_STATS)
    wait_ns, waits = ctypes.c_longlong(0), ctypes.c_longlong(0)
    V = max(1, args.views_in_flight) if (world == 1 and not scale_step) else 1
    single, value_spread = None, None
    if V > 1:
        # V independent view pipelines (gaustar_amd/pipelines.py): every step is still ONE complete forward + backward of one
        # view through the public API; step s runs on pipeline s % V, with that pipeline's own leaf tensors, stream and host
        # thread.  The K timed steps are K views, as before.  The one-view-at-a-time figure is measured next to it, the two
        # modes INTERLEAVED, R times each (single, pipelined, single, ...): the medians are what is reported -- a 20-step region
        # is 6 ms, and one hiccup of the box moved a single measurement by 10 %.
        from gaustar_amd import pipelines
        pipes = pipelines.ViewPipelines(V, device)
        leaves = pipelines.clone_leaves(dict(params, means2D=means2D), V)

        def pipe_step(t, s):
            ps = leaves[t]
            one_step(s, 0, 1, {k: v for k, v in ps.items() if k != "means2D"}, ps["means2D"], rasters, dpix)
        import gc
        clock = {}
        dts_single, dts_pipe = [], []
        single_wait_ns, single_waits = 0, 0
        pipe_error = None
        gc.collect(); gc.disable()
        try:
            pipes.run(pipe_step, list(range(max(args.warmup, 2 * V))))   # untimed: every pipeline warms its stream and allocator
            pipes.run(pipe_step, list(range(args.steps)))                # untimed: the planning pass of each pipeline's cameras
            plan_stats0 = dict(rz_plan.PLAN_STATS)
            w_ns, w_n = ctypes.c_longlong(0), ctypes.c_longlong(0)
            for rep in range(R):
                lib.gsr_debug_host_wait(None, None, 1)
                dts_single.append(timed(step, args.steps, world, device))
                lib.gsr_debug_host_wait(ctypes.byref(w_ns), ctypes.byref(w_n), 0)
                single_wait_ns += w_ns.value; single_waits += w_n.value
                gc.disable()   # (timed() switches the collector back on)
                pipes.run(pipe_step, list(range(2 * V)))   # untimed, like timed()'s own first steps
                lib.gsr_debug_host_wait(None, None, 1)
                pipes.run(pipe_step, list(range(args.steps)), before=lambda: clock.__setitem__("t0", time.perf_counter()),
                          after=lambda: clock.__setitem__("t1", time.perf_counter()))
                lib.gsr_debug_host_wait(ctypes.byref(w_ns), ctypes.byref(w_n), 0)
                wait_ns.value += w_ns.value; waits.value += w_n.value
                dts_pipe.append(clock["t1"] - clock["t0"])
        except Exception as ex:   # a box on which the threaded run fails still reports the one-at-a-time figure, and says so
            print(f"bench.py: {V} pipelines failed ({ex!r}); reporting one view at a time", file=sys.stderr)
            pipe_error = repr(ex)[:200]
        finally:
            gc.enable()
        while len(dts_single) < R:
            dts_single.append(timed(step, args.steps, world, device))
        dt_single = med(dts_single)
        single = {"value": round(args.steps / dt_single, 2), "ms_per_step": round(dt_single / args.steps * 1e3, 4), **spread(dts_single),
                  "what": "the same K steps one view at a time on one stream (how rounds 1 and 2 quoted the metric); median of the "
                          "repeats, interleaved with the pipelined ones"}
        if single_waits:   # (timed()'s untimed first steps are in both numerator and denominator)
            single["host_wait_ms_per_step"] = round(single_wait_ns / 1e6 / single_waits, 4)
        if pipe_error is None and len(dts_pipe) == R:
            dt = med(dts_pipe)
            value_spread = spread(dts_pipe)
        else:
            V, dt, value_spread = 1, dt_single, spread(dts_single)
            wait_ns.value, waits.value = 0, 0
    else:
        dts = []
        w_ns, w_n = ctypes.c_longlong(0), ctypes.c_longlong(0)
        for rep in range(R):
            lib.gsr_debug_host_wait(None, None, 1)
            dts.append(timed(step, args.steps, world, device))
            lib.gsr_debug_host_wait(ctypes.byref(w_ns), ctypes.byref(w_n), 0)
            wait_ns.value += w_ns.value; waits.value += w_n.value
        dt = med(dts)
        value_spread = spread(dts)
    ms_per_step = dt / args.steps * 1e3
    value = args.steps * world / dt
    plan_stats = {k: rz_plan.PLAN_STATS[k] - plan_stats0[k] for k in plan_stats0}
    # the same K steps with planning switched off (every view: scan + scatter + the host round trip between the stages), one
    # view at a time, median of 3 regions -- what the planned figures above are to be compared with
    exact_binning = None
    if world == 1 and not scale_step and rz_plan._PLANNED:
        rz_plan._PLANNED = False
        try:
            dts_x = [timed(step, args.steps, world, device) for _ in range(3)]
            exact_binning = {"ms_per_step": round(med(dts_x) / args.steps * 1e3, 4), "value": round(args.steps / med(dts_x), 2), **spread(dts_x),
                             "what": "single_pipeline's K steps with GSR_PLANNED=0: every view binned the exact way (tile-offset scan, "
                                     "scatter pass, one host round trip between the stages) as in rounds 1-4"}
        finally:
            rz_plan._PLANNED = True

    # the same K steps driven the way the reference's caller drives them: matrices rebuilt and uploaded per call (VERDICT r5
    # task 1).  Its `.cuda()` uploads synchronise the stream, so the host never runs ahead here -- that is the caller's cost,
    # the same with plans (`planned`) and without (`exact`); the pair shows what the plans give THIS caller.
    reference_caller = None
    if world == 1 and not scale_step:
        try:
            prepare_reference_caller(cams)
            bg_t_rc = rasters[0].raster_settings.bg
            rc_step = lambda s_: reference_caller_step(s_, cams, bg_t_rc, params, means2D, dpix)
            # (a camera's plan is keyed by the CONTENTS of its view matrix: these calls find the plans the steps above left)
            for s_ in range(min(args.steps, 8)):
                rc_step(s_)                       # untimed: allocator and module warm-up of this caller's extra kernels
            st0, k0 = dict(rz_plan.PLAN_STATS), dict(rz_plan.CAMERA_KEY_STATS)
            dts_rc = [timed(rc_step, args.steps, 1, device) for _ in range(3)]
            st1, k1 = dict(rz_plan.PLAN_STATS), dict(rz_plan.CAMERA_KEY_STATS)
            rz_plan._PLANNED = False
            try:
                dts_rx = [timed(rc_step, args.steps, 1, device) for _ in range(3)]
            finally:
                rz_plan._PLANNED = True
            # what one content read costs the host (fresh tensor each time, GPU otherwise idle)
            mats = [torch.Tensor(cams[i % len(cams)]._w2v).transpose(0, 1).cuda() for i in range(64)]
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for m_ in mats:
                rz_plan._camera_key(lib, m_, device)
            key_us = (time.perf_counter() - t0) / len(mats) * 1e6
            # what the CALLER's own per-call code costs with the GPU idle: the matrix recipe of sugar_model.py:1149-1187 alone
            # (three .cuda() uploads of pageable memory -- each waits for the stream --, a bmm, the settings object)
            def caller_only(step_):
                cam_ = cams[step_ % len(cams)]
                wv_ = torch.Tensor(cam_._w2v).transpose(0, 1).cuda()
                pj_ = torch.from_numpy(cam_._proj).transpose(0, 1).cuda()
                fp_ = (wv_.unsqueeze(0).bmm(pj_.unsqueeze(0))).squeeze(0)
                cc_ = torch.Tensor(cam_._center).cuda()
                return wv_, fp_, cc_
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for s_ in range(64):
                caller_only(s_)
            torch.cuda.synchronize(device)
            caller_ms = (time.perf_counter() - t0) / 64 * 1e3
            reference_caller = {
                "caller_matrices_ms_per_call": round(caller_ms, 4),
                "value": round(args.steps / med(dts_rc), 2), "ms_per_step": round(med(dts_rc) / args.steps * 1e3, 4), **spread(dts_rc),
                "exact": {"value": round(args.steps / med(dts_rx), 2), "ms_per_step": round(med(dts_rx) / args.steps * 1e3, 4)},
                "plan_stats": {k: st1[k] - st0[k] for k in st0}, "camera_keys": {k: k1[k] - k0[k] for k in k0},
                "camera_key_read_us": round(key_us, 2),
                "what": "single_pipeline's K steps with the settings built per call as gaustar_scene/sugar_model.py:1149-1187 builds "
                        "them (fresh transposed view matrix uploaded with .cuda(), projection, bmm, camera centre; tensors dropped after "
                        "backward); cameras are recognised by gsr_camera_key (contents of the view matrix).  `exact` = the same calls "
                        "with plans off.  The caller's two .cuda() uploads synchronise the stream (PyTorch copies pageable memory "
                        "synchronously), so unlike single_pipeline the host cannot queue a view behind the previous one's backward: "
                        "a step is the view's GPU time PLUS the host time from the uploads to the first launch.  "
                        "caller_matrices_ms_per_call = the caller's matrix recipe alone on an idle GPU (not rasterizer time)"}
        except Exception as ex:
            reference_caller = {"error": repr(ex)[:300]}

    # instrumented pass: per-kernel HIP-event durations over the same K steps (rank 0's launches)
    nst = lib.gsr_num_stages()
    names = [lib.gsr_stage_name(i).decode() for i in range(nst)]
    ms = (ctypes.c_float * nst)()
    cnt = (ctypes.c_int * nst)()
    lib.gsr_profile_enable(1)
    # (under --scale-step / N > 1 the brackets see the same six rasterizer kernels; the untimed first steps' brackets are dropped)
    dt_prof = timed(step, args.steps, world, device, on_timed_start=lambda: lib.gsr_profile_read(ms, cnt, 1))
    _lib.check(lib.gsr_profile_read(ms, cnt, 1), "gsr_profile_read")
    lib.gsr_profile_enable(0)

    # num_rendered of the views this rank rendered (for the algorithmic byte count)
    from gaustar_amd import rasterizer as rz
    R_list = []
    for s in range(min(args.steps, len(rasters))):
        rs = rasters[(s * world + rank) % len(rasters)].raster_settings
        e = torch.Tensor([])
        out = rz.rasterize_gaussians_native(rs.bg, params["means3D"].detach(), params["colors"].detach(),
                                            params["opacities"].detach(), params["scales"].detach(),
                                            params["rotations"].detach(), 1.0, e, rs.viewmatrix, rs.projmatrix,
                                            rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, e, 0, rs.campos,
                                            False, False, use_plan=False)   # (a planned view returns its plan's CAPACITY)
        R_list.append(out[0])
    R_mean = float(np.mean(R_list))

    if rank == 0:
        W, H = cams[0].W, cams[0].H
        total_b, per_kernel_b = algorithmic_bytes(gs.P, R_mean, W, H)
        kern = {}
        for i, nme in enumerate(names):
            if cnt[i]:
                launches_per_step = cnt[i] / args.steps
                mean_ms = ms[i] / cnt[i]
                kern[nme] = {"ms_per_launch": round(mean_ms, 5), "launches_per_step": launches_per_step,
                             "alg_bytes_per_step": int(per_kernel_b.get(nme, 0)),
                             "GBps": round(per_kernel_b.get(nme, 0) / launches_per_step / (mean_ms * 1e-3) / 1e9, 1)}
        dom = max(kern, key=lambda k: kern[k]["ms_per_launch"] * kern[k]["launches_per_step"])
        ach = kern[dom]["GBps"]
        # HBM traffic and instruction counts come from rocprofv3 PMC passes (separate runs, tools/pmc.sh ->
        # profiles/pmc_latest.json), which carries a hash of gaustar_amd/csrc: counters of other kernels than the ones
        # timed here are dropped (null), never reported stale.
        traffic, valu = None, None
        pmc_ok = {}
        pmc_file = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_file):
            try:
                pmc = json.load(open(pmc_file))
                if pmc.get("_csrc_sha256") == csrc_sha256():
                    pmc_ok = pmc
                if pmc.get("_csrc_sha256") == csrc_sha256() and dom in pmc:
                    traffic = pmc[dom].get("hbm_bytes_per_launch")
                    vi = pmc[dom].get("valu_insts")
                    if vi:
                        # secondary roofline: vector-ALU issue.  frac = share of the chip's VALU issue
                        # slots the kernel's own vector instructions occupy while it runs
                        # The model prices every instruction at 4 cycles; and/sub/mov-class instructions issue in
                        # 2.4-2.9 (tools/micro/valu_rates.hip), so a kernel full of them can exceed 1 under it: the
                        # uncapped figure and the 2.4-cycle floor are reported next to the capped one.
                        t_s = kern[dom]["ms_per_launch"] * 1e-3
                        f4 = vi * 4 / N_SIMD / (t_s * CLOCK_HZ)
                        valu = {"insts_per_launch": int(vi), "issue_cycles_per_simd": round(vi * 4 / N_SIMD, 1),
                                "frac": round(min(1.0, f4), 4), "frac_4_cycle_model_uncapped": round(f4, 4),
                                "frac_floor_2p4_cycles_each": round(f4 * 2.4 / 4, 4), "unit": "wave64 VALU instructions",
                                "salu_insts_per_launch": pmc[dom].get("salu_insts"),
                                "what": "share of the SIMDs' cycles the kernel's vector instructions take under a 4-cycle model (what "
                                        "SQ_ACTIVE_INST_VALU charges) and at 2.4 cycles each; round 4 measured a wave64 fma at 2.5 "
                                        "cycles with eight waves resident and 9.4 alone (tools/micro/dep_issue.hip) and -21 % "
                                        "instructions buying -5 % time: the kernel is latency-bound at its residency, read the "
                                        "floor figure, not the 4-cycle one (DESIGN.md 6)"}
            except Exception:
                traffic, valu = None, None
        path_gbs = total_b / (ms_per_step * 1e-3) / 1e9 if world == 1 else total_b * world / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "views/sec fwd+bwd @1920x1080, 500k Gaussians; HBM GB/s vs roofline",
            "value": round(value, 2), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config C: 491520 mesh-bound surface Gaussians (icosphere level 6, 6/face), "
                                   "160-camera rig @1920x1080, colours precomputed (M=0), one view per GPU per step, "
                                   "fwd+bwd" + ((", inside one refinement step per rank on the reference loop's parameters: mesh + SH "
                                                 "producers fwd/bwd, reduce-scatter of the %.0f MB of gradients from autograd hooks, Adam on "
                                                 "this rank's 1/N of the parameters, all-gather of the parameters (dist.ShardedAdam)"
                                                 % (refine[2].payload_bytes() / 1e6)) if world > 1 else ""),
                       "gaussians": gs.P, "width": W, "height": H, "views_per_step": world,
                       "num_rendered_mean": R_mean, "parallelism": f"view-parallel x{world}",
                       "views_in_flight": V,
                       "value_definition": (
                           ("MEDIAN over %d back-to-back timed regions of K = %d steps each, views per second; " % (R, args.steps)) +
                           ("a step = one complete forward + backward of one 1080p view of config C through the public API; "
                            if not (scale_step or world > 1) else
                            "a step = one whole refinement step per rank (mesh + SH producers, rasterizer forward + backward, their "
                            "backward, sharded Adam) on one view per rank; ") +
                           (("%d independent views in flight on one GPU (host thread + HIP stream each) -- a sweep / evaluation / "
                             "V-views-per-optimiser-step schedule; the reference's one-view-then-Adam loop (refine.py:538-548) has no "
                             "independent views: for it read single_pipeline (same K steps, one view at a time) and window" % V)
                            if V > 1 else "one view at a time per GPU") +
                           "; steady state of the rig: every camera has been rendered before and its views are binned by its plan "
                           "(config.binning; exact_binning = the same steps without plans)"),
                       "value_spread": value_spread,
                       "pipelines": (f"{V} independent view pipelines on the GPU (host thread + HIP stream + leaf tensors each, "
                                     "gaustar_amd.pipelines): step s = one complete forward + backward of view s on pipeline s % "
                                     f"{V}; the small kernels of one view run under the blends of another") if V > 1 else "one view at a time"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "valu": valu,
                         "path_alg_bytes_per_view": int(total_b), "path_achieved": round(path_gbs, 1),
                         "path_frac": round(path_gbs / HBM_PEAK_GBS / world, 5),
                         "instrumented_ms_per_step": round(dt_prof / args.steps * 1e3, 4), "kernels": kern,
                         "kernels_measured": "one view at a time (a kernel's duration under another view's kernels is not its own)"},
        }
        out["config"]["host_cpus_pinned"] = len(pinned)
        out["config"]["binning"] = {
            "planned_views": plan_stats["planned"], "exact_views": plan_stats["exact"], "misfits": plan_stats["misfit"],
            "what": "views of the timed regions binned by their camera's plan (no scan, no scatter: include/gsr.h gsr_forward_planned) / "
                    "rendered the exact way (no valid plan: longest list above 2 032 entries) / that outgrew their plan; every camera "
                    "was rendered once, untimed, before the timed regions (its first view leaves the plan)"}
        if exact_binning is not None:
            out["exact_binning"] = exact_binning
        if single is not None:
            out["single_pipeline"] = single
        if reference_caller is not None:
            out["reference_caller"] = reference_caller
        # The conservative figures, inside `config` (the driver's record keeps `config` and `roofline`, not the extra keys):
        # `value` above is the most favourable of the modes measured.
        pick = lambda d, *ks: None if not isinstance(d, dict) else {k: d.get(k) for k in ks if k in d}
        out["config"]["conservative"] = {
            "single_pipeline": pick(single, "value", "ms_per_step"),
            "exact_binning": pick(exact_binning, "value", "ms_per_step"),
            "reference_caller": pick(reference_caller, "value", "ms_per_step", "plan_stats", "camera_key_read_us", "caller_matrices_ms_per_call", "error"),
            "reference_caller_exact": pick((reference_caller or {}).get("exact"), "value", "ms_per_step"),
            "what": "one view at a time with plans / without plans / driven as the reference's sugar_model.py drives the rasterizer "
                    "(matrices rebuilt per call) with and without plans; window and parity figures are added below when those legs run"}
        # sum of the per-kernel means of the instrumented pass (stages that launched nothing carry no bracket)
        out["roofline"]["kernels_sum_ms_per_step"] = round(sum(k["ms_per_launch"] * k["launches_per_step"] for k in kern.values()), 4)
        # the HIP-event brackets themselves add 1 - 4 us per stage (an empty bracket reads ~5 us), so the sum above exceeds the
        # un-instrumented step; rocprofv3's own mean durations of the same kernels (profiles/, same csrc hash) do not
        rp = {k: pmc_ok[k]["rocprof_avg_ns"] * kern[k]["launches_per_step"] for k in kern if isinstance(pmc_ok.get(k), dict) and
              pmc_ok[k].get("rocprof_avg_ns")}
        # (stages that run in a few views only -- the separate sort kernels for lists above 2 048 entries -- carry no entry of
        # their own in the counter file and are left out of this sum: < 1 % of a step)
        major = [k for k in kern if kern[k]["launches_per_step"] >= 0.5]
        out["roofline"]["kernels_sum_rocprof_ms_per_step"] = round(sum(rp.values()) * 1e-6, 4) if rp and all(k in rp for k in major) else None
        # the dominant kernel's fraction spelled out, so that it can be recomputed from the line itself: algorithmic bytes
        # per launch / mean launch duration / peak -- by this run's HIP events (frac) and by rocprofv3's mean duration of the
        # same kernel of the same csrc (frac_rocprof; the event brackets add a few microseconds per launch)
        lps_dom = kern[dom]["launches_per_step"]
        out["roofline"]["alg_bytes_per_launch"] = int(round(per_kernel_b.get(dom, 0) / lps_dom))
        out["roofline"]["ms_per_launch"] = kern[dom]["ms_per_launch"]
        rp_ns = (pmc_ok.get(dom) or {}).get("rocprof_avg_ns") if isinstance(pmc_ok.get(dom), dict) else None
        out["roofline"]["rocprof_avg_ns"] = rp_ns
        out["roofline"]["frac_rocprof"] = (round(per_kernel_b.get(dom, 0) / lps_dom / (rp_ns * 1e-9) / 1e9 / HBM_PEAK_GBS, 5)
                                           if rp_ns else None)
        if world == 1 and not args.no_extras:
            try:
                iss = issue_statistics(lib, rasters[0].raster_settings, params, device)
                for kn in ("blend_fwd_kernel", "blend_bwd_kernel"):
                    vi = (pmc_ok.get(kn) or {}).get("valu_insts")
                    iss[kn + "_valu_insts_per_live_pair"] = round(vi / iss["live_pairs_per_view"], 3) if vi else None
                    iss[kn + "_lane_slots_per_live_pair"] = round(64.0 * vi / iss["live_pairs_per_view"], 1) if vi else None
                iss["what"] = ("view 0 of the rig: live pairs = set candidate bits below each pixel's last contributor; wave64 vector "
                               "instructions of profiles/pmc_latest.json (null when csrc changed since the counters were taken)")
                out["roofline"]["issue"] = iss
            except Exception as ex:   # statistics only
                out["roofline"]["issue"] = {"error": repr(ex)[:200]}
        # how long the host sat in the forward's one synchronisation per step (gsr_debug_host_wait): the slack between the
        # host's own work (Python, autograd, launches) and the GPU's -- near zero means the step is host-bound
        # (accumulated over the R pipelined repeats and over the V pipeline threads: per pipeline, a thread waits this long
        # for every view IT renders -- 1 / V of the steps)
        n_timed = max(int(waits.value), 1)   # forward calls that went through the synchronisation = views rendered
        out["host"] = {"wait_ms_per_view_per_pipeline": round(wait_ns.value / 1e6 / n_timed, 4),
                       "wait_ms_per_step": round(wait_ns.value / 1e6 / n_timed / max(V, 1), 4),
                       "views_counted": int(waits.value), "pipelines": V,
                       "what": "time the host threads sat in the forward's one synchronisation (gsr_debug_host_wait); "
                               "wait_ms_per_step = total / steps / pipelines: the share of a step's wall time a pipeline's host thread "
                               "had nothing to do but wait for its GPU work"}
        if scale_step:
            out["config"]["scale_step"] = True
            out["config"]["workload"] += (", inside one refinement step (mesh + SH producers fwd/bwd, Adam on the reference loop's "
                                          "%.0f MB of parameters): the step `--gpus N` times, here on one GPU" % (refine[2].payload_bytes() / 1e6))
        if world > 1:
            out["config"]["exchange"] = "reduce-scatter + rank-sharded Adam + all-gather"
            out["config"]["exchange_payload_MB"] = round(refine[2].payload_bytes() / 1e6, 1)
            out["config"]["optimizer_state_MB_per_rank"] = round(refine[2].state_bytes_per_rank() / 1e6, 1)
            out["config"]["buckets_issued_during_backward"] = refine[2].issued_early
            # `value` / N x this = speed-up of THIS step over one GPU running it without any exchange (max over ranks, measured
            # in this very job); the N = 1 line's value is the rasterizer alone and is not the denominator of this step
            out["same_step_on_one_gpu"] = {"ms_per_step": round(same_step_1gpu_ms, 4),
                                           "speedup_of_this_step": round(world * same_step_1gpu_ms / ms_per_step, 3)}
        if world == 1 and not args.no_cpu_baseline:
            try:   # the CPU baseline gets every core the process started with
                for tid in os.listdir("/proc/self/task"):
                    os.sched_setaffinity(int(tid), cpus_before)
            except OSError:
                pass
            out["cpu_baseline"] = cpu_baseline(gs, cams, bg)
            if pinned:   # back to the cores the GPU figures above were measured on: the legs below are GPU figures too
                try:
                    for tid in os.listdir("/proc/self/task"):
                        try:
                            os.sched_setaffinity(int(tid), pinned)
                        except OSError:
                            pass
                except OSError:
                    pass
        if world == 1 and not args.no_extras:
            oc = {}
            for cname in ("B", "D", "D_depth", "C_solid"):
                try:
                    oc[cname] = other_config(cname, device, lib)
                except Exception as ex:   # a report leg only
                    oc[cname] = {"error": repr(ex)[:200]}
            out["other_configs"] = oc
        if world == 1 and not args.no_extras and not scale_step:
            try:   # the N > 1 step (refinement step: producers + render + Adam) on ONE GPU: the denominator of its scaling
                rw = build_refinement_workload(device, 0, gs, cams, bg)
                rstep = lambda s_: refinement_step(s_, 0, 1, *rw)
                for s_ in range(5):
                    rstep(s_)
                n_r = min(args.steps, 80)
                dt_r = med([timed(rstep, n_r, 1, device) for _ in range(3)])
                # the same figure under the name the scaling curve needs: value(N) of `--gpus N` divided by N x THIS is the
                # scaling of one and the same step (`--gpus 1 --scale-step` or GSR_BENCH_SCALE_STEP=1 makes it the line's value)
                out["scale_base"] = {"value": round(n_r / dt_r, 2), "unit": "views/s", "ms_per_step": round(dt_r / n_r * 1e3, 4),
                                     "what": "the refinement step `--gpus N > 1` times, on ONE GPU without any exchange (median of 3 "
                                             "regions): value(N) / (N x this) is that step's scaling efficiency"}
                out["refinement_step_single_gpu"] = {"ms_per_step": round(dt_r / n_r * 1e3, 4), "steps": n_r,
                                                     "what": "the step bench.py times at --gpus N > 1 (producers + rasterizer fwd/bwd + "
                                                             "Adam on the reference loop's 77 MB of parameters), here on one GPU "
                                                             "without any exchange: N x (1000 / this) views/s is perfect scaling"}
                rw[2].close(); del rw
            except Exception as ex:
                out["refinement_step_single_gpu"] = {"error": repr(ex)[:200]}
            for key, fn in (("ref_gpu_baseline", lambda: ref_gpu_baseline(gs, cams, bg, device)), ("window", window_benchmark),
                            ("parity", lambda: parity_leg(gs, cams, bg, device))):
                try:
                    out[key] = fn()
                except Exception as ex:
                    out[key] = {"error": repr(ex)[:200]}
            w_, p_ = out.get("window") or {}, out.get("parity") or {}
            out["config"]["conservative"]["window_median_ms_per_iteration"] = w_.get("median_ms_per_iteration")
            out["config"]["conservative"]["window_revisits"] = pick(w_.get("revisits"), "median_ms_per_iteration", "plan_stats")
            out["config"]["conservative"]["parity_flips_per_view_median"] = (p_.get("flips_per_view") or {}).get("median") \
                if isinstance(p_.get("flips_per_view"), dict) else p_.get("flips_per_view_median")
            out["config"]["conservative"]["parity_flip_kinds"] = p_.get("flip_kinds")
        print(json.dumps(out), flush=True)
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()

"""tools/bench_window.py [--frames F] [--iters I] [--level L] -- the shape of BASELINE.json configs[4] (train_seq.py over a
tracking window: per frame, refine the mesh-bound Gaussians against that frame's images, then carry them to the next
frame, train_seq.py:101-244 / gaustar_trainers/refine.py:529-841) assembled from this package only:

  per frame:   ground truth = renders of a deformed + recoloured copy of the surface (synthetic)
  per iteration (refine.py:538-794):  camera = dist.shard_views(...) (one seeded permutation per epoch, as :534),
               harness.SurfaceGaussians -> one 4-channel render (RGB + depth-as-colour) -> l1 + dssim + masked depth L1
               -> backward -> gaustar_amd.optim.Adam.step()
  between frames: the optimised parameters are kept (the tracker's warm start), the optimiser state is rebuilt.

Not included (out of scope): mesh regularisers (pytorch3d), topology update (Open3D),
flow warp.  Prints iterations/s over all frames and the loss at the start / end of every frame."""
import argparse, ctypes, gc, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaustar_amd import GaussianRasterizer, _lib, dist as gdist, harness, losses, optim, producers, scene

MAX_DEPTH = 10.0


def render4(model, ncam, bg4):
    # RGB + depth-as-colour in one 4-channel pass through ONE autograd node (harness.SurfaceGaussians.render_channels)
    return model.render_channels(ncam, bg4, depth_channels=1)[0]


def render4_composed(model, ncam, bg4):
    """The same render composed of autograd nodes (properties -> colours -> sigmoid -> rasterizer), as rounds 1-2 ran it."""
    settings, view, campos = model._settings(ncam, bg4, 0)
    pts = model.points
    colors4 = producers.points_rgb_depth(pts, campos, model.sh_coordinates, model.sh_levels, view, depth_channels=1)
    img, _ = GaussianRasterizer(settings)(means3D=pts, means2D=torch.zeros_like(pts), opacities=model.strengths,
                                          colors_precomp=colors4, scales=model.scaling, rotations=model.quaternions)
    return img


def run(a):
    render = render4_composed if getattr(a, "composed", False) else render4
    # torchrun: one process per GPU, one camera per rank per iteration, parameter gradients averaged over the ranks right
    # before the optimiser step (gaustar_amd.dist; GSR_BENCH_BACKEND=gloo lets ranks share a GPU for tests)
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    rank, world, local = gdist.init_from_env(backend)
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    v, f = scene.icosphere(a.level, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
    verts, faces = torch.from_numpy(v).float().to(dev), torch.from_numpy(f).long().to(dev)
    model = harness.SurfaceGaussians(verts, faces, 6, 3).to(dev)
    N = model.n_points
    with torch.no_grad():
        model._sh_coordinates_dc.copy_(torch.rand(N, 1, 3, device=dev, generator=g) * 2 - 1)
    edge = float((verts[faces[:, 0]] - verts[faces[:, 1]]).norm(dim=-1).mean())
    cams = scene.ring_cameras(5, 32, a.width, a.height, focal_px=1200.0 * a.width / 1920.0)[:a.cameras]
    ncams = [harness.nerf_camera_from_scene(c) for c in cams]
    bg4 = torch.tensor([0.0, 1.0, 0.0, MAX_DEPTH], device=dev)
    target = harness.SurfaceGaussians(verts, faces, 6, 3).to(dev)
    target.load_state_dict(model.state_dict())
    frames, n_it, per_it = [], 0, []
    one = torch.ones((), device=dev)
    fused_step = bool(getattr(a, "fused_step", False))
    host_wait_ns = 0
    pts_start = model.points.detach().clone()
    t_total = 0.0
    for fi in range(a.frames):
        with torch.no_grad():   # the subject moves and changes colour from frame to frame
            target._points.add_(0.25 * edge * torch.randn(verts.shape, device=dev, generator=g))
            target._sh_coordinates_dc.add_(0.2 * torch.randn(N, 1, 3, device=dev, generator=g))
            gts = []
            for nc in ncams:
                img = render(target, nc, bg4)
                d = img[3].clone()
                d[d >= MAX_DEPTH - 1e-3] = 2 * MAX_DEPTH
                gts.append((img[:3].clone(), d))
        if os.environ.get("GSR_WINDOW_PLAN_DEBUG"):
            from gaustar_amd import rasterizer as _rzd
            print(f"[window] frame {fi} after ground truth: {_rzd.PLAN_STATS}", file=sys.stderr)
        groups = [{"params": [model._points], "lr": 2e-4},
                  {"params": [model._sh_coordinates_dc, model._sh_coordinates_rest], "lr": 5e-3},
                  {"params": [model._scales, model._quaternions, model.all_densities], "lr": 5e-3}]
        sharded = world > 1 and getattr(a, "exchange", "sharded") == "sharded"
        if sharded:   # reduce-scatter -> Adam on this rank's 1/N -> all-gather (dist.ShardedAdam), the default for N > 1
            reducer = None
            opt = gdist.ShardedAdam(groups, ready_order=model.grad_ready_order(), eps=1e-15)
            model.grad_sink = None if getattr(a, "composed", False) else opt   # gradients land in the flat buffer directly
        else:         # all-reduce + the same Adam step on every rank
            reducer = gdist.GradAllReducer(model.grad_ready_order()) if world > 1 else None
            opt = optim.Adam(groups, eps=1e-15)
        hist = []
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]   # GPU time of every iteration, no host sync
        # Python's cyclic collector is run HERE, between frames, and held off inside the frame: a generation-2 pass over the
        # few thousand tensor / autograd objects of a frame stops the host for 50 - 80 ms -- one "iteration" of 73 ms in a
        # loop whose iterations take 0.85 (measured: tools/window_diag.py; gone with the collector off)
        gc.collect(); gc.disable()
        _lib.load().gsr_debug_host_wait(None, None, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            for it in range(a.iters):
                marks[it].record()
                ci = gdist.shard_views(len(ncams), fi * a.iters + it, rank, world)
                opt.zero_grad(set_to_none=True)
                if fused_step:    # render + losses + both backward passes without an autograd graph (harness.SurfaceGaussians.rgbd_step)
                    loss = model.rgbd_step(ncams[ci], bg4, gts[ci][0], gts[ci][1], MAX_DEPTH, 0.2, 1.0, 0.5)[0]
                else:
                    img = render(model, ncams[ci], bg4)
                    loss = losses.rgb_depth_loss(img, gts[ci][0], gts[ci][1], MAX_DEPTH, 0.2, 1.0, 0.5)
                    loss.backward(one)                 # (the seed is handed over: a bare backward() fills a ones_like per call, 4.5 us on the stream)
                if reducer is not None:
                    reducer()                      # the hook in front of sugar_optimizer.py:99-101
                opt.step()
                hist.append(loss.detach())
        except BaseException:
            gc.enable()
            raise
        if reducer is not None:
            early, payload = reducer.issued_early, reducer.payload_bytes()
            reducer.close()
        if sharded:
            early, payload = opt.issued_early, opt.payload_bytes()
            opt.close()
            model.grad_sink = None
        marks[a.iters].record()
        torch.cuda.synchronize(); t_total += time.perf_counter() - t0
        if os.environ.get("GSR_WINDOW_PLAN_DEBUG"):
            from gaustar_amd import rasterizer as _rzd
            print(f"[window] frame {fi} after iterations: {_rzd.PLAN_STATS}", file=sys.stderr)
        gc.enable()
        w_ns, w_n = ctypes.c_longlong(0), ctypes.c_longlong(0)
        _lib.load().gsr_debug_host_wait(ctypes.byref(w_ns), ctypes.byref(w_n), 1)
        host_wait_ns += w_ns.value
        frame_s = time.perf_counter() - t0
        per_it.extend(marks[i].elapsed_time(marks[i + 1]) for i in range(a.iters))
        n_it += a.iters
        k = max(1, min(5, a.iters // 4))
        frames.append({"loss_first": round(float(torch.stack(hist[:k]).mean()), 5), "loss_last": round(float(torch.stack(hist[-k:]).mean()), 5),
                       "ms_per_iteration": round(frame_s / a.iters * 1e3, 3)})
    moved = float((model.points.detach() - pts_start).abs().max())
    if world > 1:
        torch.distributed.barrier()
    extra = {} if world == 1 else {"world": world, "exchange": getattr(a, "exchange", "sharded"), "allreduce_payload_MB": round(payload / 1e6, 1), "buckets_issued_during_backward": early,
                                  "views_per_iteration": world}
    from gaustar_amd import rasterizer as _rz
    extra["plan_stats"] = dict(_rz.PLAN_STATS)
    if fused_step:
        extra["fused_step"] = True
    return {**extra, "gaussians": N, "image": [a.width, a.height], "cameras": len(ncams), "frames": frames, "iterations": n_it,
            "iterations_per_s": round(n_it / t_total, 1), "ms_per_iteration": round(t_total / n_it * 1e3, 3),
            # the steady state: median over all iterations of the time between their start marks on the stream (the wall
            # figure above also carries every frame's one-off costs: optimiser state allocation, first-use allocations)
            "median_ms_per_iteration": round(float(np.median(per_it)), 3), "p90_ms_per_iteration": round(float(np.percentile(per_it, 90)), 3),
            "slowest_iterations": [[int(i), round(float(per_it[i]), 2)] for i in np.argsort(per_it)[::-1][:4]],
            # time the host spent in the forward's one wait per iteration (for the instance count): what is left of an
            # iteration after it is the host's own work -- close to preprocess + scan (~40 us) means the host is the bound
            "host_wait_ms_per_iteration": round(host_wait_ns / 1e6 / max(n_it, 1), 4),
            "geometry_moved": moved}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3); ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--level", type=int, default=6); ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080); ap.add_argument("--cameras", type=int, default=160)
    ap.add_argument("--exchange", choices=["sharded", "allreduce"], default="sharded")
    ap.add_argument("--composed", action="store_true", help="render through the composition of autograd nodes instead of the one-node render")
    ap.add_argument("--fused-step", dest="fused_step", action="store_true",
                    help="render + losses + backward through SurfaceGaussians.rgbd_step (no autograd graph) instead of loss.backward()")
    r = run(ap.parse_args())
    if gdist.rank() == 0:
        print(json.dumps(r))
    if gdist.world_size() > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

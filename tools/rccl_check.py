"""tools/rccl_check.py -- run under torchrun on N >= 2 GPUs of one node (tools/rccl_smoke.sh does): dist.ShardedAdam over
RCCL -- hook-driven reduce-scatter during backward, rank-sharded HIP Adam, all-gather (lazy for the buckets the next forward
does not need first), the gradient sink of harness.SurfaceGaussians -- against ONE torch.optim.Adam that every rank runs
on the rank-averaged gradients it computes itself.  RCCL has only ever run at world size 1 on the builder's boxes
(tests/test_gpu_dist.py); the same logic is tested at world size 2 over gloo (tests/test_dist.py).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/rccl_check.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaustar_amd import dist as gd, harness, scene  # noqa: E402


def main():
    # GSR_BENCH_BACKEND=gloo: the same script with the ranks sharing GPUs (a one-GPU box; RCCL refuses two ranks on one device)
    backend = os.environ.get("GSR_BENCH_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    rank, world, local = gd.init_from_env(backend)
    dev = torch.device("cuda", local % ndev)
    torch.cuda.set_device(dev)
    assert world >= 2, "run under torchrun with at least two ranks"
    # ---- (1) plain tensors: every rank can form every rank's gradient, so the reference needs no communication
    torch.manual_seed(0)
    shapes = [(300_000, 3), (9, 4), (100_001, 3), (77,), (250_000, 2)]
    ps = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    groups = lambda xs: [{"params": xs[:2], "lr": 1e-2}, {"params": xs[2:], "lr": 3e-3}]
    opt = gd.ShardedAdam(groups(ps), ready_order=ps, eps=1e-15, bucket_bytes=2 << 20, gather_first=[ps[3], ps[4]])
    ref = torch.optim.Adam(groups(qs), eps=1e-15)
    coeff = lambda r: [(i + 1.0) * (r + 1.0) for i in range(len(shapes))]
    for it in range(4):
        opt.zero_grad(); ref.zero_grad()
        opt.wait_params()
        sum(c * (p ** 2).sum() for c, p in zip(coeff(rank), ps)).backward()
        if it == 2:   # a second backward before the step (dirty buckets are reduced again)
            sum(0.5 * c * (p ** 2).sum() for c, p in zip(coeff(rank), ps)).backward()
        (sum(sum(c * (q ** 2).sum() for c, q in zip(coeff(r), qs)) for r in range(world)) / world).backward()
        if it == 2:
            (sum(sum(0.5 * c * (q ** 2).sum() for c, q in zip(coeff(r), qs)) for r in range(world)) / world).backward()
        opt.step(); ref.step()
        assert opt.pending_gathers() >= 1, "lazy gather did not leave a bucket in flight"
    opt.wait_params()
    torch.cuda.synchronize()
    err = max(float((p - q).abs().max()) for p, q in zip(ps, qs))
    flat = torch.cat([p.detach().reshape(-1) for p in ps])
    alls = [torch.zeros_like(flat) for _ in range(world)]
    torch.distributed.all_gather(alls, flat)
    same = all(torch.equal(alls[0], a) for a in alls)
    assert err < 5e-6 and same, (rank, err, same)
    sd = opt.state_dict()
    assert sorted(sd["state"]) == list(range(len(shapes)))
    opt.close()
    # ---- (2) the render's gradient sink: SurfaceGaussians + ShardedAdam, one view per rank, a few steps; ranks end identical
    from gaustar_amd import GaussianRasterizationSettings  # noqa: F401
    v, f = scene.icosphere(3, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
    model = harness.SurfaceGaussians(torch.from_numpy(v).float().to(dev), torch.from_numpy(f).long().to(dev), 6, sh_levels=3,
                                     surface_mesh_thickness=3.5e-6, loose_bind=True).to(dev)
    g3 = [{"params": [model._points], "lr": 1e-4},
          {"params": [model._sh_coordinates_dc, model._sh_coordinates_rest], "lr": 2e-3},
          {"params": [model._scales, model._quaternions, model.all_densities, model._delta_t, model._delta_r], "lr": 1e-3}]
    o2 = gd.ShardedAdam(g3, ready_order=model.grad_ready_order(), eps=1e-15, bucket_bytes=200_000, gather_first=model.mesh_parameters())
    model.grad_sink = o2
    cams = [harness.nerf_camera_from_scene(c) for c in scene.ring_cameras(2, 8, 320, 240, focal_px=260.0)]
    bg = torch.tensor([0.0, 1.0, 0.0], device=dev)
    dpix = torch.randn(3, 240, 320, device=dev, generator=torch.Generator(device=dev).manual_seed(7))   # same on every rank
    for it in range(5):
        o2.zero_grad()
        img, _ = model.render_channels(cams[(it * world + rank) % len(cams)], bg, depth_channels=0)
        (img * dpix).sum().backward()
        o2.step()
    o2.wait_params()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert bool(torch.isfinite(flat).all())
    alls = [torch.zeros_like(flat) for _ in range(world)]
    torch.distributed.all_gather(alls, flat)
    assert all(torch.equal(alls[0], a) for a in alls), "ranks diverged after sharded steps through the gradient sink"
    o2.close()
    torch.distributed.barrier()
    if rank == 0:
        print(f"RCCL_CHECK_OK world={world} max|p - adam|={err:.2e} early_buckets={opt.issued_early} reissued={opt.reissued}", flush=True)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

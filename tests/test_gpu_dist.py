"""Multi-process paths on a GPU box (-m gpu): RCCL itself (backend "nccl" at world size 1: a one-GPU box cannot host two
RCCL ranks), and the two-rank view-parallel paths of bench.py / tools/bench_window.py over gloo with both ranks on the
one GPU.  (World size 2 on CPU: tests/test_dist.py.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_allreduce_on_device_world_size_1():
    """init_process_group("nccl") = RCCL on ROCm; GradAllReducer's hook-driven buckets really go through
    ncclAllReduce on device tensors (run_at_world_size_1), and come back unchanged (mean over one rank)."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import torch
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
from gaustar_amd import dist as gd
dev = torch.device("cuda:0")
params = [torch.nn.Parameter(torch.randn(200_000, 3, device=dev)), torch.nn.Parameter(torch.randn(50_000, device=dev)),
          torch.nn.Parameter(torch.randn(7, 4, device=dev))]
red = gd.GradAllReducer(params, bucket_bytes=1 << 20, run_at_world_size_1=True)
for step in range(3):
    for p in params:
        p.grad = None
    loss = sum((i + 1.0) * (p * p).sum() for i, p in enumerate(params))
    loss.backward()
    want = [2.0 * (i + 1.0) * p.detach() for i, p in enumerate(params)]
    red()
    torch.cuda.synchronize()
    assert all(torch.allclose(p.grad, w) for p, w in zip(params, want)), step
    assert red.issued_early >= 1, "no bucket left during backward"
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
''' % (ROOT, str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_sharded_adam_over_rccl_equals_the_fused_adam_world_size_1():
    """dist.ShardedAdam with its collectives really issued (reduce_scatter_tensor / all_gather_into_tensor over RCCL at world
    size 1) on the harness's parameters: after five refinement-shaped steps with the trainer's learning rates the parameters
    are bit-identical to gaustar_amd.optim.Adam's (same kernel on the same numbers), buckets left during backward, and
    gather_state() matches the other optimiser's state.  Steps alternate between the one-node render with the optimiser as
    its gradient SINK (gradients written straight into the flat buffer, p.grad aliasing it) and the hook + pack path."""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import torch
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
from gaustar_amd import dist as gd, harness, optim, scene
dev = torch.device("cuda:0")
v, f = scene.icosphere(3, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
verts, faces = torch.from_numpy(v).float().to(dev), torch.from_numpy(f).long().to(dev)
def make():
    torch.manual_seed(1)
    m = harness.SurfaceGaussians(verts, faces, 6, 3, loose_bind=True).to(dev)
    with torch.no_grad():
        m._sh_coordinates_dc.copy_(torch.rand_like(m._sh_coordinates_dc) * 2 - 1)
    return m
groups = lambda m: [{"params": [m._points], "lr": 2e-4}, {"params": [m._sh_coordinates_dc, m._sh_coordinates_rest], "lr": 5e-3},
                    {"params": [m._scales, m._quaternions, m.all_densities, m._delta_t, m._delta_r], "lr": 5e-3}]
a, b = make(), make()
oa = gd.ShardedAdam(groups(a), ready_order=a.grad_ready_order(), eps=1e-15, bucket_bytes=256 << 10, run_at_world_size_1=True)
ob = optim.Adam(groups(b), eps=1e-15)
views = oa.grad_views()
aliased = []
cam = harness.nerf_camera_from_scene(scene.look_at_camera((0.5, 1.6, 3.0), scene.SUBJECT_CENTER, 320, 240, focal_px=260.0))
tgt = torch.rand(240, 320, 3, device=dev)
early = []
bg = torch.tensor([0.0, 1.0, 0.0], device=dev)
for it in range(5):
    oa.zero_grad(set_to_none=True)
    if it %% 2 == 0:     # the one-node render with the optimiser as gradient sink: gradients written into the flat buffer
        a.grad_sink = oa
        img = a.render_channels(cam, bg, depth_channels=0)[0].permute(1, 2, 0)
    else:               # the composition of autograd nodes: gradients arrive through the hooks and are packed
        a.grad_sink = None
        img = a.render_image_gaussian_rasterizer(cam, bg_color=[0.0, 1.0, 0.0])
    ((img - tgt) ** 2).mean().backward()
    aliased.append(sum(1 for p in a.parameters() if p.grad is not None and p.grad.data_ptr() == views[id(p)].data_ptr()))
    # the other optimiser steps on the SAME gradients (two renders would differ in the last bits: the backward blend sums
    # with float atomics, and Adam with eps = 1e-15 turns a sign flip of a near-zero gradient into a full-size step)
    for p, q in zip(a.parameters(), b.parameters()):
        q.grad = None if p.grad is None else p.grad.clone()
    oa.step(); ob.step()
    early.append(oa.issued_early)
torch.cuda.synchronize()
for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
    assert torch.equal(p, q), n
st = oa.gather_state()
for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
    assert torch.equal(st[p]["exp_avg"], ob.state[q]["exp_avg"]) and torch.equal(st[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"]), n
assert max(early) >= 1, early
assert len(oa.buckets) >= 3
# sink iterations: autograd adopted the views as p.grad for all eight parameters (no packing copy); hook iterations: none
assert aliased == [8, 0, 8, 0, 8], aliased
dist.barrier(); dist.destroy_process_group()
print("SHARDED_OK")
''' % (ROOT, str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARDED_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def _torchrun(script_args, env_extra, timeout=900):
    env = dict(os.environ, GSR_BENCH_BACKEND="gloo", GSR_BENCH_NO_PIN="1", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[-1])


def test_bench_two_ranks_over_gloo():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per process), the two ranks sharing
    this box's GPU over gloo: one JSON line from rank 0, aggregate views/s over both ranks, weak scaling."""
    d = _torchrun(["bench.py", "--gpus", "2", "--steps", "8", "--warmup", "2", "--no-cpu-baseline"], {})
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["views_per_step"] == 2 and "roofline" in d
    # the exchange is the reference loop's REAL optimiser payload, produced by the harness's producers' backward (SURVEY.md
    # 8e: 3 floats per mesh vertex + 39 per Gaussian), reduce-scattered, stepped on each rank's half, all-gathered
    assert abs(d["config"]["exchange_payload_MB"] - (3 * 40962 + 39 * 491520) * 4 / 1e6) < 0.1
    assert d["config"]["exchange"].startswith("reduce-scatter")
    assert d["config"]["optimizer_state_MB_per_rank"] < 0.55 * 2 * d["config"]["exchange_payload_MB"]
    assert d["config"]["buckets_issued_during_backward"] >= 1


@pytest.mark.parametrize("extra", [[], ["--fused-step"]], ids=["autograd", "graph-free step"])
def test_refinement_window_two_ranks_over_gloo(extra):
    """tools/bench_window.py under torchrun: harness parameters (the optimiser's real payload), gradients averaged by
    dist.GradAllReducer with buckets leaving during backward, every frame's loss falling on an effective batch of two views.
    --fused-step: the same with render + losses + backward as SurfaceGaussians.rgbd_step (gradients into the sink without a graph)."""
    d = _torchrun([os.path.join("tools", "bench_window.py"), "--frames", "2", "--iters", "40", "--level", "3", "--width", "320",
                   "--height", "240", "--cameras", "16"] + extra, {})
    assert d["world"] == 2 and d["views_per_iteration"] == 2 and d["buckets_issued_during_backward"] >= 1
    assert d["exchange"] == "sharded"      # reduce-scatter (gloo: all-reduce + shard) -> Adam on each rank's half -> all-gather
    for fr in d["frames"]:
        assert fr["loss_last"] < 0.95 * fr["loss_first"], d


_TWO_RANK_SINK = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from gaustar_amd import dist as gd, harness, optim, scene
rank, world, local = gd.init_from_env("gloo")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
v, f = scene.icosphere(3, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
verts, faces = torch.from_numpy(v).float().to(dev), torch.from_numpy(f).long().to(dev)
def make():
    torch.manual_seed(1)
    m = harness.SurfaceGaussians(verts, faces, 6, 3, loose_bind=True).to(dev)
    with torch.no_grad():
        m._sh_coordinates_dc.copy_(torch.rand_like(m._sh_coordinates_dc) * 2 - 1)
    return m
groups = lambda m: [{"params": [m._points], "lr": 2e-4}, {"params": [m._sh_coordinates_dc, m._sh_coordinates_rest], "lr": 5e-3},
                    {"params": [m._scales, m._quaternions, m.all_densities, m._delta_t, m._delta_r], "lr": 5e-3}]
cams = [harness.nerf_camera_from_scene(scene.look_at_camera(e, scene.SUBJECT_CENTER, 320, 240, focal_px=260.0))
        for e in ((0.5, 1.6, 3.0), (-2.0, 1.2, 2.2))]
bg = torch.tensor([0.0, 1.0, 0.0], device=dev)
tgt = torch.rand(3, 240, 320, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
a = make()
oa = gd.ShardedAdam(groups(a), ready_order=a.grad_ready_order(), eps=1e-15, bucket_bytes=256 << 10)
a.grad_sink = oa
# the reference: ONE process renders both views, averages, steps the replicated Adam
b = make()
ob = optim.Adam(groups(b), eps=1e-15)
worst = 0.0
for it in range(3):
    oa.zero_grad(set_to_none=True)
    img = a.render_channels(cams[rank], bg, depth_channels=0)[0]
    ((img - tgt) ** 2).mean().backward()
    views = oa.grad_views()
    assert all(p.grad.data_ptr() == views[id(p)].data_ptr() for p in a.parameters()), "a gradient went through a copy"
    oa.step()
    ob.zero_grad(set_to_none=True)
    for c in cams:
        img = b.render_channels(c, bg, depth_channels=0)[0]
        (0.5 * ((img - tgt) ** 2).mean()).backward()
    if it == 0:
        # identical parameters on both sides: this rank's shard of the flat gradient buffer now holds the AVERAGE of the two
        # ranks' gradients -- compare it with the one-process average, segment by segment
        ref = {id(p): q.grad.reshape(-1) for p, q in zip(a.parameters(), b.parameters())}
        off = {id(p): o for bk in oa.buckets for p, o in bk["entries"]}
        n_seg = 0
        top = max(float(r.abs().max()) for r in ref.values())
        for bk in oa.buckets:
            for p, _g, s_off, f_off, n in bk["segments"]:
                want = ref[id(p)][f_off - off[id(p)]: f_off - off[id(p)] + n]
                got = bk["shard_g"][s_off:s_off + n]
                # (a parameter whose whole gradient is cancellation noise -- the component of a rotation gradient along the
                # normalised quaternion, 1e-12 against 1e-6 elsewhere -- is held to the noise of the large ones)
                scale = max(float(ref[id(p)].abs().max()), 1e-4 * top)
                err = float((got - want).abs().max())
                assert err <= 3e-5 * scale, (err, scale)
                worst = max(worst, err / scale); n_seg += 1
        assert n_seg >= 4
    ob.step()
torch.cuda.synchronize()
for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
    assert torch.isfinite(p).all() and float((p - q).abs().max()) <= 3 * 3 * 5e-3 + 1e-6, n    # at most lr per step apart
assert oa.issued_early >= 1
dist.barrier()
if rank == 0:
    print("SINK2_OK", worst)
dist.destroy_process_group()
'''


def test_gradient_sink_two_ranks_over_gloo_equals_one_process_averaging(tmp_path):
    """Two view-parallel ranks (gloo, sharing this box's GPU), each rendering its own camera through the one-node render with
    dist.ShardedAdam as the gradient sink (gradients written straight into the flat buffer: every p.grad aliases it), three
    steps -- against one process that renders both views, averages the two gradients and steps optim.Adam."""
    script = tmp_path / "sink2.py"
    script.write_text(_TWO_RANK_SINK % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SINK2_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_rccl_check_script_two_ranks_over_gloo():
    """tools/rccl_check.py (what tools/rccl_smoke.sh runs over RCCL on a multi-GPU node), here with two ranks sharing this
    box's GPU over gloo: ShardedAdam with the HIP Adam kernel, lazy all-gather and the render's gradient sink against
    torch.optim.Adam on the rank-averaged gradients; ranks end bit-identical."""
    env = dict(os.environ, GSR_BENCH_BACKEND="gloo", GSR_BENCH_NO_PIN="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join("tools", "rccl_check.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_CHECK_OK world=2" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_scale_step_at_one_gpu():
    """`bench.py --gpus 1 --scale-step`: the N = 1 point of the step `--gpus N` times (refinement step, no exchange), so that
    the driver's value(N) / (N value(1)) can be the scaling of ONE step."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2", "--repeats", "2", "--scale-step",
                        "--no-cpu-baseline", "--no-extras"], cwd=ROOT, env=dict(os.environ, GSR_BENCH_NO_PIN="1"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["scale_step"] is True and d["config"]["views_in_flight"] == 1
    assert "refinement step" in d["config"]["workload"] and d["value"] > 0 and d["config"]["value_spread"]["repeats"] == 2

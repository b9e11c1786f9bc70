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
    # the exchange is the reference loop's optimiser payload (3 floats per mesh vertex + 39 per Gaussian)
    assert abs(d["config"]["allreduce_payload_MB"] - (3 * 40962 + 39 * 491520) * 4 / 1e6) < 0.1


def test_refinement_window_two_ranks_over_gloo():
    """tools/bench_window.py under torchrun: harness parameters (the optimiser's real payload), gradients averaged by
    dist.GradAllReducer with buckets leaving during backward, every frame's loss falling on an effective batch of two views."""
    d = _torchrun([os.path.join("tools", "bench_window.py"), "--frames", "2", "--iters", "40", "--level", "3", "--width", "320",
                   "--height", "240", "--cameras", "16"], {})
    assert d["world"] == 2 and d["views_per_iteration"] == 2 and d["buckets_issued_during_backward"] >= 1
    for fr in d["frames"]:
        assert fr["loss_last"] < 0.95 * fr["loss_first"], d

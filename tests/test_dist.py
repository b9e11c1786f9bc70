"""View-parallel gradient all-reduce (gaustar_amd/dist.py) on CPU: world_size 2, gloo backend."""
import os
import socket

import pytest
import torch

from conftest import ROOT
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from gaustar_amd import dist as gd
    r, w, _ = gd.init_from_env("gloo")
    assert (r, w) == (rank, world) and gd.world_size() == world
    torch.manual_seed(0)
    # three "parameter groups" incl. one bigger than a bucket and one with no grad on rank 1
    params = [torch.zeros(1000, 3, requires_grad=True), torch.zeros(70_000, requires_grad=True),
              torch.zeros(5, 4, requires_grad=True)]
    for i, p in enumerate(params):
        if not (rank == 1 and i == 2):
            p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    red = gd.GradAllReducer(params, bucket_bytes=64 << 10, average=True)
    assert len(red.buckets) >= 2 and red.payload_bytes() == (3000 + 70_000 + 20) * 4
    red()
    want = [1.5 * 1, 1.5 * 2, (1 * 3 + 0) / 2]
    ok = all(torch.allclose(p.grad, torch.full_like(p, w_)) for p, w_ in zip(params, want))
    # second call re-uses the flat buffers and keeps averaging correctly
    for p in params:
        p.grad = torch.ones_like(p) * (rank + 1)
    red()
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in params)
    # both ranks walk the same permutation and never collide within a step
    views = [gd.shard_views(160, s) for s in range(80)]
    q.put((rank, ok, views))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_grad_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    v0, v1 = res[0][2], res[1][2]
    assert all(a != b for a, b in zip(v0, v1))
    assert len(set(v0) | set(v1)) == 160        # one epoch of 80 steps x 2 ranks covers all 160 cameras


def _overlap_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from gaustar_amd import dist as gd
    gd.init_from_env("gloo")
    # four parameters in the order their gradients become final; buckets of <= 2 parameters
    params = [torch.nn.Parameter(torch.zeros(3000)) for _ in range(4)]
    red = gd.GradAllReducer(params, bucket_bytes=24_000, average=True)
    assert len(red.buckets) == 2
    seen = []
    ok = True
    for step in range(3):
        for p in params:
            p.grad = None
        # autograd accumulates in the reverse order of use: params[0]'s gradient lands first
        loss = sum((i + 1) * (rank + 1) * p.sum() for i, p in reversed(list(enumerate(params))))
        h = params[3].register_hook(lambda g: seen.append(sum(1 for w in red._works if w is not None)))
        loss.backward()
        h.remove()
        red()
        ok = ok and red.issued_early >= 1            # the first bucket left while backward was still running
        ok = ok and all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(params))
    q.put((rank, ok, seen))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_buckets_leave_during_backward_gloo_world2():
    """Hook-driven overlap: the bucket of the gradients that become final first is all-reduced while autograd still
    computes the rest; results equal the plain average on both ranks, step after step (flat buffers are re-used)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    # when the LAST parameter's gradient arrived, the first bucket's collective was already in flight
    assert all(s and s[-1] >= 1 for _, _, s in res), res


def test_single_process_is_a_noop():
    from gaustar_amd import dist as gd
    p = torch.zeros(4, requires_grad=True)
    p.grad = torch.ones(4)
    gd.GradAllReducer([p])()
    assert torch.equal(p.grad, torch.ones(4)) and gd.world_size() == 1 and gd.rank() == 0


def _sweep_worker(rank, world, port, n, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaustar_amd import sweep
    mine = sweep.camera_shard(n)
    local = torch.tensor([[float(i), float(i * i)] for i in mine]).reshape(len(mine), 2)
    full = sweep.gather_rows(local, n)
    q.put((rank, mine, full.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_sweep_sharding_and_gather_world_size_2():
    """Camera sweeps (gaustar_amd/sweep.py): strided shards cover every camera exactly once and the gathered table
    has row i = camera i on every rank (gloo, world size 2, odd camera count)."""
    import torch.multiprocessing as mp
    from gaustar_amd import sweep
    n = 7
    assert sorted(sweep.camera_shard(n, 0, 2) + sweep.camera_shard(n, 1, 2)) == list(range(n))
    assert sweep.camera_shard(5, 0, 1) == [0, 1, 2, 3, 4] and sweep.camera_shard(2, 1, 4) == [1] and sweep.camera_shard(2, 3, 4) == []
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650
    ps = [ctx.Process(target=_sweep_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    expect = [[float(i), float(i * i)] for i in range(n)]
    for rank, mine, full in res:
        assert mine == list(range(rank, n, 2)) and full == expect


def test_bind_to_local_cpus_in_a_subprocess():
    """Pins a child process (never the test runner) and checks the mask it ends up with: a compact subset of what it was
    allowed before, disjoint between two slots."""
    import subprocess, sys, json
    code = r'''
import json, os, sys
sys.path.insert(0, %r)
from gaustar_amd import dist
assert dist._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
before = sorted(os.sched_getaffinity(0))
a = dist.bind_to_local_cpus(0, slot=0, slots=2, cores=4)
now = sorted(os.sched_getaffinity(0))
for t in os.listdir("/proc/self/task"):
    os.sched_setaffinity(int(t), before)
b = dist.bind_to_local_cpus(0, slot=1, slots=2, cores=4)
print(json.dumps(dict(before=before, a=a, now=now, b=b)))
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    if len(d["before"]) < 8:
        pytest.skip("fewer than 8 CPUs allowed here")
    assert d["a"] and d["now"] == sorted(d["a"]) and set(d["a"]) <= set(d["before"])
    assert d["b"] and not (set(d["a"]) & set(d["b"]))

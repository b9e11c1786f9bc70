"""View-parallel multi-GPU support: one process per GPU, one camera per rank per step, and a
bucketed SUM all-reduce (then / world) of the parameter gradients right before the optimiser step.

The reference is single-GPU throughout: it draws one random camera per
iteration (gaustar_trainers/refine.py:534-548) and steps Adam in
gaustar_scene/sugar_optimizer.py:99-101.  The hook below goes immediately before that
`optimizer.step()`.  Backend "nccl" is RCCL on ROCm (xGMI between the 8 MI355X of a node); "gloo"
runs the same code on CPU for tests.

Payload sizing: ~77 MB of fp32 gradients for 491 520 Gaussians.  xGMI is
point-to-point (7 links x ~153 GB/s per GPU), so a ring is bound by ONE link; buckets are therefore
large (default 32 MB) to let RCCL's tree/direct algorithms engage every link and to amortise launch
latency, and they are issued asynchronously so the reduction of early buckets overlaps flattening
the later ones.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple:
    """Initialise torch.distributed from torchrun's RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* variables.
    Returns (rank, world_size, local_rank).  A single process (no env) is world_size 1, no init."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _gpu_local_cpus(device_index: int) -> List[int]:
    """CPUs of the NUMA node the GPU hangs off (sysfs local_cpulist of its PCI function); [] if unknown."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            return _parse_cpulist(f.read())
    except Exception:
        return []


def _physical_cores(cpus: Sequence[int]) -> List[int]:
    """One logical CPU per physical core (the lowest-numbered hardware thread), in ascending order."""
    out, seen = [], set()
    for c in sorted(cpus):
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                first = min(_parse_cpulist(f.read()))
        except Exception:
            first = c
        if first not in seen:
            seen.add(first)
            out.append(c)
    return out


def bind_to_local_cpus(device_index: int = 0, slot: int = 0, slots: int = 1, cores: int = 8) -> List[int]:
    """Pin this process (every thread it has, and those it creates later) to `cores` physical cores of the NUMA node
    its GPU is attached to.  One step of the rasterizer is a ping-pong between the Python thread, PyTorch's autograd
    thread and the HIP runtime's threads, with the GPU waiting on the host once per view (the num_rendered
    read-back): left to roam over 2 x 64 cores the scheduler now and then parks those threads sockets apart, and a
    view costs 0.365 ms instead of 0.330 ms (MI355X, 2-socket EPYC 9575F; any compact set of >= 4 cores avoids it).
    `slot`/`slots`: this process's position among the processes sharing the node (e.g. local rank among the 4 GPUs of
    a socket), so that ranks take disjoint cores.  Returns the CPUs bound to ([] = nothing changed: no sysfs entry, or
    the current affinity mask leaves no room).  The reference trainer is single-process and leaves placement to the OS."""
    try:
        allowed = set(os.sched_getaffinity(0))
    except AttributeError:
        return []
    local = [c for c in _physical_cores(_gpu_local_cpus(device_index)) if c in allowed]
    if not local:
        local = _physical_cores(sorted(allowed))
    per = max(1, min(cores, len(local) // max(slots, 1)))
    mine = local[(slot % max(slots, 1)) * per:(slot % max(slots, 1)) * per + per]
    if len(mine) < min(4, len(local)):
        return []
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), mine)
            except OSError:
                pass
        os.sched_setaffinity(0, mine)
    except OSError:
        return []
    return mine


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_views(num_views: int, step: int, rank_: int | None = None, world: int | None = None,
                seed: int = 0) -> int:
    """Camera index for (step, rank): every rank draws the SAME permutation (same seed, as
    refine.py:534 does with torch.randperm) and takes entry world*k + rank, wrapping per epoch."""
    rank_ = rank() if rank_ is None else rank_
    world = world_size() if world is None else world
    per_epoch = max(1, num_views // world)
    epoch, k = divmod(step, per_epoch)
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(num_views, generator=g)
    return int(perm[(world * k + rank_) % num_views])


class GradAllReducer:
    """Bucketed SUM all-reduce (then / world) of parameter gradients, OVERLAPPED with the backward pass.

    `params` are the tensors whose `.grad` the optimiser will consume (the param groups of SuGaROptimizer,
    sugar_optimizer.py:67-87), listed in the order their gradients become final during backward (for
    harness.SurfaceGaussians: `grad_ready_order()`).  Consecutive parameters share a flat bucket of at most
    `bucket_bytes`.  With `overlap`, every parameter carries a post-accumulate-grad hook: the moment the last
    gradient of a bucket has been accumulated, the bucket is flattened and its all-reduce is issued asynchronously
    (RCCL runs it on its own stream) while autograd goes on with the rest of the backward -- the SH coefficients'
    53 MB travel while the mesh producers' backward and the remaining kernels still run.  Calling the reducer (right
    before `optimizer.step()`, the place of sugar_optimizer.py:99-101) issues whatever has not been issued, waits,
    divides by the world size and re-points every `p.grad` at its slice of the flat buffer (no copy back).

    Parameters whose grad is None on this rank (a Gaussian set no pixel of this view touched) contribute zeros, so
    every rank issues identical collectives in identical order.  The views stay valid until the next backward;
    callers that keep gradients across steps must clone them.

    More than one backward per call (gradient accumulation, a separate regulariser backward) is supported: a bucket that
    was flattened and issued during the first backward and whose gradients were accumulated into again afterwards is
    detected at call time (every gradient's identity and version counter are recorded when its bucket is issued), waited
    for, flattened again and reduced again -- the early collective was wasted, nothing is lost.  A step that is
    abandoned after its backward (NaN loss, an evaluation backward) must call `reset()` before the next backward, or its
    in-flight buckets would be taken for the next step's; the reducer cannot tell the two apart."""

    def __init__(self, params: Iterable[torch.Tensor], bucket_bytes: int = 32 << 20, average: bool = True,
                 overlap: bool = True, run_at_world_size_1: bool = False):
        self.params: List[torch.Tensor] = [p for p in params]
        self.average = average
        self.solo = bool(run_at_world_size_1)   # tests: issue the collectives even when there is nobody to talk to
        self.buckets: List[List[torch.Tensor]] = []
        cur, cur_bytes = [], 0
        for p in self.params:
            nb = p.numel() * 4
            if cur and cur_bytes + nb > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self._flat: List[torch.Tensor] = []
        self._bucket_of = {id(p): bi for bi, bucket in enumerate(self.buckets) for p in bucket}
        self._ready = [0] * len(self.buckets)          # gradients accumulated since the last call, per bucket
        self._works: List = [None] * len(self.buckets)  # in-flight collectives
        self._fast = [False] * len(self.buckets)
        self._stamp: List = [None] * len(self.buckets)  # per issued bucket: (id, version) of every gradient at issue time
        self.reissued = 0                              # buckets reduced twice because their gradients changed after issue
        self._hooks = []
        self.issued_early = 0                          # buckets whose all-reduce started during backward (last step)
        if overlap:
            for p in self.params:
                if p.requires_grad and hasattr(p, "register_post_accumulate_grad_hook"):
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def payload_bytes(self) -> int:
        return sum(p.numel() for p in self.params) * 4

    def close(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []

    # ------------------------------------------------------------------ internals
    def _on_grad(self, p: torch.Tensor) -> None:
        if world_size() == 1 and not self.solo:
            return
        bi = self._bucket_of[id(p)]
        self._ready[bi] += 1
        # issue in bucket order only (every rank must enqueue the same sequence of collectives): a later bucket that
        # completes first waits for its predecessors
        while True:
            nxt = next((i for i, w in enumerate(self._works) if w is None), None)
            if nxt is None or self._ready[nxt] < sum(1 for q in self.buckets[nxt] if q.requires_grad):
                break
            self._issue(nxt)
            self.issued_early += 1

    @torch.no_grad()
    def _issue(self, bi: int) -> None:
        bucket = self.buckets[bi]
        n = sum(p.numel() for p in bucket)
        dev = bucket[0].device
        if bi >= len(self._flat) or self._flat[bi].numel() != n or self._flat[bi].device != dev:
            flat = torch.empty(n, dtype=torch.float32, device=dev)
            while len(self._flat) <= bi:
                self._flat.append(flat)
            self._flat[bi] = flat
        flat = self._flat[bi]
        all_there = all(p.grad is not None and p.grad.dtype == torch.float32 for p in bucket)
        self._fast[bi] = all_there
        if all_there:
            # a gradient that already is a view of this buffer (kept from the last call) must not alias the output
            parts = [(p.grad.clone() if p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() else p.grad).reshape(-1)
                     for p in bucket]
            torch.cat(parts, out=flat)
        else:
            off = 0
            for p in bucket:
                k = p.numel()
                if p.grad is None:
                    flat[off:off + k].zero_()
                else:
                    flat[off:off + k].copy_(p.grad.reshape(-1))
                off += k
        self._stamp[bi] = self._grad_stamp(bucket)
        # SUM + one divide kernel on every backend: gloo has no AVG, and a 12 us kernel is not worth a second code path
        self._works[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    @staticmethod
    def _grad_stamp(bucket):
        return tuple((None if p.grad is None else (id(p.grad), p.grad._version)) for p in bucket)

    def reset(self) -> None:
        """Forget a backward whose step will not be taken: wait for the collectives already in flight (every rank issued
        them, so they complete) and drop them, so that the next backward starts from scratch."""
        for bi, w in enumerate(self._works):
            if w is not None:
                w.wait()
        self._ready = [0] * len(self.buckets)
        self._works = [None] * len(self.buckets)
        self._stamp = [None] * len(self.buckets)

    @torch.no_grad()
    def __call__(self) -> None:
        """After this call every p.grad holds the mean (or sum) over ranks."""
        ws = world_size()
        if ws == 1 and not self.solo:
            return
        early = sum(1 for w in self._works if w is not None)
        # a gradient that changed after its bucket was issued (second backward before this call): the early reduction
        # holds a stale value -- wait for it (collectives complete in issue order on every rank) and reduce the bucket again.
        # Every rank ran the same sequence of backwards, so every rank takes the same decision for every bucket.
        for bi in range(len(self.buckets)):
            if self._works[bi] is not None and self._stamp[bi] != self._grad_stamp(self.buckets[bi]):
                self._works[bi].wait()
                self._works[bi] = None
                self.reissued += 1
                early -= 1
        for bi in range(len(self.buckets)):
            if self._works[bi] is None:
                self._issue(bi)
        for bi, bucket in enumerate(self.buckets):
            self._works[bi].wait()
            flat = self._flat[bi]
            if self.average:
                flat.div_(ws)
            off = 0
            for p in bucket:
                k = p.numel()
                g = flat[off:off + k].view_as(p)
                if self._fast[bi] or p.grad is None:
                    p.grad = g if self._fast[bi] else g.clone()
                else:
                    p.grad.copy_(g)
                off += k
        self.issued_early = early
        self._ready = [0] * len(self.buckets)
        self._works = [None] * len(self.buckets)
        self._stamp = [None] * len(self.buckets)


def allreduce_grads(params: Sequence[torch.Tensor], average: bool = True) -> None:
    """One-shot convenience wrapper (builds the buckets every call; no overlap)."""
    GradAllReducer(params, average=average, overlap=False)()

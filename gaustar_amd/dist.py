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

import functools
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
    return _epoch_permutation(num_views, seed + epoch)[(world * k + rank_) % num_views]


@functools.lru_cache(maxsize=8)
def _epoch_permutation(num_views: int, seed: int) -> tuple:
    """One permutation per epoch (drawn once: a fresh generator + randperm per ITERATION cost ~50 us of host time, a
    twentieth of a refinement iteration)."""
    g = torch.Generator().manual_seed(seed)
    return tuple(int(i) for i in torch.randperm(num_views, generator=g))


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

    More than one backward per call (gradient accumulation, a separate regulariser backward) is supported: a parameter
    whose hook fires a second time before the call marks its bucket DIRTY, and the call reduces every dirty bucket once
    more, after every bucket has had its first reduction -- on every rank in the same order, whichever buckets a rank
    happened to issue early (that differs between ranks when a parameter has no gradient on one of them, so the rule must
    not depend on it: a rank that had not issued a dirty bucket early reduces it twice at call time, the first time
    redundantly).  Assumption: all ranks run the same sequence of backwards over the same parameters.  Gradients edited
    in place between the backward and the call (clipping) are invisible to the hooks: call `mark_dirty()` first -- an
    early-issued bucket whose gradients changed without either raises instead of silently losing the change (identity and
    version counter of every gradient are recorded at issue time).  A step that is abandoned after its backward (NaN
    loss, an evaluation backward) must call `reset()` before the next backward, or its in-flight buckets would be taken for
    the next step's; the reducer cannot tell the two apart."""

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
        self._fired: dict = {}                         # id(param) -> hook firings since the last call
        self._dirty = [False] * len(self.buckets)      # a second backward (or mark_dirty) touched the bucket
        self.reissued = 0                              # buckets reduced a second time in a call
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
        n = self._fired.get(id(p), 0) + 1
        self._fired[id(p)] = n
        if n > 1:   # a second backward before the call: the bucket is reduced again at call time (rank-independent rule)
            self._dirty[bi] = True
            return
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

    def mark_dirty(self) -> None:
        """Gradients were edited in place after the backward (clipping, scaling): every bucket is reduced (again) at call time.
        Must be called on every rank."""
        self._dirty = [True] * len(self.buckets)

    def reset(self) -> None:
        """Forget a backward whose step will not be taken: issue what other ranks may have issued early (so that every
        rank's sequence of collectives stays the same), wait for everything in flight and drop it, so that the next
        backward starts from scratch."""
        if world_size() > 1 or self.solo:
            for bi in range(len(self.buckets)):
                if self._works[bi] is None:
                    self._issue(bi)
        for bi, w in enumerate(self._works):
            if w is not None:
                w.wait()
        self._clear()

    def _clear(self) -> None:
        self._ready = [0] * len(self.buckets)
        self._works = [None] * len(self.buckets)
        self._stamp = [None] * len(self.buckets)
        self._dirty = [False] * len(self.buckets)
        self._fired = {}

    @torch.no_grad()
    def __call__(self) -> None:
        """After this call every p.grad holds the mean (or sum) over ranks."""
        ws = world_size()
        if ws == 1 and not self.solo:
            return
        early = sum(1 for w in self._works if w is not None)
        # first round: every bucket that has not left yet, in bucket order
        for bi in range(len(self.buckets)):
            if self._works[bi] is None:
                self._issue(bi)
        # second round: dirty buckets once more, in bucket order (see the class docstring for why this may be redundant on
        # a rank and still has to happen)
        for bi in range(len(self.buckets)):
            if self._dirty[bi]:
                self._works[bi].wait()
                self._issue(bi)
                self.reissued += 1
            elif self._stamp[bi] != self._grad_stamp(self.buckets[bi]):
                raise RuntimeError("GradAllReducer: gradients of a bucket changed after its all-reduce was issued, outside a "
                                   "backward pass; call mark_dirty() (on every rank) before the reducer")
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
        self._clear()


def allreduce_grads(params: Sequence[torch.Tensor], average: bool = True) -> None:
    """One-shot convenience wrapper (builds the buckets every call; no overlap)."""
    GradAllReducer(params, average=average, overlap=False)()


# ------------------------------------------------------------------------------------------------------------------
# Reduce-scatter + rank-sharded Adam + all-gather: the optimiser step of N view-parallel ranks without N copies of Adam.
# ------------------------------------------------------------------------------------------------------------------
def _hip_adam_segment(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    """One contiguous segment through gaustar_amd's fused kernel (gsr_adam_step; the arithmetic of torch's
    _single_tensor_adam).  HIP tensors only -- there is no CPU path."""
    from . import _host, _lib
    if not param.is_cuda:
        raise RuntimeError("gaustar_amd.dist.ShardedAdam: parameters must live on a HIP (cuda) device -- there is no CPU path")
    lib = _lib.load()
    with _host.on_device(param.device):
        _lib.check(lib.gsr_adam_step(param.numel(), param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                     float(lr), float(beta1), float(beta2), float(eps), int(step),
                                     _host.raw_stream(param.device.index)), "gsr_adam_step")


class ShardedAdam:
    """`GradAllReducer` + `optim.Adam` for N view-parallel ranks as ONE object: reduce-scatter of the gradients, Adam on
    this rank's 1/N of every bucket, all-gather of the updated parameters (the place of sugar_optimizer.py:99-101:
    `optimizer.step()` right after the backward of refine.py:794).

    Why.  An all-reduce followed by the same Adam step on every rank moves 2 (N-1)/N S bytes per GPU and then runs N
    identical 28-byte-per-parameter updates (110 us for config C's 24.7 M parameters -- a third of a rendered view).
    Reduce-scatter + all-gather moves the SAME bytes, but between the two halves each rank only updates its own 1/N of the
    parameters (14 us at N = 8), the optimiser state shrinks to 1/N per rank, and the all-gather of an early bucket
    overlaps the Adam kernels of the later ones (it runs on RCCL's stream).

    How.  Parameters are taken in the order their gradients become final during backward (`ready_order`, e.g.
    harness.SurfaceGaussians.grad_ready_order()) and laid out back to back -- every parameter 16-byte aligned -- in flat
    buckets of at most `bucket_bytes`, each padded to N equal 16-byte-aligned shards.  The parameters are RE-POINTED at
    their slices of the flat buffers (`p.data` becomes a view; values are preserved), so the all-gather lands directly in
    the storage the next forward reads.  A post-accumulate-grad hook per parameter copies nothing: when the last gradient
    of a bucket has arrived the bucket's gradients are packed into its flat gradient buffer and the SUM reduce-scatter
    is issued asynchronously, in bucket order on every rank, while autograd goes on.  `step()` issues what is left and then,
    bucket by bucket: waits for the bucket's reduce-scatter (a stream wait, not a host wait, on RCCL), divides the shard by
    N, runs the Adam kernel on every (parameter, shard) intersection with that parameter's group's CURRENT `lr` (the
    trainer rewrites the learning rates every iteration, sugar_optimizer.py:104-118) and issues the bucket's all-gather;
    finally it waits for the all-gathers and bumps the parameters' version counters.

    Same hyper-parameters and update rule as `torch.optim.Adam(groups, eps=...)` without weight decay / amsgrad, one step
    counter per parameter; a parameter whose gradient is None on this rank contributes zeros (the step is still taken for
    it: some other rank may have seen it).  After `step()` every `p.grad` still holds this rank's LOCAL gradient --
    nothing reads it; call `zero_grad()` (set to None) before the next backward as the reference's loop does.
    `gather_state()` reassembles torch.optim.Adam-shaped state (`step`, `exp_avg`, `exp_avg_sq` per parameter) on every
    rank for checkpoints.  gloo (CPU tests) has no reduce-scatter: there the gradient half is an all-reduce of which the rank
    keeps its shard -- same values, more bytes.

    `segment_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step)`: the in-place update of one contiguous
    segment; default = the HIP kernel.  Tests inject a torch implementation to run the bookkeeping on CPU."""

    def __init__(self, param_groups, ready_order: Sequence[torch.Tensor] | None = None, betas=(0.9, 0.999), eps: float = 1e-8,
                 bucket_bytes: int = 32 << 20, average: bool = True, overlap: bool = True, run_at_world_size_1: bool = False,
                 segment_step=None, communicate: bool = True, gather_first: Sequence[torch.Tensor] | None = None):
        self._defaults = {"lr": 1e-3, "betas": tuple(betas), "eps": float(eps)}
        self.param_groups = [dict(g) for g in param_groups]
        for g in self.param_groups:
            g["params"] = list(g["params"])
            g.setdefault("lr", 1e-3)
            g.setdefault("betas", tuple(betas))
            g.setdefault("eps", float(eps))
        group_of = {id(p): g for g in self.param_groups for p in g["params"]}
        order = list(ready_order) if ready_order is not None else [p for g in self.param_groups for p in g["params"]]
        if {id(p) for p in order} != set(group_of) or len(order) != len(group_of):
            raise ValueError("ShardedAdam: ready_order must list every parameter of the groups exactly once")
        # communicate = False: this process steps ALL parameters on its own, whatever the process group (the single-GPU
        # reference of a scaling measurement taken inside a multi-rank job)
        self.world = world_size() if communicate else 1
        self.rank = rank() if communicate else 0
        self.solo = bool(run_at_world_size_1) and communicate
        self.average = average
        self._comm = self.world > 1 or self.solo
        self._segment_step = segment_step or _hip_adam_segment
        self._group_of = group_of
        self._bucket_bytes = int(bucket_bytes)
        self.buckets: List[dict] = []
        self._bucket_of = {}
        self._lay_out(order)
        self._steps = {id(p): 0 for p in order}
        self._fired: dict = {}
        self._order = order
        self.issued_early = 0
        self.reissued = 0
        self._hooks = []
        # gradient sink (grad_views / written): parameters whose gradient a backward wrote straight into the flat buffer
        self._sunk: set = set()
        self._expect_hook: set = set()
        self._views = None
        # gather_first (lazy all-gather): the parameters the NEXT forward reads first (harness.SurfaceGaussians: the mesh
        # producer's).  step() then issues the all-gathers of THEIR buckets first and waits only for those; the other buckets'
        # gathers (the SH coefficients: 70 % of the payload) are issued behind them and left in flight -- wait_params() is the
        # fence their consumers call (harness._RenderMeshBound does, right before each producer).  None: step() waits for all.
        self._gather_first = None if gather_first is None else {id(p) for p in gather_first}
        self._hooks_on = bool(overlap and self._comm)
        if self._hooks_on:
            for p in order:
                if p.requires_grad and hasattr(p, "register_post_accumulate_grad_hook"):
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _lay_out(self, order) -> None:
        """Append flat buckets for `order` (parameters in the order their gradients become final): (bucket, offset) per
        parameter, every parameter 16-byte aligned, each bucket padded to N equal 16-byte-aligned shards; the parameters are
        re-pointed at their slices (values preserved)."""
        W = max(self.world, 1)
        group_of = self._group_of
        first_new = len(self.buckets)
        cur, off = [], 0
        for p in order:
            if p.dtype != torch.float32:
                raise ValueError("ShardedAdam: float32 parameters only")
            k = (p.numel() + 3) // 4 * 4
            if cur and (off + k) * 4 > self._bucket_bytes:
                self.buckets.append(dict(entries=cur, used=off))
                cur, off = [], 0
            cur.append((p, off))
            off += k
        if cur:
            self.buckets.append(dict(entries=cur, used=off))
        for bi in range(first_new, len(self.buckets)):
            b = self.buckets[bi]
            dev = b["entries"][0][0].device
            n = (b["used"] + 4 * W - 1) // (4 * W) * (4 * W)           # N equal shards of whole float4s
            S = n // W
            flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
            with torch.no_grad():
                for p, o in b["entries"]:
                    flat_p[o:o + p.numel()].copy_(p.detach().reshape(-1))
                    p.data = flat_p[o:o + p.numel()].view(p.shape)
                    self._bucket_of[id(p)] = bi
            b.update(n=n, S=S, flat_p=flat_p, flat_g=torch.zeros(n, dtype=torch.float32, device=dev),
                     exp_avg=torch.zeros(S, dtype=torch.float32, device=dev), exp_avg_sq=torch.zeros(S, dtype=torch.float32, device=dev),
                     need=sum(1 for p, _ in b["entries"] if p.requires_grad), ready=0, rs=None, stamp=None, dirty=False, ag=None)
            lo, hi = self.rank * S, (self.rank + 1) * S
            # The reduction lands in a buffer of its OWN (1/N of the bucket), not in place: with a gradient sink p.grad aliases
            # flat_g, and a second backward before the step must find this rank's LOCAL gradient there to add to -- an in-place
            # reduce-scatter had replaced this rank's shard of it by the sum over ranks, which the re-issued reduction then
            # counted twice (found by tests/test_dist.py::test_sharded_adam_gradient_sink_gloo_world2).
            # (Nothing is exchanged without communication: the "reduced" shard then IS the local gradient.)
            b["shard_g"] = torch.zeros(S, dtype=torch.float32, device=dev) if self._comm else b["flat_g"][lo:hi]
            b["shard_lo"] = lo
            # (parameter, shard) intersections: (param, group, start in the shard, start in the flat buffer, length)
            b["segments"] = []
            for p, o in b["entries"]:
                a, e = max(o, lo), min(o + p.numel(), hi)
                if e > a:
                    b["segments"].append((p, group_of[id(p)], a - lo, a, e - a))
        self._views = None

    def add_param_group(self, param_group: dict) -> None:
        """torch.optim.Optimizer.add_param_group (the reference wrapper forwards to it, sugar_optimizer.py:117-118): the new
        parameters get flat buckets of their own behind the existing ones -- call it on every rank, between steps."""
        if any(b["rs"] is not None for b in self.buckets) or any(b["ready"] > 0 for b in self.buckets) or self._fired:
            raise RuntimeError("ShardedAdam.add_param_group: a reduction is in flight or gradients of this step have already "
                               "arrived; call it between step() and the next backward")
        self.wait_params()                      # (lazy gathers of the last step: the layout below must not move under them)
        g = dict(param_group)
        ps = g["params"]
        g["params"] = [ps] if isinstance(ps, torch.Tensor) else list(ps)
        g.setdefault("lr", self._defaults["lr"])
        g.setdefault("betas", self._defaults["betas"])
        g.setdefault("eps", self._defaults["eps"])
        # validate BEFORE touching any state: a refused group must leave the optimiser as it was
        seen = set()
        dev0 = self._order[0].device if self._order else None
        for p in g["params"]:
            if not isinstance(p, torch.Tensor):
                raise TypeError("ShardedAdam.add_param_group: parameters must be tensors")
            if id(p) in self._group_of or id(p) in seen:
                raise ValueError("some parameters appear in more than one parameter group")
            if p.dtype != torch.float32:
                raise ValueError("ShardedAdam: float32 parameters only")
            if dev0 is not None and p.device != dev0:
                raise ValueError("ShardedAdam.add_param_group: parameters must live on the optimiser's device")
            seen.add(id(p))
        self.param_groups.append(g)
        for p in g["params"]:
            self._group_of[id(p)] = g
            self._steps[id(p)] = 0
        self._order.extend(g["params"])
        self._lay_out(g["params"])
        if self._hooks_on:
            for p in g["params"]:
                if p.requires_grad and hasattr(p, "register_post_accumulate_grad_hook"):
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------ bookkeeping
    def payload_bytes(self) -> int:
        return sum(p.numel() for p in self._order) * 4

    def state_bytes_per_rank(self) -> int:
        return sum(2 * b["S"] * 4 for b in self.buckets)

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self._order:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def close(self) -> None:
        self.wait_params()
        for h in self._hooks:
            h.remove()
        self._hooks = []

    # ------------------------------------------------------------------ gradient sink
    def grad_views(self) -> dict:
        """{id(parameter): view of the flat gradient buffer, shaped like the parameter}.  A backward that controls where its
        gradients are written (harness.SurfaceGaussians with `grad_sink = this optimiser`) writes them HERE and returns the
        views to autograd: `p.grad` then aliases the buffer the reduce-scatter reads, the packing copy of the hook path (the
        whole payload read and written once more per step, one kernel per parameter) disappears, and `written()` lets a
        bucket leave while the rest of that backward is still being computed.  Only for parameters whose `.grad` is None
        when the backward runs (an accumulating second backward must go through autograd's own addition)."""
        if self._views is None:
            self._views = {id(p): b["flat_g"][o:o + p.numel()].view(p.shape) for b in self.buckets for p, o in b["entries"]}
        return self._views

    def accepts(self, p) -> bool:
        """Whether a backward may write the gradient of `p` straight into its grad_views() slot: only the FIRST delivery of a
        step, and only while `p.grad` is None.  A second render in the same backward (two sink nodes under one loss) must hand
        its gradient to autograd as an ordinary tensor -- autograd then sums the two out of place, `p.grad` stops aliasing the
        flat buffer, and the bucket is packed and reduced again in step() (see _on_grad / _pack)."""
        return p is not None and p.grad is None and id(p) not in self._sunk and p.requires_grad

    # ------------------------------------------------------------------ lazy all-gather
    def pending_gathers(self) -> int:
        return sum(1 for b in self.buckets if b["ag"] is not None)

    def wait_params(self, params=None) -> None:
        """Fence for readers of parameters whose all-gather step() left in flight (`gather_first`): after it, the flat
        buffers of the buckets holding `params` (None: all) are complete on this rank.  On RCCL `wait()` makes the CURRENT
        stream wait for the collective (no host block), so a kernel launched next is ordered behind it; on gloo it blocks.
        Costs a dictionary look-up per parameter when nothing is pending."""
        if not any(b["ag"] is not None for b in self.buckets):
            return
        # (parameters this optimiser does not hold -- frozen ones, None placeholders -- have nothing in flight)
        which = range(len(self.buckets)) if params is None else sorted({self._bucket_of[id(p)] for p in params
                                                                        if p is not None and id(p) in self._bucket_of})
        from .optim import _bump_version
        for bi in which:
            b = self.buckets[bi]
            if b["ag"] is not None:
                b["ag"].wait()
                b["ag"] = None
                for p, _ in b["entries"]:
                    _bump_version(p)

    def written(self, params) -> None:
        """The gradients of `params` now sit in their grad_views() (kernels launched on the current stream): count them as
        arrived and issue every bucket that is complete, in bucket order -- exactly what the post-accumulate hook does when
        autograd delivers a gradient, only earlier.  The hook that fires for the same parameter later in this backward is
        ignored once."""
        for p in params:
            if id(p) in self._sunk:          # a second delivery before the step: reduce the bucket again in step()
                self.buckets[self._bucket_of[id(p)]]["dirty"] = True
                continue
            self._sunk.add(id(p))
            if self._hooks:
                self._expect_hook.add(id(p))
            self._fired[id(p)] = self._fired.get(id(p), 0) + 1
            self.buckets[self._bucket_of[id(p)]]["ready"] += 1
        if self._hooks:                      # (no hooks = no overlap: step() issues everything)
            self._issue_ready()

    def mark_dirty(self) -> None:
        """Gradients were edited in place after the backward: every bucket is reduced (again) by step() (see GradAllReducer)."""
        for b in self.buckets:
            b["dirty"] = True

    def reset(self) -> None:
        """Forget a backward whose step will not be taken (see GradAllReducer.reset)."""
        for bi, b in enumerate(self.buckets):
            if b["rs"] is None and self._comm:
                self._issue_grads(bi)
        for b in self.buckets:
            if b["rs"] is not None:
                b["rs"].wait()
            b.update(rs=None, stamp=None, ready=0, dirty=False)
        self._fired = {}
        self._sunk.clear(); self._expect_hook.clear()

    def _stamp(self, b):
        # A gradient delivered through the sink may be issued before autograd has set p.grad, and p.grad then ALIASES the flat
        # buffer: its stamp is the version counter of that buffer (shared by all of its views), which an in-place edit of
        # p.grad after the backward (gradient clipping) bumps -- so the documented guard also holds with a sink.  (The
        # kernels that fill the buffer write through raw pointers and bump nothing; _pack's own copies run before the stamp.)
        fv = b["flat_g"]._version
        return tuple((("sunk", fv) if id(p) in self._sunk else None if p.grad is None else (id(p.grad), p.grad._version))
                     for p, _ in b["entries"])

    def _on_grad(self, p: torch.Tensor) -> None:
        if id(p) in self._expect_hook:   # delivered through the sink earlier in this backward: already counted
            self._expect_hook.discard(id(p))
            # ... unless autograd did NOT adopt the sink's view as p.grad: another contribution to the same parameter in this
            # backward (a regulariser, a second render) makes it sum view + other OUT OF PLACE, and the flat buffer -- possibly
            # already on its way -- holds only the sink's share.  The bucket is packed from p.grad and reduced again in step().
            if p.grad is not None and p.grad.data_ptr() != self.grad_views()[id(p)].data_ptr():
                self.buckets[self._bucket_of[id(p)]]["dirty"] = True
            return
        bi = self._bucket_of[id(p)]
        n = self._fired.get(id(p), 0) + 1
        self._fired[id(p)] = n
        if n > 1:   # second backward before the step: reduce the bucket again in step() (rank-independent rule)
            self.buckets[bi]["dirty"] = True
            return
        self.buckets[bi]["ready"] += 1
        self._issue_ready()

    def _issue_ready(self) -> None:
        while True:   # bucket order only: every rank must enqueue the same sequence of collectives
            nxt = next((i for i, b in enumerate(self.buckets) if b["rs"] is None), None)
            if nxt is None or self.buckets[nxt]["ready"] < self.buckets[nxt]["need"]:
                break
            self._issue_grads(nxt)
            self.issued_early += 1

    @torch.no_grad()
    def _pack(self, b) -> None:
        for p, o in b["entries"]:
            dst = b["flat_g"][o:o + p.numel()]
            if p.grad is not None and p.grad.data_ptr() == dst.data_ptr():
                continue                      # written in place through the sink and adopted by autograd as p.grad
            if id(p) in self._sunk and p.grad is None:
                continue                      # in place already; autograd has not delivered it yet
            if p.grad is None:
                dst.zero_()
            else:
                dst.copy_(p.grad.reshape(-1))

    @torch.no_grad()
    def _issue_grads(self, bi: int) -> None:
        b = self.buckets[bi]
        self._pack(b)
        b["stamp"] = self._stamp(b)
        lo = b["shard_lo"]
        if not self._comm:
            b["rs"] = _Done()
            return
        backend = dist.get_backend() if dist.is_initialized() else "none"
        if backend == "gloo":   # no reduce-scatter in gloo: all-reduce a copy, keep the shard (tests)
            tmp = b["flat_g"].clone()
            b["rs"] = _Then(dist.all_reduce(tmp, op=dist.ReduceOp.SUM, async_op=True),
                            lambda: b["shard_g"].copy_(tmp[lo:lo + b["S"]]))
        else:
            b["rs"] = dist.reduce_scatter_tensor(b["shard_g"], b["flat_g"], op=dist.ReduceOp.SUM, async_op=True)

    # ------------------------------------------------------------------ the step
    @torch.no_grad()
    def step(self) -> None:
        early = sum(1 for b in self.buckets if b["rs"] is not None)
        for bi, b in enumerate(self.buckets):   # first round: what has not left yet, in bucket order
            if b["rs"] is None:
                self._issue_grads(bi)
        for b in self.buckets:                  # (without hooks nobody has looked at how autograd delivered a sunk gradient)
            if b["rs"] is not None and not b["dirty"]:
                for p, o in b["entries"]:
                    if id(p) in self._sunk and p.grad is not None and p.grad.data_ptr() != b["flat_g"][o:o + 1].data_ptr():
                        b["dirty"] = True
        for bi, b in enumerate(self.buckets):   # second round: dirty buckets once more (GradAllReducer's rule)
            if b["dirty"]:
                b["rs"].wait()
                self._issue_grads(bi)
                self.reissued += 1
            elif b["stamp"] != self._stamp(b):
                raise RuntimeError("ShardedAdam: gradients of a bucket changed after its reduction was issued, outside a backward "
                                   "pass; call mark_dirty() (on every rank) before step()")
        self.issued_early = early
        self._fired = {}
        for p in self._order:
            self._steps[id(p)] += 1
        self.wait_params()                      # (a previous step's lazy gathers: nobody may update a half-gathered buffer)
        lazy = self._gather_first is not None and self._comm
        first = [any(id(p) in self._gather_first for p, _ in b["entries"]) for b in self.buckets] if lazy else None
        gathers, later = [], []
        for bi, b in enumerate(self.buckets):
            b["rs"].wait()
            if self.average and self._comm and self.world > 1:
                b["shard_g"].div_(self.world)
            lo = self.rank * b["S"]
            for p, g, s_off, f_off, n in b["segments"]:
                b1, b2 = g["betas"]
                self._segment_step(b["flat_p"][f_off:f_off + n], b["shard_g"][s_off:s_off + n], b["exp_avg"][s_off:s_off + n],
                                   b["exp_avg_sq"][s_off:s_off + n], float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                   self._steps[id(p)])
            if self._comm:
                if lazy and not first[bi]:
                    later.append(bi)         # issued behind the buckets the next forward needs first (same order on every rank)
                else:
                    gathers.append(dist.all_gather_into_tensor(b["flat_p"], b["flat_p"][lo:lo + b["S"]], async_op=True))
            b.update(rs=None, stamp=None, ready=0, dirty=False)
        for bi in later:
            b = self.buckets[bi]
            lo = self.rank * b["S"]
            b["ag"] = dist.all_gather_into_tensor(b["flat_p"], b["flat_p"][lo:lo + b["S"]], async_op=True)
        for w in gathers:
            w.wait()
        self._sunk.clear(); self._expect_hook.clear()
        from .optim import _bump_version
        for p in self._order:
            if self.buckets[self._bucket_of[id(p)]]["ag"] is None:   # (lazily gathered parameters are bumped by wait_params)
                _bump_version(p)

    # ------------------------------------------------------------------ checkpoints
    @torch.no_grad()
    def gather_state(self) -> dict:
        """{parameter: {"step", "exp_avg", "exp_avg_sq"}} with full-size tensors on every rank (a collective call): the
        state torch.optim.Adam would hold after the same steps."""
        self.wait_params()
        out = {}
        for b in self.buckets:
            full = {}
            for key in ("exp_avg", "exp_avg_sq"):
                if self._comm and self.world > 1:
                    t = torch.empty(b["n"], dtype=torch.float32, device=b[key].device)
                    dist.all_gather_into_tensor(t, b[key].contiguous())
                else:
                    t = b[key]
                full[key] = t
            for p, o in b["entries"]:
                out[p] = {"step": torch.tensor(float(self._steps[id(p)])),
                          "exp_avg": full["exp_avg"][o:o + p.numel()].view(p.shape).clone(),
                          "exp_avg_sq": full["exp_avg_sq"][o:o + p.numel()].view(p.shape).clone()}
        return out


    def _indexed_params(self):
        return [p for g in self.param_groups for p in g["params"]]

    @torch.no_grad()
    def state_dict(self) -> dict:
        """torch.optim.Adam-shaped state dict ({"state": {index: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]},
        parameters numbered over the groups in order) -- what the reference wrapper saves as `optimizer_state_dict`
        (sugar_optimizer.py:120-121).  A collective call: the moments are gathered from all ranks."""
        st = self.gather_state()
        params = self._indexed_params()
        index = {id(p): i for i, p in enumerate(params)}
        state = {index[id(p)]: {"step": v["step"], "exp_avg": v["exp_avg"], "exp_avg_sq": v["exp_avg_sq"]}
                 for p, v in st.items() if self._steps[id(p)] > 0}
        groups = []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d.setdefault("weight_decay", 0); d.setdefault("amsgrad", False); d.setdefault("maximize", False)
            d["params"] = [index[id(p)] for p in g["params"]]
            groups.append(d)
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, state_dict: dict) -> None:
        """Inverse of state_dict(); also takes a torch.optim.Adam state dict over the same parameter groups (a single-GPU
        checkpoint resumed on N ranks): every rank keeps its shard of the moments, the step counters and the groups'
        hyper-parameters (sugar_optimizer.py:123-124)."""
        self.wait_params()
        groups = state_dict["param_groups"]
        if len(groups) != len(self.param_groups) or any(len(a["params"]) != len(b["params"]) for a, b in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters per group")
        for g, src in zip(self.param_groups, groups):
            if src.get("weight_decay", 0) or src.get("amsgrad", False) or src.get("maximize", False):
                raise ValueError("ShardedAdam implements plain Adam: weight_decay / amsgrad / maximize are not supported")
            for k, v in src.items():
                if k not in ("params", "weight_decay", "amsgrad", "maximize"):
                    g[k] = tuple(v) if k == "betas" else v
        params = self._indexed_params()
        ids = [i for src in groups for i in src["params"]]
        state = state_dict.get("state", {})
        for p, i in zip(params, ids):
            e = state.get(i, state.get(str(i)))
            b = self.buckets[self._bucket_of[id(p)]]
            o = next(o_ for q, o_ in b["entries"] if q is p)
            lo, hi = b["shard_lo"], b["shard_lo"] + b["S"]
            a, z = max(o, lo), min(o + p.numel(), hi)
            if e is None:
                self._steps[id(p)] = 0
                if z > a:
                    b["exp_avg"][a - lo:z - lo].zero_(); b["exp_avg_sq"][a - lo:z - lo].zero_()
                continue
            if tuple(e["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("loaded state has a moment tensor of the wrong shape")
            self._steps[id(p)] = int(float(e["step"]))
            if z > a:
                for key in ("exp_avg", "exp_avg_sq"):
                    b[key][a - lo:z - lo].copy_(e[key].reshape(-1)[a - o:z - o].to(b[key].device, torch.float32))


class _Done:
    def wait(self):
        return True


class _Then:
    """An asynchronous work handle plus something to do once it has completed (first wait() only)."""

    def __init__(self, work, after):
        self.work, self.after = work, after

    def wait(self):
        r = self.work.wait()
        if self.after is not None:
            self.after()
            self.after = None
        return r

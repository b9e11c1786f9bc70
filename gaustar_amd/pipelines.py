"""Independent view pipelines on one GPU: T host threads, each with its own HIP stream.

The rasterizer's step is a chain of six dependent kernels, three of which (preprocess, tile scan, scatter: 48 us of a
260 us view on config C) are latency chains that leave most of the chip idle, and the two blends end in a drain.  Nothing
inside ONE view can fill those holes -- every kernel needs its predecessor's whole output -- but a SECOND, independent view
can: with two views in flight on two streams the small kernels of one run under the blends of the other (+13 % views/s on
config C, tools/bench_threads.py; a third pipeline adds 1 %).  Views are independent whenever the Gaussians do not change
between them: camera sweeps and evaluation renders (refined_mesh.py:733-775), and training schedules that take one
optimiser step per V views per GPU (the gradients of the V pipelines are summed before the step -- the view-parallel
multi-GPU mode already makes the effective batch N views; this makes it N x V).  The reference's one-view-per-iteration
loop on one GPU has no independent views and gains nothing here.

Each worker thread owns a stream for everything it does (PyTorch's caching allocator keys blocks by stream, autograd runs a
node's backward on the stream of its forward), so pipelines share no tensor that is written: hand each its own leaf tensors
(`clone_leaves`).  The C library keeps its per-call state per thread (pinned landing pad, error string) and per (device,
stream) (tile counters); it is safe to call from several threads.  The GIL is released inside every C call (ctypes), so the
threads' Python work interleaves while one of them waits for its view's totals.
"""
from __future__ import annotations

import threading
from typing import Callable, Dict, List, Sequence

import torch


def clone_leaves(tensors: Dict[str, torch.Tensor], n: int) -> List[Dict[str, torch.Tensor]]:
    """n independent sets of leaf tensors with the same values (one per pipeline: gradients accumulate per pipeline)."""
    return [{k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in tensors.items()} for _ in range(n)]


class ViewPipelines:
    """run(fn, items): items[i] is handled by pipeline i % n as fn(pipeline_index, item), each pipeline in its own thread
    under its own stream, in order within a pipeline; returns when every pipeline's stream has drained.  Exceptions of a
    worker are re-raised in the caller.  `before` / `after` (optional callables, no arguments) run in the CALLING thread
    right before the workers are released resp. right after the last one has drained its stream -- the place for a timer."""

    def __init__(self, n: int, device: torch.device):
        if n < 1:
            raise ValueError("at least one pipeline")
        self.n = int(n)
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.n)]

    def run(self, fn: Callable, items: Sequence, before: Callable | None = None, after: Callable | None = None) -> None:
        torch.cuda.synchronize(self.device)   # whatever produced the inputs (on any stream) is done before the workers start
        if self.n == 1:
            if before:
                before()
            with torch.cuda.stream(self.streams[0]):
                for it in items:
                    fn(0, it)
                self.streams[0].synchronize()
            if after:
                after()
            return
        gate = threading.Barrier(self.n + 1)
        errors: List[BaseException] = []

        def worker(t: int) -> None:
            # ANY failure of a worker -- before its first wait included (set_device, the stream context) -- is recorded and
            # breaks the barrier, so neither the caller nor the other workers can be left waiting for it
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.streams[t]):
                    gate.wait()      # ready
                    gate.wait()      # go
                    for i in range(t, len(items), self.n):
                        fn(t, items[i])
                    self.streams[t].synchronize()
                    gate.wait()      # done
            except threading.BrokenBarrierError:
                pass                 # somebody else failed (or the caller's before() did): leave
            except BaseException as ex:   # noqa: BLE001 -- handed to the caller
                errors.append(ex)
                gate.abort()

        threads = [threading.Thread(target=worker, args=(t,), daemon=True) for t in range(self.n)]
        for th in threads:
            th.start()
        caller_error = None
        try:
            gate.wait()          # every worker stands at the start line
            if before:
                before()         # (a timer started here also counts the release of the barrier: it errs on the long side)
            gate.wait()          # go
            gate.wait()          # every worker has drained its stream
            if after:
                after()
        except threading.BrokenBarrierError:
            pass                 # a worker failed: its exception is in `errors`
        except BaseException as ex:   # noqa: BLE001 -- before() / after() raised: release the parked workers, then re-raise
            caller_error = ex
            gate.abort()
        for th in threads:
            th.join()
        try:   # whatever the surviving workers had already launched is done before the caller sees the failure
            for st in self.streams:
                st.synchronize()
        except Exception:   # noqa: BLE001
            pass
        if caller_error is not None:
            raise caller_error
        if errors:
            raise errors[0]

"""Adam as GauSTAR configures it, one fused HIP kernel per parameter tensor.

Host-side mirror of the optimiser in gaustar_scene/sugar_optimizer.py:87 -- `torch.optim.Adam(groups, lr=0.0, eps=1e-15)`
with per-group learning rates that the trainer rewrites every iteration (:104-118) -- and of the update rule in
torch/optim/adam.py::_single_tensor_adam (no weight decay, no amsgrad, not maximising).  `Adam` below is a
`torch.optim.Optimizer`: same constructor arguments, same `param_groups`, same state keys (`step`, `exp_avg`,
`exp_avg_sq`), so SuGaROptimizer-style wrappers and `state_dict()` round trips with torch.optim.Adam work unchanged;
only `step()` differs: 28 bytes per parameter through one kernel for all tensors of a step (`gsr_adam_step_multi`) instead
of PyTorch's multi-tensor kernels.  There is no CPU path: parameters must be float32 HIP tensors."""
from __future__ import annotations

import ctypes

import torch

from . import _host, _lib


def _bump_version(p: torch.Tensor) -> None:
    """The kernel writes through data_ptr(), behind autograd's back: count it as the in-place update it is, so that
    version-keyed caches (harness.SurfaceGaussians._geometry) and autograd's saved-tensor checks see the step."""
    inc = getattr(torch.autograd.graph, "increment_version", None)
    if inc is not None:
        inc(p)
    else:
        torch._C._increment_version(p)


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False):
        if weight_decay != 0.0 or amsgrad:
            raise NotImplementedError("gaustar_amd.optim.Adam: weight_decay and amsgrad are not provided (GauSTAR uses neither)")
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0.0, amsgrad=False))
        self._step_views = {}   # id(parameter) -> (its state's step tensor, numpy view of it)

    def zero_grad(self, set_to_none: bool = True):
        """torch.optim.Optimizer.zero_grad; the default (drop the gradients) without the base class's per-call bookkeeping
        (profiler range, foreach grouping: ~20 us per call for eight tensors, once per iteration of a 0.55 ms loop)."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    multi_tensor = True   # one launch per 16 tensors (gsr_adam_step_multi); False: one launch per tensor (gsr_adam_step)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        # tensors that share everything but the learning rate go out together: (device, betas, eps, step count) -> entries
        batches = {}
        step_views = self.__dict__.setdefault("_step_views", {})   # (absent after unpickling)
        for group in self.param_groups:
            lr, (b1, b2), eps = float(group["lr"]), group["betas"], float(group["eps"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("gaustar_amd.optim.Adam: parameters must be contiguous float32 tensors on a HIP (cuda) "
                                       "device -- there is no CPU path")
                if p.grad.is_sparse:
                    raise RuntimeError("gaustar_amd.optim.Adam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:   # same lazy state as torch.optim.Adam (step as a CPU scalar tensor)
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                # (the step count stays what torch.optim.Adam keeps -- a CPU scalar tensor in the state --, bumped through a cached
                # numpy view of its one element: `tensor += 1` is a 3.5 us dispatch per parameter tensor and step)
                step_t = st["step"]
                view = step_views.get(id(p))
                if view is None or view[0] is not step_t:
                    view = (step_t, step_t.numpy()) if (step_t.device.type == "cpu" and step_t.dim() == 0) else (step_t, None)
                    step_views[id(p)] = view
                if view[1] is not None:
                    view[1][...] += 1
                    step_now = int(view[1])
                else:
                    step_t += 1
                    step_now = int(step_t.item())
                g = p.grad if (p.grad.dtype == torch.float32 and p.grad.is_contiguous()) else p.grad.float().contiguous()
                key = (p.device.index, float(b1), float(b2), eps, step_now)
                batches.setdefault(key, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], lr))
        for (dev_index, b1, b2, eps, step), entries in batches.items():
            with _host.on_device(entries[0][0].device):
                stream = _host.raw_stream(dev_index)
                if self.multi_tensor and len(entries) > 1:
                    n = len(entries)
                    ptrs = lambda k: (ctypes.c_void_p * n)(*[e[k].data_ptr() for e in entries])
                    _lib.check(lib.gsr_adam_step_multi(
                        n, (ctypes.c_longlong * n)(*[e[0].numel() for e in entries]), ptrs(0), ptrs(1), ptrs(2), ptrs(3),
                        (ctypes.c_double * n)(*[e[4] for e in entries]), b1, b2, eps, step, stream), "gsr_adam_step_multi")
                else:
                    for p, g, m, v, lr in entries:
                        _lib.check(lib.gsr_adam_step(p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), lr, b1, b2,
                                                     eps, step, stream), "gsr_adam_step")
            for e in entries:
                _bump_version(e[0])
        return loss

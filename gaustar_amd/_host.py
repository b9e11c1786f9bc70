"""Host-side plumbing shared by the Python mirrors of the C ABI: PyTorch's current stream as a raw handle and a
device guard that costs nothing when the device is already current.

A forward + backward step of the rasterizer is ~0.2 ms of GPU work; `torch.cuda.current_stream()` (15 us: it builds a
Stream object and resolves the device through three Python layers) and `with torch.cuda.device(dev)` (~10 us) were a
sixth of the host's share of a step (tools/host_profile.py)."""
from __future__ import annotations

import torch


def raw_stream(dev_index: int) -> int:
    """hipStream_t of PyTorch's current stream on device `dev_index`, as an int (ctypes parameters are declared c_void_p)."""
    return torch._C._cuda_getCurrentRawStream(dev_index)


class on_device:
    """`with torch.cuda.device(dev)` only when `dev` is not the current device already."""
    __slots__ = ("ctx",)

    def __init__(self, dev: torch.device):
        self.ctx = None if torch._C._cuda_getDevice() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False

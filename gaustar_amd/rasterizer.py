"""Python surface of the rasterizer -- a drop-in for the reference package
`diff_gaussian_rasterization` (DGR/diff_gaussian_rasterization/__init__.py):

    GaussianRasterizationSettings   NamedTuple, same 12 fields            (ref :157-169)
    GaussianRasterizer              nn.Module, forward(...) / markVisible (ref :171-220)
    rasterize_gaussians             functional entry                      (ref :21-42)
    _RasterizeGaussians             autograd.Function, same arg order and
                                    same gradient tuple order             (ref :44-155)

Below this file sits the C ABI of include/gsr.h instead of the reference's pybind11 module `_C`;
the glue that the reference keeps in C++ (DGR/rasterize_points.cu: shape check, .contiguous(),
output / scratch / gradient allocation) lives here, with PyTorch owning every byte of device
memory.  No CPU fallback exists: tensors must be on a HIP device and the extension must load.
"""
from __future__ import annotations

import collections
import ctypes
import os
import sys
import threading
import weakref
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _host, _lib


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _ptr(t):
    """Device pointer of an optional tensor; 0-element tensors are 'absent' (NULL), as in the
    reference where data_ptr() of an empty tensor is nullptr (rasterize_points.cu:94-111).
    (A plain int: the ctypes table declares the parameter types, so no c_void_p object is needed per argument.)"""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


_F32 = torch.float32


def _dev_f32(t, device):
    """float32, contiguous, on `device` (the reference calls .contiguous() on every argument;
    viewmatrix in particular arrives as a transposed view, sugar_model.py:1149-1150)."""
    if t is None:
        return None
    if t.dtype is _F32 and t.device == device and t.is_contiguous():   # the steady state: one test, no dispatch
        return t
    if t.numel() == 0:
        return t
    if t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_stream = _host.raw_stream
_on_device = _host.on_device


def _num_channels(colors, background) -> int:
    """3 (the reference's NUM_CHANNELS) unless precomputed colours carry 6 or 4 columns: two targets that share
    geometry (GauSTAR's RGB + depth-as-colour renders, refine.py:552 / :607) blended in one pass -- 6 = two 3-channel
    targets, 4 = RGB + one scalar target (the trainer only reads channel 0 of its depth render, refine.py:616) -- an
    extension, see include/gsr.h::gsr_forward_stage2_mt."""
    C = 3
    if colors is not None and colors.numel() != 0:
        if colors.ndimension() != 2 or int(colors.size(1)) not in (3, 4, 6):
            raise RuntimeError("colors_precomp must have dimensions (num_points, 3) or (num_points, 6) [or (num_points, 4)]")
        C = int(colors.size(1))
    if background.numel() != C:
        raise RuntimeError(f"bg must have {C} elements (one per colour channel), got {background.numel()}")
    return C


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"gaustar_amd: {what} must live on a HIP (cuda) device -- there is no CPU path")


# --------------------------------------------------------------------------------------------
# Binding-level functions: same positional signatures as the reference's `_C` module
# (DGR/rasterize_points.h:17-66), implemented over the C ABI.
# --------------------------------------------------------------------------------------------
# device index -> bytes of binning scratch to hand gsr_forward_fused up front (1.25x the largest need seen so far)
_BINNING_HINT: dict = {}
_HINT_LOCK = threading.Lock()   # trainer thread + evaluation thread may render on one device
_FUSED = os.environ.get("GSR_FUSED_FORWARD", "1") != "0"   # 0: always stage 1, allocate exactly, stage 2
# Plans (include/gsr.h, gsr_forward_planned): per camera the layout of its tiles' buckets, left behind by one view for the
# next view of that camera.  A camera is recognised by the CONTENTS of its view matrix (+ image size and field of view): the
# reference's mesh-bound caller builds a fresh `torch.Tensor(getWorld2View(...)).transpose(0, 1).cuda()` on every render call
# (gaustar_scene/sugar_model.py:1149-1150; full_proj_transform derives from it, :1163), so neither the tensor nor its address
# names the camera -- the caching allocator hands consecutive cameras the same 512-byte block.  _camera_key() below: a tensor
# seen before (gaussian_renderer.render keeps one per camera, gaussian_splatting/scene/cameras.py:212; so does harness.py) costs
# a dictionary look-up, a fresh one is read through gsr_camera_key (one-wave kernel on a side stream, ~10 us, the caller's
# stream is not waited for) -- or the caller names the camera itself (`plan_key`).  A plan is a hint: a wrong or stale one costs
# the exact path, never a wrong pixel.
_PLANNED = os.environ.get("GSR_PLANNED", "1") != "0"
_PLAN_SLOTS = int(os.environ.get("GSR_PLAN_SLOTS", "2048"))
# Slack level a camera's first plan is made with (include/gsr.h: capacity = count + (max(16, count / 8) + an eighth of what the
# largest neighbouring tile holds more) << level).  Differentiable renders belong to a training loop whose Gaussians move between
# a camera's visits: level 1.  Measured in the windowed refinement loop (tools/replan_ab.sh: tools/bench_window.py, 3 visits per
# camera and frame, 1 280 views with a plan to try): round 5's per-tile slack left 397 / 316 / 138 views outgrowing their plans at
# levels 0 / 1 / 2 -- always a handful of SILHOUETTE tiles, whose counts jump by factors when the surface's edge moves a few pixels
# their way; with the neighbour term (round 6) 104 / 5 / 2, and with every planned view re-planning from its own counts (a plan is
# one visit old) 29 / 1 / 1.  On a static scene the larger buckets cost the backward empty work items: 0.2283 / 0.2300 / 0.2315 ms
# per view at levels 0 / 1 / 2.  Forward-only renders (sweeps of a fixed model): level 0.  A view that outgrows its plan raises
# its camera's level by one.
_PLAN_LEVEL_ENV = os.environ.get("GSR_PLAN_LEVEL")
_PLANS: "collections.OrderedDict" = collections.OrderedDict()   # key -> _Plan
_PLAN_RETRY = 32   # a camera whose view could not be planned (a list too long for the in-kernel sort) is asked again after this many views
_PLAN_DIAG = bool(os.environ.get("GSR_PLAN_DIAG"))
PLAN_STATS = {"planned": 0, "exact": 0, "misfit": 0}            # views binned by a plan / without one / that outgrew theirs


class _Plan:
    """One camera's plan: the device buffer, the pinned plan_info block the library reads and writes (include/gsr.h), and the
    binding's own bookkeeping (views to render without asking, misfits in a row, views seen while unplannable)."""
    __slots__ = ("buf", "info", "skip", "misfits", "idle", "_lib")

    def __init__(self, lib, nbytes, byte_opts, level):
        self.buf = torch.empty(nbytes, **byte_opts)
        self.info = lib.gsr_plan_info_new()
        if not self.info:
            raise RuntimeError("gsr_plan_info_new failed")
        self.skip, self.misfits, self.idle, self._lib = 0, 0, 0, lib
        self.info[4] = level

    def __del__(self):
        try:
            self._lib.gsr_plan_info_free(self.info)
        except Exception:
            pass


# Plans that left the table while a builder of theirs may still be on its way to their pinned block: (event recorded on the
# plan's own stream and device when it left, plan).  The block is recycled (by _Plan.__del__) once the event has passed -- looked
# at when the next plan is made, nobody waits, and no lock is held across a device synchronisation.
_PLAN_GRAVE: list = []


def _stream_of(dev_index, raw_stream):
    return torch.cuda.ExternalStream(raw_stream, device=dev_index) if raw_stream else torch.cuda.default_stream(dev_index)


def _retire(key, ent):
    try:
        ev = torch.cuda.Event()
        ev.record(_stream_of(key[0], key[1]))
        _PLAN_GRAVE.append((ev, ent))
    except Exception:   # (no event: keep the entry alive for good rather than recycle a block that may be written)
        _PLAN_GRAVE.append((None, ent))


def _plan_entry(key, lib, nbytes, byte_opts, level):
    with _HINT_LOCK:
        ent = _PLANS.get(key)
        if ent is not None and ent.buf.numel() == nbytes:
            _PLANS.move_to_end(key)
            return ent
        if _PLAN_GRAVE:
            _PLAN_GRAVE[:] = [(ev, e) for ev, e in _PLAN_GRAVE if ev is None or not ev.query()]
        if ent is not None:
            _retire(key, ent)
        ent = _Plan(lib, nbytes, byte_opts, level)
        _PLANS[key] = ent
        while len(_PLANS) > _PLAN_SLOTS:
            _retire(*_PLANS.popitem(last=False))
        return ent


def drop_plans():
    """Forget every camera's plan (the next view of each renders the exact way and re-plans)."""
    with _HINT_LOCK:
        devs = {k[0] for k in _PLANS}
        for k, e in list(_PLANS.items()):
            _retire(k, e)
        _PLANS.clear()
    for d in devs:   # (outside the lock; every device that held a plan, not just the current one)
        torch.cuda.synchronize(d)
    with _HINT_LOCK:
        _PLAN_GRAVE[:] = [(ev, e) for ev, e in _PLAN_GRAVE if ev is None or not ev.query()]


# id(tensor) -> (weak reference, version counter, data pointer, key): view-matrix tensors whose contents have been read.
_CAM_KEYS: dict = {}
CAMERA_KEY_STATS = {"known_tensor": 0, "read": 0}


_PENDING = object()


def _camera_key_begin(lib, vm, dev):
    """First half of _camera_key: -> the key if it is known without the device (a tensor seen before, a host matrix, None for a
    matrix this binding cannot name), or _PENDING with the read launched (gsr_camera_key_begin) -- _camera_key_end then completes
    it; the caller's own host work in between hides the round trip."""
    ent = _CAM_KEYS.get(id(vm))
    if ent is not None and ent[0]() is vm and ent[1] == vm._version and ent[2] == vm.data_ptr():
        CAMERA_KEY_STATS["known_tensor"] += 1
        return ent[3]
    if vm.dim() != 2 or vm.size(0) != 4 or vm.size(1) != 4 or vm.dtype is not _F32:
        return None
    if not vm.is_cuda:
        return _remember(vm, ("host", vm.detach().contiguous().numpy().tobytes()))   # (its bytes are right here)
    if vm.device != dev:
        return None
    _lib.check(lib.gsr_camera_key_begin(vm.data_ptr(), int(vm.stride(0)), int(vm.stride(1))), "gsr_camera_key_begin")
    return _PENDING


def _camera_key_end(lib, vm):
    k = ctypes.c_ulonglong(0)
    _lib.check(lib.gsr_camera_key_end(ctypes.byref(k)), "gsr_camera_key_end")
    CAMERA_KEY_STATS["read"] += 1
    return _remember(vm, int(k.value))


def _remember(vm, key):
    i = id(vm)

    def _gone(ref, i=i):
        e = _CAM_KEYS.get(i)
        if e is not None and e[0] is ref:
            del _CAM_KEYS[i]
    try:
        _CAM_KEYS[i] = (weakref.ref(vm, _gone), vm._version, vm.data_ptr(), key)
    except TypeError:
        pass
    return key


def _camera_key(lib, vm, dev):
    """The camera's identity for its plan: a hash of the view matrix's sixteen floats.  A tensor object seen before, unchanged
    since (same version counter, same storage), is not read again; None = not a matrix this binding can name (no plan)."""
    k = _camera_key_begin(lib, vm, dev)
    return _camera_key_end(lib, vm) if k is _PENDING else k


def rasterize_gaussians_native(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                               cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                               image_width, sh, degree, campos, prefiltered, debug, need_backward=True,
                               scratch_box=None, use_plan=None, plan_key=None):
    """-> (num_rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer), like
    RasterizeGaussiansCUDA (DGR/rasterize_points.cu:35-115), plus two more elements: the longest
    per-tile instance list (informational) and the number of list segments (backward work units)."""
    lib = _lib.load()
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")   # rasterize_points.cu:57-59
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    with _on_device(dev):
        byte_opts = dict(dtype=torch.uint8, device=dev)
        if P == 0:
            # rasterize_points.cu:68-81: zero-filled image, no rasterization at all
            return (0, torch.zeros(_num_channels(colors, background), H, W, dtype=torch.float32, device=dev),
                    torch.zeros(0, dtype=torch.int32, device=dev), torch.empty(0, **byte_opts),
                    torch.empty(0, **byte_opts), torch.empty(0, **byte_opts), 0, 0)
        means3D = _dev_f32(means3D, dev)
        # (the camera's identity is read from the caller's own tensor -- the .contiguous() copy below is a kernel queued on the
        # caller's stream, the caller's tensor is final -- and the read is launched first thing: the allocations below hide it)
        vm_in = viewmatrix
        want_plan = (_PLANNED if use_plan is None else use_plan) and _FUSED and not debug
        cam_key = plan_key if (plan_key is not None or not want_plan) else _camera_key_begin(lib, vm_in, dev)
        background, viewmatrix, projmatrix, campos = (_dev_f32(x, dev) for x in (background, viewmatrix, projmatrix, campos))
        colors, opacity, scales, rotations, cov3D_precomp, sh = (
            _dev_f32(x, dev) for x in (colors, opacity, scales, rotations, cov3D_precomp, sh))
        M = int(sh.size(1)) if sh is not None and sh.numel() != 0 else 0
        C = _num_channels(colors, background)
        out_color = torch.empty(C, H, W, dtype=torch.float32, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        geom = torch.empty(lib.gsr_geom_bytes(P), **byte_opts)
        img = torch.empty(lib.gsr_image_bytes(W, H), **byte_opts)
        R, maxc, nseg, blended = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        st = _stream(dev.index)
        # The binning scratch is sized BEFORE num_rendered is known, from what earlier views on this device needed
        # (x1.25): the library then goes from the stage-1 read-back straight into the stage-2 launches, and the GPU does
        # not idle while Python allocates and re-enters.  First view, or a guess that turns out too small: blended = 0,
        # and stage 2 runs below over an exactly sized buffer (the reference's order of events, rasterize_points.cu:82-112).
        with _HINT_LOCK:
            hint = _BINNING_HINT.get(dev.index, 0) if _FUSED else 0
        binning = torch.empty(hint, **byte_opts)
        # The backward's accumulation table rides along (callers that will run a backward pass a list as scratch_box):
        # the forward blend clears it on the side, and the backward skips its own fill.
        scratch = None
        if need_backward and scratch_box is not None and hint > 0:
            scratch = torch.empty(lib.gsr_grad_scratch_bytes(P), **byte_opts)
        head = (P, int(degree), M, C, int(bool(need_backward)), _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(opacity),
                _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix),
                _ptr(campos), W, H, float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), _ptr(background), _ptr(radii),
                _ptr(geom), _ptr(img), _ptr(binning), hint, _ptr(scratch), _ptr(out_color), ctypes.byref(R),
                ctypes.byref(maxc), ctypes.byref(nseg), ctypes.byref(blended))
        plan = None
        if cam_key is _PENDING:
            cam_key = _camera_key_end(lib, vm_in)
        if want_plan:
            if cam_key is not None:
                # (a camera's differentiable renders and its forward-only ones -- ground-truth / evaluation sweeps, often of
                # another model -- keep separate plans; the field of view is part of the camera)
                key = (dev.index, int(st or 0), W, H, bool(need_backward), float(tan_fovx), float(tan_fovy), cam_key)
                plan = _plan_entry(key, lib, int(lib.gsr_plan_bytes(W, H)), byte_opts,
                                   int(_PLAN_LEVEL_ENV) if _PLAN_LEVEL_ENV is not None else (1 if need_backward else 0))
                if plan.skip > 0:         # (a camera whose views keep outgrowing their plans: left alone for a while)
                    plan.skip -= 1
                    plan = None
        if plan is None:
            _lib.check(lib.gsr_forward_fused(*head, st), "gsr_forward_fused")
        else:
            planned = ctypes.c_int(0)
            info = plan.info
            _lib.check(lib.gsr_forward_planned(*head, _ptr(plan.buf), info, ctypes.byref(planned), st), "gsr_forward_planned")
            if planned.value == 1:
                PLAN_STATS["planned"] += 1
                plan.misfits = 0
            else:
                PLAN_STATS["exact"] += 1
                if planned.value == -1:
                    # the view outgrew its plan (the library has raised the slack level and re-plans).  Misfits in a row -- a
                    # close-up whose workgroups run out of table space misfits under ANY plan -- pause planning for this
                    # camera: 2, 4, .. 64 views
                    PLAN_STATS["misfit"] += 1
                    plan.misfits += 1
                    if _PLAN_DIAG:    # (devtool: -DGSR_PLAN_DIAG build) how the view sat in the plan it outgrew
                        torch.cuda.synchronize()
                        print(f"[plan] misfit grad={need_backward} level={info[4]} tiles over empty buckets {info[18]}, over "
                              f"non-empty {info[19]}, entries over {info[20]}, worst count/capacity {info[21] / 64:.2f}", file=sys.stderr)
                        info[18] = info[19] = info[20] = info[21] = 0
                    if plan.misfits >= 2:
                        plan.skip = min(64, 1 << (plan.misfits - 1))
                if info[0] == -1:     # unplannable (longest list above the in-kernel sort): ask again after _PLAN_RETRY views
                    plan.idle += 1
                    if plan.idle >= _PLAN_RETRY:
                        info[0], plan.idle = 0, 0
        if scratch is not None and blended.value:
            scratch_box.append(scratch)   # cleared by the forward blend: good for exactly one backward
        need = int(lib.gsr_binning_bytes_mt(R.value, nseg.value, C))
        if plan is not None and plan.info[0] == 1:   # room for the plan's capacities, so that the next view of this camera can use it
            need = max(need, int(lib.gsr_binning_bytes_mt(plan.info[1], plan.info[2], C)))
            # (and for the plan this view's forward blend is building for that next view, if its header has landed already:
            # info[8..15] = the header, [16] / [17] = the builder's sequence number arrived / awaited, include/gsr.h)
            if plan.info[17] != 0 and plan.info[16] == plan.info[17] and plan.info[8] == 1 and 0 < plan.info[10] < (1 << 30):
                need = max(need, int(lib.gsr_binning_bytes_mt(plan.info[10], plan.info[11], C)))
        # The hint only moves in steps of 32 MB and only comes down when a view needs less than a QUARTER of it (then it
        # halves): every change of the size is a new block for the caching allocator -- a hipMalloc of a few hundred MB costs
        # tens of milliseconds -- and a hint that shrank by 10 % whenever a view needed less than half of it made a rig of
        # near and far cameras oscillate between shrinking and the exact-size fallback (82 ms iterations in the windowed
        # refinement loop).  288 GB of HBM make a generous buffer the cheap side of this trade.
        if need > hint or need * 4 < hint:
            with _HINT_LOCK:
                cur = _BINNING_HINT.get(dev.index, 0)
                step = 32 << 20
                if need > cur:
                    _BINNING_HINT[dev.index] = (need * 5 // 4 + step - 1) // step * step
                elif need * 4 < cur:
                    _BINNING_HINT[dev.index] = max((need * 5 // 4 + step - 1) // step * step, (cur // 2 + step - 1) // step * step)
        if not blended.value:
            # forward-only renders hand stage 2 the NEGATED segment count: no per-segment snapshots are written (gsr.h)
            nseg2 = nseg.value if need_backward else -nseg.value
            binning = torch.empty(need, **byte_opts)
            _lib.check(lib.gsr_forward_stage2_mt(
                P, R.value, maxc.value, nseg2, C, W, H, _ptr(background), _ptr(colors), _ptr(geom), _ptr(binning),
                _ptr(img), _ptr(out_color), st), "gsr_forward_stage2_mt")
        if debug:
            torch.cuda.synchronize(dev)   # surface asynchronous faults here, like CHECK_CUDA(..., debug)
    return R.value, out_color, radii, geom, binning, img, maxc.value, (nseg.value if need_backward else 0)


def rasterize_gaussians_backward_native(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                        sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug,
                                        num_segments=0, zeroed_scratch=None):
    """-> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
    dL_drotations), like RasterizeGaussiansBackwardCUDA (DGR/rasterize_points.cu:117-196).  One deviation, for every P
    including 0: dL_dcov3D is None unless `cov3D_precomp` was an input (the reference always returns a [P,6] tensor that
    autograd then drops; here the 24 B per Gaussian are neither allocated nor written)."""
    lib = _lib.load()
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if sh is not None and sh.numel() != 0 else 0
    f32 = dict(dtype=torch.float32, device=dev)
    with _on_device(dev):
        if P == 0:
            z = lambda *s: torch.zeros(*s, **f32)
            no_cov = cov3D_precomp is None or cov3D_precomp.numel() == 0
            return (z(0, 3), z(0, int(dL_dout_color.size(0))), z(0, 1), z(0, 3), None if no_cov else z(0, 6), z(0, M, 3), z(0, 3),
                    z(0, 4))
        means3D = _dev_f32(means3D, dev)
        background, viewmatrix, projmatrix, campos = (_dev_f32(x, dev) for x in (background, viewmatrix, projmatrix, campos))
        colors, scales, rotations, cov3D_precomp, sh = (_dev_f32(x, dev) for x in (colors, scales, rotations, cov3D_precomp, sh))
        dL_dout_color = _dev_f32(dL_dout_color, dev)
        has_cov = cov3D_precomp is not None and cov3D_precomp.numel() != 0
        C = _num_channels(colors, background)
        if int(dL_dout_color.size(0)) != C:
            raise RuntimeError(f"dL_dout_color must have {C} channels, got {int(dL_dout_color.size(0))}")
        # torch.empty: the library fills whatever it needs zeroed (rasterize_points.cu:151-159 uses zeros).
        dL_dmeans3D = torch.empty(P, 3, **f32)
        dL_dmeans2D = torch.empty(P, 3, **f32)
        dL_dcolors = torch.empty(P, C, **f32)
        # zeroed_scratch: the table the forward blend cleared on the side (gsr_forward_fused); otherwise the call fills it
        prezeroed = zeroed_scratch is not None and zeroed_scratch.numel() == lib.gsr_grad_scratch_bytes(P)
        grad_scratch = zeroed_scratch if prezeroed else torch.empty(lib.gsr_grad_scratch_bytes(P), dtype=torch.uint8, device=dev)
        dL_dopacity = torch.empty(P, 1, **f32)
        # (only read by autograd when the covariances were an input; the library skips the 24 B/Gaussian otherwise)
        dL_dcov3D = torch.empty(P, 6, **f32) if has_cov else None
        dL_dsh = torch.empty(P, M, 3, **f32)
        dL_dscales = torch.zeros(P, 3, **f32) if has_cov else torch.empty(P, 3, **f32)
        dL_drotations = torch.zeros(P, 4, **f32) if has_cov else torch.empty(P, 4, **f32)
        head = (P, int(degree), M, int(R), int(num_segments))
        tail = (_ptr(background), W, H, _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(scales),
                float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix),
                _ptr(campos), float(tan_fovx), float(tan_fovy), _ptr(radii), _ptr(geomBuffer), _ptr(binningBuffer),
                _ptr(imageBuffer), _ptr(dL_dout_color), _ptr(grad_scratch), _ptr(dL_dmeans2D), _ptr(dL_dopacity),
                _ptr(dL_dcolors), _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales),
                _ptr(dL_drotations))
        _lib.check(lib.gsr_backward_mt(*head, C, *tail, int(prezeroed), _stream(dev.index)), "gsr_backward_mt")
        if debug:
            torch.cuda.synchronize(dev)
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible_native(means3D, viewmatrix, projmatrix):
    """-> bool[P], like markVisible (DGR/rasterize_points.cu:198-217)."""
    lib = _lib.load()
    _require_gpu(means3D, "positions")
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros(P, dtype=torch.bool, device=dev)
    if P:
        with _on_device(dev):
            means3D, viewmatrix, projmatrix = (_dev_f32(x, dev) for x in (means3D, viewmatrix, projmatrix))
            _lib.check(lib.gsr_mark_visible(P, _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix),
                                            present.data_ptr(), _stream(dev.index)), "gsr_mark_visible")
    return present


# --------------------------------------------------------------------------------------------
# The reference's Python layer, unchanged in shape.
# --------------------------------------------------------------------------------------------
# Whether the caller records a graph at all: inside autograd.Function.forward grad mode is always off and
# ctx.needs_input_grad says True for every input that requires grad even under torch.no_grad(), so the mode is noted
# here, at call time.  A render under no_grad() (ground-truth and evaluation sweeps, refined_mesh.py:733-775) can never be
# differentiated: it runs forward-only (no per-unit snapshots or candidate words are written) and keeps a plan of its own.
_CALL = threading.local()


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    _CALL.grad = torch.is_grad_enabled()
    try:
        return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                         cov3Ds_precomp, raster_settings)
    finally:
        _CALL.grad = True


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        # Same argument restructuring as the reference (ref :60-80).
        args = (raster_settings.bg, means3D, colors_precomp, opacities, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy,
                raster_settings.image_height, raster_settings.image_width, sh, raster_settings.sh_degree,
                raster_settings.campos, raster_settings.prefiltered, raster_settings.debug)
        box = []   # receives the backward's accumulation table when the forward cleared it on the side
        need_backward = any(ctx.needs_input_grad) and getattr(_CALL, "grad", True)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)   # copy them before they can be corrupted (ref :83-90)
            try:
                out = rasterize_gaussians_native(*args, need_backward=need_backward, scratch_box=box)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            out = rasterize_gaussians_native(*args, need_backward=need_backward, scratch_box=box)
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, _max_tile, num_segments = out
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.num_segments = num_segments
        ctx.opacity_shape = opacities.shape
        ctx.zeroed_scratch = box[0] if box else None
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # no P-element zero fill for the integer output's "gradient" per backward
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        if grad_out_color is None:   # the colour image did not take part in the loss
            return (None,) * 9
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        zeroed, ctx.zeroed_scratch = ctx.zeroed_scratch, None   # one use: a second backward (retain_graph) fills its own
        args = (raster_settings.bg, means3D, radii, colors_precomp, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy, grad_out_color, sh,
                raster_settings.sh_degree, raster_settings.campos, geomBuffer, num_rendered, binningBuffer,
                imgBuffer, raster_settings.debug)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                grads_native = rasterize_gaussians_backward_native(*args, num_segments=ctx.num_segments, zeroed_scratch=zeroed)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            grads_native = rasterize_gaussians_backward_native(*args, num_segments=ctx.num_segments, zeroed_scratch=zeroed)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = grads_native

        def present(inp, g):   # absent optionals were 0-element placeholders: they get no gradient
            return g if inp is not None and inp.numel() != 0 else None

        # Same order as the reference (ref :143-153).  opacities arrive as [P,1] from both callers
        # (sugar_model.py:1215, gaussian_renderer/__init__.py:49) and the reference returns [P,1].
        return (grad_means3D, grad_means2D, present(sh, grad_sh), present(colors_precomp, grad_colors_precomp),
                grad_opacities.reshape(ctx.opacity_shape), present(scales, grad_scales),
                present(rotations, grad_rotations), present(cov3Ds_precomp, grad_cov3Ds_precomp), None)


_EMPTY = torch.Tensor([])


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = mark_visible_native(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        # (one shared 0-element placeholder instead of the reference's five `torch.Tensor([])` per call: it is never written)
        if shs is None:
            shs = _EMPTY
        if colors_precomp is None:
            colors_precomp = _EMPTY
        if scales is None:
            scales = _EMPTY
        if rotations is None:
            rotations = _EMPTY
        if cov3D_precomp is None:
            cov3D_precomp = _EMPTY

        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, raster_settings)

"""Image-space losses of the GauSTAR refinement loop as fused HIP ops.

Host-side mirror of gaustar_utils/loss_utils.py (`l1_loss`, `ssim`) and of the loss assembly in
gaustar_trainers/refine.py:451-453 (`(1 - f) * l1 + f * (1 - ssim)`, on the margin-cropped view of
:584-594) and :634-660 (masked depth + silhouette L1).  Each op is ONE call into libgsr_hip.so that returns
the loss value; the gradient pass is a second call made from autograd's backward, which hands the kernels the incoming
scalar as a DEVICE pointer, so d loss / d pred leaves them final (no elementwise multiply over the image).  The gradient
w.r.t. the ground-truth image is not provided (the trainer never needs it).

There is no CPU path: CPU tensors raise, like the rasterizer (gaustar_amd/rasterizer.py).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch

from . import _host, _lib


def _stream():
    return _host.raw_stream(torch._C._cuda_getDevice())


def _chw_view(t: torch.Tensor, what: str) -> torch.Tensor:
    """[C,H,W] or [1,C,H,W] float32 HIP tensor, any strides (views are used as they are, never copied)."""
    if not t.is_cuda:
        raise RuntimeError(f"gaustar_amd.losses: {what} must live on a HIP (cuda) device -- there is no CPU path")
    if t.dim() == 4 and t.size(0) == 1:
        t = t[0]
    if t.dim() != 3:
        raise RuntimeError(f"{what} must have dimensions (C, H, W) or (1, C, H, W), got {tuple(t.shape)}")
    if t.dtype != torch.float32:
        t = t.float()
    return t


def _crop(t: torch.Tensor, margin: Optional[Sequence[int]]) -> torch.Tensor:
    """refine.py:587: pred[..., m2:-m3, m0:-m1] with margin = (left, right, top, bottom)."""
    if margin is None:
        return t
    m0, m1, m2, m3 = (int(v) for v in margin)
    return t[..., m2:(-m3 if m3 else None), m0:(-m1 if m1 else None)]


def _scale_ptr(g_loss: torch.Tensor, dev):
    """autograd's incoming d(total)/d(loss) as a float32 device scalar the gradient kernels read (no host sync, and no
    elementwise multiply over the image afterwards); the tensor is returned to keep it alive over the launch."""
    g = g_loss
    if g.dtype != torch.float32 or g.device != dev or g.dim() != 0 or not g.is_contiguous():
        g = g.detach().to(device=dev, dtype=torch.float32).reshape(()).contiguous()
    return g, ctypes.c_void_p(g.data_ptr())


class _L1DSSIM(torch.autograd.Function):
    """Forward: the value pass (gsr_l1_ssim without a gradient buffer).  Backward: gsr_l1_ssim_backward from the workspace the
    value pass left, scaled by the incoming gradient ON THE DEVICE."""

    @staticmethod
    def forward(ctx, pred, gt, dssim_factor, margin):
        lib = _lib.load()
        p_full = _chw_view(pred, "pred")
        g_full = _chw_view(gt, "gt")
        if p_full.shape != g_full.shape:
            raise RuntimeError(f"pred {tuple(p_full.shape)} and gt {tuple(g_full.shape)} differ in shape")
        p, g = _crop(p_full, margin), _crop(g_full, margin)
        C, H, W = (int(v) for v in p.shape)
        if H <= 0 or W <= 0:
            raise RuntimeError("the margin leaves an empty image")
        dev = p.device
        with _host.on_device(dev):
            ws = torch.empty(lib.gsr_l1_ssim_workspace_bytes(C, H, W), dtype=torch.uint8, device=dev)
            out = torch.empty(3, dtype=torch.float32, device=dev)
            _lib.check(lib.gsr_l1_ssim(
                C, H, W, ctypes.c_void_p(p.data_ptr()), p.stride(0), p.stride(1), p.stride(2),
                ctypes.c_void_p(g.data_ptr()), g.stride(0), g.stride(1), g.stride(2), float(dssim_factor),
                ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(out.data_ptr()), None, 0, 0, 0, _stream()), "gsr_l1_ssim")
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(p_full, g_full, ws)
        ctx.cfg = (float(dssim_factor), margin, pred.shape)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)   # no zero-filled "gradient" of the parts vector per backward (one fill kernel)
        return out[0], out

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        if not ctx.saved_tensors or g_loss is None:
            return None, None, None, None
        lib = _lib.load()
        p_full, g_full, ws = ctx.saved_tensors
        f, margin, pred_shape = ctx.cfg
        p, g = _crop(p_full, margin), _crop(g_full, margin)
        C, H, W = (int(v) for v in p.shape)
        dev = p.device
        with _host.on_device(dev):
            # planar [C,H,W] -- the layout the backward blend reads; zero outside the crop
            grad_full = (torch.zeros if margin is not None else torch.empty)(p_full.shape, dtype=torch.float32, device=dev)
            gv = _crop(grad_full, margin)
            keep, sp = _scale_ptr(g_loss, dev)
            _lib.check(lib.gsr_l1_ssim_backward(
                C, H, W, ctypes.c_void_p(p.data_ptr()), p.stride(0), p.stride(1), p.stride(2),
                ctypes.c_void_p(g.data_ptr()), g.stride(0), g.stride(1), g.stride(2), f, ctypes.c_void_p(ws.data_ptr()), sp,
                ctypes.c_void_p(gv.data_ptr()), gv.stride(0), gv.stride(1), gv.stride(2), _stream()), "gsr_l1_ssim_backward")
        return grad_full.reshape(pred_shape), None, None, None


def l1_dssim_loss(pred: torch.Tensor, gt: torch.Tensor, dssim_factor: float = 0.2,
                  margin: Optional[Sequence[int]] = None, return_parts: bool = False):
    """(1 - f) * l1_loss(pred, gt) + f * (1 - ssim(pred, gt)) on pred[..., m2:-m3, m0:-m1] (refine.py:451-453,
    :584-594).  pred/gt: [C,H,W] or [1,C,H,W], any strides.  With return_parts also returns the device vector
    {loss, l1 mean, ssim mean} (no host sync anywhere)."""
    if gt.requires_grad:
        raise NotImplementedError("gaustar_amd.losses: no gradient w.r.t. the ground-truth image")
    loss, parts = _L1DSSIM.apply(pred, gt, float(dssim_factor), None if margin is None else tuple(margin))
    return (loss, parts) if return_parts else loss


def l1_loss(network_output: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """loss_utils.py:17-18."""
    return l1_dssim_loss(network_output, gt, 0.0)


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True) -> torch.Tensor:
    """loss_utils.py:33-43 with the defaults the trainer uses (window 11, mean over everything)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("gaustar_amd.losses.ssim: only window_size=11, size_average=True (the trainer's use)")
    return 1.0 - l1_dssim_loss(img1, img2, 1.0)


class _DepthL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, max_depth, depth_factor, mask_factor):
        lib = _lib.load()
        if not pred.is_cuda:
            raise RuntimeError("gaustar_amd.losses: pred must live on a HIP (cuda) device -- there is no CPU path")
        if pred.dim() != 2 or pred.shape != gt.shape:
            raise RuntimeError(f"pred and gt must both be (H, W), got {tuple(pred.shape)} and {tuple(gt.shape)}")
        p = pred if pred.dtype == torch.float32 else pred.float()
        g = gt if gt.dtype == torch.float32 else gt.float()
        H, W = (int(v) for v in p.shape)
        dev = p.device
        with _host.on_device(dev):
            ws = torch.empty(lib.gsr_depth_l1_workspace_bytes(), dtype=torch.uint8, device=dev)
            out = torch.empty(4, dtype=torch.float32, device=dev)
            _lib.check(lib.gsr_depth_l1(
                H, W, ctypes.c_void_p(p.data_ptr()), p.stride(0), p.stride(1), ctypes.c_void_p(g.data_ptr()),
                g.stride(0), g.stride(1), float(max_depth), float(depth_factor), float(mask_factor),
                ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(out.data_ptr()), None, 0, 0, _stream()), "gsr_depth_l1")
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(p, g, out)
        ctx.cfg = (float(max_depth), float(depth_factor), float(mask_factor), pred.dtype)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)   # no zero-filled "gradient" of the parts vector per backward (one fill kernel)
        return out[0] + out[1], out

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        if not ctx.saved_tensors or g_loss is None:
            return None, None, None, None, None
        lib = _lib.load()
        p, g, out = ctx.saved_tensors
        max_depth, depth_factor, mask_factor, in_dtype = ctx.cfg
        H, W = (int(v) for v in p.shape)
        dev = p.device
        with _host.on_device(dev):
            grad = torch.empty(H, W, dtype=torch.float32, device=dev)
            keep, sp = _scale_ptr(g_loss, dev)
            _lib.check(lib.gsr_depth_l1_backward(
                H, W, ctypes.c_void_p(p.data_ptr()), p.stride(0), p.stride(1), ctypes.c_void_p(g.data_ptr()), g.stride(0),
                g.stride(1), max_depth, depth_factor, mask_factor, ctypes.c_void_p(out.data_ptr()), sp,
                ctypes.c_void_p(grad.data_ptr()), W, 1, _stream()), "gsr_depth_l1_backward")
        return grad.to(in_dtype), None, None, None, None


def depth_mask_l1_loss(pred_depth: torch.Tensor, gt_depth: torch.Tensor, max_depth: float, depth_factor: float = 1.0,
                       mask_factor: float = 1.0, return_parts: bool = False):
    """depth_factor * |pred - gt|.mean over {gt < max_depth} + mask_factor * |pred - max_depth|.mean over
    {gt > max_depth} (refine.py:634-660, depth_alpha = False).  parts = {depth term, mask term, #fg, #bg}."""
    loss, parts = _DepthL1.apply(pred_depth, gt_depth, float(max_depth), float(depth_factor), float(mask_factor))
    return (loss, parts) if return_parts else loss


class _RGBDepthLoss(torch.autograd.Function):
    """l1 + dssim on channels 0-2 and masked depth L1 on channel 3 of ONE [6,H,W] / [4,H,W] render.  Forward: both value
    passes and ONE reduction kernel (gsr_rgb_depth_loss) -> the total as a device scalar.  Backward: the two gradient passes
    write d(total)/d(render), scaled by the incoming gradient on the device, into ONE tensor (channels 4-5 zero)."""

    @staticmethod
    def forward(ctx, img6, gt_rgb, gt_depth, dssim_factor, margin, max_depth, depth_factor, mask_factor):
        lib = _lib.load()
        if not img6.is_cuda:
            raise RuntimeError("gaustar_amd.losses: the render must live on a HIP (cuda) device -- there is no CPU path")
        if img6.dim() != 3 or img6.size(0) not in (4, 6):
            raise RuntimeError(f"the two-target render must have dimensions (6, H, W) or (4, H, W), got {tuple(img6.shape)}")
        x = img6 if img6.dtype == torch.float32 else img6.float()
        g_full = _chw_view(gt_rgb, "gt_rgb")
        if tuple(g_full.shape) != (3,) + tuple(x.shape[1:]):
            raise RuntimeError(f"gt_rgb {tuple(g_full.shape)} does not match the render {tuple(x.shape)}")
        if gt_depth.dim() != 2 or tuple(gt_depth.shape) != tuple(x.shape[1:]):
            raise RuntimeError(f"gt_depth must be (H, W) = {tuple(x.shape[1:])}, got {tuple(gt_depth.shape)}")
        gd = gt_depth if gt_depth.dtype == torch.float32 else gt_depth.float()
        p, g = _crop(x[:3], margin), _crop(g_full, margin)
        C, H, W = (int(v) for v in p.shape)
        if H <= 0 or W <= 0:
            raise RuntimeError("the margin leaves an empty image")
        d = x[3]
        Hd, Wd = (int(v) for v in d.shape)
        dev = x.device
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        with _host.on_device(dev):
            ws = torch.empty(lib.gsr_l1_ssim_workspace_bytes(C, H, W), dtype=torch.uint8, device=dev)
            wd = torch.empty(lib.gsr_depth_l1_workspace_bytes(), dtype=torch.uint8, device=dev)
            out = torch.empty(8, dtype=torch.float32, device=dev)   # {loss, l1, ssim | depth term, mask term, #fg, #bg | total}
            _lib.check(lib.gsr_rgb_depth_loss(
                C, H, W, vp(p), p.stride(0), p.stride(1), p.stride(2), vp(g), g.stride(0), g.stride(1), g.stride(2),
                float(dssim_factor), vp(ws), Hd, Wd, vp(d), d.stride(0), d.stride(1), vp(gd), gd.stride(0), gd.stride(1),
                float(max_depth), float(depth_factor), float(mask_factor), vp(wd), vp(out), _stream()), "gsr_rgb_depth_loss")
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(x, g_full, gd, ws)
        ctx.cfg = (float(dssim_factor), margin, float(max_depth), float(depth_factor), float(mask_factor), img6.dtype)
        # (what backward reads -- the pixel counts the depth gradient divides by -- is the second copy of the eight numbers the
        # reduction kernel leaves in the workspace's tail (include/gsr.h), not the vector handed to the caller: `out` goes out
        # as it is, without the 32-byte device copy that used to separate the two)
        parts = out[:7]
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)   # no zero-filled "gradient" of the parts vector per backward (one fill kernel)
        return out[7], parts

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        if not ctx.saved_tensors or g_loss is None:
            return (None,) * 8
        lib = _lib.load()
        x, g_full, gd, ws = ctx.saved_tensors
        f, margin, max_depth, depth_factor, mask_factor, in_dtype = ctx.cfg
        p, g = _crop(x[:3], margin), _crop(g_full, margin)
        C, H, W = (int(v) for v in p.shape)
        d = x[3]
        Hd, Wd = (int(v) for v in d.shape)
        dev = x.device
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        with _host.on_device(dev):
            grad6 = torch.empty_like(x, memory_format=torch.contiguous_format)
            if x.size(0) > 4:
                grad6[4:].zero_()
            if margin is not None:
                grad6[:3].zero_()
            gv, gdv = _crop(grad6[:3], margin), grad6[3]
            keep, sp = _scale_ptr(g_loss, dev)
            _lib.check(lib.gsr_rgb_depth_loss_backward(
                C, H, W, vp(p), p.stride(0), p.stride(1), p.stride(2), vp(g), g.stride(0), g.stride(1), g.stride(2), f, vp(ws),
                Hd, Wd, vp(d), d.stride(0), d.stride(1), vp(gd), gd.stride(0), gd.stride(1), max_depth, depth_factor, mask_factor,
                ctypes.c_void_p(ws.data_ptr() + ws.numel() - 256), sp, vp(gv), gv.stride(0), gv.stride(1), gv.stride(2), vp(gdv),
                gdv.stride(0), gdv.stride(1), _stream()),
                "gsr_rgb_depth_loss_backward")
        return grad6.to(in_dtype), None, None, None, None, None, None, None


def rgb_depth_loss(render6: torch.Tensor, gt_rgb: torch.Tensor, gt_depth: torch.Tensor, max_depth: float,
                   dssim_factor: float = 0.2, depth_factor: float = 1.0, mask_factor: float = 1.0,
                   margin: Optional[Sequence[int]] = None, return_parts: bool = False):
    """The image losses of one refinement iteration on the ONE-pass render ([6,H,W] or [4,H,W]: channels 0-2 RGB, channel 3
    depth-as-colour):
    l1_dssim_loss(render6[:3], gt_rgb, dssim_factor, margin) + depth_mask_l1_loss(render6[3], gt_depth, max_depth,
    depth_factor, mask_factor) (refine.py:451-453, :584-594, :634-660).  Same two kernels as the separate functions; what
    it saves is autograd's handling of the two slices (two zero-filled [6,H,W] tensors, two slice copies and their sum):
    both kernels write into one gradient tensor.  parts = {l1+dssim loss, l1 mean, ssim mean, depth term, mask term,
    #fg, #bg}."""
    if gt_rgb.requires_grad or gt_depth.requires_grad:
        raise NotImplementedError("gaustar_amd.losses: no gradient w.r.t. the ground truth")
    loss, parts = _RGBDepthLoss.apply(render6, gt_rgb, gt_depth, float(dssim_factor),
                                      None if margin is None else tuple(margin), float(max_depth), float(depth_factor),
                                      float(mask_factor))
    return (loss, parts) if return_parts else loss

"""CPU restatement of the image-space losses (TEST INFRASTRUCTURE ONLY: imported by tests/ -- never by gaustar_amd/).

Follows gaustar_utils/loss_utils.py:17-62 (l1_loss, gaussian, create_window, _ssim) and the loss assembly of
gaustar_trainers/refine.py:451-453, :584-594 (margin crop) and :634-660 (masked depth / silhouette L1), in plain
PyTorch on the CPU; gradients come from autograd.  Pinned by tests/golden/loss_*.npz, which
tests/golden/make_loss_golden.py produced by importing the reference's own loss_utils.py."""
from math import exp

import torch
import torch.nn.functional as F


def gaussian_window(window_size=11, sigma=1.5):                      # loss_utils.py:23-31
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g = g / g.sum()
    w1 = g.unsqueeze(1)
    return w1.mm(w1.t()).float()


def ssim_map(img1, img2, window_size=11):                            # loss_utils.py:45-57
    c = img1.size(-3)
    w = gaussian_window(window_size).to(img1.device, img1.dtype).expand(c, 1, window_size, window_size).contiguous()
    pad = window_size // 2
    mu1 = F.conv2d(img1, w, padding=pad, groups=c)
    mu2 = F.conv2d(img2, w, padding=pad, groups=c)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=pad, groups=c) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=pad, groups=c) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=c) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))


def l1_dssim(pred, gt, dssim_factor=0.2, margin=None):
    """pred, gt: [1,C,H,W] (or [C,H,W]); returns (loss, l1, ssim) as 0-dim tensors."""
    if pred.dim() == 3:
        pred, gt = pred[None], gt[None]
    if margin is not None:                                           # refine.py:584-588
        m = margin
        sl = (..., slice(m[2], -m[3] if m[3] else None), slice(m[0], -m[1] if m[1] else None))
        pred, gt = pred[sl], gt[sl]
    l1 = torch.abs(pred - gt).mean()                                 # loss_utils.py:17-18
    s = ssim_map(pred, gt).mean()
    return (1.0 - dssim_factor) * l1 + dssim_factor * (1.0 - s), l1, s   # refine.py:453


def depth_mask_l1(pred_depth, gt_depth, max_depth, depth_factor, mask_factor):
    """refine.py:634-660 with depth_alpha = False; returns (depth term, mask term)."""
    fg = gt_depth < max_depth
    depth_loss = depth_factor * (pred_depth[fg] - gt_depth[fg]).abs().mean()
    bg = gt_depth > max_depth
    mask_loss = mask_factor * (pred_depth[bg] - max_depth).abs().mean()
    return depth_loss, mask_loss

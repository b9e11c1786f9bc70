"""Generates tests/golden/loss_kat.npz by IMPORTING the reference's own gaustar_utils/loss_utils.py
(runs only in the build container, where /root/reference exists; torch CPU).  Inputs + expected outputs only.

Cases: full-frame and margin-cropped l1 + dssim (refine.py:451-453, :584-594) on seeded images of awkward sizes,
with d loss / d pred from autograd; plain l1 and plain ssim; the masked depth / silhouette L1 of refine.py:634-660
(that block is inline trainer code, restated here line by line on top of torch)."""
import os, sys
import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from gaustar_utils.loss_utils import l1_loss, ssim   # noqa: E402

torch.manual_seed(0)
out = {}


def smooth_image(c, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(1, c, h // 4 + 2, w // 4 + 2, generator=g)
    img = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False)
    return (img + 0.05 * torch.randn(1, c, h, w, generator=g)).clamp(0, 1)


cases = [("a", 3, 45, 70, None, 0.2), ("b", 3, 64, 96, (3, 5, 2, 7), 0.2), ("c", 1, 33, 31, None, 0.5),
         ("d", 3, 100, 37, (1, 1, 1, 1), 0.2), ("e", 3, 20, 24, None, 1.0), ("f", 3, 20, 24, None, 0.0)]
for name, c, h, w, margin, f in cases:
    pred = smooth_image(c, h, w, ord(name)).requires_grad_(True)
    gt = (smooth_image(c, h, w, ord(name)) * 0.7 + 0.3 * smooth_image(c, h, w, 7 + ord(name))).detach()
    if name == "d":
        gt[..., 10:30, 5:20] = pred.detach()[..., 10:30, 5:20]     # exact zeros of (pred - gt): sign(0) = 0
    p, g = pred, gt
    if margin is not None:                                          # refine.py:584-588
        p = pred[..., margin[2]:-margin[3], margin[0]:-margin[1]]
        g = gt[..., margin[2]:-margin[3], margin[0]:-margin[1]]
    l1 = l1_loss(p, g)
    s = ssim(p, g)
    loss = (1.0 - f) * l1 + f * (1.0 - s)                           # refine.py:453
    loss.backward()
    out[f"{name}_pred"] = pred.detach().numpy()[0]
    out[f"{name}_gt"] = gt.numpy()[0]
    out[f"{name}_margin"] = np.array(margin if margin is not None else [-1, -1, -1, -1], np.int32)
    out[f"{name}_f"] = np.float32(f)
    out[f"{name}_loss"] = np.float32(loss.item())
    out[f"{name}_l1"] = np.float32(l1.item())
    out[f"{name}_ssim"] = np.float32(s.item())
    out[f"{name}_grad"] = pred.grad.numpy()[0]

# masked depth + silhouette L1, refine.py:634-660 (depth_alpha = False)
max_depth, depth_factor, mask_factor = 10.0, 1.0, 0.5
g = torch.Generator().manual_seed(5)
gt_depth = 2.0 + 3.0 * torch.rand(60, 83, generator=g)
gt_depth[torch.rand(60, 83, generator=g) < 0.4] = 15.0             # background
gt_depth[0, :5] = max_depth                                         # exactly max_depth: in neither set
pred_depth = (gt_depth.clamp(max=max_depth) + 0.3 * torch.randn(60, 83, generator=g)).requires_grad_(True)
fg_mask = (gt_depth < max_depth)
depth_loss = depth_factor * (pred_depth[fg_mask] - gt_depth[fg_mask]).abs().mean()
bg_mask = (gt_depth > max_depth)
mask_loss = mask_factor * (pred_depth[bg_mask] - max_depth).abs().mean()
(depth_loss + mask_loss).backward()
out.update(depth_pred=pred_depth.detach().numpy(), depth_gt=gt_depth.numpy(), depth_max=np.float32(max_depth),
           depth_factor=np.float32(depth_factor), mask_factor=np.float32(mask_factor),
           depth_loss=np.float32(depth_loss.item()), mask_loss=np.float32(mask_loss.item()),
           depth_grad=pred_depth.grad.numpy())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "loss_kat.npz"), **out)
print("wrote loss_kat.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim > 1})

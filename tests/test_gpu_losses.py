"""GPU: fused image-space losses (gsr_l1_ssim, gsr_depth_l1) against the reference's own loss_utils.py vectors
(tests/golden/loss_kat.npz) and, at the trainer's full size and memory layouts, against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
KAT = os.path.join(ROOT, "tests", "golden", "loss_kat.npz")
LOSS_TOL = 2e-6          # absolute, losses are O(0.1): fp32 window sums in a different association than conv2d
GRAD_TOL = 2e-4          # relative to the gradient's largest entry (the values are ~1/N)


def _margin(z, name):
    m = z[f"{name}_margin"]
    return None if m[0] < 0 else tuple(int(v) for v in m)


def _check_grad(a, b, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert err <= GRAD_TOL, f"{what}: normalised max error {err:.3e}"


@pytest.mark.parametrize("name", list("abcdef"))
def test_l1_dssim_matches_reference_vectors(name, hip_lib):
    from gaustar_amd import losses
    z = np.load(KAT)
    pred = torch.from_numpy(z[f"{name}_pred"]).cuda().requires_grad_(True)
    gt = torch.from_numpy(z[f"{name}_gt"]).cuda()
    loss, parts = losses.l1_dssim_loss(pred, gt, float(z[f"{name}_f"]), _margin(z, name), return_parts=True)
    (3.0 * loss).backward()          # a non-unit upstream gradient
    parts = parts.cpu().numpy()
    assert abs(loss.item() - float(z[f"{name}_loss"])) <= LOSS_TOL
    assert abs(parts[1] - float(z[f"{name}_l1"])) <= LOSS_TOL and abs(parts[2] - float(z[f"{name}_ssim"])) <= LOSS_TOL
    _check_grad(pred.grad.cpu().numpy() / 3.0, z[f"{name}_grad"], name)


def test_l1_and_ssim_wrappers(hip_lib):
    from gaustar_amd import losses
    z = np.load(KAT)
    pred, gt = torch.from_numpy(z["a_pred"]).cuda(), torch.from_numpy(z["a_gt"]).cuda()
    assert abs(losses.l1_loss(pred, gt).item() - float(z["a_l1"])) <= LOSS_TOL
    assert abs(losses.ssim(pred[None], gt[None]).item() - float(z["a_ssim"])) <= LOSS_TOL
    with pytest.raises(NotImplementedError):
        losses.ssim(pred, gt, window_size=7)
    with pytest.raises(NotImplementedError):
        losses.l1_dssim_loss(pred, gt.clone().requires_grad_(True))


def test_depth_mask_l1_matches_reference_vectors(hip_lib):
    from gaustar_amd import losses
    z = np.load(KAT)
    pred = torch.from_numpy(z["depth_pred"]).cuda().requires_grad_(True)
    loss, parts = losses.depth_mask_l1_loss(pred, torch.from_numpy(z["depth_gt"]).cuda(), float(z["depth_max"]),
                                            float(z["depth_factor"]), float(z["mask_factor"]), return_parts=True)
    loss.backward()
    parts = parts.cpu().numpy()
    assert abs(parts[0] - float(z["depth_loss"])) <= 1e-6 and abs(parts[1] - float(z["mask_loss"])) <= 1e-6
    assert parts[2] == (z["depth_gt"] < z["depth_max"]).sum() and parts[3] == (z["depth_gt"] > z["depth_max"]).sum()
    np.testing.assert_allclose(pred.grad.cpu().numpy(), z["depth_grad"], rtol=1e-6, atol=0)


def test_full_size_trainer_layouts_against_oracle(hip_lib):
    """1080p, the memory layouts of refine.py:575-590: pred = the rasterizer's planar [3,H,W] seen through
    [H,W,3] and back (a view), gt = genuinely [H,W,3] storage transposed to [1,3,H,W], ActorsHQ-style margin."""
    from gaustar_amd import losses
    from oracle import loss_oracle
    H, W, margin = 1080, 1920, (24, 40, 16, 8)
    g = torch.Generator().manual_seed(3)
    small = torch.rand(1, 3, H // 8, W // 8, generator=g)
    base = torch.nn.functional.interpolate(small, size=(H, W), mode="bicubic", align_corners=False).clamp(0, 1)[0]
    pred_c = (base + 0.03 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    gt_hwc = (base + 0.03 * torch.randn(3, H, W, generator=g)).clamp(0, 1).permute(1, 2, 0).contiguous()

    raster_out = pred_c.cuda().requires_grad_(True)                                  # [3,H,W] like the rasterizer's
    pred = raster_out.transpose(0, 1).transpose(1, 2)                                # sugar_model.py:1298 -> [H,W,3]
    pred = pred.view(-1, H, W, 3).transpose(-1, -2).transpose(-2, -3)                # refine.py:575 -> [1,3,H,W] view
    gt = gt_hwc.cuda().view(-1, H, W, 3).transpose(-1, -2).transpose(-2, -3)         # refine.py:579-580
    assert not gt.is_contiguous()
    loss = losses.l1_dssim_loss(pred, gt, 0.2, margin)
    loss.backward()

    p_ref = pred_c.clone().requires_grad_(True)
    ref, _, _ = loss_oracle.l1_dssim(p_ref[None], gt_hwc.permute(2, 0, 1)[None], 0.2, margin)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= LOSS_TOL
    got = raster_out.grad.cpu().numpy()
    _check_grad(got, p_ref.grad.numpy(), "1080p l1+dssim")
    m = margin
    outside = np.ones((H, W), bool); outside[m[2]:H - m[3], m[0]:W - m[1]] = False
    assert (got[:, outside] == 0).all()

    # depth target: channel 0 of a [3,H,W] render (refine.py:616 `[..., 0]` of the [H,W,3] view)
    depth_img = (4.0 + torch.rand(3, H, W, generator=g)).cuda().requires_grad_(True)
    gt_d = 4.0 + torch.rand(H, W, generator=g); gt_d[torch.rand(H, W, generator=g) < 0.3] = 20.0
    pd = depth_img.transpose(0, 1).transpose(1, 2)[..., 0]
    dl = losses.depth_mask_l1_loss(pd, gt_d.cuda(), 10.0, 1.0, 0.7)
    dl.backward()
    p2 = depth_img.detach().cpu()[0].clone().requires_grad_(True)
    a, b = loss_oracle.depth_mask_l1(p2, gt_d, 10.0, 1.0, 0.7)
    (a + b).backward()
    assert abs(dl.item() - (a + b).item()) <= 1e-5
    np.testing.assert_allclose(depth_img.grad.cpu().numpy()[0], p2.grad.numpy(), rtol=1e-5, atol=0)
    assert (depth_img.grad.cpu().numpy()[1:] == 0).all()


@pytest.mark.parametrize("margin", [None, (3, 5, 2, 4)])
def test_rgb_depth_loss_equals_the_two_separate_losses(margin, hip_lib):
    """rgb_depth_loss on the [6,H,W] render = l1_dssim_loss on channels 0-2 + depth_mask_l1_loss on channel 3 (both pinned
    above), value and gradient; channels 4-5 receive zero gradient."""
    from gaustar_amd import losses
    g = torch.Generator(device="cuda").manual_seed(8)
    H, W = 70, 101
    img = torch.rand(6, H, W, device="cuda", generator=g)
    img[3:] = img[3:] * 12.0
    gt_rgb = torch.rand(1, 3, H, W, device="cuda", generator=g)
    gt_d = torch.rand(H, W, device="cuda", generator=g) * 14.0
    a = img.clone().requires_grad_(True)
    b = img.clone().requires_grad_(True)
    la, parts = losses.rgb_depth_loss(a, gt_rgb, gt_d, 10.0, 0.2, 0.7, 0.3, margin=margin, return_parts=True)
    lb = losses.l1_dssim_loss(b[:3], gt_rgb, 0.2, margin=margin) + losses.depth_mask_l1_loss(b[3], gt_d, 10.0, 0.7, 0.3)
    (2.0 * la).backward()
    (2.0 * lb).backward()
    assert abs(float(la) - float(lb)) < 1e-6 and parts.numel() == 7
    assert torch.equal(a.grad[4:], torch.zeros_like(a.grad[4:]))
    assert torch.allclose(a.grad, b.grad, rtol=0, atol=1e-9) and float(a.grad[:4].abs().max()) > 0


def test_gradient_pass_scales_on_the_device_and_can_run_twice(hip_lib):
    """The gradient kernels run in autograd's backward with the incoming gradient as a DEVICE scalar (ABI 12:
    gsr_l1_ssim_backward / gsr_depth_l1_backward): any scale -- also one that is itself a function of other device values --
    gives scale x the unit gradient, and the saved workspace survives a second backward over the same graph."""
    from gaustar_amd import losses
    g = torch.Generator(device="cuda").manual_seed(12)
    H, W = 64, 96
    img = torch.rand(4, H, W, device="cuda", generator=g)
    img[3] = img[3] * 12.0
    gt_rgb = torch.rand(3, H, W, device="cuda", generator=g)
    gt_d = torch.rand(H, W, device="cuda", generator=g) * 14.0
    a = img.clone().requires_grad_(True)
    la = losses.rgb_depth_loss(a, gt_rgb, gt_d, 10.0, 0.2, 0.7, 0.3, margin=(2, 3, 1, 4))
    la.backward(retain_graph=True)
    unit = a.grad.clone(); a.grad = None
    w = torch.tensor(3.0, device="cuda", requires_grad=True)
    (la * w * w).backward()                     # d/dloss = 9, delivered as a device scalar computed by other kernels
    assert torch.allclose(a.grad, 9.0 * unit, rtol=1e-5, atol=1e-9) and abs(float(w.grad) - 6.0 * float(la)) < 1e-5
    for fn in (lambda t: losses.l1_dssim_loss(t[:3], gt_rgb, 0.3), lambda t: losses.depth_mask_l1_loss(t[3], gt_d, 10.0, 0.7, 0.3)):
        b = img.clone().requires_grad_(True)
        fn(b).backward()
        u = b.grad.clone(); b.grad = None
        (fn(b) * -0.25).backward()
        assert torch.allclose(b.grad, -0.25 * u, rtol=1e-5, atol=1e-9)   # (the scale enters before the two terms are subtracted)

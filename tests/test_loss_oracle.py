"""CPU: the loss oracle (oracle/loss_oracle.py) against vectors produced by the reference's own
gaustar_utils/loss_utils.py (tests/golden/loss_kat.npz, made by tests/golden/make_loss_golden.py)."""
import os

import numpy as np
import torch

from conftest import ROOT

KAT = os.path.join(ROOT, "tests", "golden", "loss_kat.npz")
CASES = "abcdef"


def _margin(z, name):
    m = z[f"{name}_margin"]
    return None if m[0] < 0 else tuple(int(v) for v in m)


def test_l1_dssim_oracle_matches_reference_vectors():
    from oracle import loss_oracle
    z = np.load(KAT)
    for name in CASES:
        pred = torch.from_numpy(z[f"{name}_pred"]).requires_grad_(True)
        gt = torch.from_numpy(z[f"{name}_gt"])
        loss, l1, s = loss_oracle.l1_dssim(pred, gt, float(z[f"{name}_f"]), _margin(z, name))
        loss.backward()
        assert abs(loss.item() - float(z[f"{name}_loss"])) <= 1e-6, name
        assert abs(l1.item() - float(z[f"{name}_l1"])) <= 1e-6 and abs(s.item() - float(z[f"{name}_ssim"])) <= 1e-6
        np.testing.assert_allclose(pred.grad.numpy(), z[f"{name}_grad"], rtol=1e-5, atol=1e-9, err_msg=name)


def test_depth_oracle_matches_reference_vectors():
    from oracle import loss_oracle
    z = np.load(KAT)
    pred = torch.from_numpy(z["depth_pred"]).requires_grad_(True)
    d, m = loss_oracle.depth_mask_l1(pred, torch.from_numpy(z["depth_gt"]), float(z["depth_max"]), float(z["depth_factor"]),
                                     float(z["mask_factor"]))
    (d + m).backward()
    assert abs(d.item() - float(z["depth_loss"])) <= 1e-6 and abs(m.item() - float(z["mask_loss"])) <= 1e-6
    np.testing.assert_allclose(pred.grad.numpy(), z["depth_grad"], rtol=1e-6, atol=0)


def test_losses_module_has_no_cpu_path():
    import pytest
    from gaustar_amd import losses
    a = torch.rand(3, 16, 16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        losses.l1_dssim_loss(a, a.clone())
    with pytest.raises(RuntimeError, match="no CPU path"):
        losses.depth_mask_l1_loss(a[0], a[1], 10.0)


def test_points_rgb_oracle_matches_reference_vectors():
    """oracle/producers_oracle.py against vectors made with the reference's own eval_sh (producers_kat.npz)."""
    from oracle import producers_oracle
    z = np.load(os.path.join(ROOT, "tests", "golden", "producers_kat.npz"))
    for lv in (1, 2, 3, 4, 5):
        k = f"l{lv}"
        pos = torch.from_numpy(z[f"{k}_pos"]).requires_grad_(True)
        sh = torch.from_numpy(z[f"{k}_sh"]).requires_grad_(True)
        col = producers_oracle.points_rgb(pos, torch.from_numpy(z[f"{k}_cam"]), sh, lv)
        col.backward(torch.from_numpy(z[f"{k}_dL"]))
        np.testing.assert_allclose(col.detach().numpy(), z[f"{k}_colors"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(sh.grad.numpy(), z[f"{k}_dsh"], rtol=1e-6, atol=1e-7)
        dpos = pos.grad.numpy() if pos.grad is not None else np.zeros_like(z[f"{k}_dpos"])
        np.testing.assert_allclose(dpos, z[f"{k}_dpos"], rtol=1e-5, atol=1e-6)

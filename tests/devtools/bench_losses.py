"""Dev tool (uses the oracle's torch restatement as the comparison -> lives under tests/): time of the loss
assembly of refine.py:451-453 + :634-660 at 1080p on the GPU, (a) as the reference computes it (PyTorch
composite: 5 depthwise conv2d + elementwise kernels + autograd backward), (b) gaustar_amd.losses (fused HIP)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaustar_amd import losses
from oracle import loss_oracle

dev = torch.device("cuda:0")
H, W = 1080, 1920
g = torch.Generator(device=dev).manual_seed(0)
pred = torch.rand(3, H, W, device=dev, generator=g).requires_grad_(True)
gt = torch.rand(H, W, 3, device=dev, generator=g).view(-1, H, W, 3).transpose(-1, -2).transpose(-2, -3)
pd = (4 + torch.rand(H, W, device=dev, generator=g)).requires_grad_(True)
gd = 4 + torch.rand(H, W, device=dev, generator=g); gd[torch.rand(H, W, device=dev, generator=g) < 0.3] = 20.0
margin = (24, 40, 16, 8)

def composite():
    pred.grad = None; pd.grad = None
    l, _, _ = loss_oracle.l1_dssim(pred[None], gt, 0.2, margin)
    a, b = loss_oracle.depth_mask_l1(pd, gd, 10.0, 1.0, 0.7)
    (l + a + b).backward()

def fused():
    pred.grad = None; pd.grad = None
    l = losses.l1_dssim_loss(pred, gt, 0.2, margin) + losses.depth_mask_l1_loss(pd, gd, 10.0, 1.0, 0.7)
    l.backward()

def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

res = {"pytorch_composite_ms": round(timed(composite), 4), "fused_hip_ms": round(timed(fused), 4)}
res["speedup"] = round(res["pytorch_composite_ms"] / res["fused_hip_ms"], 2)
print(json.dumps(res))

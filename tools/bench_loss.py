"""tools/bench_loss.py -- the image losses of one refinement iteration (rgb_depth_loss forward + backward) on a random 1080p render:
ms per forward + backward pair by stream events.  GSR_LIB_PATH selects a variant build (-DGSR_SSIM_TILE_H=.., -DGSR_SSIM_HS=..)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaustar_amd import losses
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
H, W = 1080, 1920
img = torch.rand(4, H, W, device=dev, generator=g)
img[3] *= 5.0
gt = torch.rand(3, H, W, device=dev, generator=g)
gd = torch.rand(H, W, device=dev, generator=g) * 5.0
gd[gd > 4.0] = 20.0
one = torch.ones((), device=dev)
def step():
    x = img.detach().requires_grad_(True)
    loss = losses.rgb_depth_loss(x, gt, gd, 10.0, 0.2, 1.0, 0.5)
    loss.backward(one)
    return loss, x.grad
for _ in range(10): step()
torch.cuda.synchronize()
res = []
for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): l, gr = step()
    b.record(); torch.cuda.synchronize()
    res.append(a.elapsed_time(b) / 50)
print(os.environ.get("GSR_LIB_PATH", "product"), "ms per forward+backward:", [round(r, 4) for r in res], "loss", float(l), "grad sum", float(gr.double().abs().sum()))

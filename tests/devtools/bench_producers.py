"""Dev tool (compares against the oracle's torch restatement -> lives under tests/): per-iteration cost of the
rasterizer-input producers at config-C size on the GPU, PyTorch composite (what the reference runs, twice per
iteration: sugar_model.py:417-508 + :674-718) vs gaustar_amd.producers (fused HIP), forward + backward."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaustar_amd import producers, scene
from oracle import producers_oracle as po

dev = torch.device("cuda:0")
v, f = scene.icosphere(6, radius=0.9, center=(0.0, 1.2, 0.0))
g = torch.Generator().manual_seed(0)
verts = torch.from_numpy(v).float().to(dev).requires_grad_(True)
faces = torch.from_numpy(f).long().to(dev)
N = faces.shape[0] * 6
bary = torch.tensor([[2/3, 1/6, 1/6], [1/6, 2/3, 1/6], [1/6, 1/6, 2/3], [1/6, 5/12, 5/12], [5/12, 1/6, 5/12], [5/12, 5/12, 1/6]], device=dev)
rs = (torch.randn(N, 2, generator=g) * 0.4 - 4).to(dev).requires_grad_(True)
rc = torch.randn(N, 2, generator=g).to(dev).requires_grad_(True)
dt = (0.01 * torch.randn(N, 3, generator=g)).to(dev).requires_grad_(True)
dr = (torch.randn(N, 4, generator=g) * 0.3 + torch.tensor([1.0, 0, 0, 0])).to(dev).requires_grad_(True)
sh = (torch.rand(N, 16, 3, generator=g) - 0.5).to(dev).requires_grad_(True)
cam = torch.tensor([[0.0, 1.2, -3.0]], device=dev)
w = [torch.randn(N, k, generator=g).to(dev) for k in (3, 3, 4, 3)]
params = [verts, rs, rc, dt, dr, sh]

def run(mesh_fn, rgb_fn):
    for p in params: p.grad = None
    p_, s_, q_ = mesh_fn(verts, faces, bary, rs, rc, 3e-6, 0.001, 0.1, dt, dr)
    c_ = rgb_fn(p_, cam, sh, 4)
    ((p_ * w[0]).sum() + (s_ * w[1]).sum() + (q_ * w[2]).sum() + (c_ * w[3]).sum()).backward()

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

res = {"gaussians": N, "pytorch_composite_ms": round(timed(lambda: run(po.mesh_bound_gaussians, po.points_rgb)), 4),
       "fused_hip_ms": round(timed(lambda: run(producers.mesh_bound_gaussians, producers.points_rgb)), 4)}
res["speedup"] = round(res["pytorch_composite_ms"] / res["fused_hip_ms"], 2)
print(json.dumps(res))

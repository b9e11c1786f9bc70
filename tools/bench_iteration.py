"""tools/bench_iteration.py [--steps K] [--level L] -- one GauSTAR-style refinement iteration built entirely from this
package's fused ops (config C geometry by default):

  mesh-bound producers (points / scaling / quaternions) -> SH colours -> ONE 6-channel render (RGB + depth-as-colour)
  -> l1 + dssim on RGB + masked depth L1 -> backward through all of it -> Adam step.

Ground truth = a render of a perturbed copy of the parameters (SURVEY.md 8d config E's synthetic-GT recipe).
Prints ms per iteration and the loss at the first / last step."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, losses, optim, producers, scene

BARY6 = [[2/3, 1/6, 1/6], [1/6, 2/3, 1/6], [1/6, 1/6, 2/3], [1/6, 5/12, 5/12], [5/12, 1/6, 5/12], [5/12, 5/12, 1/6]]
MAX_DEPTH = 10.0
CH = 4 if "--six" not in sys.argv else 6   # RGB + one depth channel (default) or + three identical ones
_MEANS2D = None


def build(level, dev, seed=0):
    v, f = scene.icosphere(level, radius=0.9, center=(0.0, 1.2, 0.0))
    g = torch.Generator().manual_seed(seed)
    verts = torch.from_numpy(v).float()
    faces = torch.from_numpy(f).long()
    N = faces.shape[0] * 6
    edge = (verts[faces[:, 0]] - verts[faces[:, 1]]).norm(dim=-1).mean().item()
    p = dict(verts=verts, raw_scales=torch.full((N, 2), float(np.log(edge / (4 + 2 * np.sqrt(3))))),
             raw_complex=torch.tensor([1.0, 0.0]).repeat(N, 1), densities=torch.full((N, 1), 2.5),
             sh=torch.cat([torch.rand(N, 1, 3, generator=g) * 2 - 1, 0.1 * (torch.rand(N, 15, 3, generator=g) - 0.5)], 1))
    return {k: t.to(dev) for k, t in p.items()}, faces.to(dev), torch.tensor(BARY6, device=dev), edge


def render(p, faces, bary, cam_t, dev):
    view, proj, campos, bg6, H, W, tx, ty = cam_t
    pts, scl, quat = producers.mesh_bound_gaussians(p["verts"], faces, bary, p["raw_scales"], p["raw_complex"], 3e-6)
    colors6 = producers.points_rgb_depth(pts, campos, p["sh"], 4, view, depth_channels=CH - 3)   # rgb + depth-as-colour (refine.py:603-605)
    s = GaussianRasterizationSettings(H, W, tx, ty, bg6, 1.0, view, proj, 0, campos, False, False)
    global _MEANS2D
    if _MEANS2D is None or _MEANS2D.shape != pts.shape:   # never read by the rasterizer (it only carries a gradient in the
        _MEANS2D = torch.zeros_like(pts)                   # reference's densifier): one tensor for the whole run
    img, _ = GaussianRasterizer(s)(means3D=pts, means2D=_MEANS2D, opacities=torch.sigmoid(p["densities"]),
                                   colors_precomp=colors6, scales=scl, rotations=quat)
    return img


def run(a):
    dev = torch.device("cuda:0")
    params, faces, bary, edge = build(a.level, dev)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
    cams = scene.ring_cameras(5, 32, a.width, a.height, focal_px=1200.0 * a.width / 1920.0)
    cams = [cams[i % len(cams)] for i in range(a.steps + a.warmup)]
    bg6 = torch.tensor([0.0, 1.0, 0.0] + [MAX_DEPTH] * (CH - 3), device=dev)
    cam_ts = [(t(c.viewmatrix), t(c.projmatrix), t(c.campos), bg6, c.H, c.W, c.tanfovx, c.tanfovy) for c in cams]
    # ground truth: the same surface, slightly deformed and recoloured
    with torch.no_grad():
        g = torch.Generator(device=dev).manual_seed(1)
        gt_p = {k: v.clone() for k, v in params.items()}
        gt_p["verts"] += 0.3 * edge * torch.randn(gt_p["verts"].shape, device=dev, generator=g)
        gt_p["sh"][:, 0] += 0.3 * torch.randn(gt_p["sh"][:, 0].shape, device=dev, generator=g)
        gts = []
        for ct in cam_ts:
            img = render(gt_p, faces, bary, ct, dev)
            gt_depth = img[3].clone()
            gt_depth[gt_depth >= MAX_DEPTH - 1e-3] = 2 * MAX_DEPTH     # real captures carry "far" values behind the subject
            gts.append((img[:3].permute(1, 2, 0).contiguous().view(-1, ct[4], ct[5], 3).transpose(-1, -2).transpose(-2, -3),
                        gt_depth))
    for v in params.values():
        v.requires_grad_(True)
    groups = [{"params": [params["verts"]], "lr": 2e-4}, {"params": [params["sh"]], "lr": 5e-3},
              {"params": [params["raw_scales"], params["raw_complex"], params["densities"]], "lr": 5e-3}]
    opt = torch.optim.Adam(groups, fused=True) if getattr(a, "torch_adam", False) else optim.Adam(groups)
    hist = []

    def step(i):
        ct, (gt_rgb, gt_depth) = cam_ts[i], gts[i]
        opt.zero_grad(set_to_none=True)
        img = render(params, faces, bary, ct, dev)
        loss = losses.rgb_depth_loss(img, gt_rgb, gt_depth, MAX_DEPTH, 0.2, 1.0, 0.5)   # both image losses, one gradient tensor
        loss.backward()
        opt.step()
        hist.append(loss.detach())

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(a.warmup, a.warmup + a.steps):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    return {"gaussians": int(faces.shape[0] * 6), "image": [a.width, a.height], "ms_per_iteration": round(ms, 3),
            "iterations_per_s": round(1e3 / ms, 1), "loss_first": round(float(hist[0]), 5),
            "loss_last": round(float(torch.stack(hist[-5:]).mean()), 5)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40); ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--level", type=int, default=6); ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--six", action="store_true", help="6-channel render (depth in three identical channels) instead of 4")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of gaustar_amd.optim.Adam")
    print(json.dumps(run(ap.parse_args())))


if __name__ == "__main__":
    main()

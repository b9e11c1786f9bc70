"""tools/window_phases.py -- where the HOST spends an iteration of tools/bench_window.py's loop (config-C size): wall time of
each phase's Python call, no synchronisation added (the forward's own wait for the instance count is inside `render`)."""
import argparse, gc, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_window
from gaustar_amd import losses, optim

acc = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        acc.setdefault(name, []).append(time.perf_counter() - t)
        return r
    return w
bench_window.render4 = timed("render4 (producers + rasterizer forward, incl. its wait)", bench_window.render4)
bench_window.losses.rgb_depth_loss = timed("loss forward", losses.rgb_depth_loss)
optim.Adam.step = timed("Adam.step", optim.Adam.step)
optim.Adam.zero_grad = timed("zero_grad", optim.Adam.zero_grad)
from gaustar_amd import producers as _pr, rasterizer as _rz, harness as _hs
_pr.mesh_bound_gaussians = timed("  in render4: mesh_bound_gaussians", _pr.mesh_bound_gaussians)
_pr.points_rgb_depth = timed("  in render4: points_rgb_depth", _pr.points_rgb_depth)
_rz.rasterize_gaussians_native = timed("  in render4: rasterize_gaussians_native (C ABI forward, incl. wait)", _rz.rasterize_gaussians_native)
_rz.GaussianRasterizer.forward = timed("  in render4: GaussianRasterizer.forward (all of it)", _rz.GaussianRasterizer.forward)
_hs.SurfaceGaussians._settings = timed("  in render4: _settings", _hs.SurfaceGaussians._settings)
torch.cat = timed("  torch.cat", torch.cat)
torch.sigmoid = timed("  torch.sigmoid", torch.sigmoid)
_rz.rasterize_gaussians_backward_native = timed("  in backward: rasterize_gaussians_backward_native", _rz.rasterize_gaussians_backward_native)
_hs._rasterizer.rasterize_gaussians_backward_native = _rz.rasterize_gaussians_backward_native
_pr._sh_backward_raw = timed("  in backward: _sh_backward_raw", _pr._sh_backward_raw)
_pr._mesh_backward_raw = timed("  in backward: _mesh_backward_raw", _pr._mesh_backward_raw)
_pr._mesh_forward_raw = timed("  in render4: _mesh_forward_raw", _pr._mesh_forward_raw)
_pr._sh_forward_raw = timed("  in render4: _sh_forward_raw", _pr._sh_forward_raw)
_hs._RenderMeshBound.backward = staticmethod(timed("  in backward: _RenderMeshBound.backward (all of it)", _hs._RenderMeshBound.backward))
losses._RGBDepthLoss.backward = staticmethod(timed("  in backward: _RGBDepthLoss.backward", losses._RGBDepthLoss.backward))
_hs._RenderMeshBound.forward = staticmethod(timed("  in render4: _RenderMeshBound.forward (all of it)", _hs._RenderMeshBound.forward))
_bw = torch.Tensor.backward
torch.Tensor.backward = timed("backward (autograd engine, all backward launches)", _bw)
TINY = os.environ.get("WINDOW_PHASES_SIZE") == "tiny"   # tiny: negligible GPU work -- what is timed is the host alone
r = bench_window.run(argparse.Namespace(frames=2, iters=100, level=2 if TINY else 6, width=160 if TINY else 1920, height=96 if TINY else 1080, cameras=160))
print({k: r[k] for k in ("median_ms_per_iteration", "host_wait_ms_per_iteration")})
tot = 0.0
for k, v in acc.items():
    v = np.array(v[-100:]) * 1e3
    tot += float(np.median(v))
    print(f"{k:75s} median {np.median(v):.3f} ms  p90 {np.percentile(v, 90):.3f}")
print("sum of medians", round(tot, 3))
if TINY:
    raise SystemExit(0)

# ---- the forward's parts, one at a time (same model; each call timed on the host, GPU idle in between) ----
from gaustar_amd import GaussianRasterizer, harness, producers, scene
dev = torch.device("cuda:0")
v, f = scene.icosphere(6, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
model = harness.SurfaceGaussians(torch.from_numpy(v).float().to(dev), torch.from_numpy(f).long().to(dev), 6, 3).to(dev)
cams = scene.ring_cameras(5, 32, 1920, 1080, focal_px=1200.0)[:8]
ncams = [harness.nerf_camera_from_scene(c) for c in cams]
bg4 = torch.tensor([0.0, 1.0, 0.0, 10.0], device=dev)
parts = {}
def t(name, fn):
    t0 = time.perf_counter(); r = fn(); parts.setdefault(name, []).append(time.perf_counter() - t0); return r
gc.collect(); gc.disable()
for it in range(60):
    nc = ncams[it % 8]
    with torch.no_grad():
        for p in model.parameters():
            torch.autograd.graph.increment_version(p)
    settings, view, campos = t("_settings", lambda: model._settings(nc, bg4, 0))
    pts = t("points (mesh producer)", lambda: model.points)
    sh = t("sh_coordinates (cat)", lambda: model.sh_coordinates)
    col = t("points_rgb_depth", lambda: producers.points_rgb_depth(pts, campos, sh, model.sh_levels, view, depth_channels=1))
    op = t("strengths (sigmoid)", lambda: model.strengths)
    sc = t("scaling", lambda: model.scaling)
    qu = t("quaternions", lambda: model.quaternions)
    m2 = t("zeros_like", lambda: torch.zeros_like(pts))
    rast = t("GaussianRasterizer()", lambda: GaussianRasterizer(settings))
    img = t("rasterizer forward (incl. wait)", lambda: rast(means3D=pts, means2D=m2, opacities=op, colors_precomp=col, scales=sc, rotations=qu)[0])
    torch.cuda.synchronize()
gc.enable()
for k, vv in parts.items():
    vv = np.array(vv[10:]) * 1e3
    print(f"{k:40s} median {np.median(vv):.4f} ms")

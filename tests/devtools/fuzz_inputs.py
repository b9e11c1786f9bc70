"""Dev tool: hostile inputs (NaN / Inf / huge / tiny / degenerate) through forward + backward; every call must return
(no hang, no fault) -- run under `timeout`.  Values are not checked: the reference's behaviour on such inputs is garbage too."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer, scene

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
cam = scene.look_at_camera((0, 0, -4.0), (0, 0, 0), 200, 120, fovx=0.9, znear=0.01)
t = lambda x, g=False: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev).requires_grad_(g)

def run(gs, tag, bg=(0, 0, 0)):
    s = GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, t(np.array(bg)), 1.0, t(cam.viewmatrix), t(cam.projmatrix),
                                      0, t(cam.campos), False, False)
    m, o, c, sc, ro = t(gs.means3D, True), t(gs.opacities, True), t(gs.colors_precomp, True), t(gs.scales, True), t(gs.rotations, True)
    img, radii = GaussianRasterizer(s)(means3D=m, means2D=torch.zeros(gs.P, 3, device=dev, requires_grad=True), opacities=o,
                                       colors_precomp=c, scales=sc, rotations=ro)
    img.backward(torch.randn_like(img))
    torch.cuda.synchronize()
    print(f"{tag:28s} ok  visible {int((radii > 0).sum()):6d}  finite image {bool(torch.isfinite(img).all())}")

base = lambda n=3000: scene.random_gaussians(n, rng, scale_range=(0.01, 0.2))
g = base(); g.means3D[::7] = np.nan; run(g, "NaN means")
g = base(); g.means3D[::5] = np.inf; g.means3D[1::5] = -np.inf; run(g, "Inf means")
g = base(); g.scales[::3] = np.nan; run(g, "NaN scales")
g = base(); g.scales[::3] = 1e20; run(g, "huge scales")
g = base(); g.scales[:] = 1e-30; run(g, "denormal scales")
g = base(); g.scales[::2] = 0.0; run(g, "zero scales")
g = base(); g.scales[::2] = -0.1; run(g, "negative scales")
g = base(); g.rotations[::2] = 0.0; run(g, "zero quaternions")
g = base(); g.rotations[::3] = np.nan; run(g, "NaN quaternions")
g = base(); g.opacities[::2] = np.nan; g.opacities[1::4] = 5.0; g.opacities[3::4] = -1.0; run(g, "bad opacities")
g = base(); g.colors_precomp[::2] = np.inf; run(g, "Inf colours")
g = base(); g.means3D[:] = 0.0; g.scales[:] = 3.0; run(g, "all coincident, screen-filling")
g = base(20000); g.means3D[:, :2] *= 0.01; g.opacities[:] = 0.005; run(g, "20k stacked translucent")
g = base(1); run(g, "single Gaussian")
g = base(); g.means3D[:, 2] = -3.9; run(g, "everything at the near plane")
g = base(); g.means3D *= 1e6; run(g, "far away")
print("FUZZ_OK")

"""tools/conic_bits.py [view] -- how many of a view's per-Gaussian {mean2D, conic, opacity} differ in their BITS between this
library's preprocess and the reference build's (oracle/_ref): the inputs of every alpha >= 1/255 decision."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaustar_amd import _lib, scene
from gaustar_amd import rasterizer as R
from oracle import ref
cam_i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gs, cams, bg = scene.config_C()
cam = cams[cam_i]
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
lib = _lib.load()
e = torch.Tensor([])
m3, op, cols, sc, rot = t(gs.means3D), t(gs.opacities), t(gs.colors_precomp), t(gs.scales), t(gs.rotations)
out = R.rasterize_gaussians_native(t(bg), m3, cols, op, sc, rot, 1.0, e, t(cam.viewmatrix), t(cam.projmatrix), cam.tanfovx, cam.tanfovy, cam.H, cam.W, e, 0, t(cam.campos), False, False, use_plan=False)
Rn, _, radii, geom, binning, img, maxc, U = out
P, W, H = gs.P, cam.W, cam.H
T = ((W + 15) // 16) * ((H + 15) // 16)
m2_ = torch.zeros(P, 2, device=dev); co_ = torch.zeros(P, 4, device=dev); dep_ = torch.zeros(P, device=dev)
rng_t = torch.zeros(T, 2, dtype=torch.int32, device=dev); pl_t = torch.zeros(max(Rn, 1), dtype=torch.int32, device=dev)
fT_ = torch.zeros(H, W, device=dev); nc_ = torch.zeros(H, W, dtype=torch.int32, device=dev)
pp = lambda x: ctypes.c_void_p(x.data_ptr())
_lib.check(lib.gsr_debug_export(P, Rn, 1, W, H, pp(geom), pp(binning), pp(img), pp(m2_), pp(co_), pp(dep_), None, pp(rng_t), pp(pl_t), pp(fT_), pp(nc_), None), "export")
torch.cuda.synchronize()
rr = ref.RefRasterizer()
color, radii_r, Rr = rr.forward(gs.means3D, gs.opacities, cam.viewmatrix, cam.projmatrix, cam.campos, W, H, cam.tanfovx, cam.tanfovy, bg,
                                colors_precomp=gs.colors_precomp, scales=gs.scales, rotations=gs.rotations)
st = rr.state()
vis = radii.cpu().numpy() > 0
for name, a, b in (("means2D", m2_.cpu().numpy(), st["means2D"]), ("conic_opacity", co_.cpu().numpy(), st["conic_opacity"]),
                   ("depth", dep_.cpu().numpy(), st["depths"])):
    a, b = a[vis].reshape(vis.sum(), -1), b[vis].reshape(vis.sum(), -1)
    d = a.view(np.uint32) != b.view(np.uint32)
    rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
    print(f"{name}: {int(vis.sum())} visible; entries with different bits per column {d.sum(0).tolist()}; max relative difference {rel.max(0).tolist()}")
print("final_T differing bits:", int((fT_.cpu().numpy().view(np.uint32) != st["final_T"].view(np.uint32)).sum()), "n_contrib differing:", "n/a (list positions differ: tiles dropped)")

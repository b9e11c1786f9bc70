"""Dev tool: find where the HIP path and the reference build disagree on one view of config C."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import parity  # noqa: E402
from gaustar_amd import _lib, scene  # noqa: E402
from gaustar_amd import rasterizer as R  # noqa: E402
from oracle import ref  # noqa: E402

cam_i = int(sys.argv[1]) if len(sys.argv) > 1 else 37
gs, cams, bg = scene.config_C()
cam = cams[cam_i]
kw = dict(means3D=gs.means3D, opacities=gs.opacities, view=cam.viewmatrix, proj=cam.projmatrix, campos=cam.campos,
          W=cam.W, H=cam.H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, shs=None,
          colors_precomp=gs.colors_precomp, scales=gs.scales, rotations=gs.rotations, cov3D_precomp=None, sh_degree=0,
          scale_modifier=1.0)
rr = ref.RefRasterizer()
color, radii, Rr = rr.forward(**kw)
st = rr.state()
color = color.cpu().numpy()
radii = radii.cpu().numpy()

lib = _lib.load()
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
e = torch.Tensor([])
P, W, H = gs.P, cam.W, cam.H
out = R.rasterize_gaussians_native(t(bg), t(gs.means3D), t(gs.colors_precomp), t(gs.opacities), t(gs.scales),
                                   t(gs.rotations), 1.0, e, t(cam.viewmatrix), t(cam.projmatrix), cam.tanfovx,
                                   cam.tanfovy, H, W, e, 0, t(cam.campos), False, False, use_plan=False)
Rn, c2, r2, geom, binning, img, maxc, nseg = out
print("R ref", Rr, "R ours", Rn, "max tile", maxc, "radii mismatch", (r2.cpu().numpy() != radii).sum())
T = ((W + 15) // 16) * ((H + 15) // 16)
gx = (W + 15) // 16
rng_ = torch.zeros(T, 2, dtype=torch.int32, device=dev)
pl = torch.zeros(max(Rn, 1), dtype=torch.int32, device=dev)
fT = torch.zeros(H, W, device=dev)
nc = torch.zeros(H, W, dtype=torch.int32, device=dev)
co = torch.zeros(P, 4, device=dev)
m2 = torch.zeros(P, 2, device=dev)
p = lambda x: ctypes.c_void_p(x.data_ptr())
lib.gsr_debug_export(P, Rn, nseg, W, H, p(geom), p(binning), p(img), p(m2), p(co), None, None, p(rng_), p(pl), p(fT), p(nc), None)
torch.cuda.synchronize()
c2 = c2.cpu().numpy()
err = np.abs(c2 - color).max(0)
print("n pixels > 1e-4:", (err > 1e-4).sum(), "max", err.max())
ys, xs = np.nonzero(err > 1e-4)
tiles = sorted(set((y // 16) * gx + (x // 16) for y, x in zip(ys, xs)))
print("bad tiles", tiles[:20])
ours_r, ours_l = rng_.cpu().numpy().astype(np.int64), pl.cpu().numpy().astype(np.int64)
ref_r, ref_l = st["ranges"].astype(np.int64), st["point_list"].astype(np.int64)
m2n, con = m2.cpu().numpy(), co.cpu().numpy()
for tile in tiles[:4]:
    a = ours_l[ours_r[tile, 0]:ours_r[tile, 1]]
    b = ref_l[ref_r[tile, 0]:ref_r[tile, 1]]
    print("tile", tile, "(", tile % gx, tile // gx, ") ours n", len(a), "ref n", len(b))
    d = st["depths"]
    print("  ours sorted by depth?", (np.diff(d[a]) >= 0).all(), " ref sorted?", (np.diff(d[b]) >= 0).all())
    sb = set(b.tolist())
    print("  ours not in ref:", [x for x in a.tolist() if x not in sb][:10])
    it = iter(b.tolist())
    print("  ordered subset:", all(x in it for x in a.tolist()))
    ty, tx = tile // gx, tile % gx
    sub = err[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    yy, xx = np.unravel_index(sub.argmax(), sub.shape)
    py, px = ty * 16 + yy, tx * 16 + xx
    print("  worst pixel", px, py, "ours", c2[:, py, px], "ref", color[:, py, px], "T ours", fT[py, px].item(),
          "T ref", st["final_T"][py, px], "ncontrib ours", nc[py, px].item(), "ref", st["n_contrib"][py, px])
    # replay the reference list at this pixel and report contributing gaussians and whether ours has them
    sa = set(a.tolist())
    Tt = 1.0
    for k, g in enumerate(b.tolist()):
        x0, y0 = st["means2D"][g]
        cA, cB, cC, op = st["conic_opacity"][g]
        dx, dy = x0 - px, y0 - py
        power = -0.5 * (cA * dx * dx + cC * dy * dy) - cB * dx * dy
        if power > 0:
            continue
        alpha = min(0.99, op * np.exp(power))
        if alpha < 1 / 255:
            continue
        if Tt * (1 - alpha) < 1e-4:
            break
        if g not in sa:
            print(f"    MISSING in ours: g={g} k={k} alpha={alpha:.4f} T={Tt:.4f} d=({dx:.2f},{dy:.2f}) radius={radii[g]} "
                  f"ours m2={m2n[g]} conic={con[g]} ref m2=({x0:.3f},{y0:.3f}) conic=({cA:.4f},{cB:.4f},{cC:.4f}) op={op:.4f}")
        Tt *= (1 - alpha)

# tie analysis
for tile in tiles[:3]:
    a = ours_l[ours_r[tile, 0]:ours_r[tile, 1]]
    b = ref_l[ref_r[tile, 0]:ref_r[tile, 1]]
    d = st["depths"]
    for nm, l in (("ours", a), ("ref", b)):
        dd = d[l]
        ties = np.nonzero(np.diff(dd) == 0)[0]
        bad = [(int(l[i]), int(l[i + 1])) for i in ties if l[i] > l[i + 1]]
        print(f"tile {tile} {nm}: {len(ties)} adjacent ties, {len(bad)} with descending index, e.g. {bad[:5]}")
    # first position where the common subsequence order differs
    sa = set(a.tolist())
    bf = [x for x in b.tolist() if x in sa]
    diff = [i for i, (x, y) in enumerate(zip(a.tolist(), bf)) if x != y]
    if diff:
        i = diff[0]
        print("  first diff at", i, "ours", a[i:i + 4], "ref", bf[i:i + 4], "ref depths", d[a[i:i + 4]], d[np.array(bf[i:i + 4])])
        kb = st["keys"][ref_r[tile, 0]:ref_r[tile, 1]]
        pos = [b.tolist().index(x) for x in bf[i:i + 4]]
        print("  ref keys", [hex(int(kb[p_])) for p_ in pos])

dp = torch.zeros(P, device=dev)
lib.gsr_debug_export(P, Rn, nseg, W, H, p(geom), p(binning), p(img), None, None, p(dp), None, None, None, None, None, None)
torch.cuda.synchronize()
mine = dp.cpu().numpy()
vis = radii > 0
nb = (mine[vis].view(np.uint32) != st["depths"][vis].view(np.uint32)).sum()
print("depth bit mismatches among visible:", nb, "of", vis.sum())
for g in (295914, 295918, 474408, 474410):
    print(g, hex(mine[g:g + 1].view(np.uint32)[0]), hex(st["depths"][g:g + 1].view(np.uint32)[0]), gs.means3D[g])

"""Dev tool (GPU): the per-pixel CANDIDATE words the forward blend leaves behind (gsr_mask.h / gsr_blend_fwd.hip; read by the
backward's units) against a numpy evaluation of the reference's pair test (forward.cu:330-345): for every list position
the forward reached, the set bits of a pixel's words must be a SUPERSET of the instances with power <= 0 and
alpha >= 1/255 at that pixel; the tool also reports how tight the superset is.

usage: python tests/devtools/check_masks.py [smoke|A|C] [view]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaustar_amd import _lib, scene
from gaustar_amd import rasterizer as R


def build(which, view):
    rng = np.random.default_rng(0)
    if which == "C":
        gs, cams, bg = scene.config_C()
        return gs, cams[view], bg
    if which == "A":
        return scene.config_A()
    v, f = scene.icosphere(3, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
    gs = scene.mesh_bound_gaussians(v, f, rng, 3.5e-6)
    cam = scene.look_at_camera((0.5, 1.6, 3.0), scene.SUBJECT_CENTER, 325, 243, focal_px=260.0)
    return gs, cam, np.array([0.0, 1.0, 0.0], np.float32)


def render(gs, cam, bg, need_backward):
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
    e = torch.Tensor([])
    sh_mode = getattr(gs, "shs", None) is not None
    out = R.rasterize_gaussians_native(t(bg), t(gs.means3D), e if sh_mode else t(gs.colors_precomp), t(gs.opacities), t(gs.scales),
                                       t(gs.rotations), 1.0, e, t(cam.viewmatrix), t(cam.projmatrix), cam.tanfovx,
                                       cam.tanfovy, cam.H, cam.W, t(gs.shs) if sh_mode else e, int(gs.sh_degree) if sh_mode else 0,
                                       t(cam.campos), False, False, need_backward=need_backward,
                                       use_plan=False)   # (compact unit numbering: U is recomputed from the ranges below)
    Rn, color, radii, geom, binning, img, maxc, _ = out
    lib = _lib.load()
    P, W, H = len(gs.means3D), cam.W, cam.H
    T = ((W + 15) // 16) * ((H + 15) // 16)
    # number of units from the tile ranges
    m2 = torch.zeros(P, 2, device=dev); co = torch.zeros(P, 4, device=dev)
    rng_ = torch.zeros(T, 2, dtype=torch.int32, device=dev); pl = torch.zeros(max(Rn, 1), dtype=torch.int32, device=dev)
    fT = torch.zeros(H, W, device=dev); nc = torch.zeros(H, W, dtype=torch.int32, device=dev)
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    torch.cuda.synchronize()
    # U is needed to carve the buffer: recompute it from the ranges after a first export with a dummy value is not
    # possible (the point list sits in front of the per-unit areas, so any U works for it)
    _lib.check(lib.gsr_debug_export(P, Rn, 1, W, H, p(geom), p(binning), p(img), p(m2), p(co), None, None, p(rng_), p(pl),
                                    p(fT), p(nc), None), "export")
    torch.cuda.synchronize()
    ranges = rng_.cpu().numpy().astype(np.int64)
    U = int(((ranges[:, 1] - ranges[:, 0] + 63) // 64).sum())
    masks = torch.zeros(U, 4, 64, 2, dtype=torch.int32, device=dev)
    _lib.check(lib.gsr_debug_export_masks(Rn, U, p(binning), p(masks), None), "export masks")
    torch.cuda.synchronize()
    return dict(color=color.cpu().numpy(), ranges=ranges, list=pl.cpu().numpy().astype(np.int64), m2=m2.cpu().numpy(),
                co=co.cpu().numpy(), nc=nc.cpu().numpy(), masks=masks.cpu().numpy().view(np.uint32), U=U, W=W, H=H)


def unpack(words, reversed_bits):
    """[..., 2] uint32 -> [..., 64] bool over list positions"""
    bits = ((words[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)   # [..., 2, 32]
    if reversed_bits:
        bits = bits[..., ::-1]
    return bits.reshape(*words.shape[:-1], 64)


def verify(tr, tiles=None):
    """-> (pairs passing the reference's alpha test, candidate bits, passing pairs WITHOUT a candidate bit) over `tiles`
    (default: every non-empty tile), restricted to the list positions the forward certainly reached."""
    W, H = tr["W"], tr["H"]
    gx = (W + 15) // 16
    ranges, lst, m2, co = tr["ranges"], tr["list"], tr["m2"], tr["co"]
    cand = unpack(tr["masks"], False)     # [U, 4, 64 lanes, 64 positions]
    yy, xx = np.arange(256) // 16, np.arange(256) % 16
    blk = (yy // 8) * 2 + xx // 8
    lane = (yy % 8) * 8 + xx % 8
    if tiles is None:
        tiles = np.nonzero(ranges[:, 1] > ranges[:, 0])[0]
    unit0 = np.concatenate([[0], np.cumsum((ranges[:, 1] - ranges[:, 0] + 63) // 64)])
    n_ok = n_cand = n_missing = 0
    for t in tiles:
        a, b = ranges[t]
        ids = lst[a:b]; n = len(ids)
        ty, tx = divmod(int(t), gx)
        px = (tx * 16 + xx).astype(np.float32); py = (ty * 16 + yy).astype(np.float32)
        inside = (px < W) & (py < H)
        nu = (n + 63) // 64
        ncv = tr["nc"][ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
        reached = min(n, (int(ncv.max()) + 63) // 64 * 64)   # the forward certainly parked the units up to its deepest contributor
        ids = ids[:reached]
        dx = m2[ids, 0:1] - px[None]; dy = m2[ids, 1:2] - py[None]
        power = -0.5 * (co[ids, 0:1] * dx * dx + co[ids, 2:3] * dy * dy) - co[ids, 1:2] * dx * dy
        alpha = np.minimum(0.99, co[ids, 3:4] * np.exp(power))
        ok = (power <= 0) & (alpha >= 1 / 255) & inside[None]
        cm = cand[unit0[t]:unit0[t] + nu][:, blk, lane, :].transpose(0, 2, 1).reshape(nu * 64, 256)[:reached]
        n_ok += int(ok.sum()); n_cand += int((cm & inside[None]).sum()); n_missing += int((ok & ~cm).sum())
    return n_ok, n_cand, n_missing


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "smoke"
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    gs, cam, bg = build(which, view)
    tr = render(gs, cam, bg, need_backward=True)
    tiles = np.nonzero(tr["ranges"][:, 1] > tr["ranges"][:, 0])[0]
    if len(tiles) > 400:
        tiles = np.random.default_rng(1).choice(tiles, 400, replace=False)
    n_ok, n_cand, n_missing = verify(tr, tiles)
    print(f"{which}: tiles {len(tiles)}  pairs passing the alpha test {n_ok}  candidate bits {n_cand} "
          f"({n_cand / max(n_ok, 1):.3f} per passing pair)  passing pairs WITHOUT a candidate bit: {n_missing}")
    assert n_missing == 0


if __name__ == "__main__":
    main()

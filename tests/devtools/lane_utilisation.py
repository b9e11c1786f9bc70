"""Dev tool (uses the oracle -> lives under tests/): how many (pixel, Gaussian) pairs of a config-C view actually
blend, versus the lane slots an 8x8-block-granular walk evaluates.  Numbers quoted in DESIGN.md section 6."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle
from gaustar_amd import scene

gs, cams, bg = scene.config_C()
cam = cams[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
st = oracle.forward(gs.means3D, gs.opacities, cam.viewmatrix, cam.projmatrix, cam.campos, cam.W, cam.H, cam.tanfovx,
                    cam.tanfovy, bg, colors_precomp=gs.colors_precomp, scales=gs.scales, rotations=gs.rotations)
W, H = cam.W, cam.H
gx = (W + 15) // 16
rng = np.random.default_rng(0)
ranges = st["ranges"].astype(np.int64)
nonempty = np.nonzero(ranges[:, 1] > ranges[:, 0])[0]
sample = rng.choice(nonempty, size=min(300, len(nonempty)), replace=False)
tot = dict(list=0, blended=0, alpha_ok=0, block_pairs_any_alpha=0, block_pairs_live=0, pixels=0, nlast=0, block_hi=0)
for t in sample:
    a, b = ranges[t]
    ids = st["point_list"][a:b]
    xy = st["means2D"][ids]; co = st["conic_opacity"][ids]
    ty, tx = divmod(int(t), gx)
    px = (tx * 16 + np.arange(16))[None, :].repeat(16, 0).reshape(-1).astype(np.float32)
    py = (ty * 16 + np.arange(16))[:, None].repeat(16, 1).reshape(-1).astype(np.float32)
    inside = (px < W) & (py < H)
    dx = xy[:, 0:1] - px[None]; dy = xy[:, 1:2] - py[None]
    power = -0.5 * (co[:, 0:1] * dx * dx + co[:, 2:3] * dy * dy) - co[:, 1:2] * dx * dy
    alpha = np.minimum(0.99, co[:, 3:4] * np.exp(power))
    ok = (power <= 0) & (alpha >= 1 / 255) & inside[None]
    n = len(ids)
    T = np.ones(256, np.float32); done = ~inside.copy(); blended = np.zeros((n, 256), bool)
    for k in range(n):
        tt = T * (1 - alpha[k])
        live = ok[k] & ~done
        stop = live & (tt < 1e-4)
        upd = live & ~stop
        done |= stop
        T = np.where(upd, tt, T)
        blended[k] = upd
        if done.all():
            break
    tot["list"] += n
    tot["blended"] += int(blended.sum())
    tot["alpha_ok"] += int(ok.sum())
    tot["pixels"] += int(inside.sum())
    last = np.where(blended.any(0), n - 1 - np.argmax(blended[::-1], 0), -1)
    tot["nlast"] += int((last + 1).sum())
    blk = ((np.arange(256) // 16) // 8) * 2 + ((np.arange(256) % 16) // 8)
    for q in range(4):
        m = blk == q
        tot["block_pairs_any_alpha"] += int(ok[:, m].any(1).sum())
        tot["block_pairs_live"] += int(blended[:, m].any(1).sum())
        tot["block_hi"] += int(last[m].max() + 1)
nt = len(sample)
print(f"tiles sampled {nt}; mean list {tot['list']/nt:.1f}")
print(f"blended pairs / pixel          {tot['blended']/tot['pixels']:.2f}")
print(f"alpha-ok pairs / pixel (no termination) {tot['alpha_ok']/tot['pixels']:.2f}")
print(f"mean last-contributor position / pixel {tot['nlast']/tot['pixels']:.1f}")
print(f"8x8 blocks: mean deepest position {tot['block_hi']/(4*nt):.1f}; pairs with any alpha-ok px {tot['block_pairs_any_alpha']/(4*nt):.1f}; "
      f"pairs with a blended px {tot['block_pairs_live']/(4*nt):.1f}")
print(f"lane utilisation of blended block pairs: {tot['blended']/(64*tot['block_pairs_live']):.3f}")

# ---- what a 4x4-quadrant-granular queue would iterate over, per 8x8 block and per 64-position segment
it_block = it_quad = it_quad_sum = 0
for t in sample:
    a, b = ranges[t]
    ids = st["point_list"][a:b]
    xy = st["means2D"][ids]; co = st["conic_opacity"][ids]
    ty, tx = divmod(int(t), gx)
    px = (tx * 16 + np.arange(16))[None, :].repeat(16, 0).reshape(-1).astype(np.float32)
    py = (ty * 16 + np.arange(16))[:, None].repeat(16, 1).reshape(-1).astype(np.float32)
    dx = xy[:, 0:1] - px[None]; dy = xy[:, 1:2] - py[None]
    power = -0.5 * (co[:, 0:1] * dx * dx + co[:, 2:3] * dy * dy) - co[:, 1:2] * dx * dy
    ok = (power <= 0) & (np.minimum(0.99, co[:, 3:4] * np.exp(power)) >= 1 / 255)
    yy, xx = np.arange(256) // 16, np.arange(256) % 16
    nc = st["n_contrib"][ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    ncf = np.zeros((16, 16), np.int64); ncf[:nc.shape[0], :nc.shape[1]] = nc
    ncf = ncf.reshape(-1)
    for blk in range(4):
        mb = ((yy // 8) * 2 + xx // 8) == blk
        hi = int(ncf[mb].max())
        for s0 in range(0, hi, 64):
            s1 = min(s0 + 64, hi)
            okb = ok[s0:s1][:, mb]
            it_block += int(okb.any(1).sum())
            qs = []
            for q in range(4):
                mq = (((yy[mb] % 8) // 4) * 2 + (xx[mb] % 8) // 4) == q
                qs.append(int(okb[:, mq].any(1).sum()))
            it_quad += max(qs); it_quad_sum += sum(qs)
print(f"phase-A iterations: per-block queues {it_block}, per-quadrant queues {it_quad} (max over 4), "
      f"ratio {it_quad/it_block:.3f}; quadrant pairs total {it_quad_sum} ({it_quad_sum/it_block:.2f} per block pair)")

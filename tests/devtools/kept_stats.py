"""Dev tool (GPU): how much larger the backward's kept sets would be if the keep-set of a (unit, 8x8 block) came from the
FORWARD as one word -- OR over the block's pixels of their candidate words, cut at the block's deepest last contributor --
instead of being formed in the backward from every pixel's word cut at ITS OWN last contributor.

usage: python tests/devtools/kept_stats.py [smoke|A|C] [view]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_masks as cm


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "C"
    view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    gs, cam, bg = cm.build(which, view)
    tr = cm.render(gs, cam, bg, need_backward=True)
    W, H = tr["W"], tr["H"]
    gx = (W + 15) // 16
    ranges = tr["ranges"]
    cand = cm.unpack(tr["masks"], False)            # [U, 4 blocks, 64 lanes, 64 positions]
    n_per_tile = ranges[:, 1] - ranges[:, 0]
    units_per_tile = (n_per_tile + 63) // 64
    unit0 = np.concatenate([[0], np.cumsum(units_per_tile)])
    nc = tr["nc"]
    kept_true = kept_blockmax = kept_nolimit = ub_true = ub_blockmax = 0
    pos = np.arange(64)
    for t in np.nonzero(n_per_tile > 0)[0]:
        ty, tx = divmod(int(t), gx)
        tile_nc = np.zeros((16, 16), np.int64)
        sub = nc[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
        tile_nc[:sub.shape[0], :sub.shape[1]] = sub
        for b in range(4):
            by, bx = divmod(b, 2)
            last = tile_nc[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8].reshape(64)
            for u in range(int(units_per_tile[t])):
                s0 = u * 64
                c = cand[unit0[t] + u, b]
                lim = np.clip(last - s0, 0, 64)
                k_true = (c & (pos[None, :] < lim[:, None])).any(axis=0).sum()
                k_bm = (c.any(axis=0) & (pos < lim.max())).sum()
                kept_true += int(k_true); kept_blockmax += int(k_bm)
                ub_true += int(k_true > 0); ub_blockmax += int(k_bm > 0)
                if lim.max() > 0:
                    kept_nolimit += int(c.any(axis=0).sum())
    print(f"kept instances (pair trips): true {kept_true}, block-max cut {kept_blockmax} (+{100.0 * (kept_blockmax / kept_true - 1):.2f} %), "
          f"no cut inside reached units {kept_nolimit} (+{100.0 * (kept_nolimit / kept_true - 1):.2f} %)")
    print(f"(unit, block) pairs with work: true {ub_true}, block-max cut {ub_blockmax}")


if __name__ == "__main__":
    main()

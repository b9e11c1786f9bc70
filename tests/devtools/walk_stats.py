"""Dev tool (GPU): what a PER-LANE backward walk would cost against the instance-uniform pair loop the backward runs.

The backward's unit (gsr_blend_bwd.hip) evaluates every kept instance on all 64 pixel lanes (about 10 of them live).  The
alternative measured here: lanes walk their OWN candidate bits (as the forward does), in groups of 8 kept instances (the
MFMA tile height); a group then costs max-over-lanes(candidates of the lane inside the group) trips instead of 8.

usage: python tests/devtools/walk_stats.py [smoke|A|C] [view]
"""
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
    kept_total = trips_total = pairs_total = groups_total = 0
    hist = np.zeros(9, np.int64)
    CH = (8, 16, 24, 32, 64)
    trips_ch = {c: 0 for c in CH}
    chunks_ch = {c: 0 for c in CH}
    kept_hist = np.zeros(65, np.int64)
    maxlane_hist = np.zeros(65, np.int64)
    unit_blocks = 0
    lanes = np.arange(64)
    PARTS = {"halves 8x4 (top/bottom)": [lanes < 32, lanes >= 32],
             "halves 4x8 (left/right)": [(lanes & 7) < 4, (lanes & 7) >= 4],
             "quadrants 4x4": [((lanes & 7) < 4) & (lanes < 32), ((lanes & 7) >= 4) & (lanes < 32),
                               ((lanes & 7) < 4) & (lanes >= 32), ((lanes & 7) >= 4) & (lanes >= 32)],
             "row pairs 8x2": [(lanes >> 4) == q for q in range(4)]}
    part_trips = {k: 0 for k in PARTS}
    part_sum = {k: 0 for k in PARTS}
    for t in np.nonzero(n_per_tile > 0)[0]:
        ty, tx = divmod(int(t), gx)
        tile_nc = np.zeros((16, 16), np.int64)
        sub = nc[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
        tile_nc[:sub.shape[0], :sub.shape[1]] = sub
        for b in range(4):
            by, bx = divmod(b, 2)
            last = tile_nc[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8].reshape(64)      # 1-based last contributor per lane
            for u in range(int(units_per_tile[t])):
                s0 = u * 64
                lim = np.clip(last - s0, 0, 64)
                w = cand[unit0[t] + u, b] & (np.arange(64)[None, :] < lim[:, None])   # [lane, position]
                kept = np.nonzero(w.any(axis=0))[0][::-1]                               # back to front
                if len(kept) == 0:
                    continue
                kept_total += len(kept)
                unit_blocks += 1
                for name, sel in PARTS.items():
                    ks = [int(w[m_].any(axis=0).sum()) for m_ in sel]
                    part_trips[name] += max(ks)
                    part_sum[name] += sum(ks)
                kept_hist[len(kept)] += 1
                maxlane_hist[int(w.sum(axis=1).max())] += 1
                for c in CH:
                    for g in range(0, len(kept), c):
                        trips_ch[c] += int(w[:, kept[g:g + c]].sum(axis=1).max())
                        chunks_ch[c] += 1
                pairs_total += int(w.sum())
                for g in range(0, len(kept), 8):
                    per_lane = w[:, kept[g:g + 8]].sum(axis=1)
                    trips = int(per_lane.max())
                    trips_total += trips
                    groups_total += 1
                    hist[trips] += 1
    print(f"{which}: kept instances {kept_total}  candidate pairs {pairs_total} ({pairs_total / kept_total:.1f} per kept instance)")
    print(f"groups of 8: {groups_total}  trips if lanes walk their own bits {trips_total} "
          f"({trips_total / kept_total:.3f} per kept instance, {trips_total / groups_total:.2f} per group)")
    print("trips per group histogram (0..8):", hist.tolist())
    print(f"(unit, block) pairs with work: {unit_blocks}; kept per unit-block mean {kept_total / unit_blocks:.2f}")
    for c in CH:
        print(f"chunks of {c:2d} kept instances: chunks {chunks_ch[c]} ({chunks_ch[c] / unit_blocks:.2f} per unit-block)  "
              f"trips {trips_ch[c]} ({trips_ch[c] / unit_blocks:.2f} per unit-block; uniform loop {kept_total / unit_blocks:.2f})  "
              f"lane utilisation {pairs_total / (64.0 * trips_ch[c]):.3f}")
    for name in PARTS:
        print(f"lock-step queues per {name}: trips {part_trips[name]} ({part_trips[name] / unit_blocks:.2f} per unit-block), "
              f"queued records {part_sum[name]} ({part_sum[name] / unit_blocks:.2f} per unit-block)")
    print("kept-instances histogram (0..64):", kept_hist.tolist())
    print("busiest-lane candidate count histogram (0..64):", maxlane_hist.tolist())


if __name__ == "__main__":
    main()

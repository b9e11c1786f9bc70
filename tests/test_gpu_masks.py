"""The per-pixel CANDIDATE WORDS the forward blend leaves behind (gsr_mask.h / gsr_blend_fwd.hip) are a SUPERSET of the
pairs the reference blends: for every list position the forward reached, every (pixel, instance) pair that passes the
reference's test (power <= 0 and alpha >= 1/255, forward.cu:330-345) has its bit set -- zero misses.  The forward walks
ONLY set bits and the backward replays ONLY set bits, so a dropped bit is a dropped contributor; at full size a single
one could hide inside the image / gradient outlier allowances of tests/parity.py, hence this direct check.

Config C: three cameras from different rings of the rig, EVERY non-empty tile.  Config D (lists up to ~11 800 entries,
blended in parts): the 64 longest tiles + 600 random ones (every tile would be 2.5 G pair evaluations in numpy)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tests", "devtools"))


def _check(gs, cam, bg, what, sample=None):
    import check_masks as cm
    tr = cm.render(gs, cam, bg, need_backward=True)
    tiles = None
    if sample is not None:
        n = tr["ranges"][:, 1] - tr["ranges"][:, 0]
        nz = np.nonzero(n > 0)[0]
        longest = nz[np.argsort(-n[nz])[:64]]
        rnd = np.random.default_rng(1).choice(nz, min(sample, len(nz)), replace=False)
        tiles = np.unique(np.concatenate([longest, rnd]))
    n_ok, n_cand, n_missing = cm.verify(tr, tiles)
    print(f"[masks] {what}: pairs passing the alpha test {n_ok}, candidate bits {n_cand} ({n_cand / max(n_ok, 1):.4f} per passing "
          f"pair), passing pairs WITHOUT a bit: {n_missing}")
    assert n_ok > 0
    assert n_missing == 0, f"{what}: {n_missing} pairs the reference blends have no candidate bit"
    assert n_cand <= 1.05 * n_ok, f"{what}: the words are no longer tight ({n_cand} bits for {n_ok} passing pairs)"


@pytest.mark.parametrize("view", [3, 70, 141])
def test_candidate_words_cover_every_blended_pair_config_c(view):
    from gaustar_amd import scene
    gs, cams, bg = scene.config_C()
    _check(gs, cams[view], bg, f"config C view {view}")


def test_candidate_words_cover_every_blended_pair_config_d():
    from gaustar_amd import scene
    gs, cam, bg = scene.config_D()
    _check(gs, cam, bg, "config D", sample=600)

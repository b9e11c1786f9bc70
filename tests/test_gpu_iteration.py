"""GPU integration: producers -> 6-channel render -> fused losses -> backward -> Adam, i.e. one GauSTAR-style
refinement iteration assembled from this package only (tools/bench_iteration.py).  The loss must go down."""
import argparse
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_refinement_iterations_reduce_the_loss(hip_lib):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_iteration
    r = bench_iteration.run(argparse.Namespace(steps=60, warmup=0, level=3, width=320, height=240))
    assert r["gaussians"] == 20 * 4 ** 3 * 6
    assert r["loss_first"] == r["loss_first"] and r["loss_last"] == r["loss_last"], "NaN loss"
    assert r["loss_last"] < 0.8 * r["loss_first"], r

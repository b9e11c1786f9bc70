"""GPU: the one-line contract of bench.py at N = 1 as the driver runs it (reduced step counts): metric / unit / config names of
BASELINE.json, medians over interleaved repeats with their spread, the like-for-like single-pipeline figure, roofline and
(without --no-extras) the other single-GPU configs with their own roofline."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*args):
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", *args], cwd=ROOT, env=dict(os.environ, GSR_BENCH_NO_PIN="1"),
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_at_one_gpu():
    d = _bench("--steps", "6", "--warmup", "2", "--repeats", "3", "--no-cpu-baseline", "--no-extras")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].replace("x", "×") == base["metric"] or d["metric"] == base["metric"].replace("×", "x")
    assert d["unit"] == "views/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    c = d["config"]
    assert "config C" in c["workload"] and c["gaussians"] == 491520 and (c["width"], c["height"]) == (1920, 1080)
    assert c["views_in_flight"] == 2 and "MEDIAN" in c["value_definition"] and c["value_spread"]["repeats"] == 3
    sp = d["single_pipeline"]
    assert sp["repeats"] == 3 and sp["ms_per_step_min"] <= sp["ms_per_step"] <= sp["ms_per_step_max"]
    assert c["value_spread"]["ms_per_step_min"] <= d["ms_per_step"] <= c["value_spread"]["ms_per_step_max"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-3            # value = views per second of the median region
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0.0 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    # (views binned by their camera's plan launch neither scan nor scatter: config.binning says how many were)
    assert set(rf["kernels"]) >= {"preprocess_kernel", "blend_fwd_kernel", "blend_bwd_kernel", "geom_bwd_kernel"}
    bn = c["binning"]
    assert bn["planned_views"] + bn["exact_views"] > 0 and bn["misfits"] == 0
    if bn["exact_views"] == 0:   # cameras 0-5 of the rig are all plannable
        assert "tile_scan_kernel" not in rf["kernels"] and "scatter_kernel" not in rf["kernels"]
        assert d["exact_binning"]["ms_per_step"] > 0.0   # (6-step regions are too noisy to order the two figures: profiles/ has them)
    assert rf["kernel"] in ("blend_bwd_kernel", "blend_fwd_kernel") and "host" in d
    # the fraction follows from the line's own numbers: algorithmic bytes of ONE launch / its mean duration / peak, and the
    # brackets count the K timed steps only (round 4 divided by 22 / 20 launches per step: timed()'s untimed first steps)
    for k in ("preprocess_kernel", "blend_fwd_kernel", "blend_bwd_kernel", "geom_bwd_kernel"):
        assert rf["kernels"][k]["launches_per_step"] == 1.0, (k, rf["kernels"][k])
    recomputed = rf["alg_bytes_per_launch"] / (rf["ms_per_launch"] * 1e-3) / 8e12
    assert abs(recomputed - rf["frac"]) <= 2e-3 * rf["frac"], (recomputed, rf["frac"])
    assert abs(rf["kernels_sum_ms_per_step"] - sum(v["ms_per_launch"] for v in rf["kernels"].values())) < 2e-3
    assert "frac_rocprof" in rf and (rf["frac_rocprof"] is None or 0.0 < rf["frac_rocprof"] < 1.0)


def test_bench_other_configs_leg():
    """`other_configs` (B, D, D-depth: BASELINE.json configs[1] and configs[3]) and `scale_base` ride in the default N = 1 line."""
    import bench
    import torch
    from gaustar_amd import _lib
    dev = torch.device("cuda:0")
    o = bench.other_config("B", dev, _lib.load(), steps=4, repeats=2)
    assert o["gaussians"] == 200400 and o["beta_P"] == 464 and o["sh_coeffs_in_kernel"] == 0 and o["num_rendered"] > 100_000
    assert 0.0 < o["path_frac"] < 1.0 and o["dominant_kernel"] in o["kernels"] and o["best_frac"] >= o["dominant_frac"]
    assert all(v["launches_per_step"] == 1.0 for k_, v in o["kernels"].items() if k_ in ("blend_fwd_kernel", "blend_bwd_kernel"))
    total, per = bench.algorithmic_bytes(200400, o["num_rendered"], 1920, 1080, 0)
    assert o["alg_bytes_per_view"] == int(total)
    t16, _ = bench.algorithmic_bytes(1000, 2000, 64, 64, 16)
    assert t16 == 1000 * (494 + 48 * 16) + 2000 * 160 + 64 * 64 * 40 + 16 * 16       # SURVEY.md 8(d) with in-kernel SH

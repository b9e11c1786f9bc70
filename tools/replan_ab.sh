#!/usr/bin/env bash
# tools/replan_ab.sh [iters] -- the windowed refinement loop (tools/bench_window.py, 2 frames x ITERS iterations = ITERS / 160 visits per
# camera and frame, the mesh moving between them) with and without re-planning from planned views (GSR_REPLAN, gsr_api.hip), per
# slack level of a camera's first plan: median ms per iteration and how the views were binned.  GPU box:
#   gpurun -- 'bash tools/replan_ab.sh 480 > gpurun_out/replan_ab.txt 2>&1'
cd "$(dirname "$0")/.."
IT="${1:-480}"
# (neighbour-aware slack, gsr_plan.h GSR_PLAN_NB_SLACK, is a compile-time switch:
#   python -m gaustar_amd.build --variant nbs0 -DGSR_PLAN_NB_SLACK=0)
for lvl in 0 1 2; do
  for cfg in "0 nbs0" "1 nbs0" "0 -" "1 -"; do
    set -- $cfg; rp=$1; lib=$2
    [ "$lib" = "-" ] && libpath=$PWD/gaustar_amd/libgsr_hip.so || libpath=$PWD/gaustar_amd/libgsr_hip_$lib.so
    [ -f "$libpath" ] || continue
    echo "== GSR_REPLAN=$rp GSR_PLAN_LEVEL=$lvl lib=$lib iters=$IT"
    GSR_LIB_PATH=$libpath GSR_REPLAN=$rp GSR_PLAN_LEVEL=$lvl python - <<PY 2>&1 | tail -1
import argparse, json, sys, os
sys.path.insert(0, "tools")
import bench_window
from gaustar_amd import rasterizer as rz
bench_window.run(argparse.Namespace(frames=1, iters=8, level=6, width=1920, height=1080, cameras=16))
rz.drop_plans()
before = dict(rz.PLAN_STATS)
r = bench_window.run(argparse.Namespace(frames=2, iters=$IT, level=6, width=1920, height=1080, cameras=160))
st = {k: rz.PLAN_STATS[k] - before[k] for k in before}
print(json.dumps({"median_ms": r["median_ms_per_iteration"], "mean_ms": r["ms_per_iteration"], "p90": r["p90_ms_per_iteration"], "plan_stats": st}))
PY
  done
done

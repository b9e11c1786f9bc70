#!/usr/bin/env bash
# tools/walk_ab.sh -- the per-lane-walk backward (gsr_blend_bwd_walk.hip) against the uniform pair loop: parity suite under
# GSR_BWD_WALK, instruction counts (one PMC pass per build) and an in-process interleaved timing A/B.  Run on the GPU box:
#   gpurun -- 'bash tools/walk_ab.sh > gpurun_out/walk_ab.log 2>&1'
# The walk kernel lives in tools/variants/ (round 5: out of the product library); build the variant libraries first:
#   for r in 8 16 32; do python -m gaustar_amd.build --variant walk$r --with tools/variants/gsr_blend_bwd_walk.hip; done
# and select the table height with GSR_BWD_WALK=<rows> when loading one of them (GSR_LIB_PATH=gaustar_amd/libgsr_hip_walk16.so).
cd "$(dirname "$0")/.."
export GSR_LIB_PATH=$PWD/gaustar_amd/libgsr_hip_walk16.so
echo "== parity, GSR_BWD_WALK=16 (whole GPU suite)"
GSR_BWD_WALK=16 timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for r in 8 32; do
  echo "== parity, GSR_BWD_WALK=$r (parity + multitarget)"
  GSR_BWD_WALK=$r timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multitarget.py -x -q -m gpu 2>&1 | tail -3
done
for v in "" _walk8 _walk16 _walk32; do
  echo "== instruction counts libgsr_hip$v.so"
  bash tools/valu_count.sh gaustar_amd/libgsr_hip$v.so | grep -i "blend\|sum"
done
echo "== timing A/B (ms per fwd+bwd step, medians of 6 interleaved rounds of 160 steps)"
bash tools/ab3.sh gaustar_amd/libgsr_hip.so gaustar_amd/libgsr_hip_walk8.so gaustar_amd/libgsr_hip_walk16.so gaustar_amd/libgsr_hip_walk32.so

#!/usr/bin/env bash
# tools/pk_ab.sh -- the packed-math backward (gsr_blend_bwd_pk.hip, two kept instances per trip on v_pk_*_f32) against the
# product's scalar pair loop: parity under GSR_BWD_PK=1, instruction counts and an in-process interleaved timing A/B.
#   python -m gaustar_amd.build --variant pk --with tools/variants/gsr_blend_bwd_pk.hip      (round 5: the kernel lives in tools/variants/)
#   export GSR_LIB_PATH=$PWD/gaustar_amd/libgsr_hip_pk.so
#   gpurun -- 'bash tools/pk_ab.sh > gpurun_out/pk_ab.log 2>&1'
cd "$(dirname "$0")/.."
echo "== parity, GSR_BWD_PK=1 (parity + multitarget + harness)"
GSR_BWD_PK=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multitarget.py tests/test_gpu_harness.py -x -q -m gpu 2>&1 | tail -5
for v in "" _pk; do
  echo "== instruction counts libgsr_hip$v.so"
  bash tools/valu_count.sh gaustar_amd/libgsr_hip$v.so | grep -i "blend\|sum"
done
echo "== timing A/B (ms per fwd+bwd step, medians of 6 interleaved rounds)"
bash tools/ab3.sh gaustar_amd/libgsr_hip.so gaustar_amd/libgsr_hip_pk.so
bash tools/ab3.sh gaustar_amd/libgsr_hip.so gaustar_amd/libgsr_hip_pk.so

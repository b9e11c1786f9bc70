#!/usr/bin/env bash
# tools/quad_ab.sh [variant names ...] -- the four-instances-per-trip backward (tools/variants/gsr_blend_bwd_quad.hip) against the
# product's uniform pair loop: parity suite under the variant library, instruction counts (one PMC pass per build), rocprofv3
# kernel durations and an in-process interleaved timing A/B.  Build first:
#   python -m gaustar_amd.build --variant quad --with tools/variants/gsr_blend_bwd_quad.hip [-DGSR_PAIRS_Q=8 ...]
# Run on the GPU box:  gpurun -- 'bash tools/quad_ab.sh quad > gpurun_out/quad_ab.log 2>&1'
cd "$(dirname "$0")/.."
VARS="${@:-quad}"
for v in $VARS; do
  echo "== parity under libgsr_hip_$v.so (parity + multitarget + pipelines + harness)"
  GSR_LIB_PATH=$PWD/gaustar_amd/libgsr_hip_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multitarget.py \
      tests/test_gpu_pipelines_full.py tests/test_gpu_harness.py tests/test_gpu_iteration.py -x -q -m gpu 2>&1 | tail -15
done
for v in "" $VARS; do
  lib=gaustar_amd/libgsr_hip${v:+_$v}.so
  echo "== instruction counts $lib"
  bash tools/valu_count.sh $lib | grep -i "blend\|sum"
  echo "== kernel durations (rocprofv3 --kernel-trace --stats) $lib"
  ( export TMPDIR=/tmp GSR_LIB_PATH=$PWD/$lib; R=$PWD; rm -rf gpurun_out/kt_tmp; cd /tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_tmp -o k -- python $R/bench.py --steps 20 --warmup 5 \
        --no-cpu-baseline --no-extras --views-in-flight 1 --repeats 1 > /dev/null 2>&1
    cd $R; f=$(find gpurun_out/kt_tmp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "blend\|Name" $f | cut -c1-200 )
done
echo "== timing A/B (ms per fwd+bwd step, medians of 6 interleaved rounds of 160 steps)"
libs=""; for v in $VARS; do libs="$libs gaustar_amd/libgsr_hip_$v.so"; done
bash tools/ab3.sh gaustar_amd/libgsr_hip.so $libs

#!/usr/bin/env bash
# tools/quad_pmc.sh -- counters and a residency sweep of the backward blend: four instances per trip (the product since round 6,
# gsr_blend_bwd.hip) against the uniform pair loop (rounds 1-5) inside the build that carries both --
#   python -m gaustar_amd.build --variant uniform --with tools/variants/gsr_blend_bwd_uniform.hip   (GSR_BWD_UNIFORM=1 / 0) --:
# two rocprofv3 --pmc passes per build (kernel-trace only) and the kernel's rocprofv3 duration under GSR_BWD_LDS_PAD.
cd "$(dirname "$0")/.."
R="$PWD"; export TMPDIR=/tmp
kt() {   # kt LIB PAD -> average ns of the backward blend kernel
  ( export GSR_LIB_PATH=$R/gaustar_amd/libgsr_hip_uniform.so GSR_BWD_UNIFORM=$1 GSR_BWD_LDS_PAD=$2; rm -rf $R/gpurun_out/kt_tmp; cd /tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_tmp -o k -- python $R/bench.py --steps 12 --warmup 3 \
        --no-cpu-baseline --no-extras --views-in-flight 1 --repeats 1 > /dev/null 2>&1
    f=$(find $R/gpurun_out/kt_tmp -name "*kernel_stats.csv" | head -1)
    python - "$f" "$1" "$2" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "blend_bwd" in r["Name"]:
        print(f"{sys.argv[2]:40s} pad {sys.argv[3]:>6s}  {r['Name'].split('(')[0]:36s} avg {float(r['AverageNs'])/1e3:8.1f} us  calls {r['Calls']}")
PY
  )
}
echo "== residency sweep (dynamic LDS padding; waves per SIMD = floor(160 KB / (static + pad)) / 4)"
# (both kernels hold ~5.2 KB of static LDS and 72 registers: 7 / 5 / 4 / 3 / 2 waves per SIMD at these paddings)
for pad in 0 2800 4800 8200 15000; do kt 0 $pad; done
for pad in 0 2800 4800 8200 15000; do kt 1 $pad; done
for lib in 0 1; do
  echo "== counters GSR_BWD_UNIFORM=$lib (millions per launch)"
  i=0
  for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
    ( export GSR_LIB_PATH=$R/gaustar_amd/libgsr_hip_uniform.so GSR_BWD_UNIFORM=$lib; rm -rf $R/gpurun_out/pmc_quad/p$i; cd /tmp
      rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$R/gpurun_out/pmc_quad/p$i" -o "p$i" -- python "$R/bench.py" --steps 4 --warmup 2 \
          --no-cpu-baseline --no-extras --views-in-flight 1 --repeats 1 > /dev/null 2>&1 )
    i=$((i+1))
  done
  python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_quad/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "blend_bwd" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print("  ", k, "  ".join(f"{c}={sum(v)/len(v)/1e6:.2f}" for c, v in sorted(agg[k].items())))
PY
done

#!/usr/bin/env bash
# quick: VALU/SALU of blend_fwd for a given lib
export TMPDIR=/tmp; R=$PWD
for v in "" _nowalk _nomask; do
  cd /tmp; rm -rf /tmp/pmcq
  GSR_LIB_PATH=$R/gaustar_amd/libgsr_hip$v.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/pmcq -o q -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  cd $R
  python - "$v" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "blend_fwd_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("variant [%s]" % sys.argv[1], {k: round(sum(v)/len(v)/1e6, 2) for k, v in agg.items()})
PY
done

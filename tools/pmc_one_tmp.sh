#!/usr/bin/env bash
# tools/pmc_one_tmp.sh KERNEL_SUBSTR [LIB_SUFFIX...] -- a few counters of one kernel
export TMPDIR=/tmp; R=$PWD; K=$1; shift
for v in "$@"; do
 for P in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  cd /tmp; rm -rf /tmp/pmcq
  GSR_LIB_PATH=$R/gaustar_amd/libgsr_hip$v.so rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmcq -o q -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  cd $R
  python - "$v" "$K" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("variant [%s]" % sys.argv[1], {k: round(sum(v)/len(v)/1e6, 2) for k, v in agg.items()})
PY
 done
done

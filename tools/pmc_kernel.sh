#!/usr/bin/env bash
# tools/pmc_kernel.sh KERNEL_SUBSTR "COUNTERS ..." ["COUNTERS ..." ...] -- mean per launch of a few counters of one kernel
# (one rocprofv3 run per quoted counter set, kernel-trace only); GSR_LIB_PATH selects the build.
export TMPDIR=/tmp; R=$PWD; K=$1; shift
for P in "$@"; do
  cd /tmp; rm -rf /tmp/pmcq
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmcq -o q -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --views-in-flight 1 --repeats 1 > /dev/null 2>&1
  cd $R
  python - "$K" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v) / 1e6, 3) for k, v in agg.items()}, "(millions per launch)")
PY
done

#!/usr/bin/env bash
# tools/pmc_loss.sh "COUNTERS ..." ["COUNTERS ..."] -- mean per launch of a few counters of the two SSIM kernels (tools/bench_loss.py;
# one rocprofv3 run per quoted counter set, kernel-trace only)
export TMPDIR=/tmp; R=$PWD
for P in "$@"; do
  cd /tmp; rm -rf /tmp/pmcq
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmcq -o q -- python $R/tools/bench_loss.py > /dev/null 2>&1
  cd $R
  python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("ssim_stats", "ssim_grad"):
            if k in r["Kernel_Name"]:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]; print(k, round(sum(v) / len(v) / 1e6, 3), "M per launch")
PY
done

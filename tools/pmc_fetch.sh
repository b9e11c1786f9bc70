#!/usr/bin/env bash
# tools/pmc_fetch.sh [LIB] -- FETCH_SIZE / WRITE_SIZE (KiB, per launch) of the blend kernels for one library build
set -uo pipefail
R="$PWD"; export TMPDIR=/tmp
[ $# -ge 1 ] && export GSR_LIB_PATH="$1"
TAG=$(basename "${GSR_LIB_PATH:-default}" .so)
i=0
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  cd /tmp
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$R/gpurun_out/pmcf_$TAG/p$i" -o "p$i" -- python "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-extras --views-in-flight 1 --repeats 1 > /dev/null 2>&1 || echo "pass $i failed"
  cd "$R"; i=$((i+1))
done
python - "$TAG" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmcf_{sys.argv[1]}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "blend" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    f_, w_ = (sum(agg[k][c]) / max(len(agg[k][c]), 1) for c in ("FETCH_SIZE", "WRITE_SIZE"))
    print(f"{sys.argv[1]:24s} {k:28s} fetch {f_:10.0f} KiB  write {w_:10.0f} KiB  -> HBM bytes (2*fetch + write) {(2*f_ + w_) * 1024 / 1e6:8.1f} MB")
PY

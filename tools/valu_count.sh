#!/usr/bin/env bash
# tools/valu_count.sh [LIB.so] -- wave64 vector / scalar instruction counts per kernel launch of a short bench run (one PMC pass):
# the deterministic figure of merit for instruction-level work on kernels that are issue-bound.
set -uo pipefail
R="$PWD"; export TMPDIR=/tmp
[ $# -ge 1 ] && export GSR_LIB_PATH="$(realpath "$1")"
rm -rf "$R/gpurun_out/valu_tmp"; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$R/gpurun_out/valu_tmp" -o v -- \
    python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-extras --views-in-flight 1 --repeats 1 > /dev/null 2>&1
cd "$R"
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/valu_tmp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("gsr::"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = 0
for k in sorted(agg):
    v = {c: sum(x) / len(x) for c, x in agg[k].items()}
    tot += v.get("SQ_INSTS_VALU", 0)
    print(f"{k:28s} VALU {v.get('SQ_INSTS_VALU', 0) / 1e6:8.3f} M  SALU {v.get('SQ_INSTS_SALU', 0) / 1e6:7.3f} M  LDS {v.get('SQ_INSTS_LDS', 0) / 1e6:7.3f} M")
print(f"{'sum (one launch each)':28s} VALU {tot / 1e6:8.3f} M")
PY

#!/usr/bin/env bash
# tools/valu_channels.sh [LIB.so] -- wave64 instruction counts per launch of the two blends for 3, 4 and 6 colour channels
# (tools/bench_channels.py under one PMC pass; kernel names keep their template arguments here).
set -uo pipefail
R="$PWD"; export TMPDIR=/tmp
[ $# -ge 1 ] && export GSR_LIB_PATH="$(realpath "$1")"
rm -rf "$R/gpurun_out/valuc_tmp"; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d "$R/gpurun_out/valuc_tmp" -o v -- \
    python "$R/tools/bench_channels.py" --steps 4 --warmup 2 > /dev/null 2>&1
cd "$R"
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/valuc_tmp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("gsr::blend"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = {c: sum(x) / len(x) for c, x in agg[k].items()}
    print(f"{k:40s} " + "  ".join(f"{c.replace('SQ_INSTS_', '')} {v[c] / 1e6:8.3f} M" for c in sorted(v)))
PY

#!/usr/bin/env bash
# tools/pmc.sh TAG -- rocprofv3 PMC passes (separate runs, kernel-trace only) of a short bench run.
# Per-kernel averages are written to gpurun_out/pmc_TAG_summary.txt
set -uo pipefail
TAG="$1"; shift
R="$PWD"
export TMPDIR=/tmp
mkdir -p "$R/gpurun_out"
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum"
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH"
)
i=0
for P in "${PASSES[@]}"; do
  cd /tmp
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$R/gpurun_out/pmc_${TAG}/p$i" -o "p$i" -- \
      python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-extras --views-in-flight 1 --repeats 1 > "$R/gpurun_out/pmc_${TAG}_p$i.log" 2>&1 || echo "pass $i failed"
  cd "$R"
  i=$((i+1))
done
python - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("gsr::"):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"gpurun_out/pmc_{tag}_summary.txt", "w") as o:
    for k in sorted(agg):
        o.write(k + "\n")
        for c in sorted(agg[k]):
            v = agg[k][c]
            o.write(f"   {c:24s} mean {sum(v)/len(v):16.1f}  n={len(v)}\n")
print(open(f"gpurun_out/pmc_{tag}_summary.txt").read())
PY

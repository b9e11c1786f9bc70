set -u
R="$PWD"; export TMPDIR=/tmp
bash tools/quick_bench.sh GSR_BWD_LDS_PAD=3500 | tail -1
bash tools/quick_bench.sh GSR_BWD_LDS_PAD=7000 | tail -1
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  cd /tmp
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$R/gpurun_out/pmc_bwd/p$i" -o "p$i" -- python "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-extras --views-in-flight 1 --repeats 1 > "$R/gpurun_out/pmc_bwd_p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$R/gpurun_out/pmc_bwd_p$i.log"; }
  cd "$R"; i=$((i+1))
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_bwd/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "blend" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]; print(f"   {c:28s} {sum(v)/len(v):16.1f}")
PY

export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_win -o win -- python $R/tools/bench_window.py --frames 2 --iters 50 > $R/gpurun_out/prof_win.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_win/**/win_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "plan_build" in r["Kernel_Name"]]
print("plan_build launches", len(idx))
i0 = idx[-20] - 12
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = None
for r in rows[i0:i0 + 60]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{(s - t0)/1e3:9.1f} us  dur {(e - s)/1e3:7.1f}  gap {gap:6.1f}  q{r['Queue_Id']} {r['Kernel_Name'].split('(')[0][-50:]}")
    prev_end = max(prev_end or 0, e)
PY

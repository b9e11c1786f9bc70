#!/usr/bin/env bash
# tools/window_timeline.sh -- rocprofv3 kernel trace of tools/bench_window.py (2 frames x 60 iterations, config-C size): the kernels
# of one steady-state iteration in launch order with start offsets and durations, and the per-iteration totals by kernel.
set -uo pipefail
R="$PWD"; export TMPDIR=/tmp
rm -rf "$R/gpurun_out/wtl"; cd /tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/wtl" -o w -- python "$R/tools/bench_window.py" --frames 2 --iters 60 "$@" > "$R/gpurun_out/wtl.log" 2>&1
cd "$R"
python - <<'PY'
import csv, glob, re, collections
fs = glob.glob("gpurun_out/wtl/**/*kernel_trace.csv", recursive=True)
if not fs:
    print("no kernel trace written"); raise SystemExit(0)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(fs[0]))), key=lambda t: t[0])
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", ""))[:90]
idx = [i for i, r in enumerate(rows) if "blend_bwd" in r[2]]
a, b = idx[-12], idx[-11]
print("one iteration (us from its backward blend's start):")
for s, e, n in rows[a:b]:
    print(f"{(s - rows[a][0]) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {short(n)}")
span = rows[idx[-2]][0] - rows[idx[-42]][0]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows[idx[-42]:idx[-2]]:
    k = short(n); agg[k][0] += 1; agg[k][1] += e - s
print(f"\n40 iterations: {span / 40 / 1e3:.1f} us per iteration, busy {sum(v[1] for v in agg.values()) / 40 / 1e3:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{v[1] / 40 / 1e3:8.1f} us  x{v[0] / 40:4.1f}  {k}")
PY

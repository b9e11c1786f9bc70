"""tools/trace_summary.py KERNEL_TRACE.csv [skip_fraction] -- GPU busy vs idle and the top kernels of the LAST part of a
rocprofv3 --kernel-trace run (the first `skip_fraction` of the kernels -- set-up, warm-up -- is ignored)."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
short = lambda n: n.split("(")[0].replace("void ", "").replace("gsr::", "")[:70]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"kernels {len(rows)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms ({100*busy/span:.1f} %)")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{100*t/busy:5.1f} %  n={n:5d}  mean {t/n/1e3:8.2f} us  {k}")

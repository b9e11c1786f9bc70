"""tools/trace_overlap.py KERNEL_TRACE.csv -- how much of a multi-stream run the GPU spends with 0 / 1 / 2+ kernels in
flight, and the busy time per kernel name (rocprofv3 --kernel-trace output)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
per = collections.defaultdict(float)
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    if not name.startswith("gsr::"):
        continue
    ev.append((a, 1)); ev.append((b, -1)); per[name] += b - a
ev.sort()
# skip the first 20 % (warm-up, setup)
t_lo = ev[0][0] + (ev[-1][0] - ev[0][0]) * 2 // 5   # skip the first 40 % (set-up, warm-up)
depth, last, hist = 0, None, collections.defaultdict(int)
for t, d in ev:
    if last is not None and t > t_lo:
        hist[min(depth, 3)] += t - max(last, t_lo)
    depth += d; last = t
tot = sum(hist.values())
print({k: round(v / tot, 3) for k, v in sorted(hist.items())}, "fraction of wall time with 0 / 1 / 2 / 3+ kernels in flight")
print({k: round(v / 1e3) for k, v in per.items()}, "us busy per kernel name (whole trace)")

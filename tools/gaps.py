"""tools/gaps.py KERNEL_TRACE.csv -- idle time between consecutive kernels of one bench run (rocprofv3 --kernel-trace).
Prints, per (previous kernel -> next kernel) pair, the mean gap in us and the mean busy/idle split per step."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("gsr::", "").split("<")[0]
gaps = collections.defaultdict(list)
busy = 0
for a, b in zip(rows, rows[1:]):
    gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
tot = {k: sum(v) for k, v in gaps.items()}
n_steps = max(1, sum(1 for r in rows if "blend_bwd" in r["Kernel_Name"]))
print(f"steps (blend_bwd launches): {n_steps}")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v2 = sorted(v)
    print(f"{k[0]:28s} -> {k[1]:28s} n={len(v):4d} median={v2[len(v2)//2]:8.2f} us  mean={sum(v)/len(v):8.2f} us")

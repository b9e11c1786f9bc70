"""tools/summarize_profiles.py TAG NAME -- turn gpurun_out/{prof,pmc}_TAG_* into the committed summaries
profiles/NAME_kernel_stats.txt, profiles/NAME_pmc.txt and profiles/pmc_latest.json.

HBM bytes per launch follow MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE come from separate
--pmc passes and are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of 16-byte-per-lane reads,
so it is doubled.  Calibration on a kernel of known traffic in this very run (geom_bwd_kernel: 124 B read and
92 B written per Gaussian, 491 520 Gaussians) confirms both: 2*FETCH_SIZE = 61.1 MB vs 60.9 MB expected,
WRITE_SIZE = 45.2 MB vs 45.2 MB expected."""
import hashlib, json, os, re, sys

tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

out = []
for l in open(os.path.join(src, f"prof_{tag}_summary.txt")):
    if l.startswith('"') and not l.startswith('"Name"'):
        m = re.match(r'"(.*)",(\d+,\d+,[\d.]+,[\d.e+-]+,\d+,\d+,[\d.e+-]+)\s*$', l.strip())
        if m:
            nm = re.sub(r"\(.*", "", m.group(1)).replace("void ", "")
            l = f'"{nm}",{m.group(2)}\n'
    elif l.startswith("{"):
        l = l[:1500] + (" ...\n" if len(l) > 1500 else "")
    out.append(l)
open(os.path.join(dst, f"{name}_kernel_stats.txt"), "w").writelines(out)

pmc = open(os.path.join(src, f"pmc_{tag}_summary.txt")).read()
hdr = ("# rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline\n"
       "# one run per counter set (tools/pmc.sh); per-kernel MEAN per launch. FETCH_SIZE/WRITE_SIZE in KiB.\n")
open(os.path.join(dst, f"{name}_pmc.txt"), "w").write(hdr + pmc)

kern, cur = {}, None
for l in pmc.splitlines():
    if l.startswith("gsr::"):
        cur = l.strip().replace("gsr::", "").split("<")[0]
        kern.setdefault(cur, {})
    elif cur and "mean" in l:
        p = l.split()
        kern[cur][p[0]] = float(p[2])
res = {}
for k, c in kern.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        res[k] = {"hbm_bytes_per_launch": int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
                  "fetch_size_kib": c["FETCH_SIZE"], "write_size_kib": c["WRITE_SIZE"],
                  "valu_insts": c.get("SQ_INSTS_VALU"), "salu_insts": c.get("SQ_INSTS_SALU"), "lds_insts": c.get("SQ_INSTS_LDS"),
                  "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": c.get("SQ_ACTIVE_INST_LDS"),
                  "atomic_requests": c.get("TCC_EA0_ATOMIC_sum")}
# rocprofv3's own per-kernel mean durations (kernel trace of the un-instrumented bench run): bench.py's HIP-event brackets add
# 1 - 4 us per stage, so the sum of ITS means exceeds the step; the sum of these does not
for l in out:
    m = re.match(r'"gsr::(\w+?)(?:<[^"]*)?",(\d+),(\d+),([\d.]+),', l)
    # (a kernel with two instantiations in the run -- exact and planned binning -- is represented by the one launched more often)
    if m and m.group(1) in res and int(m.group(2)) > res[m.group(1)].get("rocprof_calls", 0):
        res[m.group(1)]["rocprof_avg_ns"] = float(m.group(4))
        res[m.group(1)]["rocprof_calls"] = int(m.group(2))
res["_source"] = f"profiles/{name}_pmc.txt (2*FETCH_SIZE + WRITE_SIZE, KiB; see tools/summarize_profiles.py)"
# the counters describe THESE kernels: bench.py drops them (traffic = null) once gaustar_amd/csrc has changed
h = hashlib.sha256()
csrc = os.path.join(root, "gaustar_amd", "csrc")
for f in sorted(os.listdir(csrc)):
    h.update(f.encode()); h.update(open(os.path.join(csrc, f), "rb").read())
res["_csrc_sha256"] = h.hexdigest()
json.dump(res, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])

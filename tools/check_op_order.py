"""tools/check_op_order.py -- is this library's per-Gaussian projection the SAME tree of IEEE operations as the reference build's?

    python tools/check_op_order.py                 compile gaustar_amd/csrc/gsr_preprocess.hip to assembly, run tools/symfp.py over
                                                   preprocess_kernel<false> and <true>, compare the operation trees of
                                                   {pixel x, pixel y, conic a, b, c} with tests/golden/preprocess_op_order.txt
    python tools/check_op_order.py --make-fixture  (dev container only: needs /root/reference and its hipify'd build, as
                                                   oracle/build_ref.sh makes it) derive that fixture from the REFERENCE build's
                                                   preprocessCUDA

The fixture is data about the reference build -- a straight-line program over named inputs, one IEEE operation per line -- not
source text; equal programs <=> bit-identical results on every input (gsr_ref_order.h says why that matters).
"""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import symfp  # noqa: E402

FIXTURE = os.path.join(ROOT, "tests", "golden", "preprocess_op_order.txt")
OUTS = ("px", "py", "conic_a", "conic_b", "conic_c")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def program(stores, picks, rename):
    """picks: {output name: (store line, component)} -> the SSA text of those outputs over renamed leaves."""
    sel = []
    for name in OUTS:
        ln, comp = picks[name]
        vals = next(v for l, _t, v in stores if l == ln)
        sel.append((name, vals[comp]))
    prog, res = symfp.ssa(sel, rename)
    return "\n".join(prog + [f"OUT {label} = {nm}" for label, nm in res]) + "\n"


def find_stores(M):
    """The store that carries (px, py, ...) -- its first component is the f64 ndc2Pix expression -- and the one with the conic."""
    xy = conic = None
    for ln, _text, vals in M.stores:
        heads = [symfp.show(symfp.canon(v))[:12] for v in vals]
        if len(vals) >= 2 and heads[0].startswith("f32(mul64(") and heads[1].startswith("f32(mul64("):
            xy = (ln, vals)
        if any(h.startswith("mul(") and "div(1.0" in symfp.show(symfp.canon(v))[:200000] for h, v in zip(heads, vals)) and len(vals) == 4:
            conic = (ln, vals) if conic is None or ln > conic[0] else conic
    return xy, conic


def ours(kernel_prefix, asm):
    kargs = {0x10: "pts", 0x30: "scales", 0x40: "rot", 0x28: "opac", 0x48: "cov3Dpre", 0x50: "view", 0x58: "proj", 0x60: "campos"}
    M = symfp.run(symfp.extract(asm, kernel_prefix), kargs)
    # g0 = {px, py, conic a, conic b}; g1 = {conic c, opacity, tau, 0}
    g0 = next((ln, v) for ln, _t, v in M.stores if len(v) == 4 and symfp.show(symfp.canon(v[0])).startswith("f32(mul64("))
    g1 = next((ln, v) for ln, _t, v in M.stores if len(v) == 4 and symfp.show(v[1]) == "opac[+0]")
    picks = {"px": (g0[0], 0), "py": (g0[0], 1), "conic_a": (g0[0], 2), "conic_b": (g0[0], 3), "conic_c": (g1[0], 0)}
    rename = {"pts[+0]": "X", "pts[+1]": "Y", "pts[+2]": "Z", "rot[+0]": "qr", "rot[+1]": "qx", "rot[+2]": "qy", "rot[+3]": "qz",
              "scales[+0]": "sx", "scales[+1]": "sy", "scales[+2]": "sz", "karg[0x38]": "mod", "karg[0x78]": "fx", "karg[0x7c]": "fy",
              "karg[0x70]": "tanx", "karg[0x74]": "tany", "karg[0x68]": "W", "karg[0x6c]": "H"}
    return program(M.stores, picks, rename)


def compile_ours(tmp):
    from gaustar_amd import build
    asm = os.path.join(tmp, "pre.s")
    cmd = [HIPCC, *[f for f in build.FLAGS if f != "-Wall"], "-I", build.CSRC, "--offload-device-only", "-S",
           os.path.join(build.CSRC, "gsr_preprocess.hip"), "-o", asm]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return asm


def make_fixture():
    ref = os.path.join(os.environ.get("GSR_REFERENCE_ROOT", "/root/reference"), "gaussian_splatting", "submodules",
                       "diff-gaussian-rasterization")
    with tempfile.TemporaryDirectory() as tmp:   # the same translation oracle/build_ref.sh does; nothing of it is kept
        for f in os.listdir(os.path.join(ref, "cuda_rasterizer")):
            if f.endswith((".cu", ".h")):
                src = subprocess.run(["/opt/rocm/bin/hipify-perl", os.path.join(ref, "cuda_rasterizer", f)], capture_output=True, text=True).stdout
                src = "\n".join(l for l in src.split("\n") if '#include ""' not in l and "cooperative_groups/reduce.h" not in l
                                and "cub/device/device_radix_sort.cuh" not in l)
                import re
                src = re.sub(r"<< *<", "<<<", src)
                src = re.sub(r">> *>", ">>>", src)
                open(os.path.join(tmp, f), "w").write(src)
        asm = os.path.join(tmp, "forward.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-D__trap=__builtin_trap",
                        "-I" + os.path.join(ref, "third_party", "glm"), "-I" + tmp, "-x", "hip", "--offload-device-only", "-S",
                        os.path.join(tmp, "forward.cu"), "-o", asm], check=True)
        kargs = {0x10: "pts", 0x18: "scales", 0x28: "rot", 0x30: "opac", 0x48: "cov3Dpre", 0x58: "view", 0x60: "proj", 0x68: "campos"}
        lines = symfp.extract(asm, "_Z14preprocessCUDA")
        # the reference reads x, y, z of the point with three separate loads at computed addresses: name them by line
        M0 = symfp.run(lines, kargs)
        singles = [ln for ln, text in ((l, t.split(";")[0].strip()) for l, t in lines)
                   if text.startswith("global_load_dword ") and any(ln == l2 for l2, _ in lines)]
        M0 = None
        # (find them: the three single-dword loads in front of the first dwordx3 load of the scales)
        pts_loads = []
        for ln, raw in lines:
            t = raw.split(";")[0].strip()
            if t.startswith("global_load_dword ") and "off" in t and len(pts_loads) < 3 and ln > lines[0][0] + 60:
                pts_loads.append(ln)
        symfp.BY_LINE.update(pts_loads)
        M = symfp.run(lines, kargs)
        xy, conic = find_stores(M)
        picks = {"px": (xy[0], 0), "py": (xy[0], 1), "conic_a": (conic[0], 0), "conic_b": (conic[0], 1), "conic_c": (conic[0], 2)}
        rename = {f"load@{pts_loads[0]}[0]": "X", f"load@{pts_loads[1]}[0]": "Y", f"load@{pts_loads[2]}[0]": "Z",
                  "rot[+0]": "qr", "rot[+1]": "qx", "rot[+2]": "qy", "rot[+3]": "qz", "scales[+0]": "sx", "scales[+1]": "sy",
                  "scales[+2]": "sz", "karg[0x20]": "mod", "karg[0x80]": "fx", "karg[0x84]": "fy", "karg[0x78]": "tanx",
                  "karg[0x7c]": "tany", "karg[0x70]": "W", "karg[0x74]": "H"}
        text = program(M.stores, picks, rename)
    assert "load@" not in text and "?" not in text, "an input of the reference's program was not named"
    head = ("# Operation tree of {px, py, conic a, b, c} as the reference's preprocessCUDA computes them in its gfx950 build (hipcc -O3):\n"
            "# one IEEE operation per line over the inputs X Y Z (point), q* (quaternion), s* (scale), mod, view[], proj[], fx fy, tanx tany, W H.\n"
            "# Derived by tools/check_op_order.py --make-fixture (tools/symfp.py over the reference build's assembly); data, not source.\n")
    open(FIXTURE, "w").write(head + text)
    print(f"wrote {FIXTURE}: {text.count(chr(10))} lines")


def check():
    want = "".join(l for l in open(FIXTURE) if not l.startswith("#"))
    with tempfile.TemporaryDirectory() as tmp:
        asm = compile_ours(tmp)
        bad = 0
        for name, prefix in (("preprocess_kernel<false>", "_ZN3gsr17preprocess_kernelILb0EE"), ("preprocess_kernel<true>", "_ZN3gsr17preprocess_kernelILb1EE")):
            got = ours(prefix, asm)
            if got == want:
                print(f"{name}: operation tree of px, py, conic = the reference build's ({want.count(chr(10))} lines)")
            else:
                bad += 1
                import difflib
                d = list(difflib.unified_diff(want.split("\n"), got.split("\n"), "reference build", name, lineterm="", n=1))
                print(f"{name}: DIFFERS from the reference build's operation tree ({len(d)} diff lines):")
                print("\n".join(d[:60]))
    return bad


if __name__ == "__main__":
    if "--make-fixture" in sys.argv:
        make_fixture()
    else:
        sys.exit(1 if check() else 0)

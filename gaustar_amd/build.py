"""Build the HIP extension (libgsr_hip.so) for gfx950, in-tree, with plain hipcc.

The product of this script is the ONLY compute path of the package: there is no CPU or
PyTorch fallback (gaustar_amd._lib raises if the library is missing).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgsr_hip.so")
SOURCES = ["gsr_api.hip", "gsr_preprocess.hip", "gsr_binning.hip", "gsr_blend_fwd.hip", "gsr_blend_bwd.hip",
           "gsr_geom_bwd.hip", "gsr_loss.hip", "gsr_producers.hip", "gsr_optim.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: left on, clang packs neighbouring scalar f32 operations of the per-pair loops into v_pk_*_f32 and
# pays for it in register moves and s_nops (blend_bwd 0.139 -> 0.136 ms, blend_fwd 0.097 -> 0.094 ms on config C).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-Wall",
         "-Wno-unused-function"]


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    d.append(os.path.join(HERE, "..", "include", "gsr.h"))
    return d


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = OUT, extra_sources=()) -> str:
    """extra_flags / out / extra_sources: experiment variants (e.g. -DGSR_EXP_..., tools/variants/*.hip) built next to the
    product library.  A variant source defines launch_blend_bwd_variant and is compiled with -DGSR_BWD_VARIANT."""
    if not force and not extra_flags and not extra_sources and not needs_build():
        return OUT
    objdir = os.path.join(HERE, "build", os.path.basename(out).replace(".so", ""))
    os.makedirs(objdir, exist_ok=True)

    if extra_sources:
        extra_flags = [*extra_flags, "-DGSR_BWD_VARIANT"]

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        path = src if os.path.isabs(src) else os.path.join(CSRC, src)
        cmd = [HIPCC, *FLAGS, *extra_flags, "-I", CSRC, "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)
        return obj

    sources = [*SOURCES, *(os.path.abspath(p) for p in extra_sources)]
    with ThreadPoolExecutor(max_workers=len(sources)) as ex:
        objs = list(ex.map(cc, sources))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m gaustar_amd.build --variant NAME [--with tools/variants/x.hip] -DFOO ...
        i = sys.argv.index("--variant")
        name, rest = sys.argv[i + 1], sys.argv[i + 2:]
        extra = []
        while "--with" in rest:
            j = rest.index("--with")
            extra.append(rest[j + 1])
            del rest[j:j + 2]
        print(build(force=True, extra_flags=rest, out=os.path.join(HERE, f"libgsr_hip_{name}.so"), extra_sources=extra))
    else:
        print(build(force="--force" in sys.argv, verbose=True))

"""The C-ABI boundary: include/gsr.h, the ctypes table and the built library must agree, the
library must export every declared symbol, and argument validation must fail loudly -- all
without a GPU (no compute is launched)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "gsr.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b([A-Za-z_][\w \*]*?)\b(gsr_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        n = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
        decls[name] = (ret, n, args)
    return decls


def test_header_declares_the_expected_entry_points():
    d = _declared()
    for name in ("gsr_abi_version", "gsr_last_error", "gsr_geom_bytes", "gsr_image_bytes", "gsr_binning_bytes",
                 "gsr_grad_scratch_bytes", "gsr_forward_stage1", "gsr_forward_stage2", "gsr_forward", "gsr_backward", "gsr_mark_visible",
                 "gsr_debug_export", "gsr_binning_bytes_mt", "gsr_forward_stage2_mt", "gsr_backward_mt"):
        assert name in d, name
    src = open(HEADER).read()
    # every entry point cites the reference interface it replaces
    assert src.count("rasterizer.h:") >= 3 and "rasterize_points.cu" in src
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)          # declarations only: comments may name pytorch3d / autograd
    assert "torch" not in code and "at::" not in code and "std::" not in code


def test_ctypes_table_matches_header():
    from gaustar_amd import _lib
    d = _declared()
    assert set(d) == set(_lib.SIGNATURES), set(d) ^ set(_lib.SIGNATURES)
    for name, (ret, n, args) in d.items():
        res, argtypes = _lib.SIGNATURES[name]
        assert len(argtypes) == n, f"{name}: header has {n} parameters, ctypes table {len(argtypes)}"
        # floats sit at the same positions on both sides
        hdr_float = [i for i, a in enumerate(args.split(",")) if re.match(r"\s*float\s+\w+$", a)]
        tab_float = [i for i, t in enumerate(argtypes) if t is ctypes.c_float]
        assert hdr_float == tab_float, f"{name}: float parameter positions differ"


def test_library_loads_and_exports_every_symbol(hip_lib):
    for name in _declared():
        assert hasattr(hip_lib, name), name
    assert hip_lib.gsr_abi_version() == 16


def test_scratch_sizes(hip_lib):
    # 44 B/Gaussian geometry state + 12 B SH-colour slot + what a 256-Gaussian preprocess workgroup leaves for scatter
    # (2 048 instance records of 16 B, a 512-slot tile table of 8 B, four counts), 8 B/pixel (+ alignment slack)
    g = hip_lib.gsr_geom_bytes(500_000)
    nwg = (500_000 + 255) // 256
    assert 56 * 500_000 + nwg * (16 * 2048 + 8 * 512 + 16) <= g <= 56 * 500_000 + nwg * (16 * 2048 + 8 * 512 + 16) + 8192
    i = hip_lib.gsr_image_bytes(1920, 1080)
    assert 8 * 1920 * 1080 <= i <= 8 * 1920 * 1080 + (8 + 64 + 12) * 8160 + 8192   # ranges, 2x8 shard counters, order, seg_off, totals + view tokens
    assert hip_lib.gsr_grad_scratch_bytes(500_000) == 48 * 500_000 + 256
    # per instance: 8 B key + 4 B id + the forward's 36-byte record in list order (16 + 16 + 4); per unit: a 16-byte unit
    # table entry + one 64-bit mask word and one snapshot per pixel of the tile; per PART of a list above 1 024 entries (at
    # most U / 8 + R / 1024 + 2 of them): 8 B in the part list + a 4-byte ticket + 4 + 16 B per pixel of the tile
    al = lambda x: (x + 255) & ~255
    def expect(R, U, sv=1, tail=4):
        parts = U // 8 + R // 1024 + 2
        return (al(8 * R) + al(4 * R) + 2 * al(16 * R) + al(16 * U) + al(8 * 256 * U) + al(8 * parts) + al(4 * parts) + al(4 * 256 * parts)
                + al(16 * sv * 256 * parts) + al(16 * sv * 256 * U) + al(tail * R) + 256)
    assert hip_lib.gsr_binning_bytes(1_000_000, 0) == expect(1_000_000, 0)
    assert hip_lib.gsr_binning_bytes(1_000_000, 1000) == expect(1_000_000, 1000)
    assert hip_lib.gsr_geom_bytes(0) > 0 and hip_lib.gsr_binning_bytes(0, 0) > 0
    # multi-target: 3 channels = the plain size; 6 channels double every per-pixel (T, C) record (two float4 instead of one)
    # and carry four more colours in the record tail (16 instead of 4 bytes per instance)
    assert hip_lib.gsr_binning_bytes_mt(1_000_000, 1000, 3) == hip_lib.gsr_binning_bytes(1_000_000, 1000)
    assert hip_lib.gsr_binning_bytes_mt(1_000_000, 1000, 6) == expect(1_000_000, 1000, sv=2, tail=16)


def test_validation_errors_without_gpu(hip_lib):
    R, mx, ns = ctypes.c_int(7), ctypes.c_int(7), ctypes.c_int(7)
    null = None
    rc = hip_lib.gsr_forward_stage1(10, 0, 0, null, null, null, null, null, 1.0, null, null, null, null, null, 64, 64,
                                    0.5, 0.5, 0, null, null, null, ctypes.byref(R), ctypes.byref(mx), ctypes.byref(ns), null)
    assert rc != 0 and b"null" in hip_lib.gsr_last_error()
    rc = hip_lib.gsr_forward_stage1(10, 0, 0, null, null, null, null, null, 1.0, null, null, null, null, null, 0, 64,
                                    0.5, 0.5, 0, null, null, null, ctypes.byref(R), ctypes.byref(mx), ctypes.byref(ns), null)
    assert rc != 0 and b"positive" in hip_lib.gsr_last_error()
    assert R.value == 0
    rc = hip_lib.gsr_backward(5, 0, 0, 0, 0, null, 8, 8, *([null] * 4), 1.0, *([null] * 5), 0.5, 0.5, *([null] * 15))
    assert rc != 0 and hip_lib.gsr_last_error()
    # P == 0 backward is a no-op success (rasterize_points.cu:161)
    assert hip_lib.gsr_backward(0, 0, 0, 0, 0, null, 8, 8, *([null] * 4), 1.0, *([null] * 5), 0.5, 0.5, *([null] * 15)) == 0


def test_fused_forward_and_new_entry_points_validate_without_gpu(hip_lib):
    """gsr_forward_fused / gsr_backward_mt / gsr_sh_to_rgbd / gsr_adam_step: argument checks come before any device work."""
    null = None
    R, mx, ns, bl = ctypes.c_int(7), ctypes.c_int(7), ctypes.c_int(7), ctypes.c_int(7)
    outs = (ctypes.byref(R), ctypes.byref(mx), ctypes.byref(ns))
    # P > 0 with null arrays: stage 1's own validation fires, nothing was blended
    rc = hip_lib.gsr_forward_fused(10, 0, 0, 3, 1, *([null] * 5), 1.0, *([null] * 5), 64, 64, 0.5, 0.5, 0, *([null] * 5), 0, null,
                                   null, *outs, ctypes.byref(bl), null)
    assert rc != 0 and b"null" in hip_lib.gsr_last_error() and bl.value == 0 and R.value == 0
    rc = hip_lib.gsr_forward_fused(10, 0, 0, 5, 1, *([null] * 5), 1.0, *([null] * 5), 64, 64, 0.5, 0.5, 0, *([null] * 5), 0, null,
                                   null, *outs, ctypes.byref(bl), null)
    assert rc != 0 and b"num_channels" in hip_lib.gsr_last_error()
    rc = hip_lib.gsr_forward_fused(10, 0, 0, 3, 1, *([null] * 5), 1.0, *([null] * 5), 64, 64, 0.5, 0.5, 0, *([null] * 5), 0, null,
                                   null, *outs, null, null)
    assert rc != 0 and b"null output" in hip_lib.gsr_last_error()
    # planned forward (ABI 13): its three plan pointers are required; with them and null arrays the same validation as fused fires
    pl = ctypes.c_int(7)
    rc = hip_lib.gsr_forward_planned(10, 0, 0, 3, 1, *([null] * 5), 1.0, *([null] * 5), 64, 64, 0.5, 0.5, 0, *([null] * 5), 0, null,
                                     null, *outs, ctypes.byref(bl), null, None, ctypes.byref(pl), null)
    assert rc != 0 and b"null plan" in hip_lib.gsr_last_error()
    assert hip_lib.gsr_plan_bytes(1920, 1080) >= 16 * 8160 and hip_lib.gsr_plan_bytes(0, 5) == 0
    # backward_mt: the extra flag does not relax the pointer checks; P == 0 stays a no-op success
    rc = hip_lib.gsr_backward_mt(5, 0, 0, 0, 0, 3, null, 8, 8, *([null] * 4), 1.0, *([null] * 5), 0.5, 0.5, *([null] * 14), 1, null)
    assert rc != 0 and hip_lib.gsr_last_error()
    assert hip_lib.gsr_backward_mt(0, 0, 0, 0, 0, 3, null, 8, 8, *([null] * 4), 1.0, *([null] * 5), 0.5, 0.5, *([null] * 14), 1, null) == 0
    assert hip_lib.gsr_sh_to_rgbd(4, 0, 1, *([null] * 4), 3, null, null) != 0 and b"null" in hip_lib.gsr_last_error()
    assert hip_lib.gsr_sh_to_rgbd(4, 0, 1, *([null] * 4), 2, null, null) != 0 and b"depth_channels" in hip_lib.gsr_last_error()
    assert hip_lib.gsr_sh_to_rgbd(0, 0, 1, *([null] * 4), 1, null, null) == 0
    assert hip_lib.gsr_adam_step(8, *([null] * 4), 1e-3, 0.9, 0.999, 1e-8, 1, null) != 0 and b"null" in hip_lib.gsr_last_error()
    assert hip_lib.gsr_adam_step(0, *([null] * 4), 1e-3, 0.9, 0.999, 1e-8, 1, null) == 0
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert hip_lib.gsr_adam_step(8, p, p, p, p, 1e-3, 0.9, 0.999, 1e-8, 0, null) != 0 and b"step" in hip_lib.gsr_last_error()

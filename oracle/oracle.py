"""ctypes/numpy front end of oracle/gsr_oracle.c (the CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (gaustar_amd/) never imports this.

`forward()` / `backward()` mirror the two entry points of the reference binding
(DGR/rasterize_points.cu:35-115 and :117-196): same argument meaning, numpy arrays
instead of torch tensors, "absent" optionals passed as None.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_float, c_int, c_uint8, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "gsr_oracle.c")
_SO = os.path.join(_HERE, "_build", "libgsr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (OpenMP, no FMA contraction)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-o", _SO, _SRC, "-lm"]
        subprocess.run(cmd, check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.gsr_oracle_preprocess.restype = c_int
    return _lib


def set_threads(n: int) -> None:
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def tile_grid(W, H):
    return (W + 15) // 16, (H + 15) // 16


def forward(means3D, opacities, view, proj, campos, W, H, tanfovx, tanfovy, bg, *, shs=None, colors_precomp=None,
            scales=None, rotations=None, cov3D_precomp=None, sh_degree=0, scale_modifier=1.0):
    """Returns a dict with color[3,H,W], radii[P] and every intermediate the reference keeps
    in its geometry/binning/image buffers (rasterizer_impl.h:31-63)."""
    L = lib()
    means3D = _f(means3D).reshape(-1, 3)
    P = means3D.shape[0]
    opacities = _f(opacities).reshape(-1)
    view, proj, campos, bg = _f(view).reshape(16), _f(proj).reshape(16), _f(campos).reshape(3), _f(bg).reshape(3)
    shs, colors_precomp = _f(shs), _f(colors_precomp)
    scales, rotations, cov3D_precomp = _f(scales), _f(rotations), _f(cov3D_precomp)
    M = 0 if shs is None else shs.shape[1]
    st = dict(P=P, W=W, H=H, M=M, D=sh_degree)
    st["radii"] = np.zeros(P, np.int32)
    st["means2D"] = np.zeros((P, 2), np.float32)
    st["depths"] = np.zeros(P, np.float32)
    st["cov3D"] = np.zeros((P, 6), np.float32)
    st["rgb"] = np.zeros((P, 3), np.float32)
    st["conic_opacity"] = np.zeros((P, 4), np.float32)
    st["tiles_touched"] = np.zeros(P, np.uint32)
    st["clamped"] = np.zeros((P, 3), np.uint8)
    gx, gy = tile_grid(W, H)
    R = 0
    if P:
        R = L.gsr_oracle_preprocess(
            c_int(P), c_int(sh_degree), c_int(M), _p(means3D), _p(scales), c_float(scale_modifier), _p(rotations),
            _p(opacities), _p(shs), _p(cov3D_precomp), _p(colors_precomp), _p(view), _p(proj), _p(campos),
            c_int(W), c_int(H), c_float(tanfovx), c_float(tanfovy), _p(st["radii"]), _p(st["means2D"]),
            _p(st["depths"]), _p(st["cov3D"]), _p(st["rgb"]), _p(st["conic_opacity"]), _p(st["tiles_touched"]),
            _p(st["clamped"]))
    st["num_rendered"] = R
    st["keys"] = np.zeros(max(R, 1), np.uint64)
    st["point_list"] = np.zeros(max(R, 1), np.uint32)
    st["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    if P:
        L.gsr_oracle_bin(c_int(P), c_int(W), c_int(H), _p(st["means2D"]), _p(st["depths"]), _p(st["radii"]),
                         c_int(R), _p(st["keys"]), _p(st["point_list"]), _p(st["ranges"]))
    feats = colors_precomp if colors_precomp is not None else st["rgb"]
    st["final_T"] = np.zeros((H, W), np.float32)
    st["n_contrib"] = np.zeros((H, W), np.uint32)
    st["color"] = np.zeros((3, H, W), np.float32)
    if P:
        L.gsr_oracle_render_fwd(c_int(W), c_int(H), _p(st["ranges"]), _p(st["point_list"]), _p(st["means2D"]),
                                _p(feats), _p(st["conic_opacity"]), _p(bg), _p(st["final_T"]), _p(st["n_contrib"]),
                                _p(st["color"]))
    st["_in"] = dict(means3D=means3D, view=view, proj=proj, campos=campos, bg=bg, shs=shs,
                     colors_precomp=colors_precomp, scales=scales, rotations=rotations,
                     cov3D_precomp=cov3D_precomp, tanfovx=tanfovx, tanfovy=tanfovy, scale_modifier=scale_modifier)
    return st


def backward(st, dL_dout_color):
    """Gradients in the reference binding's return order (rasterize_points.cu:195):
    dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations
    (+ dL_dconic for stage-level checks)."""
    L = lib()
    i = st["_in"]
    P, W, H, M, D = st["P"], st["W"], st["H"], st["M"], st["D"]
    dpix = _f(dL_dout_color).reshape(3, H, W)
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
        dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
        dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32),
        dL_dconic=np.zeros((P, 4), np.float32))
    if P == 0:
        return g
    feats = i["colors_precomp"] if i["colors_precomp"] is not None else st["rgb"]
    L.gsr_oracle_render_bwd(c_int(P), c_int(W), c_int(H), _p(st["ranges"]), _p(st["point_list"]), _p(i["bg"]),
                            _p(st["means2D"]), _p(st["conic_opacity"]), _p(feats), _p(st["final_T"]),
                            _p(st["n_contrib"]), _p(dpix), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]),
                            _p(g["dL_dopacity"]), _p(g["dL_dcolors"]))
    cov3D = i["cov3D_precomp"] if i["cov3D_precomp"] is not None else st["cov3D"]
    L.gsr_oracle_preprocess_bwd(
        c_int(P), c_int(D), c_int(M), _p(i["means3D"]), _p(st["radii"]), _p(i["shs"]), _p(st["clamped"]),
        _p(i["scales"]), _p(i["rotations"]), c_float(i["scale_modifier"]), _p(cov3D), _p(i["view"]), _p(i["proj"]),
        c_int(W), c_int(H), c_float(i["tanfovx"]), c_float(i["tanfovy"]), _p(i["campos"]), _p(g["dL_dmeans2D"]),
        _p(g["dL_dconic"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcolors"]), _p(g["dL_dcov3D"]), _p(g["dL_dsh"]),
        _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def mark_visible(means3D, view, proj):
    means3D = _f(means3D).reshape(-1, 3)
    out = np.zeros(means3D.shape[0], np.uint8)
    if means3D.shape[0]:
        lib().gsr_oracle_mark_visible(c_int(means3D.shape[0]), _p(means3D), _p(_f(view).reshape(16)),
                                      _p(_f(proj).reshape(16)), _p(out))
    return out.astype(bool)

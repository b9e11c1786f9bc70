"""ctypes front end of oracle/_ref/libgsr_ref.so -- the REFERENCE rasterizer itself, built for
gfx950 by oracle/build_ref.sh from the sources under /root/reference.

TEST INFRASTRUCTURE ONLY (tests/, tests/golden/make_golden.py, bench.py's comparison leg).
Needs a GPU; tensors are torch HIP tensors.  Argument meaning follows
DGR/rasterize_points.cu:35-196; absent optionals are None.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_void_p

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libgsr_ref.so")
_lib = None


def available() -> bool:
    return os.path.exists(SO)


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(SO)
        L.ref_create.restype = c_void_p
        L.ref_destroy.argtypes = [c_void_p]
        L.ref_forward.restype = c_int
        L.ref_forward.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_float, c_float, c_int, c_void_p, c_void_p]
        L.ref_backward.restype = None
        L.ref_backward.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p,
                                   c_void_p] + [c_void_p] * 9
        L.ref_mark_visible.restype = None
        L.ref_mark_visible.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p]
        L.ref_state_ptr.restype = c_void_p
        L.ref_state_ptr.argtypes = [c_void_p, c_int]
        _lib = L
    return _lib


def _p(t):
    return None if t is None or t.numel() == 0 else c_void_p(t.data_ptr())


def _f(t, dev):
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
    return t.to(dev, torch.float32).contiguous()


class RefRasterizer:
    """One reference forward (+ optional backward) with access to the reference's scratch."""

    def __init__(self, device="cuda:0"):
        self.dev = torch.device(device)
        self.h = c_void_p(lib().ref_create())
        self.keep = None

    def __del__(self):
        try:
            if self.h:
                lib().ref_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def forward(self, means3D, opacities, view, proj, campos, W, H, tanfovx, tanfovy, bg, *, shs=None,
                colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=0,
                scale_modifier=1.0):
        d = self.dev
        a = dict(means3D=_f(means3D, d).reshape(-1, 3), opacities=_f(opacities, d).reshape(-1), view=_f(view, d),
                 proj=_f(proj, d), campos=_f(campos, d).reshape(-1), bg=_f(bg, d), shs=_f(shs, d),
                 colors=_f(colors_precomp, d), scales=_f(scales, d), rotations=_f(rotations, d),
                 cov3D=_f(cov3D_precomp, d))
        P = a["means3D"].shape[0]
        M = 0 if a["shs"] is None else a["shs"].shape[1]
        self.P, self.W, self.H, self.M, self.D = P, W, H, M, sh_degree
        self.tan = (float(tanfovx), float(tanfovy))
        self.scale_modifier = float(scale_modifier)
        color = torch.zeros(3, H, W, device=d)
        radii = torch.zeros(P, dtype=torch.int32, device=d)
        torch.cuda.synchronize()
        with torch.cuda.device(d):
            R = lib().ref_forward(self.h, P, sh_degree, M, _p(a["bg"]), W, H, _p(a["means3D"]), _p(a["shs"]),
                                  _p(a["colors"]), _p(a["opacities"]), _p(a["scales"]), self.scale_modifier,
                                  _p(a["rotations"]), _p(a["cov3D"]), _p(a["view"]), _p(a["proj"]), _p(a["campos"]),
                                  self.tan[0], self.tan[1], 0, _p(color), _p(radii))
        self.keep, self.R, self.radii = a, R, radii
        return color, radii, R

    def backward(self, dL_dpix):
        d, a, P, M = self.dev, self.keep, self.P, self.M
        dpix = _f(dL_dpix, d)
        z = lambda *s: torch.zeros(*s, device=d)
        g = dict(dL_dmeans2D=z(P, 3), dL_dconic=z(P, 4), dL_dopacity=z(P, 1), dL_dcolors=z(P, 3), dL_dmeans3D=z(P, 3),
                 dL_dcov3D=z(P, 6), dL_dsh=z(P, M, 3), dL_dscales=z(P, 3), dL_drotations=z(P, 4))
        torch.cuda.synchronize()
        with torch.cuda.device(d):
            lib().ref_backward(self.h, self.D, M, _p(a["bg"]), _p(a["means3D"]), _p(a["shs"]), _p(a["colors"]),
                               _p(a["scales"]), self.scale_modifier, _p(a["rotations"]), _p(a["cov3D"]), _p(a["view"]),
                               _p(a["proj"]), _p(a["campos"]), self.tan[0], self.tan[1], _p(self.radii), _p(dpix),
                               _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]),
                               _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), c_void_p(g["dL_dsh"].data_ptr()) if M else None,
                               _p(g["dL_dscales"]), _p(g["dL_drotations"]))
        return g

    def state(self):
        """Copies of the reference's scratch contents (rasterizer_impl.h:31-63) as numpy arrays."""
        P, W, H, R = self.P, self.W, self.H, self.R
        T = ((W + 15) // 16) * ((H + 15) // 16)
        spec = {"depths": (0, np.float32, (P,)), "clamped": (1, np.uint8, (P, 3)), "means2D": (2, np.float32, (P, 2)),
                "cov3D": (3, np.float32, (P, 6)), "conic_opacity": (4, np.float32, (P, 4)),
                "rgb": (5, np.float32, (P, 3)), "tiles_touched": (6, np.uint32, (P,)),
                "final_T": (8, np.float32, (H, W)), "n_contrib": (9, np.uint32, (H, W)),
                "ranges": (10, np.uint32, (T, 2)), "point_list": (11, np.uint32, (R,)),
                "keys": (12, np.uint64, (R,))}
        out = {}
        hip = ctypes.CDLL("libamdhip64.so")
        for name, (which, dt, shape) in spec.items():
            n = int(np.prod(shape))
            arr = np.zeros(shape, dt)
            ptr = lib().ref_state_ptr(self.h, which)
            if n and ptr:
                rc = hip.hipMemcpy(c_void_p(arr.ctypes.data), c_void_p(ptr), ctypes.c_size_t(arr.nbytes), c_int(2))
                assert rc == 0, f"hipMemcpy D2H failed ({rc})"
            out[name] = arr
        return out


def mark_visible(means3D, view, proj, device="cuda:0"):
    d = torch.device(device)
    m, v, p = _f(means3D, d).reshape(-1, 3), _f(view, d), _f(proj, d)
    out = torch.zeros(m.shape[0], dtype=torch.bool, device=d)
    with torch.cuda.device(d):
        lib().ref_mark_visible(m.shape[0], _p(m), _p(v), _p(p), c_void_p(out.data_ptr()))
    return out

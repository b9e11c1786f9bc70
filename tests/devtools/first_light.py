"""Bring-up check on the GPU box: HIP path vs the reference build (oracle/_ref) on the golden cases
and on config B/C, printing errors and timings instead of asserting.  Dev tool, not a test."""
from __future__ import annotations

import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402

import parity  # noqa: E402
from gaustar_amd import scene  # noqa: E402
from make_golden import cases, kwargs_of  # noqa: E402
from oracle import ref  # noqa: E402


def err_stats(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64).reshape(np.asarray(a).shape)
    if a.size == 0:
        return "empty"
    m = np.abs(b).max()
    e = np.abs(a - b)
    return f"max|ref|={m:.3e} maxerr={e.max():.3e} norm={e.max() / (m + 1e-30):.2e} n>1e-4*max={(e > 1e-4 * max(m, 1e-30)).sum()}/{e.size}"


def main():
    print("device:", torch.cuda.get_device_name(0), flush=True)
    for name, (gs, cam, bg, sm) in cases().items():
        kw = kwargs_of(gs, cam, bg, sm)
        try:
            rr = ref.RefRasterizer()
            color, radii, R = rr.forward(**kw)
            dpix = np.random.default_rng(1).normal(size=(3, cam.H, cam.W)).astype(np.float32)
            g = rr.backward(dpix)
            hip = parity.run_hip(kw, dpix)
            print(f"== {name}: P={gs.P} R_ref={R} radii_mismatch={(hip['radii'] != radii.cpu().numpy()).sum()}")
            print("   color ", err_stats(hip["color"], color.cpu().numpy()))
            for k in parity.GRAD_KEYS:
                if k in hip["_has"] and not hip["_has"][k]:
                    continue
                print(f"   {k:14s}", err_stats(hip[k], g[k].cpu().numpy()))
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()

    # timing on config B and C (view 0)
    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    for label, cfg in (("B", scene.config_B), ("C", scene.config_C)):
        gs, cam, bg = cfg()
        if isinstance(cam, list):
            cam = cam[0]
        kw = kwargs_of(gs, cam, bg, 1.0)
        rr = ref.RefRasterizer()
        color_r, radii_r, R = rr.forward(**kw)
        dpix = np.random.default_rng(2).normal(size=(3, cam.H, cam.W)).astype(np.float32)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            rr.forward(**kw)
            torch.cuda.synchronize()
        t1 = time.time()
        g_r = rr.backward(dpix)
        torch.cuda.synchronize()
        t2 = time.time()
        for _ in range(5):
            rr.backward(dpix)
        torch.cuda.synchronize()
        t3 = time.time()
        print(f"== config {label}: P={gs.P} R_ref={R} ref fwd {(t1 - t0) / 5 * 1e3:.2f} ms  bwd {(t3 - t2) / 5 * 1e3:.2f} ms (incl. H2D of inputs)", flush=True)
        hip = parity.run_hip(kw, dpix)
        print("   radii mismatch", (hip["radii"] != radii_r.cpu().numpy()).sum())
        print("   color ", err_stats(hip["color"], color_r.cpu().numpy()))
        for k in parity.GRAD_KEYS:
            if k in hip["_has"] and not hip["_has"][k]:
                continue
            print(f"   {k:14s}", err_stats(hip[k], g_r[k].cpu().numpy()))
        # HIP timing with resident tensors
        t = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
        means3D = t(gs.means3D).requires_grad_(True)
        opac = t(gs.opacities).requires_grad_(True)
        cols = t(gs.colors_precomp).requires_grad_(True)
        scales = t(gs.scales).requires_grad_(True)
        rots = t(gs.rotations).requires_grad_(True)
        means2D = torch.zeros(gs.P, 3, device=dev, requires_grad=True)
        settings = GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, t(bg), 1.0, t(cam.viewmatrix),
                                                 t(cam.projmatrix), 0, t(cam.campos), False, False)
        rast = GaussianRasterizer(settings)
        dp = t(dpix)
        for it in range(3):
            c, r = rast(means3D, means2D, opac, None, cols, scales, rots, None)
            c.backward(dp)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        fw = bw = 0.0
        N = 10
        for it in range(N):
            e0.record()
            c, r = rast(means3D, means2D, opac, None, cols, scales, rots, None)
            e1.record()
            c.backward(dp)
            e2.record()
            torch.cuda.synchronize()
            fw += e0.elapsed_time(e1)
            bw += e1.elapsed_time(e2)
        print(f"   HIP fwd {fw / N:.3f} ms  bwd {bw / N:.3f} ms  -> {1e3 / ((fw + bw) / N):.1f} views/s", flush=True)


if __name__ == "__main__":
    main()

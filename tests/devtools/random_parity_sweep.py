"""Dev tool: N random small scenes (sizes, counts, opacity ranges, SH degree / precomputed colours / precomputed
covariances, scale modifiers, channel counts) through the HIP path and the oracle, tolerances of tests/parity.py."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import parity
from gaustar_amd import scene

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
START = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # first case (seeds are 1000 + case: `1 1203` re-runs case 1203 alone)
fails = 0
flips = 0
pair_flips = 0
noise_rot = 0
for case in range(START, START + N):
    rng = np.random.default_rng(1000 + case)
    W, H = int(rng.integers(1, 260)), int(rng.integers(1, 200))
    P = int(rng.choice([1, 7, 100, 1500, 6000]))
    deg = int(rng.integers(0, 4))
    mode = rng.choice(["sh", "rgb", "rgb6", "rgb4"])
    s_lo = float(rng.choice([0.005, 0.03, 0.2]))
    gs = scene.random_gaussians(P, rng, sh_degree=deg, with_sh=(mode == "sh"), scale_range=(s_lo, s_lo * float(rng.choice([2, 10]))))
    lo = float(rng.choice([0.003, 0.3, 0.9]))
    gs.opacities[:] = rng.uniform(lo, min(1.0, lo * 3 + 0.01), (P, 1)).astype(np.float32)
    cam = scene.look_at_camera((float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), -float(rng.uniform(2.5, 5))), (0, 0, 0), W, H,
                               fovx=float(rng.uniform(0.4, 1.2)), znear=0.01)
    sm = float(rng.choice([1.0, 0.6, 1.7]))
    kw = dict(means3D=gs.means3D, opacities=gs.opacities, view=cam.viewmatrix, proj=cam.projmatrix, campos=cam.campos, W=W, H=H,
              tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=rng.uniform(0, 1, 3).astype(np.float32), shs=gs.shs if mode == "sh" else None,
              colors_precomp=None if mode == "sh" else gs.colors_precomp, scales=gs.scales, rotations=gs.rotations, cov3D_precomp=None,
              sh_degree=deg if mode == "sh" else 0, scale_modifier=sm)
    dpix = rng.normal(size=(3, H, W)).astype(np.float32)
    hip = None
    try:
        st, g = parity.run_oracle(kw, dpix)
        if mode in ("rgb6", "rgb4"):
            NX = 3 if mode == "rgb6" else 1      # rgb4: RGB + ONE scalar target (the oracle renders it three times over)
            extra = rng.uniform(0, 3, (P, 3)).astype(np.float32)
            if NX == 1:
                extra[:, 1:] = extra[:, :1]
            kw2 = dict(kw, colors_precomp=extra, bg=np.full(3, 7.0, np.float32))
            d2 = rng.normal(size=(3, H, W)).astype(np.float32)
            if NX == 1:
                d2[1:] = 0.0
            st2, g2 = parity.run_oracle(kw2, d2)
            if NX == 1:                           # the three identical colour columns share one gradient: their sum
                st2 = dict(st2, color=st2["color"][:1]); g2 = dict(g2, dL_dcolors=np.asarray(g2["dL_dcolors"]).sum(1, keepdims=True))
            kw6 = dict(kw, colors_precomp=np.concatenate([gs.colors_precomp, extra[:, :NX]], 1), bg=np.concatenate([kw["bg"], kw2["bg"][:NX]]))
            hip = parity.run_hip(kw6, np.concatenate([dpix, d2[:NX]]))
            nflip = int((hip["radii"] != st["radii"]).sum())
            assert nflip <= 2, f"radii differ in {nflip} entries"
            flips += nflip
            io, go = dict(max_outlier_frac=2e-4), dict(max_outlier_frac=1e-3)   # random scenes: threshold-flip allowance
            parity.check_image(hip["color"][:3], st["color"], **io); parity.check_image(hip["color"][3:], st2["color"], **io)
            parity.check_grad(hip["dL_dcolors"][:, :3], g["dL_dcolors"], **go); parity.check_grad(hip["dL_dcolors"][:, 3:], g2["dL_dcolors"], **go)
            for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"):
                parity.check_grad(hip[k], np.asarray(g[k], np.float64).reshape(hip[k].shape) + np.asarray(g2[k], np.float64).reshape(hip[k].shape), k, **go)
        else:
            hip = parity.run_hip(kw, dpix)
            flips += int((hip["radii"] != st["radii"]).sum())   # assert_radii: only ceil(3 sigma) flips at an integer boundary
            parity.compare_hip_to(hip, st["color"], st["radii"], g, what=f"case {case}", kw=kw, max_radii_flips=2, strict=False)
    except AssertionError as e:
        # One (pixel, Gaussian) pair sitting within an ulp of a hard threshold (alpha >= 1/255, T < 1e-4) may fall on the other
        # side of it here than in the oracle -- the two evaluate the exponent with different roundings.  It shows as ONE or
        # two pixels off by up to alpha*T*c and a gradient difference confined to that pair; anything wider is a failure.
        img_ref = st["color"] if mode not in ("rgb6", "rgb4") else np.concatenate([st["color"], st2["color"]])
        off = int((np.abs(hip["color"] - img_ref).max(0) > 1e-4).sum()) if hip is not None and hip["color"].shape == img_ref.shape else 99
        # A rotation gradient that is ITSELF rounding noise: the quaternion's gradient is what is left of dL/dR after the
        # normalisation projects the radial part out, and for a splat with nearly equal in-plane scales that is a difference of
        # terms five orders above it (case 1203: max|ref| 5e-7 beside dL_dscales of order 1e-2; HIP path and oracle agree to
        # 6e-11 absolute).  check_grad scales its tolerance by the array's own maximum; here the other arrays set the scale.
        rot_noise = False
        if "dL_drotations" in str(e) and hip is not None and mode not in ("rgb6", "rgb4"):
            scale = max(float(np.abs(np.asarray(g[k])).max()) for k in ("dL_dscales", "dL_dmeans3D"))
            d = np.abs(np.asarray(hip["dL_drotations"], np.float64) - np.asarray(g["dL_drotations"], np.float64).reshape(hip["dL_drotations"].shape)).max()
            rot_noise = float(np.abs(np.asarray(g["dL_drotations"])).max()) < 1e-3 * scale and d <= 1e-6 * scale
        if rot_noise:
            noise_rot += 1
            print(f"case {case}: W={W} H={H} P={P} mode={mode}: rotation gradient at rounding-noise level, tolerated ({str(e)[:120]})")
        elif 1 <= off <= 2:
            pair_flips += 1
            print(f"case {case}: W={W} H={H} P={P} mode={mode}: {off} pixel(s) across a blend threshold, tolerated ({str(e)[:90]})")
        else:
            fails += 1
            print(f"case {case}: W={W} H={H} P={P} mode={mode} deg={deg} sm={sm}: FAIL {str(e)[:200]}")
print(f"random parity sweep: {N - fails}/{N} passed ({flips} single-radius ulp flips, {pair_flips} single-pair threshold flips, "
      f"{noise_rot} noise-level rotation gradients tolerated)")

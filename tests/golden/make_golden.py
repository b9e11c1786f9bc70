"""Generate tests/golden/*.npz: golden input/output vectors of the REFERENCE rasterizer.

The vectors are outputs of the reference's own kernels (oracle/_ref/libgsr_ref.so, built from
/root/reference by oracle/build_ref.sh) run on an MI355X:

    gpurun -- 'python tests/golden/make_golden.py gpurun_out/golden'      # on the GPU box
    cp gpurun_out/golden/*.npz tests/golden/                               # back in the repo

A fixture is data only: the seeded inputs, every forward output / intermediate the reference keeps
(rasterizer_impl.h:31-63) and, for a seeded dL_dpix, all gradient tensors.  The reference's float
atomics make gradients order-dependent in the last bits, so gradients carry
~1e-6 relative noise by construction.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gaustar_amd import scene  # noqa: E402


def _cam(W, H, fovx=0.9, eye=(0.0, 0.0, -4.0), target=(0.0, 0.0, 0.0)):
    return scene.look_at_camera(eye, target, W, H, fovx=fovx, znear=0.01)


def cases():
    """name -> (GaussianSet, Camera, bg, scale_modifier).  Deterministic."""
    out = {}
    for deg in (0, 1, 2, 3):
        rng = np.random.default_rng(100 + deg)
        gs = scene.random_gaussians(350, rng, sh_degree=deg, with_sh=True, scale_range=(0.02, 0.25))
        out[f"sh{deg}_random"] = (gs, _cam(100, 70), np.array([0.1, 0.5, 0.2], np.float32), 1.0)
    # SH buffer wider than the active degree (GauSTAR renders deg-2 out of a 16-coefficient buffer)
    rng = np.random.default_rng(105)
    gs = scene.random_gaussians(300, rng, sh_degree=3, with_sh=True, scale_range=(0.02, 0.2))
    gs.sh_degree = 2
    out["sh2_of_16"] = (gs, _cam(96, 80), np.array([0.0, 0.0, 0.0], np.float32), 1.0)

    rng = np.random.default_rng(7)
    gs = scene.random_gaussians(1500, rng, scale_range=(0.01, 0.12))
    out["colors_rgb"] = (gs, _cam(128, 128), np.array([0.0, 1.0, 0.0], np.float32), 1.0)

    # GauSTAR's depth-as-colour pass: colours = view depth x3, bg = 10, odd image size
    rng = np.random.default_rng(8)
    gs = scene.random_gaussians(600, rng, scale_range=(0.02, 0.2))
    cam = _cam(70, 45)
    gs.colors_precomp = scene.view_depth_colors(gs, cam)
    out["depth_color_bg10"] = (gs, cam, np.array([10.0, 10.0, 10.0], np.float32), 1.0)

    # precomputed 3D covariance
    rng = np.random.default_rng(9)
    gs = scene.random_gaussians(500, rng, scale_range=(0.02, 0.2))
    q, s = gs.rotations.astype(np.float64), gs.scales.astype(np.float64)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rm = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
                   np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
                   np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], 1)
    Sg = Rm @ (s[:, :, None] ** 2 * np.transpose(Rm, (0, 2, 1)))
    gs.cov3D_precomp = np.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).astype(np.float32)
    gs.scales = None
    gs.rotations = None
    out["cov3d_precomp"] = (gs, _cam(80, 64), np.array([0.3, 0.3, 0.3], np.float32), 1.0)

    # edge cases: behind the near plane, far off screen, sub-threshold opacity, opacity 1 (0.99 clamp),
    # huge splats (long lists, early termination), needle-thin splats, un-normalised quaternions
    rng = np.random.default_rng(10)
    gs = scene.random_gaussians(900, rng, scale_range=(0.02, 0.2))
    gs.means3D[:40, 2] = -4.5 + rng.uniform(-0.5, 0.5, 40).astype(np.float32)       # behind / at the camera
    gs.means3D[40:80, 0] += 30.0                                                     # off screen
    gs.opacities[80:120] = rng.uniform(0.0, 0.004, (40, 1)).astype(np.float32)       # < 1/255
    gs.opacities[120:220] = 1.0                                                      # alpha clamp
    gs.scales[220:260] = rng.uniform(0.8, 2.0, (40, 3)).astype(np.float32)           # huge
    gs.scales[260:300, 0] = 1e-6                                                     # thin
    gs.rotations[300:340] *= rng.uniform(0.5, 1.5, (40, 1)).astype(np.float32)        # raw quaternions
    out["edge_cases"] = (gs, _cam(112, 72), np.array([1.0, 0.0, 1.0], np.float32), 1.0)

    # mesh-bound surface Gaussians (SuGaR binding), camera as in the 1080p configs but small
    rng = np.random.default_rng(11)
    v, f = scene.icosphere(2, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
    gs = scene.mesh_bound_gaussians(v, f, rng, 3.5e-6)
    cam = scene.look_at_camera((0.4, 1.5, 3.0), scene.SUBJECT_CENTER, 160, 120, focal_px=130.0)
    out["mesh_sphere"] = (gs, cam, np.array([0.0, 1.0, 0.0], np.float32), 1.0)

    rng = np.random.default_rng(12)
    gs = scene.random_gaussians(400, rng, scale_range=(0.02, 0.2))
    out["scale_modifier"] = (gs, _cam(64, 64), np.array([0.2, 0.2, 0.2], np.float32), 0.6)
    return out


def kwargs_of(gs, cam, bg, scale_modifier):
    return dict(means3D=gs.means3D, opacities=gs.opacities, view=cam.viewmatrix, proj=cam.projmatrix,
                campos=cam.campos, W=cam.W, H=cam.H, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, shs=gs.shs,
                colors_precomp=gs.colors_precomp, scales=gs.scales, rotations=gs.rotations,
                cov3D_precomp=gs.cov3D_precomp, sh_degree=gs.sh_degree, scale_modifier=scale_modifier)


def main(outdir):
    from oracle import ref
    os.makedirs(outdir, exist_ok=True)
    for name, (gs, cam, bg, sm) in cases().items():
        kw = kwargs_of(gs, cam, bg, sm)
        rr = ref.RefRasterizer()
        color, radii, R = rr.forward(**kw)
        st = rr.state()
        rng = np.random.default_rng(sum(map(ord, name)))
        dpix = rng.normal(size=(3, cam.H, cam.W)).astype(np.float32)
        g = rr.backward(dpix)
        d = {f"in_{k}": (np.zeros(0, np.float32) if v is None else np.asarray(v)) for k, v in kw.items()}
        d.update(out_color=color.cpu().numpy(), out_radii=radii.cpu().numpy(), out_num_rendered=np.int64(R),
                 in_dL_dpix=dpix)
        vis = radii.cpu().numpy() > 0
        for k, v in st.items():
            d[f"state_{k}"] = v
        d["state_visible"] = vis
        for k, v in g.items():
            d[f"grad_{k}"] = v.cpu().numpy()
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **d)
        print(f"{name}: P={gs.P} R={R} visible={int(vis.sum())} |color|={float(color.abs().mean()):.4f}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden"))

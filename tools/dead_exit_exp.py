"""tools/dead_exit_exp.py [view] -- what would a per-(unit, block) live flag from the forward buy the backward blend?  (GPU; needs
python -m gaustar_amd.build --variant explive -DGSR_EXP_LIVE and GSR_LIB_PATH=gaustar_amd/libgsr_hip_explive.so.)  24 % of the
backward's waves find no work after their head (no candidate below the block's last contributors); in this build the kernel first
RECORDS which (unit, block) waves found work, then runs with a table that lets the others leave after one scalar load -- the upper bound
of VERDICT r5 task 6's first lever, measured with the library's HIP-event brackets over interleaved rounds."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from gaustar_amd import _lib, scene
from gaustar_amd import rasterizer as R

view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gs, cams, bg = scene.config_C()
cam = cams[view]
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
lib = _lib.load()
W, H = cam.W, cam.H
m3, op, cols, sc, rot = t(gs.means3D), t(gs.opacities), t(gs.colors_precomp), t(gs.scales), t(gs.rotations)
vm, pm, cp, bgt = t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos), t(bg)
e = torch.Tensor([])
dp = torch.randn(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
out = R.rasterize_gaussians_native(bgt, m3, cols, op, sc, rot, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, H, W, e, 0, cp, False, False, use_plan=False)
Rn, _, radii, geom, binning, img, maxc, U = out

def bwd():
    return R.rasterize_gaussians_backward_native(bgt, m3, radii, cols, sc, rot, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, dp, e, 0, cp, geom, Rn,
                                                 binning, img, False, num_segments=U)
ref = [g.clone() if g is not None else None for g in bwd()]
torch.cuda.synchronize()
tab = torch.zeros(U + 1, dtype=torch.int32, device=dev)
lib.gsr_debug_set_bwd_order(ctypes.c_void_p(tab.data_ptr()))
bwd(); torch.cuda.synchronize()                       # record
live = tab[:U].cpu().numpy()
n_live = int(np.bitwise_count(live.astype(np.uint32)).sum())
print(f"view {view}: U {U}, (unit, block) waves {4 * U}, with work {n_live} ({100.0 * n_live / (4 * U):.1f} %)")
tab[U] = 1
got = bwd(); torch.cuda.synchronize()
for a, b in zip(got, ref):
    if a is not None and a.numel():
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-30
nst = lib.gsr_num_stages()
names = [lib.gsr_stage_name(i).decode() for i in range(nst)]
ib = names.index("blend_bwd_kernel")
ms = (ctypes.c_float * nst)(); cnt = (ctypes.c_int * nst)()

def timed(mode, n=24):
    lib.gsr_debug_set_bwd_order(ctypes.c_void_p(tab.data_ptr()) if mode else None)
    for _ in range(2):
        bwd()
    torch.cuda.synchronize()
    lib.gsr_profile_enable(1); lib.gsr_profile_read(ms, cnt, 1)
    for _ in range(n):
        bwd()
    torch.cuda.synchronize()
    lib.gsr_profile_read(ms, cnt, 1); lib.gsr_profile_enable(0)
    lib.gsr_debug_set_bwd_order(None)
    return ms[ib] / max(cnt[ib], 1) * 1e3
res = {0: [], 1: []}
for r in range(6):
    for mode in ((0, 1) if r % 2 == 0 else (1, 0)):
        res[mode].append(round(timed(mode), 1))
print("blend_bwd us (HIP events): every wave runs its head", res[0], "median", float(np.median(res[0])))
print("blend_bwd us (HIP events): dead waves leave at once ", res[1], "median", float(np.median(res[1])))

"""tools/pre_phases.py [view] -- where the planned preprocess's time goes, from a -DGSR_PRE_PHASES build (thread 0 of every
workgroup stamps: start, state stored, own wave walked, all waves walked, cursor atomics returned, keys stored)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from gaustar_amd import _lib, scene
from gaustar_amd import rasterizer as R
view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
gs, cams, bg = scene.config_C(); cam = cams[view]
dev = torch.device("cuda:0"); lib = _lib.load()
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
args = (t(bg), t(gs.means3D), t(gs.colors_precomp), t(gs.opacities), t(gs.scales), t(gs.rotations), 1.0, torch.Tensor([]), t(cam.viewmatrix),
        t(cam.projmatrix), cam.tanfovx, cam.tanfovy, cam.H, cam.W, torch.Tensor([]), 0, t(cam.campos), False, False)
for _ in range(4):
    b = dict(R.PLAN_STATS)
    out = R.rasterize_gaussians_native(*args)
    torch.cuda.synchronize()
print("last view planned:", R.PLAN_STATS["planned"] - b["planned"])
P = gs.P
al = lambda x: (x + 255) & ~255
off = al(16 * P); off = al(off + 16 * P); off = al(off + 4 * P); off = al(off + 8 * P); off = al(off + 12 * P)
nwg = (P + 255) // 256
cap = (int(lib.gsr_geom_bytes(P)) - off) // nwg   # rough: records dominate; the record area of a workgroup is 16 * WG_REC_CAP bytes
geom = out[3].cpu().numpy()
REC = 16 * 2048
st = np.stack([geom[off + w * REC: off + w * REC + 48].view(np.uint64) for w in range(nwg)]).astype(np.float64) / 100.0
t0 = st[:, 0].min()
names = ["load + project + store", "walk (own wave)", "wait for the other waves", "cursor atomics", "key pass"]
d = np.diff(st, axis=1)
print(f"workgroups {nwg}; kernel span (first start to last end) {st[:, 5].max() - t0:.1f} us; start spread p50/p99 {np.percentile(st[:, 0] - t0, 50):.1f}/{np.percentile(st[:, 0] - t0, 99):.1f} us")
for k, n in enumerate(names):
    print(f"  {n:28s} mean {d[:, k].mean():5.2f} us  p50 {np.percentile(d[:, k], 50):5.2f}  p99 {np.percentile(d[:, k], 99):5.2f}")
print(f"  workgroup life mean {(st[:, 5] - st[:, 0]).mean():.2f} us, p99 {np.percentile(st[:, 5] - st[:, 0], 99):.2f}")
s0 = np.sort(st[:, 0] - t0)
print("start times us: ", [round(float(x), 1) for x in np.percentile(s0, [50, 75, 90, 93, 95, 97, 99, 100])], " workgroups starting after 5 us:", int((s0 > 5).sum()), " after 10 us:", int((s0 > 10).sum()))
late = np.nonzero(st[:, 0] - t0 > 5)[0]
print("late workgroups (index % 8 histogram):", np.bincount(late % 8, minlength=8).tolist(), " index range", (int(late.min()), int(late.max())) if len(late) else None)
e0 = st[:, 5] - t0
print("end times us p50/p90/p99/max:", [round(float(x), 1) for x in np.percentile(e0, [50, 90, 99, 100])])

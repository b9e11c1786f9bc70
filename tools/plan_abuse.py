"""tools/plan_abuse.py -- a plan is a HINT: render views under plans that were made for OTHER cameras / scenes (same image size,
same plan_key) and compare with the exact path.  GPU."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from gaustar_amd import scene
from gaustar_amd import rasterizer as R
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
e = torch.Tensor([])

def tensors(gs, cam):
    cols = t(gs.colors_precomp) if gs.colors_precomp is not None else t(scene.view_depth_colors(gs, cam))
    return dict(m3=t(gs.means3D), op=t(gs.opacities), sc=t(gs.scales), rot=t(gs.rotations), cols=cols,
                vm=t(cam.viewmatrix), pm=t(cam.projmatrix), cp=t(cam.campos), cam=cam)

def render(x, bg, key, use_plan=None, bwd=True):
    c = x["cam"]
    b = dict(R.PLAN_STATS)
    out = R.rasterize_gaussians_native(t(bg), x["m3"], x["cols"], x["op"], x["sc"], x["rot"], 1.0, e, x["vm"], x["pm"], c.tanfovx, c.tanfovy,
                                       c.H, c.W, e, 0, x["cp"], False, False, need_backward=bwd, use_plan=use_plan, plan_key=key)
    torch.cuda.synchronize()
    d = {k: R.PLAN_STATS[k] - b[k] for k in b}
    g = None
    if bwd:
        dp = torch.ones(3, c.H, c.W, device=dev)
        g = R.rasterize_gaussians_backward_native(t(bg), x["m3"], out[2], x["cols"], x["sc"], x["rot"], 1.0, e, x["vm"], x["pm"], c.tanfovx,
                                                  c.tanfovy, dp, e, 0, x["cp"], out[3], out[0], out[4], out[5], False, num_segments=out[7])
        torch.cuda.synchronize()
    return out[1], d, g

gsC, camsC, bgC = scene.config_C()
gsB, camB, bgB = scene.config_B()
srcs = {"C0": tensors(gsC, camsC[0]), "C90": tensors(gsC, camsC[90]), "C141": tensors(gsC, camsC[141]), "B": tensors(gsB, camB)}
v, f = scene.icosphere(3, scene.SUBJECT_RADIUS, scene.SUBJECT_CENTER)
gsS = scene.mesh_bound_gaussians(v, f, np.random.default_rng(0), 3.5e-6)
srcs["small"] = tensors(gsS, camsC[5])
tgts = {"C3": tensors(gsC, camsC[3]), "C70": tensors(gsC, camsC[70]), "small@C40": tensors(gsS, camsC[40]), "C0": srcs["C0"]}
for sn, sx in srcs.items():
    for tn, tx in tgts.items():
        R.drop_plans()
        key = ("abuse", sn, tn)
        for _ in range(3):
            render(sx, bgC, key)          # source view: exact, plan, planned
        ref, _, gref = render(tx, bgC, None, use_plan=False)
        for k in range(3):
            img, d, g = render(tx, bgC, key)
            same = torch.equal(img, ref)
            gerr = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(g, gref) if a is not None and a.numel())
            print(f"plan of {sn:6s} -> view {tn:10s} try {k}: {d}  image equal {same}  grad rel err {gerr:.2e}", flush=True)
            assert same and gerr < 1e-4
print("ok")

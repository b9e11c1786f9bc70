"""tools/bench_forward_only.py -- forward-only renders (no input requires grad: sweeps, evaluation) of config C."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gaustar_amd import dist as gdist
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); gdist.bind_to_local_cpus(0)
gs, cams, bg, params, means2D, rasters, dpix = bench.build_workload(dev, 0)
p = {k: v.detach() for k, v in params.items()}
m2 = means2D.detach()
def step(s):
    r = rasters[s % len(rasters)]
    with torch.no_grad():
        return r(means3D=p["means3D"], means2D=m2, opacities=p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
for s in range(170): step(s)
out = []
for rnd in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(160): step(s)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 160 * 1e3)
ms = sorted(out)[1]
print(json.dumps({"forward_only_ms_per_view": round(ms, 4), "views_per_s": round(1e3 / ms, 1)}))
# the same with two views in flight (gaustar_amd.pipelines: what ForwardSweep.sweep does by default)
from gaustar_amd import pipelines
pipes = pipelines.ViewPipelines(2, dev)
pipes.run(lambda t, s: step(s), list(range(20)))
clock = {}
pipes.run(lambda t, s: step(s), list(range(480)), before=lambda: clock.__setitem__("a", time.perf_counter()),
          after=lambda: clock.__setitem__("b", time.perf_counter()))
ms2 = (clock["b"] - clock["a"]) / 480 * 1e3
print(json.dumps({"forward_only_two_in_flight_ms_per_view": round(ms2, 4), "views_per_s": round(1e3 / ms2, 1)}))

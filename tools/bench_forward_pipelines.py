import json, os, sys, time
import torch
sys.path.insert(0, "/root/repo")
import bench
from gaustar_amd import dist as gdist, pipelines
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); gdist.bind_to_local_cpus(0)
gs, cams, bg, params, means2D, rasters, dpix = bench.build_workload(dev, 0)
p = {k: v.detach() for k, v in params.items()}
m2 = means2D.detach()
def step(s):
    r = rasters[s % len(rasters)]
    with torch.no_grad():
        return r(means3D=p["means3D"], means2D=m2, opacities=p["opacities"], colors_precomp=p["colors"], scales=p["scales"], rotations=p["rotations"])
for V in (1, 2, 3, 4):
    pipes = pipelines.ViewPipelines(V, dev)
    pipes.run(lambda t, s: step(s), list(range(40)))
    clock = {}
    pipes.run(lambda t, s: step(s), list(range(640)), before=lambda: clock.__setitem__("a", time.perf_counter()),
              after=lambda: clock.__setitem__("b", time.perf_counter()))
    ms2 = (clock["b"] - clock["a"]) / 640 * 1e3
    print(V, round(ms2, 4), round(1e3 / ms2, 1), flush=True)

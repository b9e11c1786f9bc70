import cProfile, pstats, sys, os, io
sys.argv = ["x"]
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import runpy
# run host_floor once to warm up and get the figure, then profile 2000 steps
ns = runpy.run_path("/root/repo/tools/host_floor.py")
step = ns["step"]; torch = ns["torch"]
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])

"""tools/profile_window.py -- host-side profile (cProfile, main thread) of tools/bench_window.py's loop at config-C size."""
import argparse, cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_window
a = argparse.Namespace(frames=2, iters=50, level=6, width=1920, height=1080, cameras=160)
print(bench_window.run(a))
pr = cProfile.Profile(); pr.enable()
r = bench_window.run(a)
pr.disable()
print(r)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])

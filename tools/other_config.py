import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import bench
from gaustar_amd import _lib
dev = torch.device("cuda:0")
for c in sys.argv[1:]:
    o = bench.other_config(c, dev, _lib.load(), steps=20, repeats=3)
    print(c, o["ms_per_view"], {k: round(v["ms_per_launch"] * 1e3, 1) for k, v in o["kernels"].items()})

"""tools/flip_kinds.py [every] [name] -> gpurun_out/<name, default r06_flip_kinds>.json (GSR_LIB_PATH picks the library): the threshold flips of every `every`-th view of config C's rig
against the reference build, attributed to the decision that can have flipped (oracle/rig_parity.py::classify_flips).  GPU."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaustar_amd import scene
from oracle import rig_parity
every = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gs, cams, bg = scene.config_C()
rows = rig_parity.compare_views(gs, cams, bg, range(0, len(cams), every), classify=True)
s = rig_parity.summarise(rows)
print(json.dumps(s, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
name = sys.argv[2] if len(sys.argv) > 2 else "r06_flip_kinds"
json.dump({"summary": s, "per_view": rows}, open(os.path.join(ROOT, "gpurun_out", name + ".json"), "w"))

"""tools/bench_cumask.py -- two (or more) independent views in flight with the CUs PARTITIONED between their streams
(hipExtStreamCreateWithCUMask) against the shared-chip pipelines bench.py's `value` uses.  Config C, fwd + bwd, planned binning."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from gaustar_amd import pipelines

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
from gaustar_amd import dist as gdist
print("pinned to", len(gdist.bind_to_local_cpus(0)), "cores")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
NCU = torch.cuda.get_device_properties(0).multi_processor_count
print("CUs", NCU)

def masked_stream(bits):
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)

gs, cams, bg, params, means2D, rasters, dpix = bench.build_workload(dev, 0)
STEPS = 40

def measure(streams, label, reps=5):
    V = len(streams)
    pipes = pipelines.ViewPipelines(V, dev)
    pipes.streams = streams
    leaves = pipelines.clone_leaves(dict(params, means2D=means2D), V)
    def step(t, s):
        ps = leaves[t]
        bench.one_step(s, 0, 1, {k: v for k, v in ps.items() if k != "means2D"}, ps["means2D"], rasters, dpix)
    pipes.run(step, list(range(STEPS)))       # plans of these streams' cameras
    pipes.run(step, list(range(STEPS)))
    clock = {}; dts = []
    for _ in range(reps):
        pipes.run(step, list(range(2 * V)))
        pipes.run(step, list(range(STEPS)), before=lambda: clock.__setitem__("t0", time.perf_counter()), after=lambda: clock.__setitem__("t1", time.perf_counter()))
        dts.append(clock["t1"] - clock["t0"])
    ms = float(np.median(dts)) / STEPS * 1e3
    print(f"{label:58s} {ms:.4f} ms per view  {1e3 / ms:7.1f} views/s   regions {[round(d / STEPS * 1e3, 4) for d in dts]}", flush=True)

allb = list(range(NCU))
xcd = lambda sel: [b for b in allb if (b % 8) in sel]
for rnd in range(2):
    measure([torch.cuda.Stream(dev)], "one view at a time")
    measure([torch.cuda.Stream(dev) for _ in range(2)], "two in flight, shared chip (bench.py)")
    measure([masked_stream(xcd({0, 1, 2, 3})), masked_stream(xcd({4, 5, 6, 7}))], "two in flight, XCDs 0-3 / 4-7")
    measure([masked_stream(xcd({0, 2, 4, 6})), masked_stream(xcd({1, 3, 5, 7}))], "two in flight, even / odd XCDs")
    measure([masked_stream(xcd({0, 1, 2})), masked_stream(xcd({3, 4, 5})), masked_stream(xcd({6, 7}))], "three in flight, XCDs 3 + 3 + 2")
    measure([masked_stream(xcd({0, 1, 2, 3, 4})), masked_stream(xcd({3, 4, 5, 6, 7}))], "two in flight, XCDs 0-4 / 3-7 (two shared)")
    measure([masked_stream(xcd({0, 1, 2, 3, 4, 5})), masked_stream(xcd({2, 3, 4, 5, 6, 7}))], "two in flight, XCDs 0-5 / 2-7 (four shared)")

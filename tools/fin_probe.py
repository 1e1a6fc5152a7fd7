"""Phase clocks of k_insert_par_fin (-DCC_FIN_STATS build as libcc_hip_finstats.so): what the kernel behind the block-parallel insertion spends its
~55 us on at 32 streams. usage: CC_HIP_LIB=libcc_hip_finstats.so python tools/fin_probe.py [streams ...]"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continuous_clustering_amd as cca
if os.environ.get("CC_HIP_LIB"):
    cca.LIB_PATH = os.path.join(os.path.dirname(cca.LIB_PATH), os.environ["CC_HIP_LIB"])
import bench
from continuous_clustering_amd import capi, synth
sizes = [int(a) for a in sys.argv[1:]] or [32, 64]
dev = torch.device("cuda:0")
ctx = bench.Ctx(torch, None, False, 1, 0, dev, 0, False)
sensor = synth.SensorModel(num_rows=64, num_columns=2200)
cfg = capi.Config.kitti()
L = cca.load_library(); L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for S in sizes:
    inputs = bench.gen_inputs(torch, dev, sensor, [1000 + j for j in range(S)], 2200, 23)
    r, e, _ = bench.run_throughput(ctx, sensor, cfg, list(range(S)), 2200, 20, 3, 0, inputs=inputs)
    out = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, 0, out.ctypes.data)
    n = float(out[12]) or 1.0
    print(f"streams {S}: {r['value']:.0f} Mpoints/s, {r['ms_per_step']:.3f} ms per step; launches of k_insert_par_fin seen {int(n)}")
    for name, v in zip(["entry -> state read, offsets, barrier", "thread 0: par_close_stream (the stream's state, the batch descriptor)", "wavefront 1: table_from_partials (from the barrier on)"], out[8:11]):
        print(f"  {name:72s} {v / n:10.0f} clocks = {v / n / 100:8.2f} us (100 MHz)")
    e.close()

"""Phase clocks of k_insert_par (wavefront 0 of a stream's first block): -DCC_IP_STATS build as libcc_hip_ipstats.so.
usage: CC_HIP_LIB=libcc_hip_ipstats.so python tools/ip_probe.py [streams ...]"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import continuous_clustering_amd as cca
from continuous_clustering_amd import capi, synth
sizes = [int(a) for a in sys.argv[1:]] or [32, 256]
dev = torch.device("cuda:0")
ctx = bench.Ctx(torch, None, False, 1, 0, dev, 0, False)
sensor = synth.SensorModel(num_rows=64, num_columns=2200)
cfg = capi.Config.kitti()
L = cca.load_library(); L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for S in sizes:
    inputs = bench.gen_inputs(torch, dev, sensor, [1000 + j for j in range(S)], 2200, 15)
    r, e, _ = bench.run_throughput(ctx, sensor, cfg, list(range(S)), 2200, 12, 3, 0, inputs=inputs)
    out = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, 0, out.ctypes.data)
    n = float(out[13]) or 1.0
    print(f"streams {S}: {r['value']:.0f} Mpoints/s, {r['ms_per_step']:.3f} ms per step, prep_ms {r['kernel_ms_per_step']['prep_ms']:.3f}; launches seen {int(n)}")
    for name, v in zip(["entry, clearing, steady test", "0: column of every firing (first valid return)", "B: prefix sum of the column advances", "D: cells + fused segmentation (wavefront 0's firings)", "wait for the block's other wavefronts"], out[8:13]):
        print(f"  {name:60s} {v / n:10.0f} clocks = {v / n / 2400:8.2f} us")
    e.close()

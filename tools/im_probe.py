"""Phase clocks of k_insert_multi (one wavefront of every stream's block; -DCC_IM_STATS [-DCC_IM_STATS_WAVE=w] build, e.g. libcc_hip_imstats.so).
usage: CC_HIP_LIB=libcc_hip_imstats.so python tools/im_probe.py [streams ...]"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import continuous_clustering_amd as cca
from continuous_clustering_amd import capi, synth
sizes = [int(a) for a in sys.argv[1:]] or [256]
dev = torch.device("cuda:0")
ctx = bench.Ctx(torch, None, False, 1, 0, dev, 0, False)
sensor, cfg = synth.SensorModel.s128(), capi.Config.vls128()
L = cca.load_library(); L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for S in sizes:
    inputs = bench.gen_inputs(torch, dev, sensor, [5678 + j for j in range(S)], 1700, 11)
    for opts in ({}, {"pipeline": 0}):
        r, e, _ = bench.run_throughput(ctx, sensor, cfg, list(range(S)), 1700, 8, 3, 0, inputs=inputs, options=opts)
        out = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, 0, out.ctypes.data)
        n = float(out[14]) or 1.0
        print(f"streams {S} {opts}: {r['value']:.0f} Mpoints/s, {r['ms_per_step']:.3f} ms per step, {r['kernel_ms_per_step']}; chunks seen {int(n)}")
        tot = 0.0
        for name, v in zip(["prepare (load, transform, columns of the firing)", "wait 1", "walk + collision rule", "wait 2", "cells", "carry + wait 3"], out[8:14]):
            print(f"  {name:60s} {v / n:10.0f} clocks per chunk = {v / n / 2400:8.2f} us")
            tot += v / n
        print(f"  {'chunk':60s} {tot:10.0f} clocks = {tot / 2400:8.2f} us")
        nl = float(out[5]) or 1.0
        print(f"  per launch: prologue (entry, deferred clearing, the rows' last columns) {out[15] / nl / 2400:8.1f} us, chunk loop {tot * n / nl / 2400:8.1f} us ({n / nl:.1f} chunks)")
        per = []
        for st in range(S):
            o = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, st, o.ctypes.data)
            per.append((float(o[8:14].sum()) + float(o[15])) / (float(o[5]) or 1.0) / 2400)
        per = np.sort(np.array(per))
        print("  block time per launch over the streams, us: min %.0f  p25 %.0f  median %.0f  p75 %.0f  p95 %.0f  max %.0f" % (per[0], per[S // 4], per[S // 2], per[3 * S // 4], per[int(S * 0.95)], per[-1]))
        e.close()

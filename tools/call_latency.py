"""Per-call latency of cc_engine_add_firings (host buffers, one stream) against the number of firings per call: the graph path (n <= 8) and
the general path. usage: python tools/call_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from continuous_clustering_amd import Engine, capi, synth
cfg = capi.Config.kitti()
st = synth.make_stream(2200 * 4, seed=5, motion=synth.Motion.translate())
for n in (1, 2, 4, 8, 16, 32, 64, 128):
    e = Engine(cfg, 64)
    e.add_firings(st.xyz[:2200], st.intensity[:2200], st.poses[:2200])
    lat = []
    f = 2200
    while f + n <= 2200 * 4 and len(lat) < 600:
        t0 = time.perf_counter()
        e.add_firings(st.xyz[f:f + n], st.intensity[f:f + n], st.poses[f:f + n])
        lat.append(time.perf_counter() - t0)
        f += n
    lat = np.array(lat[20:]) * 1e6
    print(f"n={n:4d}  p50 {np.percentile(lat, 50):7.1f} us  p99 {np.percentile(lat, 99):7.1f} us  -> {n / np.percentile(lat, 50) * 1e6:9.0f} firings/s")
    e.close()

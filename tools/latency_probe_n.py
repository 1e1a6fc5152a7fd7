"""single-stream host-path calls of n firings: latency, for rocprofv3 (tools/lat_prof_n.sh). usage: python tools/latency_probe_n.py <n>"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from continuous_clustering_amd import Engine, capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = capi.Config.kitti()
st = synth.make_stream(2200 + 400 * n, seed=5, motion=synth.Motion.translate())
e = Engine(cfg, 64)
e.add_firings(st.xyz[:2200], st.intensity[:2200], st.poses[:2200]); e.drain_events()
lat = []
for k in range(2200, 2200 + 400 * n, n):
    t = time.perf_counter(); e.add_firings(st.xyz[k:k+n], st.intensity[k:k+n], st.poses[k:k+n]); lat.append(time.perf_counter() - t)
lat = np.array(lat[20:]) * 1e6
print("n=%d call latency us: p50 %.1f p99 %.1f" % (n, np.percentile(lat, 50), np.percentile(lat, 99)))

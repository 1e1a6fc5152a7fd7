import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import continuous_clustering_amd as _cca
if len(sys.argv) > 1:  # another build of the library (A/B)
    _cca.LIB_PATH = os.path.join(os.path.dirname(_cca.LIB_PATH), sys.argv[1])
from continuous_clustering_amd import Engine, capi, synth
cfg = capi.Config.kitti()
st = synth.make_stream(2200 + 800, seed=5, motion=synth.Motion.translate())
e = Engine(cfg, 64)
for kv in filter(None, os.environ.get('CC_OPTS', '').split(',')):  # e.g. CC_OPTS=small_direct=0,small_all=0
    e.set_option(kv.split('=')[0], int(kv.split('=')[1]))
e.add_firings(st.xyz[:2200], st.intensity[:2200], st.poses[:2200]); e.drain_events()
lat = []
for k in range(2200, 3000):
    t = time.perf_counter(); e.add_firings(st.xyz[k:k+1], st.intensity[k:k+1], st.poses[k:k+1]); lat.append(time.perf_counter() - t)
lat = np.array(lat[50:]) * 1e6
print("single-firing call latency us: p50 %.1f p99 %.1f" % (np.percentile(lat, 50), np.percentile(lat, 99)))

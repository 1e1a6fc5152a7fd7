import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from continuous_clustering_amd import Engine, capi, synth
cfg = capi.Config.kitti()
st = synth.make_stream(2200 * 2, seed=1234, sensor=synth.SensorModel.s64(), motion=synth.Motion.translate())
for rep in range(2):
    for mv in (1, 0):
        e = Engine(cfg, 64, 1); e.set_option("mirror_views", mv)
        e.add_firings(st.xyz[:200], st.intensity[:200], st.poses[:200])
        lat = []
        for f in range(200, 3200):
            t0 = time.perf_counter(); e.add_firings(st.xyz[f:f+1], st.intensity[f:f+1], st.poses[f:f+1]); lat.append(time.perf_counter() - t0)
        lat = np.array(lat[300:]) * 1e6
        print("mirror_views", mv, "p50 %.2f p99 %.2f" % (np.percentile(lat, 50), np.percentile(lat, 99)))
        e.close()

import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from continuous_clustering_amd import Engine, capi, synth
cfg = capi.Config.kitti()
st = synth.make_stream(2200 * 3, seed=1234, sensor=synth.SensorModel.s64(), motion=synth.Motion.translate())
e = Engine(cfg, 64, 1)
fields = [k for k in capi.COLUMN_FIELDS if k != "number_of_child_points"]
tot = []; t_hit = []; t_miss = []
for f in range(st.n_firings):
    e.add_firings(st.xyz[f:f+1], st.intensity[f:f+1], st.poses[f:f+1])
    ev = e.drain_events()
    rng = [(int(x["a"]), int(x["b"])) for x in ev if x["type"] != capi.EV_CLUSTER and x["b"] >= x["a"]]
    if not rng: continue
    rng = sorted(set(rng))[:8]
    n = sum(b - a + 1 for a, b in rng)
    before = e.view_counters()["mirror"]
    t0 = time.perf_counter(); e.read_column_ranges(rng, fields=fields); dt = time.perf_counter() - t0
    (t_hit if e.view_counters()["mirror"] > before else t_miss).append(dt * 1e6)
    if f > 2200: tot.append(n)
tot = np.array(tot)
print("columns named per call: mean %.2f p50 %d p90 %d p99 %d max %d; share <= 8: %.3f, <= 16: %.3f, <= 32: %.3f" % (tot.mean(), np.percentile(tot,50), np.percentile(tot,90), np.percentile(tot,99), tot.max(), (tot<=8).mean(), (tot<=16).mean(), (tot<=32).mean()))
print("served from mirror %d (%.1f us p50 incl. numpy alloc) / by kernel %d (%.1f us p50)" % (len(t_hit), np.percentile(t_hit,50), len(t_miss), np.percentile(t_miss,50) if t_miss else 0))

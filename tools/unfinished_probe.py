"""How many point trees are unfinished at a time (what k_assocb's lanes have to hold): a stream fed in small calls, StreamState::n_unfinished
after every call. usage: python tools/unfinished_probe.py [scene: bench|cluttered|sparse|near] [density]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from continuous_clustering_amd import Engine, capi, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "cluttered"
d = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
scene = {"bench": synth.SceneModel(), "cluttered": synth.SceneModel.cluttered(d), "sparse": synth.SceneModel.sparse_clutter(d), "near": synth.SceneModel.near_clutter(d)}[kind]
cfg = capi.Config.kitti()
st = synth.make_stream(2200 * 3, seed=4321, scene=scene, motion=synth.Motion.translate(10.0))
e = Engine(cfg, 64, 1)
vals = []
for f in range(0, st.n_firings, 20):
    assert e.add_firings(st.xyz[f:f + 20], st.intensity[f:f + 20], st.poses[f:f + 20]) == 0
    vals.append(e.state()["n_unfinished_trees"])
v = np.array(vals[110:])
print(kind, d, "unfinished trees: mean %.1f p50 %d p90 %d p99 %d max %d; share of samples > 64: %.3f, > 128: %.3f, > 256: %.3f" % (v.mean(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(), (v > 64).mean(), (v > 128).mean(), (v > 256).mean()))

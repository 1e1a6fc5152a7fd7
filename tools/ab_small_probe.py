"""Phase clocks of k_assocb in one-firing calls (one column per launch). usage: CC_HIP_LIB=libcc_hip_abstats.so [AB_WORKER=1] python tools/ab_small_probe.py"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import continuous_clustering_amd as cca
from continuous_clustering_amd import Engine, capi, synth
cfg = capi.Config.kitti()
st = synth.make_stream(2200 + 800, seed=5, motion=synth.Motion.translate())
e = Engine(cfg, 64)
e.add_firings(st.xyz[:2200], st.intensity[:2200], st.poses[:2200]); e.drain_events()
L = cca.load_library(); L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
a = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, 0, a.ctypes.data)
N = 700
for k in range(2200, 2200 + N):
    e.add_firings(st.xyz[k:k+1], st.intensity[k:k+1], st.poses[k:k+1])
b = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, 0, b.ctypes.data)
d = (b - a).astype(float) / N
if os.environ.get("AB_WORKER"):
    names = {11: "own points: where, records requested", 8: "roots, ring renumbering", 0: "pointers", 1: "wait B1", 2: "jumping", 3: "records", 4: "links", 5: "wait B2", 7: "iterations", 9: "loop total", 13: "prologue", 14: "entry -> loop end"}
else:
    names = {0: "own alive words", 4: "pass, first half", 1: "wait B1", 2: "jumping (following)", 6: "pass, second half", 3: "header", 5: "wait B2", 7: "iterations", 9: "loop total", 13: "prologue", 14: "entry -> loop end"}
for k, n in names.items():
    print(f"  {n:40s} {d[k]:9.0f} clocks per launch = {d[k] / 2400:6.2f} us")

"""Phase clocks of k_small_all / k_small_front (one-firing calls): -DCC_SF_STATS build as libcc_hip_sfstats.so. usage: CC_HIP_LIB=libcc_hip_sfstats.so python tools/sf_probe.py"""
import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import continuous_clustering_amd as cca
from continuous_clustering_amd import Engine, capi, synth
cfg = capi.Config.kitti()
st = synth.make_stream(2200 + 800, seed=5, motion=synth.Motion.translate())
e = Engine(cfg, 64)
e.add_firings(st.xyz[:2200], st.intensity[:2200], st.poses[:2200]); e.drain_events()
for k in range(2200, 3000):
    e.add_firings(st.xyz[k:k+1], st.intensity[k:k+1], st.poses[k:k+1])
L = cca.load_library(); L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
out = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, 0, out.ctypes.data)
if out[6] > 0:  # k_small_all (the whole call in one launch)
    n = float(out[6])
    for name, v in zip(["begin + ego + prep (inputs from pinned host memory) | idle wavefronts: association state -> LDS", "serial insertion (insert2_body)", "segmentation (seg_small_body)",
                        "window scan (scan_body)", "association (assocb_body)", "results to pinned host memory | cluster ids"], out[:6]):
        print(f"  {name:100s} {v / n:9.0f} clocks = {v / n / 2400:6.2f} us")
    print(f"  {'kernel, first to last mark':100s} {out[:6].sum() / n:9.0f} clocks = {out[:6].sum() / n / 2400:6.2f} us")
    sys.exit(0)
n = float(out[4])
for name, v in zip(["begin + ego + prep (inputs from pinned host memory)", "serial insertion (insert2_body)", "segmentation (seg_small_body)", "window scan (scan_body)"], out[:4]):
    print(f"  {name:55s} {v / n:9.0f} clocks = {v / n / 2400:6.2f} us")

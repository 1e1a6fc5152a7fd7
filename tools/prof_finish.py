"""Finish-pass sub-sections of k_assoc_lds (build with -DCC_PROFILE_SECTIONS -DCC_PROFILE_ASSOC_FINE as libcc_hip_prof.so)."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continuous_clustering_amd as cca
cca.LIB_PATH = cca.LIB_PATH.replace("libcc_hip.so", "libcc_hip_prof.so")
from continuous_clustering_amd import Engine, capi, synth
import bench
sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
S, F, NB = 256, 2200, 4
xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, S, F, NB, 1234)
torch.cuda.synchronize()
e = Engine(cfg, 64, S); e.record_events(False); e.set_option("pipeline", 0)
for b in range(NB): e.add_firings_device(F, xyz[b], inten[b], poses[b])
print(e.sync())
L = cca.load_library(); L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
tot = np.zeros(16)
for s in range(0, S, 16):
    out = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, s, out.ctypes.data); tot += out
tot /= (S / 16)
cols = F * NB
names = ["full passes (count)", "sum n_unf at passes", "trees removed (count)", "init+find+flags", "cluster id loop", "compaction", "remaps",
         "may_finish calls (count)", "assoc init", "loop top", "issue prefetch", "resolve", "apply/links", "(unused)", "C+P", "ballots"]
for n, v in zip(names, tot): print(f"{n:28s} {v:14.0f}   per column {v / cols:10.3f}")
if tot[0] > 0:
    print("avg n_unf at a pass", tot[1] / tot[0], " cycles per pass", (tot[3] + tot[4] + tot[5] + tot[6]) / tot[0])

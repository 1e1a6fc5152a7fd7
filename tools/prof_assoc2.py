"""Per-section cycles of wave B of k_assoc2: one build per section (libcc_hip_a2s<k>.so, -DCC_A2_SECTION=k), pipeline off."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continuous_clustering_amd as cca
k = int(sys.argv[1])
cca.LIB_PATH = cca.LIB_PATH.replace("libcc_hip.so", f"libcc_hip_a2s{k}.so")
from continuous_clustering_amd import Engine, capi, synth
import bench
sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
S, F, NB = 64, 2200, 3
xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, S, F, NB, 1234)
torch.cuda.synchronize()
e = Engine(cfg, 64, S); e.record_events(False); e.set_option("pipeline", 0)
for b in range(NB): e.add_firings_device(F, xyz[b], inten[b], poses[b])
rc = e.sync()
L = cca.load_library(); L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
tot = np.zeros(16)
for s in range(0, S, 8):
    out = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, s, out.ctypes.data); tot += out
tot /= (S / 8)
names = ["loop top + staged copies (vmcnt wait)", "prefetch issue + a_done + info", "ballots + verification", "tree init + root store", "apply (find/atomics/links)",
         "finish check + publish", "whole iteration", "full finish pass"]
print(f"section {k} {names[k]:40s} {tot[k] / (F * NB):9.1f} cycles per column (rc {rc})")

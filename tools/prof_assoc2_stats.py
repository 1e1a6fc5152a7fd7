import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continuous_clustering_amd as cca
cca.LIB_PATH = cca.LIB_PATH.replace("libcc_hip.so", "libcc_hip_a2stats.so")
from continuous_clustering_amd import Engine, capi, synth
import bench
sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
S, F, NB = 64, 2200, 3
xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, [1234 + k for k in range(S)], F, NB)
torch.cuda.synchronize()
e = Engine(cfg, 64, S); e.record_events(False); e.set_option("pipeline", 0)
for b in range(NB): e.add_firings_device(F, xyz[b], inten[b], poses[b])
rc = e.sync()
L = cca.load_library(); L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
tot = np.zeros(16)
for s in range(0, S, 8):
    out = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, s, out.ctypes.data); tot += out
tot /= (S / 8)
cols = tot[12]
print(f"columns {cols:.0f}  sub-batches/col {tot[8]/cols:.3f}  check cuts/col {tot[9]/cols:.3f}  live cuts/col {tot[11]/cols:.4f}  "
      f"B cycles/col {tot[10]/cols:.0f}  B waits for A /col {tot[13]/cols:.3f}")
print(f"  full finish passes/col {tot[14]/cols:.3f}  passes that retired trees/col {tot[15]/cols:.3f}")
print(f"  k_assoc3 only: wave A busy cycles/col {tot[6]/max(tot[7],1):.0f} (over {tot[7]:.0f} columns incl. re-resolved ones)  wave R busy cycles/col {tot[5]/cols:.0f}")
names = ["ids (re)load", "verification", "scalar walk", "batch apply", "cut column + b_done"]
for i, n in enumerate(names): print(f"  {n:22s} {tot[i]/cols:8.0f} cycles/col   {tot[i]/max(tot[8],1):8.0f} per sub-batch")

import sys, ctypes as C, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
import continuous_clustering_amd as cca
cca.LIB_PATH = cca.LIB_PATH.replace("libcc_hip.so","libcc_hip_prof.so")
from continuous_clustering_amd import Engine, capi, synth
import bench
sensor = synth.SensorModel.s128(); cfg = capi.Config.vls128()
S,F,NB = 64,1700,3
xyz,inten,poses = bench.gen_inputs(torch, torch.device("cuda",0), sensor, [1234 + k for k in range(S)], F, NB)
torch.cuda.synchronize()
e = Engine(cfg, 128, S); e.record_events(False); e.set_option("pipeline", 0)
for b in range(NB): e.add_firings_device(F, xyz[b], inten[b], poses[b])
print(e.sync())
L = cca.load_library(); L.cc_engine_debug_counters.argtypes=[C.c_void_p, C.c_int, C.c_void_p]
tot = np.zeros(16)
for s in range(0,S,16):
    out = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, s, out.ctypes.data); tot += out
tot /= (S/16)
names = ["ins wait","ins gcol(general)","ins peel(general)","ins window+stores(general)","ins rear/fore(general)","ins total loop","-","-","assoc init","assoc loop top","assoc issue prefetch","assoc resolve","assoc apply/links","assoc (unused)","assoc C+P","assoc ballots"]
for n,v in zip(names,tot): print(f"{n:24s} {v:14.0f} ticks  per column {v/(F*NB):10.1f}")

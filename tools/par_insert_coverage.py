"""How many firings of every batch k_insert_par takes (StreamState::dbg[6] / dbg[7]) on the bench streams."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import continuous_clustering_amd as cca
from continuous_clustering_amd import Engine, capi, synth
import bench
sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
S, F, NB = 8, 2200, 5
xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, [1234 + k for k in range(S)], F, NB)
torch.cuda.synchronize()
L = cca.load_library(); L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
e = Engine(cfg, 64, S); e.record_events(False)
prev = np.zeros((S, 16), dtype=np.uint64)
for b in range(NB):
    e.add_firings_device(F, xyz[b], inten[b], poses[b]); e.sync()
    cur = np.zeros((S, 16), dtype=np.uint64)
    for s in range(S):
        L.cc_engine_debug_counters(e.h, s, cur[s].ctypes.data)
    print("batch", b, "taken per stream:", (cur[:, 6] - prev[:, 6]).tolist())
    prev = cur

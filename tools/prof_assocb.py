"""Per-phase clock counters of k_assocb (build with -DCC_AB_STATS into libcc_hip_abstats.so: tools/build_variant.sh abstats -DCC_AB_STATS).
usage: python tools/prof_assocb.py [streams]"""
import sys, ctypes as C, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continuous_clustering_amd as cca
cca.LIB_PATH = cca.LIB_PATH.replace("libcc_hip.so", os.environ.get("AB_LIB", "libcc_hip_abstats.so"))
from continuous_clustering_amd import Engine, capi, synth
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
F, NB = 2200, 4
xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, [1234 + k for k in range(S)], F, NB)
torch.cuda.synchronize()
for pipe in (0,):
    e = Engine(cfg, 64, S); e.record_events(False); e.set_option("pipeline", pipe)
    for b in range(NB): e.add_firings_device(F, xyz[b], inten[b], poses[b])
    assert e.sync() == 0
    L = cca.load_library(); L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    tot = np.zeros(16)
    for s in range(0, S, max(1, S // 16)):
        out = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, s, out.ctypes.data); tot += out
    tot /= len(range(0, S, max(1, S // 16)))
    # seen from the timeline wavefront: what it waits for is what the workers do
    names = ["loop top", "old trees' words + wait for the workers' pointers (B1)", "pointer jumping rounds", "-", "wait for records / links (B2)",
             "timeline + commit + next header (to B3)", "-", "groups", "-", "kernel total"]
    if os.environ.get("AB_WORKER"):  # -DCC_AB_STATS -DCC_AB_STATS_W: the counters of worker wavefront 1
        names = ["4b of the previous group + pointers (to B1)", "wait at B1", "pointer jumping rounds", "records", "links", "wait at B2 (slowest worker)",
                 "prefetch + wait for the timeline (B3)", "groups", "4b: roots, ring remap", "kernel total"]
    groups = tot[7]
    print("pipeline", pipe, "streams", S, e.batch_counters())
    for n, v in zip(names, tot):
        print(f"  {n:24s} {v:14.0f}   per group {v / groups:10.1f}")
    e.close()

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
F, NB = 2200, int(os.environ.get("AB_NB", "4"))
scene = synth.SceneModel.cluttered(0.1) if os.environ.get("AB_SCENE") == "cluttered" else None  # (AB_SCENE=cluttered: the bench's vegetation streams)
xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, [(4321 if scene else 1234) + k for k in range(S)], F, NB, scene=scene)
torch.cuda.synchronize()
for pipe in (0,):
    e = Engine(cfg, 64, S); e.record_events(False); e.set_option("pipeline", pipe); e.enable_timing(True)
    for b in range(NB): e.add_firings_device(F, xyz[b], inten[b], poses[b])
    assert e.sync() == 0
    L = cca.load_library(); L.cc_engine_debug_counters.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    tot = np.zeros(16)
    for s in range(0, S, max(1, S // 16)):
        out = np.zeros(16, dtype=np.uint64); L.cc_engine_debug_counters(e.h, s, out.ctypes.data); tot += out
    tot /= len(range(0, S, max(1, S // 16)))
    # (cc_assocb.h: AB_PH / AB_PHW indices)
    if os.environ.get("AB_WORKER"):  # -DCC_AB_STATS -DCC_AB_STATS_W: the counters of worker wavefront 1
        names = {6: "tile -> packed order (stage write)", 8: "roots of group i - 2, ring renumbering", 10: "wait at Bq", 0: "pointers (waits for the records)",
                 1: "wait at B1", 2: "pointer jumping rounds", 11: "own 64 points: where, records requested", 3: "records", 4: "links", 5: "wait at B2", 7: "groups", 9: "kernel total", 13: "prologue", 14: "entry -> loop end", 15: "entry -> loop end in 10 ns units"}
    else:
        names = {0: "own alive words of the old trees", 4: "pass of group i - 1, first half", 6: "pass, second half", 3: "header of group i + 1", 1: "wait at Bq + B1", 2: "pointer jumping rounds (following)",
                 5: "wait at B2", 7: "groups", 9: "kernel total", 13: "prologue", 14: "entry -> loop end", 15: "entry -> loop end in 10 ns units"}
    order = [11, 8, 0, 1, 2, 3, 4, 5, 7, 9, 13, 14, 15] if os.environ.get("AB_WORKER") else [0, 4, 1, 2, 6, 3, 5, 7, 9, 13, 14, 15]
    groups = tot[7]
    print("pipeline", pipe, "streams", S, e.batch_counters(), {k: round(v / NB, 4) for k, v in e.kernel_times().items() if k.endswith("_ms")})
    for k in order:
        print(f"  {names[k]:44s} {tot[k]:14.0f}   per group {tot[k] / groups:10.1f}")
    e.close()

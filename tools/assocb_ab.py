"""A/B of libcc_hip builds on the association chain: kernel times at several stream counts. usage: python tools/assocb_ab.py lib1.so lib2.so ..."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, time
sys.path.insert(0, %r)
import continuous_clustering_amd as cca
cca.LIB_PATH = os.path.join(os.path.dirname(cca.LIB_PATH), %r)
import torch
from continuous_clustering_amd import Engine, capi, synth
import bench
sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
F, NB = 2200, 6
for S in (32, 128, 256):
    xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, [1234 + k for k in range(S)], F, NB)
    torch.cuda.synchronize()
    for pipe in (0, 1):
        e = Engine(cfg, 64, S); e.record_events(False); e.set_option("pipeline", pipe)
        e.add_firings_device(F, xyz[0], inten[0], poses[0]); e.sync()
        e.enable_timing(True)
        t0 = time.time()
        for b in range(1, NB): e.add_firings_device(F, xyz[b], inten[b], poses[b])
        e.sync(); dt = (time.time() - t0) / (NB - 1)
        k = e.kernel_times(); bc = e.batch_counters(); tot = e.totals()
        print(%r, "streams", S, "pipeline", pipe, "ms/step %%.3f" %% (dt * 1e3), "Mpoints/s %%.0f" %% (S * F * 64 / dt / 1e6), "bails", bc["batch_bails"],
              "clusters", tot["clusters_finished"], {n: round(v / k["batches"], 3) for n, v in k.items() if n.endswith("_ms")}, flush=True)
        e.close()
    del xyz, inten, poses
'''
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", code % (ROOT, lib, lib)], capture_output=True, text=True, cwd=ROOT)
    print(r.stdout, r.stderr[-500:] if r.returncode else "")

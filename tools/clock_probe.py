"""Shader clock and package power while the 256-stream pipelined workload runs for ~12 s (rocm-smi sampled from a side thread).
usage (GPU box): python tools/clock_probe.py"""
import os, sys, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from continuous_clustering_amd import Engine, capi, synth
import bench
S, F, NB = 256, 2200, 4
sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, [1234 + k for k in range(S)], F, NB)
torch.cuda.synchronize()
e = Engine(cfg, 64, S); e.record_events(False)
samples = []; stop = False
def sample():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        s = [l.split(":")[-1].strip() for l in out.splitlines() if "sclk" in l or "Power (W)" in l]
        samples.append((time.time(), s))
        time.sleep(0.3)
th = threading.Thread(target=sample); th.start()
# NOTE: the streams are replayed rotation after rotation with the same poses; only the load matters here
t0 = time.time(); n = 0
while time.time() - t0 < 12.0:
    for b in range(NB):
        e.add_firings_device(F, xyz[b], inten[b], poses[b]); n += 1
    e.sync()
dt = time.time() - t0
stop = True; th.join()
print("steps", n, "ms_per_step", dt / n * 1e3, "Mpoints/s", S * F * 64 * n / dt / 1e6)
for t, s in samples: print(round(t - t0, 1), s)

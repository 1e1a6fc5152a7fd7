"""bench.py's live_multi_stream leg alone (256 live S64 streams, calls of n firings), optionally with another build of the library:
python tools/live_probe.py [libcc_hip_<variant>.so] [sizes, e.g. 8,128,550]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import continuous_clustering_amd as cca
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    cca.LIB_PATH = os.path.join(os.path.dirname(cca.LIB_PATH), sys.argv[1])
sizes = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (8, 128, 550)
import torch
import bench
from continuous_clustering_amd import capi, synth
sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, [1234 + k for k in range(256)], 2200, 4)
torch.cuda.synchronize()
out = bench.live_multi_stream(torch, cfg, sensor, xyz, inten, poses, 0, sizes=sizes)
for k, v in out.items():
    if isinstance(v, dict):
        print(os.path.basename(cca.LIB_PATH), "n", k, "Mpoints/s", round(v["Mpoints_per_s"]), "period_us", round(v["call_period_us"]), "p50", round(v["call_latency_us_p50"]), "p99", round(v["call_latency_us_p99"]))

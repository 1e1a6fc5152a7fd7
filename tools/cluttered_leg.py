"""bench.py's `cluttered` leg alone (256 vegetation-like streams: value, stops per reason, share of columns on the fast path, the floor).
usage: python tools/cluttered_leg.py [streams [steps]]   (engine options through the environment with CC_ENABLE_ENV_OPTS=1, library through CC_HIP_LIB)"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from continuous_clustering_amd import capi, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
args = argparse.Namespace(no_cpu_baseline=True, cpu_rotations=8, cpu_procs=16, no_verify=False, verify_streams=2)
dev = torch.device("cuda:0")
ctx = bench.Ctx(torch, None, False, 1, 0, dev, 0, False)
sensor = synth.SensorModel(num_rows=64, num_columns=2200)
cfg = capi.Config.kitti()
out = bench.cluttered_report(ctx, sensor, cfg, 2200, S, steps, 16800.0, args)
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}))

import sys, json, os
sys.path.insert(0,'.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from continuous_clustering_amd import Engine, capi, synth
import bench
sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
S,F,NB = 256,2200,4
xyz,inten,poses = bench.gen_inputs(torch, torch.device("cuda",0), sensor, [1234 + k for k in range(S)], F, NB)
torch.cuda.synchronize()
for pipe, par in [(0, 0), (0, 1), (1, 0), (1, 1)]:
    flags = 0
    e = Engine(cfg, 64, S); e.record_events(False); e.set_option("pipeline", pipe); e.set_option("parallel_insert", par)
    if os.environ.get("CC_ASSOC_WAVES"): e.set_option("assoc_waves", int(os.environ["CC_ASSOC_WAVES"]))
    e.add_firings_device(F, xyz[0], inten[0], poses[0]); e.sync()
    e.enable_timing(True)
    for b in range(1,NB): e.add_firings_device(F, xyz[b], inten[b], poses[b])
    e.sync(); k = e.kernel_times()
    print("pipeline", pipe, "parallel_insert", par, {k2: e.state(0)[k2] for k2 in ("firings_consumed",)}, {n: round(v/k["batches"],3) for n,v in k.items() if n.endswith("_ms")})
    e.close()

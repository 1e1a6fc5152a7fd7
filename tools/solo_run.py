"""3 batches of the bench workload with the pipeline off (kernels run back to back on one HIP stream): the subject of PMC passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continuous_clustering_amd as _cca
if len(sys.argv) > 2:  # another build of the library (A/B)
    _cca.LIB_PATH = os.path.join(os.path.dirname(_cca.LIB_PATH), sys.argv[2])
from continuous_clustering_amd import Engine, capi, synth
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s128 = os.environ.get("SOLO_SENSOR", "s64") == "s128"  # SOLO_SENSOR=s128: the 128-row workload of the bench's s128 leg
sensor = synth.SensorModel.s128() if s128 else synth.SensorModel.s64()
cfg = capi.Config.vls128() if s128 else capi.Config.kitti()
F, NB = (1700 if s128 else 2200), 3
xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, [1234 + k for k in range(S)], F, NB)
torch.cuda.synchronize()
e = Engine(cfg, sensor.num_rows, S); e.record_events(False); e.set_option("pipeline", 0)
# (the counter passes run on 64 streams but stand for the 256-stream bench, where the engine takes the packed window scan: pin it)
if os.environ.get("SOLO_SCAN_PACKED", "1") == "1": e.set_option("scan_packed", 1)
for b in range(NB): e.add_firings_device(F, xyz[b], inten[b], poses[b])
print("rc", e.sync())

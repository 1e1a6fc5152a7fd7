for p in 1 2; do CC_PIPE=$p timeout 300 python - <<'PY'
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from continuous_clustering_amd import Engine, capi, synth
import bench
sensor = synth.SensorModel.s64(); cfg = capi.Config.kitti()
S, F, NB = 256, 2200, 10
xyz, inten, poses = bench.gen_inputs(torch, torch.device("cuda", 0), sensor, S, F, 4, 1234)
torch.cuda.synchronize()
import time
e = Engine(cfg, 64, S); e.record_events(False); e.set_option("pipeline", int(os.environ["CC_PIPE"]))
for b in range(2): e.add_firings_device(F, xyz[b % 4], inten[b % 4], poses[b % 4])
e.sync(); e.enable_timing(True)
t0 = time.time()
for b in range(2, NB): e.add_firings_device(F, xyz[b % 4], inten[b % 4], poses[b % 4])
rc = e.sync(); dt = time.time() - t0
k = e.kernel_times()
print("pipeline", os.environ["CC_PIPE"], "rc", rc, "ms/step", round(dt / (NB - 2) * 1e3, 3), {n: round(v / k["batches"], 2) for n, v in k.items() if n.endswith("_ms")})
PY
done

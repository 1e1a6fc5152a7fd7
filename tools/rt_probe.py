import sys, os, struct, subprocess, re
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from continuous_clustering_amd import capi, synth
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
DEMO = os.path.join(ROOT, "tests", "cpp", "dropin_demo")
cfg = capi.Config.kitti()
stream = synth.make_stream(2200 * 3, seed=77, motion=synth.Motion.translate())
inp = "/tmp/in.bin"
with open(inp, "wb") as f:
    f.write(struct.pack("<iiii", 64, cfg.num_columns, stream.n_firings, 1))
    f.write(stream.xyz.astype(np.float32).tobytes()); f.write(stream.intensity.astype(np.uint8).tobytes()); f.write(stream.poses.astype(np.float64).tobytes())
for rep in range(6):
    r = subprocess.run([DEMO, inp, "/dev/null", "0", "22000"], capture_output=True, text=True, timeout=600)
    m = re.search(r"firings_per_s=(\d+) latency_us_p50=([\d.]+) p99=([\d.]+) max=([\d.]+)", r.stdout)
    print(rep, m.groups() if m else r.stdout[-200:] + r.stderr[-200:], flush=True)

"""The drop-in class's asynchronous mode fed like a live sensor (tests/cpp/dropin_demo, mode -1, 22 000 firings/s) with CC_ASYNC_TRACE=1:
delivery latency and the longest hand-overs of the worker thread (when, how many firings, how long). usage: python tools/async_trace.py [runs]"""
import os, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from continuous_clustering_amd import synth
st = synth.make_stream(2200 * 3, seed=1234, motion=synth.Motion.translate(10.0))
with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
    f.write(struct.pack("<iiii", 64, 2200, st.n_firings, 1))
    f.write(st.xyz.astype(np.float32).tobytes()); f.write(st.intensity.astype(np.uint8).tobytes()); f.write(st.poses.astype(np.float64).tobytes())
    path = f.name
demo = os.path.join(ROOT, "tests", "cpp", "dropin_demo")
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    p = subprocess.run([demo, path, "/dev/null", "-1", "22000"], capture_output=True, text=True, env=dict(os.environ, CC_ASYNC_TRACE="1"))
    print([l for l in p.stdout.splitlines() if l.startswith("feed")])
    print([l for l in p.stderr.splitlines() if "async trace" in l])
os.unlink(path)

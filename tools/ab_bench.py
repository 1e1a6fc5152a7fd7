"""A/B two builds of libcc_hip on the same box: python tools/ab_bench.py libA.so libB.so [steps]  (paths relative to the package dir)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[3] if len(sys.argv) > 3 else "40"
code = """
import sys, os
sys.path.insert(0, %r)
import continuous_clustering_amd as cca
cca.LIB_PATH = os.path.join(os.path.dirname(cca.LIB_PATH), %%r)
sys.argv = ['bench.py', '--steps', %r, '--warmup', '3', '--no-cpu-baseline', '--no-latency']
import bench
bench.main()
""" % (ROOT, steps)
for rep in range(int(sys.argv[4]) if len(sys.argv) > 4 else 2):
    for lib in sys.argv[1:3]:
        r = subprocess.run([sys.executable, "-c", code % lib], capture_output=True, text=True, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            d = json.loads(line[-1]); print(lib, round(d["value"]), round(d["ms_per_step"], 3), flush=True)
        else:
            print(lib, "FAILED", r.stderr[-300:])

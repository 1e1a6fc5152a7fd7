"""A/B builds of libcc_hip on the S128 workload: python tools/ab_s128.py libA.so libB.so ...  (names relative to the package dir)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = """
import sys, os
sys.path.insert(0, %r)
import continuous_clustering_amd as cca
cca.LIB_PATH = os.path.join(os.path.dirname(cca.LIB_PATH), %%r)
sys.argv = ['bench.py', '--sensor', 's128', '--firings', '1700', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-latency', '--no-s128', '--no-verify']
import bench
bench.main()
""" % ROOT
for rep in range(2):
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, "-c", code % lib], capture_output=True, text=True, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            d = json.loads(line[-1]); print(lib, round(d["value"]), round(d["ms_per_step"], 3), {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()}, flush=True)
        else:
            print(lib, "FAILED", r.stderr[-300:])

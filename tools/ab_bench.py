"""A/B builds of libcc_hip on the same box: python tools/ab_bench.py libA.so libB.so [steps [reps]]  (paths relative to the package dir),
or python tools/ab_bench.py --multi <steps> <reps> libA.so libB.so libC.so ..."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
multi = sys.argv[1] == "--multi"
steps = sys.argv[2] if multi else (sys.argv[3] if len(sys.argv) > 3 else "40")
libs = sys.argv[4:] if multi else sys.argv[1:3]
reps = int(sys.argv[3]) if multi else (int(sys.argv[4]) if len(sys.argv) > 4 else 2)
code = """
import sys, os
sys.path.insert(0, %r)
import continuous_clustering_amd as cca
cca.LIB_PATH = os.path.join(os.path.dirname(cca.LIB_PATH), %%r)
sys.argv = ['bench.py', '--steps', %r, '--warmup', '3', '--no-cpu-baseline', '--no-latency']
import bench
bench.main()
""" % (ROOT, steps)
for rep in range(reps):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", code % lib], capture_output=True, text=True, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if line:
            d = json.loads(line[-1]); print(lib, round(d["value"]), round(d["ms_per_step"], 3), flush=True)
        else:
            print(lib, "FAILED", r.stderr[-300:])

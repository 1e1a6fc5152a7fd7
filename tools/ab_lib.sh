#!/bin/bash
# usage: tools/ab_lib.sh [-s "<bench args>"] <lib1.so> <lib2.so> ...   same-box alternation of bench.py (256 x S64, 40 steps) over several builds of the library
args="--steps 40 --warmup 3 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-verify"
if [ "$1" = "-s" ]; then args="$2"; shift 2; fi
for rep in 1 2 3; do
for lib in "$@"; do
timeout 300 python -c "
import sys, os, runpy
sys.path.insert(0, os.getcwd())
import continuous_clustering_amd as c
c.LIB_PATH = os.path.join(os.path.dirname(c.LIB_PATH), '$lib')
sys.argv = ['bench.py'] + '$args'.split()
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done; done

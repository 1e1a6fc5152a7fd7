#!/bin/bash
# usage: tools/ab_lib_streams.sh <lib1.so> <lib2.so> ...   same-box alternation of bench.py at 32 / 64 / 256 streams over several builds of the library
for S in 32 64 256; do for rep in 1 2 3; do for lib in "$@"; do
timeout 300 python -c "
import sys, os, runpy
sys.path.insert(0, os.getcwd())
import continuous_clustering_amd as c
c.LIB_PATH = os.path.join(os.path.dirname(c.LIB_PATH), '$lib')
sys.argv = ['bench.py', '--streams', '$S', '--steps', '60', '--no-cpu-baseline', '--no-latency', '--no-s128', '--no-verify', '--no-few-streams', '--no-host-fed']
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'streams', $S, round(d['value']), round(d['ms_per_step'],3), 'bails', d['association']['batch_bails'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items() if k.startswith('assoc')})"
done; done; done

#!/bin/bash
# usage: tools/ab_env.sh "<ENV=val ...>" "<ENV=val ...>" ...   same-box alternation of bench.py (256 x S64, 40 steps) under different engine options
for rep in 1 2 3; do
for cfg in "$@"; do
env $cfg timeout 200 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-latency --no-s128 --no-verify --no-host-fed --no-few-streams 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['value']), round(d['ms_per_step'],3), d.get('association',{}).get('batch_bails'), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done; done

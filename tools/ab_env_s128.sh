#!/bin/bash
# usage: tools/ab_env_s128.sh "<ENV=val ...>" ...   same-box alternation of the S128 leg (256 x 128 rows x 1700 firings, 12 steps) under different engine options
for rep in 1 2 3; do
for cfg in "$@"; do
env $cfg timeout 300 python bench.py --sensor s128 --firings 1700 --steps 12 --warmup 3 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['value']), round(d['ms_per_step'],3), d.get('association',{}).get('batch_bails'), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done; done

#!/bin/bash
# usage: tools/ab_env_s.sh <streams> "<ENV=val ...>" ...   like ab_env.sh at another stream count
S=$1; shift
for rep in 1 2 3; do
for cfg in "$@"; do
env $cfg timeout 200 python bench.py --streams $S --steps 60 --warmup 3 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $S $cfg', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done; done

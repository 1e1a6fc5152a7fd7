#!/bin/bash
# usage: tools/low_stream_probe.sh [ENV=val ...]   bench.py at 32 / 64 / 128 streams (60 steps, twice each): throughput, step, per-chain kernel times, association counters
for S in 32 64 128; do for r in 1 2; do
env "$@" python bench.py --streams $S --steps 60 --no-cpu-baseline --no-latency --no-verify --no-s128 --no-few-streams --no-host-fed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'streams', $S, round(d['value']), round(d['ms_per_step'],3), 'bails', d['association']['batch_bails'], d['association']['bail_reasons'][:7], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done; done

#!/bin/bash
# same-box A/B of how stops of k_assocb are handled (batches of three rounds after a stop) at several stream counts
for S in 32 64 128 256; do for rep in 1 2; do for cfg in "CC_ASSOC_COOLDOWN=16" "CC_ASSOC_COOLDOWN=4" "CC_ASSOC_COOLDOWN=0"; do
env $cfg python bench.py --streams $S --steps 60 --no-cpu-baseline --no-latency --no-verify --no-s128 --no-few-streams --no-host-fed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'streams', $S, round(d['value']), round(d['ms_per_step'],3), 'bails', d['association']['batch_bails'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items() if k.startswith('assoc')})"
done; done; done

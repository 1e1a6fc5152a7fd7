for rep in 1 2 3; do
for cfg in "1 1" "2 1" "2 0" "1 0"; do
set -- $cfg
CC_PIPELINE=$1 CC_SKIP_FALLBACKS=$2 timeout 200 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline', $1, 'skip', $2, round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done; done

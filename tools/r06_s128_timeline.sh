repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s128tl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s128tl -o bench -- python $repo/bench.py --sensor s128 --firings 1700 --steps 12 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-verify --no-cluttered > /tmp/prof_s128tl.log 2>&1
t=$(find /tmp/prof_s128tl -name "*kernel_trace.csv" | head -1)
mkdir -p $repo/gpurun_out/r06q
python $repo/tools/timeline.py $t 70 > $repo/gpurun_out/r06q/s128_timeline.txt
grep '^{' /tmp/prof_s128tl.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s128 value', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])"
tail -45 $repo/gpurun_out/r06q/s128_timeline.txt

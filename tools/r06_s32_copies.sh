#!/bin/bash
# round 6: kernel + memory-copy timeline at 32 (and 64) streams, per library given as arguments (default libcc_hip.so)
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/r06q
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for n in 32 64; do
  rm -rf /tmp/prof_s$n
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_s$n -o bench -- python $repo/bench.py --streams $n --steps 40 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-verify --no-cluttered --no-strong-split > /tmp/prof_s$n.log 2>&1
  t=$(find /tmp/prof_s$n -name "*kernel_trace.csv" | head -1)
  c=$(find /tmp/prof_s$n -name "*memory_copy_trace.csv" | head -1)
  python $repo/tools/timeline.py $t 130 $c > $out/s${n}_timeline_copies.txt
  grep '^{' /tmp/prof_s$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s$n value', round(d['value']), d['ms_per_step'])"
  head -2 $c
done

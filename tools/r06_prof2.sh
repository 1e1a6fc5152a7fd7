#!/bin/bash
# round 6: (1) per-kernel rocprofv3 stats of the headline leg with the long scans apart (CC_SCAN_SPLIT=1) and in one pass (0); (2) kernel timeline at 32 streams
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/r06p2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for sp in 1 0; do
  rm -rf /tmp/prof_sp$sp
  CC_ENABLE_ENV_OPTS=1 CC_SCAN_SPLIT=$sp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sp$sp -o bench -- python $repo/bench.py --steps 40 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-verify --no-cluttered > /tmp/prof_sp$sp.log 2>&1
  grep '^{' /tmp/prof_sp$sp.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $sp value', round(d['value']), d['kernel_ms_per_step'])" > $out/split${sp}_line.txt
  f=$(find /tmp/prof_sp$sp -name "*kernel_stats.csv" | head -1)
  grep -E 'cck::' $f | sed 's/(ccd::Geometry[^"]*"/"/; s/(ccd::StreamState[^"]*"/"/' | cut -d, -f1-4,6,7 | head -14 > $out/split${sp}_kernel_stats.csv
  t=$(find /tmp/prof_sp$sp -name "*kernel_trace.csv" | head -1)
  python $repo/tools/timeline.py $t 70 > $out/split${sp}_timeline.txt
done
rm -rf /tmp/prof_s32
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s32 -o bench -- python $repo/bench.py --streams 32 --steps 40 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-verify --no-cluttered --no-strong-split > /tmp/prof_s32.log 2>&1
t=$(find /tmp/prof_s32 -name "*kernel_trace.csv" | head -1)
python $repo/tools/timeline.py $t 90 > $out/s32_timeline.txt
grep '^{' /tmp/prof_s32.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s32 value', round(d['value']), d['ms_per_step'], d['kernel_ms_per_step'])" > $out/s32_line.txt
cat $out/split1_line.txt $out/split0_line.txt $out/s32_line.txt; cat $out/split1_kernel_stats.csv

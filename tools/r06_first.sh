#!/bin/bash
# round 6, first GPU call: the tree as round 5 left it, on this round's box — driver-shape line, kernel timeline of the pipelined step, occupancy counters, GPU tests
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/r06a
mkdir -p $out
cd $repo
python bench.py --steps 20 --warmup 3 > $out/bench_line.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r06a
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r06a -o bench -- python $repo/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-verify > /tmp/prof_r06a.log 2>&1
grep '^{' /tmp/prof_r06a.log > $out/prof_bench_line.json
f=$(find /tmp/prof_r06a -name "*kernel_trace.csv" | head -1)
python $repo/tools/timeline.py $f 120 > $out/timeline.txt
cp $(find /tmp/prof_r06a -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
cd $repo
tools/pmc_sq.sh r06a "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY" "SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU" > $out/sq_solo64.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
tail -3 $out/pytest_gpu.log
cat $out/sq_solo64.log | cut -c1-400
cat $out/bench_line.json | cut -c1-600

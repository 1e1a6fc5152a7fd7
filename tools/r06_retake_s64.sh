#!/bin/bash
# round 6: the S64 part of the profile set again on the final tree (the in-kernel fin / gate experiment had doubled k_insert_par's scalar spills in the first take)
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/round_r06g
mkdir -p $out
cd $repo
python bench.py --steps 20 --warmup 3 > $out/bench_line.json 2> $out/bench.err
cp $repo/gpurun_out/bench_detail.json $out/bench_detail.json 2>/dev/null
tools/prof.sh r06g --steps 40 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-cluttered > $out/prof.log 2>&1
tools/pmc_sq.sh r06g "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" > $out/sq.log 2>&1
cp $repo/gpurun_out/prof_r06g/*kernel_stats.csv $out/kernel_stats_s64.csv 2>/dev/null
cp $repo/gpurun_out/prof_r06g/bench_line.json $out/prof_bench_line.json 2>/dev/null
grep -E "^k_" $out/sq.log | cut -c1-200; tail -2 $out/bench.err

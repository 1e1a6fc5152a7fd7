#!/bin/bash
# usage: tools/profile_round.sh <tag>   (GPU box, through gpurun)  — every measurement the round's profiles/ entries come from, in one call:
#  1 default bench line   2 rocprofv3 --kernel-trace --stats of the same S64 leg   3 TCC FETCH_SIZE / WRITE_SIZE passes (S64 and S128)
#  4 SQ / LDS / L2 counter passes on tools/solo_run.py   5 bench under torch.distributed.run at world size 1 (RCCL)
tag=$1
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/round_$tag
mkdir -p $out
cd $repo
python bench.py --steps 20 --warmup 3 > $out/bench_line.json 2> $out/bench.err
cp $repo/gpurun_out/bench_detail.json $out/bench_detail.json 2>/dev/null
CC_ASSOC_ROUNDS=1 tools/prof.sh $tag --steps 40 --warmup 3 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed > $out/prof.log 2>&1
CC_ASSOC_ROUNDS=1 tools/prof.sh ${tag}_s128 --sensor s128 --firings 1700 --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed > $out/prof_s128.log 2>&1
tools/pmc.sh $tag > $out/pmc.log 2>&1
tools/pmc.sh ${tag}_s128 --sensor s128 --firings 1700 > $out/pmc_s128.log 2>&1
tools/pmc_sq.sh $tag "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
                     "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
                     "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC" \
                     "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" > $out/sq.log 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 3 \
  --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed > $out/bench_rccl_world1.json 2> $out/bench_rccl_world1.err
# every kernel alone (pipeline 0), association with / without its links wavefront; stream-count sweep; per-wave cycle counters (instrumented build)
CC_ASSOC_WAVES=4 python tools/kernel_times.py 2>&1 | grep "^pipeline" > $out/kernel_times_assoc_waves4.txt
CC_ASSOC_WAVES=3 python tools/kernel_times.py 2>&1 | grep "^pipeline" > $out/kernel_times_assoc_waves3.txt
for S in 32 64 128 256 384 512; do
  python bench.py --streams $S --steps 30 --no-cpu-baseline --no-latency --no-verify --no-s128 --no-few-streams --no-host-fed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', $S, 'Mpoints/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'dominant', d['roofline']['kernel'], 'launch_ms', round(d['roofline']['launch_ms'],3), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()})"
done > $out/stream_sweep.txt
[ -f continuous_clustering_amd/libcc_hip_abstats.so ] && { AB_NB=12 python tools/prof_assocb.py 32; AB_NB=12 python tools/prof_assocb.py 256; } > $out/assocb_phase_clocks.txt 2>&1
[ -f continuous_clustering_amd/libcc_hip_abstatsw.so ] && { AB_NB=12 AB_LIB=libcc_hip_abstatsw.so AB_WORKER=1 python tools/prof_assocb.py 32; AB_NB=12 AB_LIB=libcc_hip_abstatsw.so AB_WORKER=1 python tools/prof_assocb.py 256; } >> $out/assocb_phase_clocks.txt 2>&1
# round 5: the measured vector-issue roof, the phase clocks of a one-firing call, trees unfinished at a time (bench scene / vegetation)
[ -x tools/ubench/valu_issue ] && (cd tools/ubench && ./valu_issue) > $out/valu_issue.txt 2>&1
{ python tools/latency_probe.py | tail -1
  [ -f continuous_clustering_amd/libcc_hip_sfstats.so ] && CC_HIP_LIB=libcc_hip_sfstats.so python tools/sf_probe.py
  [ -f continuous_clustering_amd/libcc_hip_abstats.so ] && CC_HIP_LIB=libcc_hip_abstats.so python tools/ab_small_probe.py
  [ -f continuous_clustering_amd/libcc_hip_abstatsw.so ] && CC_HIP_LIB=libcc_hip_abstatsw.so AB_WORKER=1 python tools/ab_small_probe.py; } 2>&1 | grep -v amdgpu.ids > $out/call_latency.txt
for a in "bench 0" "cluttered 0.1" "sparse 0.1" "near 0.1"; do python tools/unfinished_probe.py $a 2>&1 | tail -1; done > $out/unfinished_trees.txt
cp $repo/gpurun_out/prof_$tag/*kernel_stats.csv $out/kernel_stats_s64.csv 2>/dev/null
cp $repo/gpurun_out/prof_${tag}_s128/*kernel_stats.csv $out/kernel_stats_s128.csv 2>/dev/null
cp $repo/gpurun_out/prof_$tag/bench_line.json $out/prof_bench_line.json 2>/dev/null
cp $repo/gpurun_out/prof_${tag}_s128/bench_line.json $out/prof_bench_line_s128.json 2>/dev/null
ls -la $out; tail -3 $out/bench.err; grep -E "^k_" $out/sq.log | head -60

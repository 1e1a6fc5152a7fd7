#!/bin/bash
# usage: tools/r06_profile_final.sh <tag>   (GPU box, through gpurun) — every measurement the round's profiles/ entries come from, in one call:
#  1 default bench line (driver shape)  2 rocprofv3 --kernel-trace --stats of the S64 leg, the S128 leg and the vegetation leg
#  3 TCC FETCH_SIZE / WRITE_SIZE passes (S64 and S128)  4 SQ / LDS / L2 / occupancy counter passes on tools/solo_run.py
#  5 bench under torch.distributed.run at world size 1 (RCCL)  6 every kernel alone  7 stream-count sweep
tag=$1
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/round_$tag
mkdir -p $out
cd $repo
python bench.py --steps 20 --warmup 3 > $out/bench_line.json 2> $out/bench.err
cp $repo/gpurun_out/bench_detail.json $out/bench_detail.json 2>/dev/null
tools/prof.sh $tag --steps 40 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-cluttered > $out/prof.log 2>&1
tools/prof.sh ${tag}_s128 --sensor s128 --firings 1700 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-cluttered > $out/prof_s128.log 2>&1
# the vegetation leg alone under rocprofv3 (bench.py: cluttered)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_veg && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_veg -o veg -- python $repo/tools/cluttered_leg.py 256 12 > $out/veg_line.json 2> /tmp/prof_veg.err
  f=$(find /tmp/prof_veg -name "*kernel_stats.csv" | head -1); head -1 $f > $out/kernel_stats_vegetation.csv; grep -E "cck::|k_begin_batch|k_gate_out" $f >> $out/kernel_stats_vegetation.csv )
tools/pmc.sh $tag --repeats 1 --no-cluttered > $out/pmc.log 2>&1
tools/pmc.sh ${tag}_s128 --sensor s128 --firings 1700 --repeats 1 --no-cluttered > $out/pmc_s128.log 2>&1
tools/pmc_sq.sh $tag "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
                     "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
                     "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY" \
                     "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC" \
                     "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" > $out/sq.log 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 3 \
  --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-cluttered > $out/bench_rccl_world1.json 2> $out/bench_rccl_world1.err
tools/solo_prof.sh $tag 256 > $out/kernels_alone.txt 2>&1
for S in 32 64 128 256 384 512; do
  python bench.py --streams $S --steps 30 --repeats 1 --no-cpu-baseline --no-latency --no-verify --no-s128 --no-few-streams --no-host-fed --no-cluttered 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', $S, 'Mpoints/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'dominant', d['roofline']['kernel'], 'launch_ms', round(d['roofline']['launch_ms'],3), {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()})"
done > $out/stream_sweep.txt
cp $repo/gpurun_out/prof_$tag/*kernel_stats.csv $out/kernel_stats_s64.csv 2>/dev/null
cp $repo/gpurun_out/prof_${tag}_s128/*kernel_stats.csv $out/kernel_stats_s128.csv 2>/dev/null
cp $repo/gpurun_out/prof_$tag/bench_line.json $out/prof_bench_line.json 2>/dev/null
cp $repo/gpurun_out/prof_${tag}_s128/bench_line.json $out/prof_bench_line_s128.json 2>/dev/null
ls -la $out; tail -3 $out/bench.err; cat $out/stream_sweep.txt; grep -E "^k_" $out/sq.log | head -40 | cut -c1-200

#!/bin/bash
# usage: tools/pmc.sh <tag>   (run on the GPU box through gpurun)
# HBM traffic of our kernels from the TCC counters, one counter per pass (MI355X_MICROARCH.md "HBM", "rocprofv3 PMC slots").
tag=$1; shift; extra="$@"
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $repo/gpurun_out/pmc_$tag
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -o pmc -- python $repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-verify $extra > /tmp/pmc_$ctr.log 2>&1
  f=$(find /tmp/pmc_$ctr -name "*counter_collection.csv" | head -1)
  echo "$ctr -> $f"
  head -1 $f > $repo/gpurun_out/pmc_$tag/${ctr}.csv
  grep "cck::" $f >> $repo/gpurun_out/pmc_$tag/${ctr}.csv
  wc -l $repo/gpurun_out/pmc_$tag/${ctr}.csv
done
head -3 $repo/gpurun_out/pmc_$tag/FETCH_SIZE.csv | cut -c1-400

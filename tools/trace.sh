#!/bin/bash
# usage: tools/trace.sh <tag> <bench args...>  -> gpurun_out/trace_<tag>/kernel_trace.csv (+ stats) of a short bench run
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$tag -o bench -- python $repo/bench.py "$@" > /tmp/trace_$tag.log 2>&1
mkdir -p $repo/gpurun_out/trace_$tag
grep '^{' /tmp/trace_$tag.log > $repo/gpurun_out/trace_$tag/bench_line.json
for f in $(find /tmp/trace_$tag -type f -name "*kernel_stats.csv"); do cp $f $repo/gpurun_out/trace_$tag/; done
f=$(find /tmp/trace_$tag -type f -name "*kernel_trace.csv" | head -1)
head -1 $f > $repo/gpurun_out/trace_$tag/kernel_trace.csv
grep "cck::\|k_begin_batch\|k_clear" $f >> $repo/gpurun_out/trace_$tag/kernel_trace.csv
ls -la $repo/gpurun_out/trace_$tag

#!/bin/bash
# usage: tools/prof.sh <tag> <bench args...>   (run on the GPU box through gpurun)
# rocprofv3 --kernel-trace --stats of bench.py; only the stats CSVs travel back (the raw trace is large).
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o bench -- python $repo/bench.py "$@" > /tmp/prof_$tag.log 2>&1
mkdir -p $repo/gpurun_out/prof_$tag
grep '^{' /tmp/prof_$tag.log > $repo/gpurun_out/prof_$tag/bench_line.json
find /tmp/prof_$tag -type f | head -20
for f in $(find /tmp/prof_$tag -type f -name "*stats*"); do cp $f $repo/gpurun_out/prof_$tag/; done
ls -la $repo/gpurun_out/prof_$tag; cat $repo/gpurun_out/prof_$tag/*kernel_stats.csv | head -12
tail -5 /tmp/prof_$tag.log | cut -c1-300

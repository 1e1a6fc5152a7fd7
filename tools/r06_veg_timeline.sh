#!/bin/bash
# kernel timeline of the vegetation leg (bench.py: cluttered) under rocprofv3 --kernel-trace
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_vegtl
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_vegtl -o veg -- python $repo/tools/cluttered_leg.py 256 12 > /tmp/vegtl.json 2> /tmp/vegtl.err
t=$(find /tmp/prof_vegtl -name "*kernel_trace.csv" | head -1)
mkdir -p $repo/gpurun_out/r06q
python $repo/tools/timeline.py $t 260 > $repo/gpurun_out/r06q/veg_timeline.txt
sed -n 60,130p $repo/gpurun_out/r06q/veg_timeline.txt

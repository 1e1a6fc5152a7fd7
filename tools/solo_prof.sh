#!/bin/bash
# usage: tools/solo_prof.sh <tag> [streams] [libcc_hip_<variant>.so]   (GPU box) — tools/solo_run.py (pipeline off) under rocprofv3: kernel stats, then FETCH_SIZE / WRITE_SIZE passes
tag=$1; S=${2:-256}; lib=$3
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$repo/gpurun_out/solo_$tag; mkdir -p $out
rm -rf /tmp/solo_st
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/solo_st -o solo -- python $repo/tools/solo_run.py $S $lib > /tmp/solo_st.log 2>&1
cp $(find /tmp/solo_st -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
grep "cck::" $out/kernel_stats.csv | awk -F'","|",|,' '{print $1, $2, $4}' | cut -c1-60,200- | head -20
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/solo_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/solo_$ctr -o pmc -- python $repo/tools/solo_run.py $S $lib > /tmp/solo_$ctr.log 2>&1
  f=$(find /tmp/solo_$ctr -name "*counter_collection.csv" | head -1)
  head -1 $f > $out/${ctr}.csv; grep "cck::" $f >> $out/${ctr}.csv
done
python $repo/tools/solo_sum.py $out

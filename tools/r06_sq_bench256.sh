#!/bin/bash
# occupancy counters of the headline leg's kernels (256 streams, bench.py): rocprofv3 serialises dispatches while it collects counters, so these are the
# kernels ALONE in the launch shapes of the 256-stream step, not the pipelined step
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$repo/gpurun_out/r06q; mkdir -p $out; : > $out/sq_bench256.txt
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LEVEL_WAVES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/sqb_$i
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/sqb_$i -o pmc -- python $repo/bench.py --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-latency --no-s128 --no-few-streams --no-host-fed --no-verify --no-cluttered > /tmp/sqb_$i.log 2>&1
  f=$(find /tmp/sqb_$i -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "pass $i: no output" >> $out/sq_bench256.txt; continue; }
  python - "$f" >> $out/sq_bench256.txt <<'PY'
import csv, sys, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_\w+)", row["Kernel_Name"])
    if m: acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sorted(v)[len(v)//2]) for c, v in acc[k].items()})
PY
done
cat $out/sq_bench256.txt | grep -E "k_insert_par |k_assocb|k_scan2 |k_seg_scan|k_publish|k_ego"

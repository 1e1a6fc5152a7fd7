#!/bin/bash
# usage: tools/pmc_sq.sh <tag> "<counters of pass 1>" "<counters of pass 2>" ...   (GPU box, through gpurun)
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $repo/gpurun_out/sq_$tag
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rm -rf /tmp/sq_$i
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/sq_$i -o pmc -- python $repo/tools/solo_run.py 64 > /tmp/sq_$i.log 2>&1
  f=$(find /tmp/sq_$i -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $i: no output"; tail -5 /tmp/sq_$i.log; continue; fi
  python - "$f" <<'PY'
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

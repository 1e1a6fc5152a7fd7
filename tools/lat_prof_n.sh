#!/bin/bash
# usage (GPU box): tools/lat_prof_n.sh <n>  — rocprofv3 kernel stats of calls of n firings on one stream
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/latp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/latp -o lat -- python $GRAFT_REPO_ROOT/tools/latency_probe_n.py $1 > /tmp/latp.log 2>&1
tail -1 /tmp/latp.log
python - <<'P'
import csv,glob,re
f=glob.glob('/tmp/latp/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    m=re.search(r'(cck::|anonymous namespace.::)(\w+)',r['Name'])
    if m and int(r['Calls']) > 10: print(f"{m.group(2):18s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
P

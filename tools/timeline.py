"""Print the kernel timeline (start/end in ms, relative) of the last few batches from a rocprofv3 kernel-trace CSV.

usage: python tools/timeline.py <kernel_trace.csv> [rows] [<memory_copy_trace.csv>]  — with the third argument the copies of a
`--memory-copy-trace` run are merged into the same timeline (rows "copy <direction> <bytes>").
"""
import csv, sys, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    if m: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1), r.get("Stream_Id", r.get("Queue_Id", "?"))))
if len(sys.argv) > 3:
    for r in csv.DictReader(open(sys.argv[3])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy %s %s" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?"))),
                     r.get("Stream_Id", r.get("Queue_Id", "-"))))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = rows[0][0]
for s, e, k, q in rows:
    print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f}  {(e - s) / 1e6:7.3f} ms  q{q:>3}  {k}")

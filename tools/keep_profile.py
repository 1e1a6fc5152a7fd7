#!/usr/bin/env python3
"""Copy the judged part of a rocprofv3 --stats run from gpurun_out/ into profiles/ (tracked):
the rows of our own kernels (cck::*, engine helper kernels) from *_kernel_stats.csv plus the bench JSON line."""
import csv, json, os, sys
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", f"prof_{tag}")
rows = list(csv.reader(open(os.path.join(src, "bench_kernel_stats.csv"))))
hdr, body = rows[0], rows[1:]
mine = [r for r in body if "cck::" in r[0] or "k_begin_batch" in r[0] or "k_clear" in r[0]]
other_ns = sum(int(r[2]) for r in body if r not in mine)
os.makedirs("profiles", exist_ok=True)
with open(os.path.join("profiles", f"{name}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f); w.writerow(hdr)
    for r in mine: w.writerow(r)
    w.writerow(["(all other kernels: torch input generation before the timed region)", "", other_ns, "", "", "", "", ""])
line = open(os.path.join(src, "bench_line.json")).read().strip()
open(os.path.join("profiles", f"{name}_bench_line.json"), "w").write(line + "\n")
for r in mine: print(r[0][:60], r[1], "avg_ms", float(r[3]) / 1e6)

#!/usr/bin/env python
"""usage: tools/pmc_to_traffic.py <tag>   gpurun_out/pmc_<tag>/{FETCH,WRITE}_SIZE.csv -> profiles/traffic.json (+ copies the CSVs).

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB... per MI355X_MICROARCH.md: both counters are in KB units and the
gfx950 FETCH_SIZE under-counts wide reads by 2x; each counter comes from its own --pmc pass (tools/pmc.sh).
"""
import csv, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
round_tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
out_name = sys.argv[3] if len(sys.argv) > 3 else "traffic.json"
src = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
ALIAS = {"k_insert_multi": "insert_multi", "k_scan2": "scan_packed", "k_ego": "ego", "k_prep": "prep", "k_insert_par": "insert_parallel", "k_insert2": "insert", "k_seg_pre": "segment_pre", "k_seg_scan": "segment", "k_scan": "scan",
         "k_assoc_lds": "assoc_lds_1wave", "k_assoc2": "assoc_2wave", "k_assoc3": "assoc_serial", "k_assocb": "assoc_lds", "k_associate": "assoc_global", "k_publish": "publish", "k_table": "table"}
vals = {}
counts = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(list)
    with open(os.path.join(src, ctr + ".csv")) as f:
        for row in csv.DictReader(f):
            m = re.search(r"cck::(k_\w+)", row["Kernel_Name"])
            if m and row["Counter_Name"] == ctr:
                acc[m.group(1)].append(float(row["Counter_Value"]))
    # the first launch of every kernel processes a cold ring; use the median launch. Kernels that were launched fewer times than the batches
    # of the run (k_insert_multi / k_prep / k_insert2 with skip_idle_fallbacks: only where a stream needed them, i.e. the start-up batch) are
    # not part of a steady-state step
    n_ref = max(len(v) for v in acc.values())
    counts[ctr] = {k: len(v) for k, v in acc.items()}
    vals[ctr] = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
    shutil.copy(os.path.join(src, ctr + ".csv"), os.path.join(ROOT, "profiles", f"{round_tag}_pmc_{ctr}{'' if out_name == 'traffic.json' else '_s128'}.csv"))
out = {}
note = ("FETCH_SIZE doubled (gfx950 wide-read correction of MI355X_MICROARCH.md); separate --pmc passes; median launch; "
        "256 streams x one rotation of firings per launch")
for k in sorted(vals["FETCH_SIZE"]):
    f, w = vals["FETCH_SIZE"][k], vals["WRITE_SIZE"].get(k, 0.0)
    n_ref = max(counts["FETCH_SIZE"].values())
    steady = counts["FETCH_SIZE"][k] >= n_ref
    rec = {"kernel": k, "fetch_size_kb_raw": f, "write_size_kb": w, "hbm_bytes_per_launch": (2 * f + w) * 1024.0,
           "launches_in_run": counts["FETCH_SIZE"][k], "in_steady_state_step": bool(steady), "note": note}
    out[k] = rec
    if k in ALIAS:
        out[ALIAS[k]] = rec
json.dump(out, open(os.path.join(ROOT, "profiles", out_name), "w"), indent=1)
tot = sum(r["hbm_bytes_per_launch"] * (2 if k == "k_assoc3" else 1) * 0 if not r["in_steady_state_step"] else r["hbm_bytes_per_launch"] for k, r in out.items() if k.startswith("k_"))
for k, r in out.items():
    if k.startswith("k_"):
        print(f"{k:14s} {r['hbm_bytes_per_launch'] / 1e9:7.3f} GB/launch" + ("" if r["in_steady_state_step"] else "   (start-up batch only: not in a steady-state step)"))
print(f"total {tot / 1e9:.3f} GB per step")

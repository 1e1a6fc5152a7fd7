"""Markdown table of profiles/ROOFLINE.md from the committed files: python tools/roofline_table.py r03_final [s128]
frac = algorithmic bytes per launch / rocprofv3 average launch duration / 8 TB/s; traffic from profiles/traffic*.json."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
s128 = len(sys.argv) > 2
alg = 1044.5e6 if s128 else 702.9e6
stats = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats{'_s128' if s128 else ''}.csv")
traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic_s128.json" if s128 else "traffic.json")))
rows = []
for r in csv.reader(open(stats)):
    m = re.search(r"cck::(k_\w+)", r[0])
    if m:
        rows.append((m.group(1), int(r[1]), float(r[3]) / 1e6))
rows.sort(key=lambda x: -x[2] * x[1])
print("| kernel | launches | avg launch, pipelined (rocprofv3) | frac of 8 TB/s | HBM traffic / launch | traffic / algorithmic |")
print("|---|---|---|---|---|---|")
for k, n, ms in rows:
    t = traffic.get(k, {})
    tb = t.get("hbm_bytes_per_launch")
    steady = t.get("in_steady_state_step", True)
    frac = alg / (ms * 1e-3) / 8e12
    print(f"| `{k}` | {n} | {ms:.3f} ms | {frac:.4f} | " + (f"{tb / 1e9:.3f} GB" + ("" if steady else " (start-up batch only)") if tb else "—") + " | " +
          (f"{tb / alg:.2f}" if tb and steady else "—") + " |")

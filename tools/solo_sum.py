"""Sum the counter CSVs of tools/solo_prof.sh per kernel (KB -> GB; FETCH doubled per MI355X_MICROARCH.md) and print them next to the kernel times."""
import csv, sys, collections, re
out = sys.argv[1]
def name(n):
    m = re.search(r"cck::(\w+)", n); return m.group(1) if m else n[:30]
tot = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f"{out}/{ctr}.csv")):
        d[name(r["Kernel_Name"])] += float(r["Counter_Value"]); n[name(r["Kernel_Name"])] += 1
    tot[ctr] = (d, n)
t = {}
for r in csv.DictReader(open(f"{out}/kernel_stats.csv")):
    if "cck::" in r["Name"]:
        t[name(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
for k in sorted(t, key=lambda k: -t[k][1]):
    f, nf = tot["FETCH_SIZE"][0].get(k, 0), max(tot["FETCH_SIZE"][1].get(k, 1), 1)
    w = tot["WRITE_SIZE"][0].get(k, 0)
    # counter rows are per dispatch (and per XCD instance on some versions): normalise by launches of the kernel stats
    calls = t[k][0]
    print(f"{k:18s} calls {calls:3d} avg {t[k][1]:7.3f} ms  fetch {2 * f * 1024 / calls / 1e9:7.3f} GB  write {w * 1024 / calls / 1e9:7.3f} GB per launch")

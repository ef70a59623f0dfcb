"""Condenses the rocprofv3 CSVs written by tools/profile.sh into one text summary (committed under profiles/)."""
import csv, glob, sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel-trace --stats ==")
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:110]:110s} calls={r['Calls']:>4s} avg_ns={float(r['AverageNs']):14.0f} pct={r['Percentage']}")
print()
print("== PMC (per dispatch of the dominant accel kernel; mean over its dispatches) ==")
vals = defaultdict(list)
for f in glob.glob(f"{out}/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ("accel_f" in r["Kernel_Name"] or "accel_sym_f" in r["Kernel_Name"]) and "reduce" not in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in vals.items()}
for k in sorted(m):
    print(f"{k:28s} {m[k]:.6e}  (n={len(vals[k])})")
if "FETCH_SIZE" in m:
    print(f"HBM read bytes/launch  ~ {2 * m['FETCH_SIZE'] * 1024:.4e}  (FETCH_SIZE KiB x 2: gfx950 correction, MI355X_MICROARCH.md §HBM)")
if "WRITE_SIZE" in m:
    print(f"HBM write bytes/launch ~ {m['WRITE_SIZE'] * 1024:.4e}")

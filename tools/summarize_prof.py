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

# derived: executed VALU flop rate and HBM GB/s of the dominant kernel (packed instructions count once and do two lanes' worth of work per lane)
dur = None
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ("accel_f" in r["Name"] or "accel_sym_f" in r["Name"]) and "reduce" not in r["Name"]:
            dur = float(r["AverageNs"]) * 1e-9
            break
for fp, peak in (("F32", 157.3), ("F64", 78.6)):
    k = [f"SQ_INSTS_VALU_{op}_{fp}" for op in ("FMA", "ADD", "MUL", "TRANS")]
    if all(x in m for x in k) and m[k[3]] > 0 and dur:
        pk = 2.0 if fp == "F32" else 1.0          # the fp32 kernels issue their full-rate math as v_pk_*_f32; v_rsq_f32 is scalar per lane
        flop = (2 * pk * m[k[0]] + pk * (m[k[1]] + m[k[2]]) + m[k[3]]) * 64
        print(f"executed VALU flop/launch ~ {flop:.4e}  ({fp}; fma = 2, packed = 2 lanes)  -> {flop / dur / 1e12:.1f} TFLOP/s executed = {flop / dur / 1e12 / peak:.3f} of the {peak} TF/s vector peak "
              f"(kernel-trace duration {dur * 1e3:.2f} ms)")
        print(f"pair terms evaluated/launch ~ {m[k[3]] * 64:.4e}  (one rsqrt each)")
if dur and "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    print(f"HBM traffic rate ~ {(2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024 / dur / 1e9:.1f} GB/s of ~8000 (the partial-sum slabs; the position stream itself is 16 N (fp32) / 32 N (fp64) bytes per launch, L2-resident)")

"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/summarize_kernel_stats.py <dir> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms in %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%-100s calls=%7s avg_us=%9.2f pct=%s" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))

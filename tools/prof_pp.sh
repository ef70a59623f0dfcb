#!/bin/bash
# rocprofv3 kernel trace of the comoving Ewald `pp` run (run on the GPU box from the repo root). Usage: tools/prof_pp.sh [nsteps] [f32|f64]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out/prof_pp; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/run_pp_reference.py ${1:-20} 64 1 ${2:-f32} > $OUT/trace.out 2>&1 || exit 1
F=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY' | tee $OUT/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:10]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:10.2f} {100*float(r['TotalDurationNs'])/tot:6.1f}")
PY

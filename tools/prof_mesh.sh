#!/bin/bash
# rocprofv3 kernel trace of the PM / P3M force evaluation (run on the GPU box from the repo root)
# Usage: tools/prof_mesh.sh <n_side> <pm|p3m|p3m2> <lattice|clustered>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
N=${1:-128}; export NB_MESH_SOLVER=${2:-p3m}; export NB_MESH_KIND=${3:-lattice}; export NB_MESH_REPS=5
OUT=gpurun_out/prof_mesh_${N}_${NB_MESH_SOLVER}_${NB_MESH_KIND}; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_mesh.py $N > $OUT/trace.out 2>&1 || exit 1
cat $OUT/trace.out | tail -3
F=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY' | tee $OUT/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:10.2f} {100*float(r['TotalDurationNs'])/tot:6.1f}")
PY

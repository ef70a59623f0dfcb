#!/bin/bash
# Second roofline ("HBM-bandwidth-bound scatter/FFT"): every kernel of one PM force evaluation against the HBM
# roofline — rocprofv3 kernel durations, the algorithmic bytes each kernel has to move (tools/summarize_mesh_roofline.py states them),
# fraction of 8 TB/s. One JSON line per configuration.   usage: tools/mesh_roofline.sh [n_side ...] > gpurun_out/…/mesh_roofline.jsonl
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for N in ${@:-256}; do
  for KIND in lattice clustered; do
    OUT=gpurun_out/prof_mesh_roofline_${N}_${KIND}; rm -rf $OUT; mkdir -p $OUT
    NB_MESH_SOLVER=pm NB_MESH_KIND=$KIND NB_MESH_REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_mesh.py $N > $OUT/trace.out 2>&1 || exit 1
    python3 tools/summarize_mesh_roofline.py "$(find $OUT/trace -name '*kernel_stats.csv' | head -1)" $N $KIND 6 || exit 1
  done
done

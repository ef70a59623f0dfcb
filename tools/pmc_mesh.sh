#!/bin/bash
# PMC counters of the P3M short-range kernel (run on the GPU box from the repo root). Usage: tools/pmc_mesh.sh [n_side]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export NB_MESH_SOLVER=p3m NB_MESH_KIND=lattice NB_MESH_REPS=2
OUT=gpurun_out/pmc_mesh; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -- python3 tools/bench_mesh.py ${1:-128} > $OUT/a.out 2>&1 || { tail -5 $OUT/a.out; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/b -- python3 tools/bench_mesh.py ${1:-128} > $OUT/b.out 2>&1 || { tail -5 $OUT/b.out; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for sub in ("a", "b"):
    f = glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if "short_range_kernel" in row["Kernel_Name"] or "short_range_tile_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:28s} {sum(v)/len(v):.4e}  (n={len(v)})")
PY

#!/bin/bash
# PMC counters of the PM brick kernels (deposit, assemble, gather) — what they wait for. Usage (GPU box, repo root): tools/pmc_brick.sh [n_side]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export NB_MESH_SOLVER=pm NB_MESH_KIND=${NB_MESH_KIND:-lattice} NB_MESH_REPS=2
OUT=gpurun_out/pmc_brick; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -- python3 tools/bench_mesh.py ${1:-256} > $OUT/a.out 2>&1 || { tail -5 $OUT/a.out; exit 1; }
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 tools/bench_mesh.py ${1:-256} > $OUT/b.out 2>&1 || { tail -5 $OUT/b.out; exit 1; }

python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("a", "b"):
    for f in glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            for key in ("brick_deposit", "brick_assemble", "brick_gather"):
                if key in row["Kernel_Name"]:
                    acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key in acc:
    print("==", key)
    for k, v in sorted(acc[key].items()):
        print(f"  {k:24s} {sum(v)/len(v):.4e}  (n={len(v)})")
PY

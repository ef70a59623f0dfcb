export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc_bh
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_bh/a -- python3 tools/bench_bh.py 0.5 4194304 > gpurun_out/pmc_bh/a.out 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_bh/b -- python3 tools/bench_bh.py 0.5 4194304 > gpurun_out/pmc_bh/b.out 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/pmc_bh/c -- python3 tools/bench_bh.py 0.5 4194304 > gpurun_out/pmc_bh/c.out 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
v=defaultdict(list)
for f in glob.glob('gpurun_out/pmc_bh/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'walk_list' in r['Kernel_Name']:
            v[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(v): print(f"{k:28s} {sum(v[k])/len(v[k]):.4e} (n={len(v[k])})")
PY

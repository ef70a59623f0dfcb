#!/bin/bash
# rocprofv3 passes for the headline benchmark (run on the GPU box from the repo root):
#   1. --kernel-trace --stats   (per-kernel time)            -> gpurun_out/prof/trace
#   2-5. --pmc passes (own runs, counters only)               -> gpurun_out/prof/pmc{A,B,C,D}
# Usage: tools/profile.sh [extra bench.py args]     (PROF_OUT=dir overrides the output directory; "--dtype f64" switches pass B to the F64 counters)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=${PROF_OUT:-gpurun_out/prof}; rm -rf $OUT; mkdir -p $OUT
FP=F32; case "$*" in *f64*) FP=F64;; esac
B="python3 bench.py --no-cpu-baseline --no-clock-probe --traffic none $*"   # --traffic none: bench.py must not start its own rocprofv3 children under this profiler
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 3 --warmup 1 > $OUT/trace.out 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcA -- $B --steps 1 --warmup 0 > $OUT/pmcA.out 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_$FP SQ_INSTS_VALU_FMA_$FP SQ_INSTS_VALU_MUL_$FP SQ_INSTS_VALU_ADD_$FP SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmcB -- $B --steps 1 --warmup 0 > $OUT/pmcB.out 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcC -- $B --steps 1 --warmup 0 > $OUT/pmcC.out 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcD -- $B --steps 1 --warmup 0 > $OUT/pmcD.out 2>&1 || exit 1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

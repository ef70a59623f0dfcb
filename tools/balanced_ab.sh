#!/bin/bash
# A/B of the block count of the pair-symmetric fp32 kernel on ONE box (NB_F32_SYM_BLOCKS / NB_F32_SYM_ROUNDS are read once per process):
# default = the device's resident block slots (512), equal shares of steps; 4104 blocks = the round-3 launch shape at N = 2^20 (342 x 12 equal
# blocks on 512 slots: 8.02 rounds); multiples of the slots stay balanced. Then rocprofv3 kernel traces of the bench at N = 65 536 and 2^20.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for n in 1048576 65536; do
  for spec in "0 1" "4104 1" "0 2" "0 4" "0 8" "1024 1" "2176 1"; do
    set -- $spec
    echo "## n=$n NB_F32_SYM_BLOCKS=$1 NB_F32_SYM_ROUNDS=$2"
    NB_F32_SYM_BLOCKS=$1 NB_F32_SYM_ROUNDS=$2 python tools/sym_check.py f32 $n 2>&1 | grep uniform || exit 1
  done
done
for n in 65536 1048576; do
  OUT=gpurun_out/r04/trace_n$n; rm -rf $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-clock-probe --traffic none --n $n --steps 5 --warmup 2 > $OUT.out 2>&1 || exit 1
  echo "## kernel stats n=$n"; python3 tools/summarize_kernel_stats.py $OUT 2>/dev/null | head -12 || find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -8
done

#!/bin/bash
# Priority turns between the two co-resident waves of a SIMD (time slices of 2^s ticks of the 100 MHz counter, looked at every 8 steps), one box:
# fp64 with and without; fp32 rounds of equal shares with the turns on (does the rule for the round count still hold?)
cd "$(dirname "$0")/.."
for n in 24576 65536 131072 262144 524288; do
  for ts in 0 13 14 15; do
    echo "## f64 n=$n NB_F64_SYM_TURN_SHIFT=$ts $(NB_F64_SYM_TURN_SHIFT=$ts python tools/sym_check.py f64 $n 2>&1 | grep uniform | awk '{print $(NF-3), $(NF-2), $(NF-1)}')"
  done
done
for n in 65536 131072 262144; do
  for r in 0 1 2 4 8; do
    echo "## f32 n=$n NB_F32_SYM_ROUNDS=$r (turns on) $(NB_F32_SYM_ROUNDS=$r python tools/sym_check.py f32 $n 2>&1 | grep uniform | awk '{print $(NF-3), $(NF-2), $(NF-1)}')"
  done
done
for n in 65536 1048576; do echo "## f32 general n=$n turns 0/14: $(NB_F32_SYM_TURN_SHIFT=0 python tools/sym_check.py f32 $n 2>&1 | grep general | awk '{print $(NF-3), $(NF-2)}') / $(python tools/sym_check.py f32 $n 2>&1 | grep general | awk '{print $(NF-3), $(NF-2)}')"; done

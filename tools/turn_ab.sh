#!/bin/bash
# A/B of the priority turn-taking between the two co-resident waves of a SIMD (NB_F32_SYM_TURN_SHIFT: 0 = off, s = time slices of 2^s ticks of the 100 MHz counter), one box
cd "$(dirname "$0")/.."
for n in 65536 131072 262144 1048576; do
  for ts in 0 8 10 12 14; do
    echo "## n=$n NB_F32_SYM_TURN_SHIFT=$ts"
    NB_F32_SYM_TURN_SHIFT=$ts python tools/sym_check.py f32 $n 2>&1 | grep uniform || exit 1
  done
done
echo "## block timeline with the default turn"; python tools/block_trace.py 65536 1048576
for n in 65536 262144 524288; do
  for ts in 0 10; do
    echo "## f64 n=$n NB_F64_SYM_TURN_SHIFT=$ts"
    NB_F64_SYM_TURN_SHIFT=$ts python tools/sym_check.py f64 $n 2>&1 | grep uniform || exit 1
  done
done

#!/bin/bash
# Rounds of equal shares (NB_F32_SYM_ROUNDS; 0 = the built-in rule) at mid-size n, one box: does a finer deal shorten the tail of two
# co-resident blocks (the first-dispatched one finishes far ahead, the other then runs alone) more than the extra block set-ups cost?
cd "$(dirname "$0")/.."
for n in 100001 131072 200000 262144 524288; do
  for r in 0 1 2 4 8 16 32; do
    echo "## n=$n NB_F32_SYM_ROUNDS=$r"
    NB_F32_SYM_ROUNDS=$r python tools/sym_check.py f32 $n 2>&1 | grep uniform || exit 1
  done
done

#!/bin/bash
# Rounds of equal shares (NB_F32_SYM_ROUNDS; 0 = the built-in rule) with the priority turns on, one box, two repetitions
cd "$(dirname "$0")/.."
for rep in 1 2; do
for n in 131072 262144 524288 1048576; do
  for r in 0 1 2 4 8; do
    echo "## n=$n NB_F32_SYM_ROUNDS=$r $(NB_F32_SYM_ROUNDS=$r python tools/sym_check.py f32 $n 2>&1 | grep uniform | awk '{print $(NF-3), $(NF-2), $(NF-1)}')"
  done
done
done

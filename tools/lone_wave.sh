#!/bin/bash
# How fast is ONE wave per SIMD? 256 blocks (one per CU, alone for the whole launch) against 512 (two per CU) — the pair-symmetric fp32 kernel
cd "$(dirname "$0")/.."
for n in 65536 1048576; do
  for b in 256 512; do
    echo "## n=$n NB_F32_SYM_BLOCKS=$b"
    NB_F32_SYM_BLOCKS=$b python tools/sym_check.py f32 $n 2>&1 | grep uniform | awk '{print $(NF-3), $(NF-2), $(NF-1)}'
    NB_F32_SYM_BLOCKS=$b python tools/block_trace.py $n 2>&1 | grep "makespan\|shorter of"
  done
done

#!/bin/bash
cd "$(dirname "$0")/.."
for n in 262144; do for probe in "" 1; do for ts in 0 14; do
  echo "## n=$n probe=${probe:-0} NB_F32_SYM_TURN_SHIFT=$ts"
  BLOCK_TRACE_PROBE=$probe NB_F32_SYM_TURN_SHIFT=$ts python tools/block_trace.py $n 2>&1 | grep "makespan\|CU busy\|clock probe" | cut -c1-330
done; done; done

#!/bin/bash
# Block-count quantisation of the pair-symmetric fp32 kernel: a launch of nsets x groups equal blocks on 512 resident slots (2 per CU) takes
# ceil(blocks / 512) rounds. Sweep the group count (NB_F32_SYM_GROUPS, read once per process) at the headline size and at configs[1]'s.
# usage: tools/groups_sweep.sh > gpurun_out/r04/groups_sweep.txt
cd "$(dirname "$0")/.."
for n in 1048576 65536; do
  for g in 0 13 16 19 22 25 28 31 47; do
    echo "## n=$n NB_F32_SYM_GROUPS=$g"
    NB_F32_SYM_GROUPS=$g python tools/sym_check.py f32 $n 2>&1 | grep uniform || exit 1
  done
done

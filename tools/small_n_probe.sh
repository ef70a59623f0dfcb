#!/bin/bash
# Why does the pair-symmetric fp32 kernel need 26 cycles per 64 delivered interactions at N = 65 536 but 20 at 2^20? (round 4)
#  1. odd against even set counts with 8 sinks per lane (an even count has TWO one-sided offsets — the set on itself and the set half-way
#     round — an odd count one), and the asymptote at large n
#  2. the counter passes of tools/profile.sh at N = 65 536
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for n in 63488 65536 129024 131072 260096 262144 1048576; do
  echo "## NB_F32_SYM_KS=8 n=$n ($(( (n + 2047) / 2048 )) sets)"
  NB_F32_SYM_KS=8 python tools/sym_check.py f32 $n 2>&1 | grep uniform || exit 1
done
PROF_OUT=gpurun_out/r04/prof_n65536 tools/profile.sh --n 65536

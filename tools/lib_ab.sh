#!/bin/bash
# A/B of two builds of libnbody_amd.so on one box (NBODY_AMD_LIB selects the library): bachelor_thesis_n_body_amd/libnbody_amd_prev.so against the current one
cd "$(dirname "$0")/.."
for rep in 1 2; do
for lib in libnbody_amd_prev.so libnbody_amd.so; do
  for n in 1048576 65536 262144; do
    echo "## $lib n=$n"
    NBODY_AMD_LIB=$PWD/bachelor_thesis_n_body_amd/$lib python tools/sym_check.py f32 $n 2>&1 | grep "uniform\|general" || exit 1
  done
done
done

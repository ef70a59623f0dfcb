#!/bin/bash
# where the pair-symmetric kernels overtake the one-sided ones: bench.py --n N with NB_F32_SYM_MIN_N / NB_F64_SYM_MIN_N = 0 (pair-symmetric wherever
# it can run) against a threshold above N (one-sided), one box
cd "$(dirname "$0")/.."
run() { python3 bench.py --n $1 --steps $2 --warmup 5 --dtype $3 --no-cpu-baseline --traffic none 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f %s' % (j['ms_per_step'], j['roofline']['kernel'].replace('accel_','').replace('_kernel','')))"; }
for rep in 1 2; do
for cfg in "f32 6144 400" "f32 8192 400" "f32 12288 400" "f32 16384 400" "f32 20480 300" "f32 24576 300" "f64 4096 400" "f64 6144 400" "f64 8192 400" "f64 12288 300" "f64 16384 300"; do
  set -- $cfg
  echo "## $1 n=$2 ms/step: pair-symmetric $(NB_F32_SYM_MIN_N=0 NB_F64_SYM_MIN_N=0 run $2 $3 $1) | one-sided $(NB_F32_SYM_MIN_N=100000000 NB_F64_SYM_MIN_N=100000000 run $2 $3 $1)"
done
done

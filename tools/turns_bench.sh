#!/bin/bash
# bench.py WITH its clock probe (the form the driver runs), priority turns off (NB_*_SYM_TURN_SHIFT=0) against the default, on one box.
# The turns were first adopted on kernel-only timings (tools/turn_ab.sh) and then cost 4-8 % in bench.py: the probe's sleeping wave shifted the
# wave-slot ids on one SIMD per XCD. This is the check that belongs to any change of the turn logic. Output: gpurun_out/r04/turns_bench.txt
cd "$(dirname "$0")/.."
run() { python3 bench.py --n $1 --steps $2 --warmup 3 --dtype $3 --no-cpu-baseline --traffic none $4 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % (j['ms_per_step']))"; }
for rep in 1 2; do
for cfg in "f32 65536 40" "f32 131072 40" "f32 262144 40" "f32 524288 16" "f32 1048576 8" "f64 65536 40" "f64 131072 40" "f64 524288 8"; do
  set -- $cfg
  off=$(NB_F32_SYM_TURN_SHIFT=0 NB_F64_SYM_TURN_SHIFT=0 run $2 $3 $1)
  on=$(run $2 $3 $1)
  offn=$(NB_F32_SYM_TURN_SHIFT=0 NB_F64_SYM_TURN_SHIFT=0 run $2 $3 $1 --no-clock-probe)
  onn=$(run $2 $3 $1 --no-clock-probe)
  echo "## $1 n=$2 ms/step with probe: turns off $off  on $on | without probe: off $offn  on $onn"
done
done

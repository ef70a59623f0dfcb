#!/bin/bash
# slice length of the priority turns (2^shift ticks of the 100 MHz counter) at the small end of the pair-symmetric range: bench.py --n N, one box
cd "$(dirname "$0")/.."
run() { python3 bench.py --n $1 --steps $2 --warmup 5 --no-cpu-baseline --traffic none 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % (j['ms_per_step']))"; }
for rep in 1 2; do
for cfg in "24576 300" "32768 300" "49152 200" "65536 100" "98304 60" "131072 40" "262144 20" "524288 10"; do
  set -- $cfg
  line="## n=$1 ms/step:"
  for ts in 0 12 13 14 15 16 -1; do line="$line shift $ts: $(NB_F32_SYM_TURN_SHIFT=$ts run $1 $2) |"; done
  echo "$line"
done
done

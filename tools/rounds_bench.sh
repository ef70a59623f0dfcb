#!/bin/bash
# bench.py (driver form) for several round counts on ONE box, with the per-XCD clocks the box holds: which round count tolerates the XCDs' clock spread?
cd "$(dirname "$0")/.."
for r in 0 1 4 8 16; do
  echo "## NB_F32_SYM_ROUNDS=$r $(NB_F32_SYM_ROUNDS=$r python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%.2f ms  %.1f MHz  %.2f cycles  xcd %s' % (j['ms_per_step'], r['sustained_clock_mhz'], r['cycles_per_64_interactions_per_simd'], [round(v) for v in r['sustained_clock_mhz_per_xcd']]))")"
done
for n in 262144 524288; do for r in 0 1 2 8; do
  echo "## n=$n NB_F32_SYM_ROUNDS=$r $(NB_F32_SYM_ROUNDS=$r python3 bench.py --n $n --steps 20 --warmup 3 --no-cpu-baseline --traffic none 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('%.3f ms  %.1f MHz  %.2f cycles' % (j['ms_per_step'], r['sustained_clock_mhz'], r['cycles_per_64_interactions_per_simd']))")"
done; done

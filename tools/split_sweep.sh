#!/bin/bash
# A/B of the source split of the fp32 force kernel at the headline size (run on the GPU box): NB_F32_SPLIT = 1, 2, 4, 8 and the default heuristic.
for s in 0 1 2 4 8 16 32; do
  echo "== NB_F32_SPLIT=$s"
  NB_F32_SPLIT=$s python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms_per_step %.3f  avg_launch_ms %.3f  value %.4e' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], j['value']))"
done

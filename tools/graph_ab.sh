#!/bin/bash
# A/B of nb_kdk_run's captured-graph replay against eager stepping, P3M, 400 steps, by particle count (run time of jobs 2-3, seconds)
for n in 16 24 32 48 64; do
  g=$(python tools/farm_job_profile.py 400 $n 2>/dev/null | sed -n 2,3p | awk '{print $5}' | tr '\n' ' ')
  e=$(NB_NO_GRAPH=1 python tools/farm_job_profile.py 400 $n 2>/dev/null | sed -n 2,3p | awk '{print $5}' | tr '\n' ' ')
  echo "n_side $n  graph: $g  eager: $e"
done

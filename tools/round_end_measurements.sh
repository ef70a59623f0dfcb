#!/bin/bash
# The round's record runs on one MI355X (from the repo root on the GPU box); outputs under gpurun_out/r02_final/, copied to profiles/ by hand.
O=gpurun_out/r02_final; mkdir -p $O
python3 bench.py --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --general-mass --no-cpu-baseline > $O/bench_n1_general_mass.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --dtype f64 --n 524288 --no-cpu-baseline > $O/bench_f64_n512k.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --dtype f64 --n 524288 --general-mass --no-cpu-baseline > $O/bench_f64_n512k_general_mass.json 2>> $O/bench.err
python3 bench.py --steps 100 --warmup 10 --n 65536 --no-cpu-baseline > $O/bench_n65536_f32.json 2>> $O/bench.err
python3 bench.py --gpus 2 --backend gloo --same-device --steps 5 --warmup 2 > $O/bench_2ranks_gloo_same_device.json 2>> $O/bench.err
python3 bench.py --gpus 2 --single-process --same-device --steps 5 --warmup 2 > $O/bench_2shards_single_process_same_device.json 2>> $O/bench.err
python3 bench.py --gpus 8 --single-process --same-device --steps 5 --warmup 2 > $O/bench_8shards_single_process_same_device.json 2>> $O/bench.err
python3 tools/farm_job_profile.py 400 > $O/farm_job_profile.txt 2>&1
python3 tools/bench_mesh.py 64 128 256 > $O/mesh_bench.txt 2>&1
python3 tools/bench_bh.py > $O/bench_bh.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export NB_NO_GRAPH=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_farm -- python3 tools/farm_job_profile.py 400 > $O/prof_farm.out 2>&1
python3 tools/summarize_kernel_stats.py $O/prof_farm 16 > $O/farm_job_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bh -- python3 tools/bench_bh.py 0.5 16777216 > $O/prof_bh.out 2>&1
python3 tools/summarize_kernel_stats.py $O/prof_bh 14 > $O/bh_n16M_kernel_stats.txt 2>&1
for f in $O/bench_*.json; do echo "$f: $(python3 -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4e int/s  %.3f ms/step  frac %.3f' % (j['value'], j['ms_per_step'], j['roofline']['frac']))")"; done
head -6 $O/farm_job_profile.txt; cat $O/farm_job_kernel_stats.txt | head -6

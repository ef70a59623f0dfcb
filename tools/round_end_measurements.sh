#!/bin/bash
# The round's record runs on one MI355X (from the repo root on the GPU box); outputs under gpurun_out/r03_final/, copied to profiles/ by hand.
O=gpurun_out/r03_final; mkdir -p $O
python3 bench.py --steps 10 --warmup 3 > $O/bench_n1.json 2> $O/bench.err
NB_F32_SYM=0 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none > $O/bench_n1_one_sided_kernel_same_box.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --general-mass --no-cpu-baseline > $O/bench_n1_general_mass.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --dtype f64 --n 524288 --no-cpu-baseline > $O/bench_f64_n512k.json 2>> $O/bench.err
NB_F64_SYM=0 python3 bench.py --steps 5 --warmup 2 --dtype f64 --n 524288 --no-cpu-baseline --traffic none > $O/bench_f64_n512k_one_sided_kernel_same_box.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --dtype f64 --n 524288 --general-mass --no-cpu-baseline > $O/bench_f64_n512k_general_mass.json 2>> $O/bench.err
python3 bench.py --steps 100 --warmup 10 --n 65536 --no-cpu-baseline > $O/bench_n65536_f32.json 2>> $O/bench.err
python3 bench.py --gpus 2 --backend gloo --same-device --steps 5 --warmup 2 > $O/bench_2ranks_gloo_same_device.json 2>> $O/bench.err
python3 bench.py --gpus 4 --backend gloo --same-device --steps 5 --warmup 2 > $O/bench_4ranks_gloo_same_device.json 2>> $O/bench.err
python3 bench.py --gpus 2 --single-process --same-device --steps 5 --warmup 2 > $O/bench_2shards_single_process_same_device.json 2>> $O/bench.err
python3 bench.py --gpus 8 --single-process --same-device --steps 5 --warmup 2 > $O/bench_8shards_single_process_same_device.json 2>> $O/bench.err
for f in $O/bench_*.json; do echo "$f: $(python3 -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); r=j['roofline']; print('%.4e int/s  %.3f ms/step  frac %.3f  kernel %s  clock %s' % (j['value'], j['ms_per_step'], r['frac'], r.get('kernel'), r.get('sustained_clock_mhz')))")"; done

#!/bin/bash
# The round's record runs on one MI355X (from the repo root on the GPU box); outputs under gpurun_out/r04_final/, copied to profiles/ by hand.
# Part 1 (default): the bench lines. Part 2 (`profiles`): rocprofv3 passes of the headline command and of the fp64 one. Part 3 (`multi`): the
# N-rank rehearsals on one GPU.
O=gpurun_out/r04_final; mkdir -p $O
part=${1:-bench}
if [ "$part" = bench ]; then
python3 bench.py --steps 20 --warmup 5 > $O/bench_n1_driver_form.json 2> $O/bench.err
NB_F32_SYM=0 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none > $O/bench_n1_one_sided_kernel_same_box.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --general-mass --no-cpu-baseline > $O/bench_n1_general_mass.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --dtype f64 --n 524288 --no-cpu-baseline > $O/bench_f64_n512k.json 2>> $O/bench.err
NB_F64_SYM=0 python3 bench.py --steps 5 --warmup 2 --dtype f64 --n 524288 --no-cpu-baseline --traffic none > $O/bench_f64_n512k_one_sided_kernel_same_box.json 2>> $O/bench.err
python3 bench.py --steps 5 --warmup 2 --dtype f64 --n 524288 --general-mass --no-cpu-baseline > $O/bench_f64_n512k_general_mass.json 2>> $O/bench.err
python3 bench.py --steps 100 --warmup 10 --n 65536 --no-cpu-baseline > $O/bench_n65536_f32.json 2>> $O/bench.err
python3 bench.py --steps 40 --warmup 5 --n 131072 --no-cpu-baseline --traffic none > $O/bench_n131072_f32.json 2>> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --n 262144 --no-cpu-baseline --traffic none > $O/bench_n262144_f32.json 2>> $O/bench.err
python3 tools/clock_trace.py > $O/clock_trace_n1M_f32.txt 2>> $O/bench.err
fi
if [ "$part" = profiles ]; then
PROF_OUT=$O/prof_n1M tools/profile.sh > $O/prof_n1M_summary.txt 2>&1
PROF_OUT=$O/prof_f64 tools/profile.sh --dtype f64 --n 524288 > $O/prof_f64_summary.txt 2>&1
PROF_OUT=$O/prof_n65536 tools/profile.sh --n 65536 > $O/prof_n65536_summary.txt 2>&1
for d in prof_n1M prof_f64 prof_n65536; do find $O/$d/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${d}_kernel_stats.csv; done
fi
if [ "$part" = multi ]; then
python3 bench.py --gpus 2 --backend gloo --same-device --steps 5 --warmup 2 > $O/bench_2ranks_gloo_same_device.json 2>> $O/bench.err
python3 bench.py --gpus 4 --backend gloo --same-device --steps 5 --warmup 2 > $O/bench_4ranks_gloo_same_device.json 2>> $O/bench.err
python3 bench.py --gpus 2 --single-process --same-device --steps 5 --warmup 2 > $O/bench_2shards_single_process_same_device.json 2>> $O/bench.err
python3 bench.py --gpus 8 --single-process --same-device --steps 5 --warmup 2 > $O/bench_8shards_single_process_same_device.json 2>> $O/bench.err
python3 tools/bench_shard_shapes.py > $O/shard_shapes_emulated_on_one_gpu.txt 2>> $O/bench.err
python3 tools/bench_shard_shapes.py 4194304 f64 >> $O/shard_shapes_emulated_on_one_gpu.txt 2>> $O/bench.err
HSA_ENABLE_IPC_MODE_LEGACY=0 python3 tools/rccl_virtual_ring.py > $O/rccl_virtual_ring_rank0_share_n1M_f32.txt 2>> $O/bench.err
HSA_ENABLE_IPC_MODE_LEGACY=0 python3 tools/rccl_virtual_ring.py 1048576 f32 pairwise >> $O/rccl_virtual_ring_rank0_share_n1M_f32.txt 2>> $O/bench.err
fi
for f in $O/bench_*.json; do echo "$f: $(python3 -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); r=j['roofline']; print('%.4e int/s  %.3f ms/step  frac %.3f  kernel %s  clock %s  cycles/64 %s' % (j['value'], j['ms_per_step'], r['frac'], r.get('kernel'), r.get('sustained_clock_mhz'), r.get('cycles_per_64_interactions_per_simd')))" 2>&1 | tail -1)"; done

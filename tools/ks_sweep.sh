#!/bin/bash
# Sinks per lane (set size) at small and mid-size n with the equal-share block partition of round 4, and the crossover to the one-sided kernels.
# Each line: forced KS (NB_F32_SYM_KS / NB_F64_SYM_KS; 0 = the built-in choice), n, dtype, max rel err, ms, interactions/s. One box.
cd "$(dirname "$0")/.."
T=$(mktemp)
cat > $T <<'PY'
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import bachelor_thesis_n_body_amd as nb
dtype = np.float32 if sys.argv[1] == "f32" else np.float64
for n in [int(v) for v in sys.argv[2:]]:
    s = nb.plummer_sphere(n, 3, dtype)
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        e.set_option("sym_min_n", 0)
        e.accel(0.01, fetch=False)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); e.accel(0.01, fetch=False); ts.append(time.perf_counter() - t0)
        print(f"{n} {np.dtype(dtype).name} sym={int(e.last_accel_symmetric())} {min(ts) * 1e3:.3f} ms  {n * n / min(ts):.4e}", flush=True)
PY
for ks in 0 4 8 12; do echo "F32 KS $ks"; NB_F32_SYM_KS=$ks python $T f32 8192 16384 24576 32768 40000 65536 100001 131072 200000 262144 || exit 1; done
echo "F32 one-sided"; NB_F32_SYM=0 python $T f32 8192 16384 24576 32768 40000 65536 || exit 1
for ks in 0 2 4 8; do echo "F64 KS $ks"; NB_F64_SYM_KS=$ks python $T f64 8192 16384 24576 32768 65536 100001 131072 262144 || exit 1; done
echo "F64 one-sided"; NB_F64_SYM=0 python $T f64 8192 16384 24576 32768 || exit 1
rm -f $T

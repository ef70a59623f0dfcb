"""Time per periodic (Ewald) force evaluation at 64^3 particles, fp32, against the split parameter (alpha L, kmax) — the balance between
the cell-list real-space sum (mesh.hip pair kernels) and the reciprocal sum. usage: ewald_alpha_sweep.py [n_side]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bachelor_thesis_n_body_amd as nb

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, L = n_side ** 3, 500.0
rng = np.random.default_rng(0)
g = (np.arange(n_side) + 0.5) * (L / n_side)
pos = np.mod(np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T + rng.normal(scale=0.3 * L / n_side, size=(n, 3)), L).astype(np.float32)
s = nb.State(pos, np.zeros_like(pos), np.full(n, 1.0, np.float32))
for alpha_L, kmax in ((9.9, 11), (11.5, 13), (13.2, 15), (14.9, 17), (16.5, 19), (18.2, 21), (19.8, 23)):
    with nb.Engine(n, np.float32) as e:
        e.set_periodic(L, alpha_L, kmax)
        e.set_state(s)
        e.accel(0.03 * L / n_side, fetch=False)
        t0 = time.perf_counter()
        for _ in range(5):
            e.accel(0.03 * L / n_side, fetch=False)
        print(json.dumps({"n": n, "alpha_L": alpha_L, "kmax": kmax, "ms_per_force_evaluation": (time.perf_counter() - t0) / 5 * 1e3}), flush=True)

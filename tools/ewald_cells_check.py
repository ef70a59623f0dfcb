"""Accuracy and speed of the cell-list real-space Ewald path (fp32, alpha_L >= 9.9) against the all-pairs fp64 Ewald sum."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bachelor_thesis_n_body_amd as nb

def accel(pos, m, L, dtype, aL, kmax, eps, reps=1):
    s = nb.State(pos.astype(dtype), np.zeros_like(pos, dtype=dtype), m.astype(dtype))
    with nb.Engine(s.n, dtype) as e:
        e.set_periodic(L, aL, kmax); e.set_state(s)
        a = e.accel(eps)
        t0 = time.perf_counter()
        for _ in range(reps): e.accel(eps, fetch=False)
        e.cached_acc()
        return a.astype(np.float64), (time.perf_counter() - t0) / max(reps, 1)

rng = np.random.default_rng(0)
L, n = 100.0, 30000
pos = rng.uniform(0, L, (n, 3)); pos[: n // 3] = np.mod(50 + 3 * rng.normal(size=(n // 3, 3)), L)
m = rng.uniform(0.5, 1.5, n)
ref, _ = accel(pos, m, L, np.float64, 7.5, 13, 0.05)
scale = np.sqrt((ref ** 2).sum(1).mean())
for aL, kmax in ((6.2, 7), (13.2, 15), (16.5, 19), (19.8, 23), (23.1, 27), (26.4, 30)):
    a, _ = accel(pos, m, L, np.float32, aL, kmax, 0.05)
    e = np.linalg.norm(a - ref, axis=1)
    print(json.dumps({"alpha_L": aL, "kmax": kmax, "rms_err_over_rms_force": float(np.sqrt((e ** 2).mean()) / scale), "max_err_over_rms_force": float(e.max() / scale)}), flush=True)
n_side = 64
g = (np.arange(n_side) + 0.25) * (500.0 / n_side)
q = np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T + rng.normal(scale=2.0, size=(n_side ** 3, 3))
q = np.mod(q, 500.0); mm = np.ones(len(q))
for aL, kmax in ((6.2, 7), (13.2, 15), (16.5, 19), (19.8, 23), (23.1, 27), (26.4, 30)):
    _, t = accel(q, mm, 500.0, np.float32, aL, kmax, 0.23, reps=5)
    print(json.dumps({"n": len(q), "alpha_L": aL, "kmax": kmax, "ms_per_force_evaluation": 1e3 * t}), flush=True)

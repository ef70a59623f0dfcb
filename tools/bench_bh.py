"""Barnes-Hut measurement (BASELINE.json configs[4]: theta = 0.5, N = 16M, one MI355X, fp32): wall time of one force
evaluation (bounding box + Morton sort + radix tree + multipoles + walk + scatter), best of 3, state resident in HBM."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bachelor_thesis_n_body_amd as nb

theta = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
sizes = [int(a) for a in sys.argv[2:]] or [1 << 20, 1 << 22, 1 << 24]
for n in sizes:
    s = nb.plummer_sphere(n, 0, np.float32)
    with nb.Engine(n, np.float32) as e:
        e.set_solver("tree", theta)
        e.set_state(s)
        e.accel(0.01, fetch=False)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); e.accel(0.01, fetch=False); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(json.dumps({"workload": f"plummer_sphere N={n} fp32 eps=0.01, Barnes-Hut theta={theta}", "ms_per_force_evaluation": t * 1e3,
                      "particles_per_s": n / t, "direct_sum_equivalent_interactions_per_s": float(n) * n / t}), flush=True)

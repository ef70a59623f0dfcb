"""Energy conservation of the headline configuration: N = 2^20 Plummer sphere, fp32, eps = 0.01, dt = 1/256, KDK steps on one
GPU; prints relative energy error after 0, 25, 50, 100 steps."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bachelor_thesis_n_body_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dtype = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else np.float32
s = nb.plummer_sphere(n, 0, dtype)
out = []
with nb.Engine(n, dtype) as e:
    e.set_state(s)
    k, u = e.energy(0.01); e0 = k + u
    out.append({"step": 0, "K": k, "U": u, "virial_2K_over_U": 2 * k / abs(u)})
    done = 0
    t0 = time.perf_counter()
    for upto in (25, 50, 100):
        e.step(1 / 256, upto - done, 0.01); done = upto
        k, u = e.energy(0.01)
        out.append({"step": upto, "dE_over_E": (k + u - e0) / abs(e0)})
    wall = time.perf_counter() - t0
print(json.dumps({"n": n, "dtype": np.dtype(dtype).name, "wall_s_100_steps_incl_3_energy_evals": wall, "trace": out}))

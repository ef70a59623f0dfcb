"""Per-step energy curve of a KDK run with the potential fused into the force kernel (nb_step_energy): BASELINE.json
configs[3] — N = 2^22 direct fp64, long-run energy-conservation check — on ONE GPU. Prints a progress line per block of steps
and one JSON summary. usage: energy_curve.py [n] [f32|f64] [nsteps] [block]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bachelor_thesis_n_body_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
dtype = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else np.float32
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 64
block = int(sys.argv[4]) if len(sys.argv) > 4 else 4
eps, dt = 0.01, 1.0 / 256.0
s = nb.plummer_sphere(n, 0, dtype)
K, U = [], []
t0 = time.perf_counter()
with nb.Engine(n, dtype) as e:
    e.set_state(s)
    k1, u1 = e.step_with_energy(dt, 1, eps)          # E after step 1 is the reference point (E_0 would need a separate pass)
    K += list(k1); U += list(u1)
    print(f"step 1: K = {k1[0]:.12f} U = {u1[0]:.12f} 2K/|U| = {2*k1[0]/abs(u1[0]):.6f}  ({time.perf_counter()-t0:.1f} s)", flush=True)
    done = 1
    while done < nsteps:
        m = min(block, nsteps - done)
        k, u = e.step_with_energy(dt, m, eps)
        K += list(k); U += list(u); done += m
        E = np.array(K) + np.array(U)
        print(f"step {done}: dE/E = {(E[-1]-E[0])/abs(E[0]):+.3e}  max|dE/E| so far {np.abs(E-E[0]).max()/abs(E[0]):.3e}  ({time.perf_counter()-t0:.1f} s)", flush=True)
wall = time.perf_counter() - t0
E = np.array(K) + np.array(U)
print(json.dumps({"n": n, "dtype": np.dtype(dtype).name, "eps": eps, "dt": dt, "steps": nsteps, "wall_s": wall,
                  "s_per_step_incl_energy": wall / nsteps, "interactions_per_s": float(n) * n * nsteps / wall,
                  "max_abs_dE_over_E": float(np.abs(E - E[0]).max() / abs(E[0])), "final_dE_over_E": float((E[-1] - E[0]) / abs(E[0])),
                  "virial_2K_over_U_first_last": [float(2 * K[0] / abs(U[0])), float(2 * K[-1] / abs(U[-1]))],
                  "E": [float(x) for x in E]}))

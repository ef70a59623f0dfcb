"""Time per force evaluation of the PM / P3M solvers (csrc/mesh.hip) at the sizes of the reference's parameter files:
64^3 (PPPMP3M), 128^3, 256^3 (Different_Sizes/Giant.param: N = 256^3, PM, gridsize cbrt(N)). Particles: a jittered lattice
(early-time) or a strongly clustered set (late-time-like: half the particles in Gaussian blobs).
usage: bench_mesh.py [n_side ...]   env: NB_MESH_REPS"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bachelor_thesis_n_body_amd as nb

sides = [int(a) for a in sys.argv[1:]] or [64, 128, 256]
reps = int(os.environ.get("NB_MESH_REPS", "5"))
only_kind = os.environ.get("NB_MESH_KIND")        # "lattice" | "clustered"
only_solver = os.environ.get("NB_MESH_SOLVER")    # "pm" | "p3m" | "p3m2" (mesh 2 n_side)
for n_side in sides:
    n = n_side ** 3
    L = 500.0 * n_side / 64
    rng = np.random.default_rng(0)
    g = (np.arange(n_side) + 0.25) * (L / n_side)
    q = np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T
    for kind in ("lattice", "clustered"):
        if only_kind and kind != only_kind:
            continue
        pos = q + rng.normal(scale=0.2 * L / n_side, size=q.shape)
        if kind == "clustered":
            k = n // 2
            centres = rng.uniform(0, L, (64, 3))
            pos[:k] = centres[rng.integers(0, 64, k)] + rng.normal(scale=0.01 * L, size=(k, 3))
        pos = np.mod(pos, L).astype(np.float32)
        s = nb.State(pos, np.zeros_like(pos), np.full(n, 1.0, np.float32))
        for solver, ngrid in (("pm", n_side), ("p3m", n_side), ("p3m", 2 * n_side)):
            if only_solver and only_solver != (solver if ngrid == n_side else solver + "2"):
                continue
            with nb.Engine(n, np.float32) as e:
                e.set_mesh(solver, L, ngrid)
                e.set_state(s)
                e.accel(0.03 * L / n_side, fetch=False)
                t0 = time.perf_counter()
                for _ in range(reps):
                    e.accel(0.03 * L / n_side, fetch=False)
                dt = (time.perf_counter() - t0) / reps
            print(json.dumps({"n": n, "particles": kind, "solver": solver, "mesh": ngrid, "ms_per_force_evaluation": 1e3 * dt,
                              "particles_per_s": n / dt}), flush=True)

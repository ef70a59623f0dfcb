"""Pair-symmetric self-interaction kernels (accel_sym_f32.hip / accel_sym_f64.hip) against the C oracle and against the one-sided kernels:
sampled sinks x all sources through the OpenMP C oracle, timing of repeated force evaluations. Run once per mode
(NB_F32_SYM / NB_F64_SYM = 0 | 1: the switches are read once per process).   python tools/sym_check.py f32|f64 [n ...]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bachelor_thesis_n_body_amd as nb

ORA = C.CDLL(os.path.join(ROOT, "oracle", "libnbody_oracle.so"))


def oracle_sample(pos, mass, idx, eps, G=1.0):
    p = np.ascontiguousarray(pos.T, np.float64); m = np.ascontiguousarray(mass, np.float64)
    q = np.ascontiguousarray(pos[idx].T, np.float64)
    n, k = p.shape[1], len(idx)
    a = np.empty((3, k))
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    ORA.nbo_accel_f64_omp(C.c_int64(k), vp(q[0]), vp(q[1]), vp(q[2]), C.c_int64(n), vp(p[0]), vp(p[1]), vp(p[2]), vp(m),
                          C.c_double(eps), C.c_double(G), vp(a[0]), vp(a[1]), vp(a[2]))
    return a.T


def run(n, dtype, general, eps=0.01, reps=3):
    s = nb.plummer_sphere(n, 3, dtype)
    mass = s.mass.copy()
    if general:
        rng = np.random.default_rng(5)
        mass = (mass * rng.uniform(0.5, 1.5, n)).astype(dtype)
    with nb.Engine(n, dtype) as e:
        e.set_state(nb.State(s.pos, s.vel, mass))
        a = e.accel(eps)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); e.accel(eps, fetch=False); ts.append(time.perf_counter() - t0)
        a2 = e.accel(eps)
    rng = np.random.default_rng(9)
    idx = np.unique(np.concatenate([rng.integers(0, n, 192), [0, 1, n // 2, n - 2, n - 1]]))
    ref = oracle_sample(s.pos.astype(np.float64), mass.astype(np.float64), idx, eps)
    err = np.linalg.norm(a[idx].astype(np.float64) - ref, axis=1) / np.linalg.norm(ref, axis=1)
    mom = np.abs((mass[:, None].astype(np.float64) * a.astype(np.float64)).sum(0)).max() / np.abs(mass[:, None] * a).sum(0).max()
    print(f"n={n:8d} {np.dtype(dtype).name} {'general' if general else 'uniform'}: max rel err {err.max():.3e} median {np.median(err):.3e}  "
          f"|sum m a|/sum|m a| {mom:.2e}  finite {bool(np.isfinite(a).all())}  repeat-identical {bool((a == a2).all())}  "
          f"{min(ts) * 1e3:9.3f} ms  {n * n / min(ts):.4e} int/s", flush=True)


if __name__ == "__main__":
    dt = np.float32 if sys.argv[1] == "f32" else np.float64
    ns = [int(v) for v in sys.argv[2:]] or [40000, 65536, 100001, 262144, 1048576]
    print("NB_F32_SYM =", os.environ.get("NB_F32_SYM"), " NB_F64_SYM =", os.environ.get("NB_F64_SYM"))
    for n in ns:
        for general in (False, True):
            run(n, dt, general)

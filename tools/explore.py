"""GPU exploration driver (not part of the product): parity of the HIP kernels against the C oracle at small N and
throughput of kernel variants selected through NB_F32_KS / NB_F64_KS / NB_F64_RSQ / NB_*_SPLIT."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bachelor_thesis_n_body_amd as nb

ORA = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libnbody_oracle.so"))

def oracle_acc(pos, mass, eps, G=1.0):
    p = np.ascontiguousarray(pos.T, np.float64); m = np.ascontiguousarray(mass, np.float64)
    n = p.shape[1]; a = np.empty((3, n))
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    ORA.nbo_accel_f64_omp(C.c_int64(n), vp(p[0]), vp(p[1]), vp(p[2]), C.c_int64(n), vp(p[0]), vp(p[1]), vp(p[2]), vp(m),
                          C.c_double(eps), C.c_double(G), vp(a[0]), vp(a[1]), vp(a[2]))
    return a.T

def relerr(a, ref):
    return np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)

def parity(n, dtype, eps=0.01):
    s = nb.plummer_sphere(n, 1, dtype)
    ref = oracle_acc(s.pos.astype(np.float64), s.mass.astype(np.float64), eps)
    a = nb.compute_forces(s.pos, s.mass, eps)
    e = relerr(a.astype(np.float64), ref)
    print(f"parity n={n} {np.dtype(dtype).name}: max rel {e.max():.3e}  median {np.median(e):.3e}", flush=True)

def rate(n, dtype, reps=3, eps=0.01, label=""):
    s = nb.plummer_sphere(n, 0, dtype)
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        e.accel(eps, fetch=False)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); e.accel(eps, fetch=False); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"rate n={n} {np.dtype(dtype).name} {label}: {t*1e3:9.3f} ms  {n*n/t:.4e} int/s", flush=True)
    return n * n / t

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "parity"):
        for n in (1000, 4096, 30000):
            parity(n, np.float32); parity(n, np.float64)
    if what in ("all", "f32"):
        for ks in (2, 4, 8):
            os.environ["NB_F32_KS"] = str(ks)
            for n in (65536, 1048576):
                rate(n, np.float32, label=f"KS={ks}")
        os.environ.pop("NB_F32_KS")
    if what == "tune":
        for n in (16384, 65536, 262144):
            for ks in (2, 4, 8):
                for split in (0, 4, 8, 16, 32, 64):
                    os.environ["NB_F32_KS"] = str(ks); os.environ["NB_F32_SPLIT"] = str(split)
                    rate(n, np.float32, reps=5, label=f"KS={ks} SPLIT={split}")
    if what in ("all", "f64"):
        for mode in (0, 1):
            for ks in (1, 2, 4):
                os.environ["NB_F64_KS"] = str(ks); os.environ["NB_F64_RSQ"] = str(mode)
                for n in (65536, 262144):
                    rate(n, np.float64, label=f"KS={ks} RSQ={mode}")

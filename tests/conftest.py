import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # a fresh clone has no built artefacts (*.so are git-ignored): build them once, as __graft_entry__.build() does
    lib = ROOT / "bachelor_thesis_n_body_amd" / "libnbody_amd.so"
    if not lib.exists() and Path("/opt/rocm/bin/hipcc").exists():
        subprocess.check_call(["make", "-C", str(ROOT / "bachelor_thesis_n_body_amd" / "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    for so in ("libnbody_oracle.so", "libbh_oracle.so"):
        if not (ROOT / "oracle" / so).exists():
            subprocess.check_call(["make", "-C", str(ROOT / "oracle")], stdout=subprocess.DEVNULL)
            break


def _gpu_available() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


@pytest.fixture(scope="session")
def has_gpu():
    return _gpu_available()


def pytest_collection_modifyitems(config, items):
    # GPU tests skip (not fail) on a box without a device; CPU tests run everywhere
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


class COracle:
    """ctypes view of oracle/libnbody_oracle.so (the scalar-loop C restatement) — checker only."""

    def __init__(self):
        so = ROOT / "oracle" / "libnbody_oracle.so"
        if not so.exists():
            subprocess.check_call(["make", "-C", str(ROOT / "oracle")], stdout=subprocess.DEVNULL)
        self.lib = C.CDLL(str(so))
        self.lib.nbo_potential_f64.restype = C.c_double
        self.lib.nbo_kinetic_f64.restype = C.c_double

    @staticmethod
    def _vp(a):
        return a.ctypes.data_as(C.c_void_p)

    def accel(self, pos, mass, eps, G=1.0, sinks=None, sequential=False):
        p = np.ascontiguousarray(np.asarray(pos, np.float64).T)
        m = np.ascontiguousarray(mass, np.float64)
        s = p if sinks is None else np.ascontiguousarray(np.asarray(sinks, np.float64).T)
        n_src, n_snk = p.shape[1], s.shape[1]
        a = np.zeros((3, n_snk))
        fn = self.lib.nbo_accel_f64 if sequential else self.lib.nbo_accel_f64_omp
        v = self._vp
        fn(C.c_int64(n_snk), v(s[0]), v(s[1]), v(s[2]), C.c_int64(n_src), v(p[0]), v(p[1]), v(p[2]), v(m),
           C.c_double(eps), C.c_double(G), v(a[0]), v(a[1]), v(a[2]))
        return np.ascontiguousarray(a.T)

    def potential(self, pos, mass, eps, G=1.0):
        p = np.ascontiguousarray(np.asarray(pos, np.float64).T)
        m = np.ascontiguousarray(mass, np.float64)
        v = self._vp
        return float(self.lib.nbo_potential_f64(C.c_int64(p.shape[1]), v(p[0]), v(p[1]), v(p[2]), v(m),
                                                C.c_double(eps), C.c_double(G)))

    def phi(self, pos, mass, eps, sinks):
        """phi_i = sum_j m_j / sqrt(r_ij^2 + eps^2) of `sinks` over all sources (a sink's own m_i / eps included when it is a source)."""
        p = np.ascontiguousarray(np.asarray(pos, np.float64).T)
        m = np.ascontiguousarray(mass, np.float64)
        s = np.ascontiguousarray(np.asarray(sinks, np.float64).T)
        out = np.zeros(s.shape[1])
        v = self._vp
        self.lib.nbo_phi_f64_omp(C.c_int64(s.shape[1]), v(s[0]), v(s[1]), v(s[2]), C.c_int64(p.shape[1]), v(p[0]), v(p[1]), v(p[2]), v(m),
                                 C.c_double(eps), v(out))
        return out

    def kinetic(self, vel, mass):
        q = np.ascontiguousarray(np.asarray(vel, np.float64).T)
        m = np.ascontiguousarray(mass, np.float64)
        v = self._vp
        return float(self.lib.nbo_kinetic_f64(C.c_int64(q.shape[1]), v(q[0]), v(q[1]), v(q[2]), v(m)))


@pytest.fixture(scope="session")
def coracle():
    return COracle()


def rel_err(a, ref):
    """per-particle |a - ref| / |ref|"""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)


def note(label, value, tol=None):
    """Measured-error log for calibrating tolerances (NB_TEST_ERRLOG=<file>): label, measured value, tolerance in force."""
    path = os.environ.get("NB_TEST_ERRLOG")
    if path:
        with open(path, "a") as f:
            f.write(f"{label}\t{value:.3e}\t{'' if tol is None else format(tol, '.1e')}\n")

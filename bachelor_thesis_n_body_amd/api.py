"""Host-side surface of the direct-summation engine: compute_forces() / step() / energy().

These are the in-process functions SURVEY.md §8(a) rows ⊕1-⊕3 define for the path that the reference only
reaches by spawning CONCEPT with `gravity: 'pp'` (SnapVsPK/Ucertain.py:89-97, Different_Sizes/Giant.param:66-68).
Signatures and semantics are the same as the CPU oracle's (oracle/nbody_oracle.py) so parity tests call both
with identical arguments — but everything here runs on the MI355X through the C-ABI (include/nbody_amd.h);
there is no CPU path in this package.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib
from .state import State

_DT = {np.dtype(np.float32): _lib.NB_F32, np.dtype(np.float64): _lib.NB_F64}


def _code(dtype) -> int:
    try:
        return _DT[np.dtype(dtype)]
    except KeyError:
        raise TypeError(f"dtype must be float32 or float64, got {dtype}") from None


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    """One nb_ctx: `n` particles resident on one MI355X, state kept as SoA in HBM between calls.
    `devices=[d0, d1, ...]` spreads the sinks over several GPUs of the node inside ONE process (nb_create_multi: sink decomposition,
    peer pulls of the source blocks over xGMI, see include/nbody_amd.h); the same calls then take and return GLOBAL arrays. The
    one-process-per-GPU path over RCCL is `ShardedEngine` (dist.py)."""

    def __init__(self, n: int, dtype=np.float32, device: int = 0, devices=None):
        self._lib = _lib.load()
        self.n, self.dtype, self.device = int(n), np.dtype(dtype), int(device)
        self._ctx = C.c_void_p()
        if devices is None:
            self.devices = [self.device]
            rc = self._lib.nb_create(C.byref(self._ctx), self.n, _code(self.dtype), self.device)
            what = "nb_create"
        else:
            self.devices = [int(d) for d in devices]
            ids = (C.c_int * len(self.devices))(*self.devices)
            rc = self._lib.nb_create_multi(C.byref(self._ctx), self.n, _code(self.dtype), len(self.devices), ids)
            what = "nb_create_multi"
        if rc != _lib.NB_OK:
            self._ctx = C.c_void_p()
            _lib.check(None, rc, what)

    def layout(self):
        """(counts, offsets) of the contiguous sink ranges owned by the context's devices."""
        cap = 64
        cnt, off = (C.c_int64 * cap)(), (C.c_int64 * cap)()
        p = self._lib.nb_multi_layout(self._ctx, cnt, off, cap)
        return list(cnt[:p]), list(off[:p])

    # -- lifetime -------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.nb_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self) -> C.c_void_p:
        return self._ctx

    def kdk_run(self, kick1, drift, kick2, eps: float, G: float = 1.0) -> None:
        """len(kick1) generalised KDK steps enqueued back to back (one host synchronisation for the whole run)."""
        k1, d, k2 = (np.ascontiguousarray(a, np.float64) for a in (kick1, drift, kick2))
        if not (k1.shape == d.shape == k2.shape and k1.ndim == 1):
            raise ValueError("kick1, drift, kick2 must be 1-D arrays of equal length")
        _lib.check(self._ctx, self._lib.nb_kdk_run(self._ctx, k1.size, k1.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                                   k2.ctypes.data_as(C.c_void_p), eps, G), "nb_kdk_run")

    def rung_step(self, levels: int, kc, dc, dt_base: float, crit: float, eps: float, G: float = 1.0):
        """One base step with a rung hierarchy for the short-range force (nb_rung_step; P3M contexts). Returns the rung populations."""
        kc, dc = np.ascontiguousarray(kc, np.float64), np.ascontiguousarray(dc, np.float64)
        need = 2 * (1 << (int(levels) - 1)) + 1
        if kc.shape != (need,) or dc.shape != (need,):
            raise ValueError(f"kc and dc must have {need} entries for {levels} levels")
        pop = (C.c_int64 * 8)()
        _lib.check(self._ctx, self._lib.nb_rung_step(self._ctx, int(levels), _ptr(kc), _ptr(dc), float(dt_base), float(crit), float(eps), float(G), pop),
                   "nb_rung_step")
        return [int(v) for v in pop[: int(levels)]]

    def last_run_used_graph(self) -> bool:
        """True if the last kdk_run() replayed a captured hipGraph of one step (small-N, launch-bound runs)."""
        return bool(self._lib.nb_last_run_used_graph(self._ctx))

    def set_option(self, name: str, value: int) -> None:
        """Options of the pair-symmetric all-pairs kernels (nb_set_option): 'pair_symmetry' 0/1, 'sym_min_n', 'sym_slab_bytes'; value < 0
        restores the default."""
        _lib.check(self._ctx, self._lib.nb_set_option(self._ctx, _lib.OPTIONS[name], int(value)), "nb_set_option")

    def last_accel_symmetric(self) -> bool:
        """True if the last direct-summation force evaluation ran the pair-symmetric kernel (every unordered pair once); a multi-device
        engine answers for the own-chunk launch of its first device."""
        return bool(self._lib.nb_last_accel_symmetric(self._ctx))

    def get_option(self, name: str) -> int:
        """nb_get_option: an option as set (-1 = default in force) or a read-only query — 'multi_schedule' (1 = this multi-device engine
        runs the pair-symmetric schedule), 'own_chunk_symmetric'."""
        v = C.c_int64()
        code = _lib.OPTIONS.get(name, _lib.QUERIES.get(name))
        _lib.check(self._ctx, self._lib.nb_get_option(self._ctx, code, C.byref(v)), "nb_get_option")
        return int(v.value)

    def set_stream(self, hip_stream: int) -> None:
        _lib.check(self._ctx, self._lib.nb_set_stream(self._ctx, C.c_void_p(hip_stream)), "nb_set_stream")

    def set_solver(self, method: str = "direct", theta: float = 0.5) -> None:
        """'direct' = O(N^2) summation (default); 'tree' = Barnes-Hut with opening angle theta (fp32 engines only)."""
        code = {"direct": _lib.NB_SOLVER_DIRECT, "tree": _lib.NB_SOLVER_TREE}[method]
        _lib.check(self._ctx, self._lib.nb_set_solver(self._ctx, code, float(theta)), "nb_set_solver")

    def set_periodic(self, boxsize: float, alpha_L: float = 6.2, kmax: int = 7) -> None:
        """Cubic periodic box (Ewald summation, uniform background subtracted); boxsize <= 0 switches it off."""
        _lib.check(self._ctx, self._lib.nb_set_periodic(self._ctx, float(boxsize), float(alpha_L), int(kmax)), "nb_set_periodic")

    def set_mesh(self, solver: str, boxsize: float, ngrid: int, scale_cells: float = 1.25, range_scales: float = 4.5,
                 interlace: bool = True, part: str = "both") -> None:
        """Periodic box solved with 'pm' or 'p3m' on an ngrid^3 mesh (CONCEPT's `'gravity': 'pm' | 'p3m'`; defaults are its
        documented short-range parameters). part = 'long' / 'short' evaluates one half of the P3M split only (diagnostics)."""
        code = {"pm": _lib.NB_SOLVER_PM, "p3m": _lib.NB_SOLVER_P3M}[solver]
        flags = (0 if interlace else _lib.NB_MESH_NO_INTERLACE) | {"both": 0, "long": _lib.NB_MESH_LONG_ONLY,
                                                                     "short": _lib.NB_MESH_SHORT_ONLY}[part]
        _lib.check(self._ctx, self._lib.nb_set_mesh(self._ctx, code, float(boxsize), int(ngrid), float(scale_cells),
                                                    float(range_scales), flags), "nb_set_mesh")

    def kdk_step(self, kick1: float, drift: float, kick2: float, eps: float, G: float = 1.0) -> None:
        """v += a*kick1; x += v*drift; a = a(x); v += a*kick2 (comoving leapfrog with kick = int dt/a, drift = int dt/a^2)."""
        _lib.check(self._ctx, self._lib.nb_kdk_step(self._ctx, kick1, drift, kick2, eps, G), "nb_kdk_step")

    def synchronize(self) -> None:
        _lib.check(self._ctx, self._lib.nb_synchronize(self._ctx), "nb_synchronize")

    # -- state ----------------------------------------------------------------------------------
    def _soa(self, a: np.ndarray, name: str):
        a = np.asarray(a)
        if a.shape != (self.n, 3):
            raise ValueError(f"{name} must have shape ({self.n}, 3), got {a.shape}")
        t = np.ascontiguousarray(a.T, dtype=self.dtype)  # (3, N): x, y, z rows contiguous
        return t

    def set_state(self, state: State) -> None:
        p = self._soa(state.pos, "pos")
        v = self._soa(state.vel, "vel")
        m = np.ascontiguousarray(state.mass, dtype=self.dtype)
        if m.shape != (self.n,):
            raise ValueError(f"mass must have shape ({self.n},), got {m.shape}")
        rc = self._lib.nb_set_state(self._ctx, _ptr(p[0]), _ptr(p[1]), _ptr(p[2]), _ptr(v[0]), _ptr(v[1]), _ptr(v[2]), _ptr(m))
        _lib.check(self._ctx, rc, "nb_set_state")

    def get_state(self) -> State:
        p = np.empty((3, self.n), self.dtype)
        v = np.empty((3, self.n), self.dtype)
        m = np.empty(self.n, self.dtype)
        rc = self._lib.nb_get_state(self._ctx, _ptr(p[0]), _ptr(p[1]), _ptr(p[2]), _ptr(v[0]), _ptr(v[1]), _ptr(v[2]), _ptr(m))
        _lib.check(self._ctx, rc, "nb_get_state")
        return State(np.ascontiguousarray(p.T), np.ascontiguousarray(v.T), m)

    # -- compute --------------------------------------------------------------------------------
    def accel(self, eps: float, G: float = 1.0, fetch: bool = True) -> Optional[np.ndarray]:
        """a(x) of the resident particles; (N,3) array, or None with fetch=False (device-side refresh only)."""
        if not fetch:
            _lib.check(self._ctx, self._lib.nb_accel(self._ctx, eps, G, None, None, None), "nb_accel")
            return None
        a = np.empty((3, self.n), self.dtype)
        _lib.check(self._ctx, self._lib.nb_accel(self._ctx, eps, G, _ptr(a[0]), _ptr(a[1]), _ptr(a[2])), "nb_accel")
        return np.ascontiguousarray(a.T)

    def cached_acc(self) -> np.ndarray:
        """The acceleration left on the device by the last accel()/step() — no recomputation."""
        a = np.empty((3, self.n), self.dtype)
        _lib.check(self._ctx, self._lib.nb_get_acc(self._ctx, _ptr(a[0]), _ptr(a[1]), _ptr(a[2])), "nb_get_acc")
        return np.ascontiguousarray(a.T)

    def step(self, dt: float, nsteps: int = 1, eps: float = 0.0, G: float = 1.0) -> None:
        _lib.check(self._ctx, self._lib.nb_step(self._ctx, dt, int(nsteps), eps, G), "nb_step")

    def step_with_energy(self, dt: float, nsteps: int = 1, eps: float = 0.0, G: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
        """`nsteps` KDK steps; returns (kinetic[nsteps], potential[nsteps]) after every step. The potential is a by-product
        of the force kernel (fused accumulator), not a second O(N^2) pass."""
        k = np.empty(int(nsteps), np.float64); u = np.empty(int(nsteps), np.float64)
        _lib.check(self._ctx, self._lib.nb_step_energy(self._ctx, dt, int(nsteps), eps, G, k.ctypes.data_as(C.c_void_p),
                                                       u.ctypes.data_as(C.c_void_p)), "nb_step_energy")
        return k, u

    def energy(self, eps: float, G: float = 1.0) -> Tuple[float, float]:
        k, u = C.c_double(), C.c_double()
        _lib.check(self._ctx, self._lib.nb_energy(self._ctx, eps, G, C.byref(k), C.byref(u)), "nb_energy")
        return k.value, u.value


def _infer_dtype(a, dtype):
    if dtype is not None:
        return np.dtype(dtype)
    d = np.asarray(a).dtype
    return d if d in _DT else np.dtype(np.float64)


def compute_forces(pos, mass, eps: float, G: float = 1.0, dtype=None, device: int = 0, method: str = "direct",
                   theta: float = 0.5) -> np.ndarray:
    """Softened all-pairs acceleration a_i = G sum_j m_j (r_j-r_i)/(r_ij^2+eps^2)^{3/2} on the GPU.
    `dtype` defaults to the dtype of `pos` (float32 -> fp32 kernel, anything else -> fp64 kernel).
    method="tree" evaluates the sum with the Barnes-Hut tree (opening angle `theta`, fp32 only)."""
    pos = np.asarray(pos)
    dt = _infer_dtype(pos, dtype)
    n = pos.shape[0]
    if n == 0:
        return np.zeros((0, 3), dt)
    with Engine(n, dt, device) as e:
        if method != "direct":
            e.set_solver(method, theta)
        e.set_state(State(pos, np.zeros_like(pos), np.asarray(mass)))
        return e.accel(eps, G)


def step(state: State, dt: float, nsteps: int = 1, eps: float = 0.0, G: float = 1.0, dtype=None,
         device: int = 0) -> State:
    """`nsteps` kick-drift-kick steps on the GPU, written back into `state` (pos, vel, acc); returns it."""
    d = _infer_dtype(state.pos, dtype)
    with Engine(state.n, d, device) as e:
        e.set_state(state)
        e.step(dt, nsteps, eps, G)
        out = e.get_state()
        acc = e.cached_acc() if nsteps > 0 else None
    state.pos, state.vel = out.pos.astype(state.pos.dtype, copy=False), out.vel.astype(state.vel.dtype, copy=False)
    state.acc = acc
    return state


def energy(state: State, eps: float, G: float = 1.0, dtype=None, device: int = 0) -> Tuple[float, float]:
    d = _infer_dtype(state.pos, dtype)
    with Engine(state.n, d, device) as e:
        e.set_state(state)
        return e.energy(eps, G)


class ClockProbe:
    """Sustained shader clock of one GPU over a time window (nb_clock_probe_begin / _end): start it, launch the kernels of interest,
    `probe.stop(stream)`, then `mhz, window_ms = probe.end()`. A measurement aid for bench.py's roofline line; no counterpart in the reference."""

    def __init__(self, device: int, window_ms: float):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        _lib.check(None, self._lib.nb_clock_probe_begin(int(device), float(window_ms), C.byref(self._h)), "nb_clock_probe_begin")

    def stop(self, stream: int = 0) -> None:
        """End the window when the work enqueued so far on `stream` (a hipStream_t handle, e.g. torch.cuda.current_stream().cuda_stream)
        has finished — without this the window lasts the full `window_ms`, and a device-wide synchronisation waits for it."""
        _lib.check(None, self._lib.nb_clock_probe_stop(self._h, C.c_void_p(int(stream))), "nb_clock_probe_stop")

    def end(self) -> Tuple[float, float]:
        if not self._h:
            raise RuntimeError("ClockProbe.end() called twice")
        mhz, win = C.c_double(), C.c_double()
        h, self._h = self._h, C.c_void_p()
        _lib.check(None, self._lib.nb_clock_probe_end(h, C.byref(mhz), C.byref(win)), "nb_clock_probe_end")
        return mhz.value, win.value

    def end_xcd(self) -> Tuple[float, float, list]:
        """(mean MHz, window ms, [MHz of XCD 0, 1, ...]): the probe runs one wave per XCD (nb_clock_probe_end_xcd)."""
        if not self._h:
            raise RuntimeError("ClockProbe.end() called twice")
        mhz, win = C.c_double(), C.c_double()
        per = (C.c_double * 8)()
        h, self._h = self._h, C.c_void_p()
        n = self._lib.nb_clock_probe_end_xcd(h, C.byref(mhz), C.byref(win), per, 8)
        if n < 0:
            _lib.check(None, n, "nb_clock_probe_end_xcd")
        return mhz.value, win.value, [float(per[i]) for i in range(n)]

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                self.end()
            except Exception:
                pass

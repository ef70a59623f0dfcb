"""Host-side mirror of the reference's field-analysis functions, running on the GPU through include/nbody_field.h.

Same names, argument meaning and return values as the reference:
    cic_density(positions, boxsize, N)              TheFeedingPipeline/FraP_kTilDelta.py:93-123
    compute_delta(density)                          TheFeedingPipeline/FraP_kTilDelta.py:126-127
    deconvolve_cic(delta_grid, boxsize, ngrid)      TheFeedingPipeline/FraP_kTilDelta.py:87-91
    delta_field(positions, boxsize, ngrid, ...)     the per-snapshot body of Feeder_pipe.py:174-195 (one device pass)
    compute_power_spectrum_from_snapshot(...)       SnapVsPK/Power_compare.py:135-200
The small 1-D tables (window along one axis, k values, histogram edges) are computed here with exactly the
reference's NumPy expressions and handed to the device; everything O(N_particles) / O(ngrid^3) runs in HIP kernels
and hipFFT. No CPU fallback: without the library or a GPU these raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _lib
from . import snapshot as _snap

_vp, _i64, _dbl, _int = C.c_void_p, C.c_int64, C.c_double, C.c_int
FIELD_SIGNATURES = {
    "nbf_create": (_int, [C.POINTER(_vp), _int, _dbl, _int]),
    "nbf_destroy": (_int, [_vp]),
    "nbf_last_error": (C.c_char_p, [_vp]),
    "nbf_set_window": (_int, [_vp, _vp]),
    "nbf_set_kbins": (_int, [_vp, _vp, _vp, _int]),
    "nbf_cic": (_int, [_vp, _vp, _i64]),
    "nbf_grid_mean": (_int, [_vp, C.POINTER(_dbl)]),
    "nbf_grid_affine": (_int, [_vp, _dbl, _dbl, _dbl]),
    "nbf_deconvolve": (_int, [_vp]),
    "nbf_deconvolve_clip": (_int, [_vp, _dbl]),
    "nbf_grid_round_f32": (_int, [_vp]),
    "nbf_get_grid": (_int, [_vp, _vp]),
    "nbf_set_grid": (_int, [_vp, _vp]),
    "nbf_power_bins": (_int, [_vp, _dbl, _vp, _vp]),
    "nbf_power_modes": (_int, [_vp, _vp, _i64, _int, _vp]),
    "nbf_lpt2_gradient": (_int, [_vp, _vp, _vp]),
}
_bound = False


def _load():
    global _bound
    lib = _lib.load()
    if not _bound:
        for name, (res, args) in FIELD_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _bound = True
    return lib


def _ptr(a):
    return a.ctypes.data_as(_vp)


def window_1d(ngrid: int, boxsize: float) -> np.ndarray:
    """sinc(k/(2 k_Nyq)) along one axis — the expressions of FraP_kTilDelta.py:74-80, kept verbatim."""
    d = boxsize / ngrid
    k_vals = 2 * np.pi * np.fft.fftfreq(ngrid, d=d)
    k_quist = np.pi * (d) ** (-1)
    return np.ascontiguousarray(np.sinc(k_vals / (2 * k_quist)))


def kbin_edges(L: float, Nmesh: int, nbins: int = 300) -> Tuple[np.ndarray, np.ndarray]:
    """k values along one axis and the log-spaced edges of Power_compare.py:179-191 (k_max = the corner mode)."""
    k_vals = 2 * np.pi * np.fft.fftfreq(Nmesh, d=L / Nmesh)
    a = np.abs(k_vals).max()
    k_max = np.sqrt(a ** 2 + a ** 2 + a ** 2)
    k_min = 2 * np.pi / L
    return np.ascontiguousarray(k_vals), np.ascontiguousarray(np.logspace(np.log10(k_min), np.log10(k_max), nbins))


class FieldPlan:
    """One nbf_plan: an ngrid^3 real + complex grid in HBM and a hipFFT plan, reusable across snapshots."""

    def __init__(self, ngrid: int, boxsize: float, device: int = 0):
        self._lib = _load()
        self.ngrid, self.boxsize = int(ngrid), float(boxsize)
        self._p = _vp()
        rc = self._lib.nbf_create(C.byref(self._p), self.ngrid, self.boxsize, int(device))
        if rc != 0:
            self._p = _vp()
            raise RuntimeError(f"nbf_create failed (code {rc}): {self._lib.nbf_last_error(None).decode()}")
        self._chk(self._lib.nbf_set_window(self._p, _ptr(window_1d(self.ngrid, self.boxsize))), "nbf_set_window")
        self._nedges = 0

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed (code {rc}): {self._lib.nbf_last_error(self._p).decode()}")

    def close(self):
        if getattr(self, "_p", None) is not None and self._p:
            self._lib.nbf_destroy(self._p)
            self._p = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- building blocks (device-resident grid) --------------------------------------------------------
    def cic(self, positions) -> None:
        p = np.ascontiguousarray(positions, np.float64)
        if p.ndim != 2 or p.shape[1] != 3:
            raise ValueError("positions must have shape (N, 3)")
        self._chk(self._lib.nbf_cic(self._p, _ptr(p), p.shape[0]), "nbf_cic")

    def mean(self) -> float:
        m = _dbl()
        self._chk(self._lib.nbf_grid_mean(self._p, C.byref(m)), "nbf_grid_mean")
        return m.value

    def affine(self, div1: float, div2: float, sub: float) -> None:
        self._chk(self._lib.nbf_grid_affine(self._p, div1, div2, sub), "nbf_grid_affine")

    def deconvolve(self, clip: float = 0.0) -> None:
        self._chk(self._lib.nbf_deconvolve_clip(self._p, float(clip)), "nbf_deconvolve_clip")

    def round_f32(self) -> None:
        """grid <- float32(grid): the density of the scripts that deposit into a float32 array (Different_Sizes/Plotter.py:167)."""
        self._chk(self._lib.nbf_grid_round_f32(self._p), "nbf_grid_round_f32")

    def get(self) -> np.ndarray:
        out = np.empty((self.ngrid,) * 3, np.float64)
        self._chk(self._lib.nbf_get_grid(self._p, _ptr(out)), "nbf_get_grid")
        return out

    def set(self, grid) -> None:
        g = np.ascontiguousarray(grid, np.float64)
        if g.shape != (self.ngrid,) * 3:
            raise ValueError(f"grid must have shape {(self.ngrid,) * 3}")
        self._chk(self._lib.nbf_set_grid(self._p, _ptr(g)), "nbf_set_grid")

    def power_bins(self, scale: float, nbins: int = 300):
        k_vals, edges = kbin_edges(self.boxsize, self.ngrid, nbins)
        self._chk(self._lib.nbf_set_kbins(self._p, _ptr(k_vals), _ptr(edges), len(edges)), "nbf_set_kbins")
        psum = np.empty(len(edges) - 1, np.float64)
        counts = np.empty(len(edges) - 1, np.int64)
        self._chk(self._lib.nbf_power_bins(self._p, scale, _ptr(psum), _ptr(counts)), "nbf_power_bins")
        return edges, psum, counts

    def power_modes(self, positions, interlace: bool = True) -> np.ndarray:
        """|delta_k / W|^2 L^3 / N^6 of the CIC-deposited overdensity for every mode, (N, N, N) in np.fft order; interlace: two lattices half a
        cell apart combined in k-space (CONCEPT's estimator settings, NN.param:30-41). Deposit, FFT and deconvolution on the GPU."""
        p = np.ascontiguousarray(positions, np.float64)
        if p.ndim != 2 or p.shape[1] != 3:
            raise ValueError("positions must have shape (N, 3)")
        out = np.empty((self.ngrid,) * 3, np.float64)
        self._chk(self._lib.nbf_power_modes(self._p, _ptr(p), p.shape[0], 1 if interlace else 0, _ptr(out)), "nbf_power_modes")
        return out

    def lpt2_gradient(self, psi) -> np.ndarray:
        """grad(phi2) of 2LPT from the first-order displacement psi (3, N, N, N): 13 spectral transforms on the GPU (nbf_lpt2_gradient)."""
        a = np.ascontiguousarray(psi, np.float64)
        if a.shape != (3,) + (self.ngrid,) * 3:
            raise ValueError(f"psi must have shape {(3,) + (self.ngrid,) * 3}")
        out = np.empty_like(a)
        self._chk(self._lib.nbf_lpt2_gradient(self._p, _ptr(a), _ptr(out)), "nbf_lpt2_gradient")
        return out

    # -- the reference's per-snapshot chain (Feeder_pipe.py:174-195), one device pass --------------------
    def delta_field(self, positions, subtract_mean: bool = False) -> np.ndarray:
        self.cic(positions)
        self.affine(self.mean(), 1.0, 1.0)            # compute_delta: density / density.mean() - 1.0
        self.deconvolve()
        if subtract_mean:
            self.affine(1.0, 1.0, self.mean())        # delta -= delta.mean()
        return self.get()


def cic_density(positions, boxsize: float, N: int, device: int = 0) -> np.ndarray:
    with FieldPlan(N, boxsize, device) as p:
        p.cic(positions)
        return p.get()


def compute_delta(density) -> np.ndarray:
    d = np.asarray(density, np.float64)
    with FieldPlan(d.shape[0], 1.0) as p:
        p.set(d)
        p.affine(p.mean(), 1.0, 1.0)
        return p.get()


def deconvolve_cic(delta_grid, boxsize: float, ngrid: int, device: int = 0) -> np.ndarray:
    with FieldPlan(ngrid, boxsize, device) as p:
        p.set(delta_grid)
        p.deconvolve()
        return p.get()


def delta_field(positions, boxsize: float, ngrid: int, subtract_mean: bool = False, device: int = 0) -> np.ndarray:
    with FieldPlan(ngrid, boxsize, device) as p:
        return p.delta_field(positions, subtract_mean)


def power_spectrum(positions, L: float, Nmesh: int, deconvolve: bool = False, use_fac: bool = False, nbins: int = 300,
                   device: int = 0):
    """(k_centers[valid], Pk_binned[valid], counts[valid]) — the body of Power_compare.py:148-200 after the h5py read."""
    pos = np.ascontiguousarray(positions, np.float64)
    with FieldPlan(Nmesh, L, device) as p:
        p.cic(pos)
        if not use_fac:
            cell_vol = (L / Nmesh) ** 3
            rho_mean = pos.shape[0] / (L ** 3)
            p.affine(cell_vol, rho_mean, 1.0)        # (delta / cell_vol) / rho_mean - 1.0
        else:
            p.affine(1.0, 1.0, 1.0)                  # delta -= 1
        if deconvolve:
            p.deconvolve()                           # Power_compare.py:93-97 (norm='ortho' twice == plain fft/ifft)
        V = L ** 3
        if use_fac:
            fac = Nmesh ** 3 * np.sqrt(2 * np.pi / V)
            scale = 1.0 / (fac ** 2)
        else:
            scale = V / (Nmesh ** 6)
        kbins, psum, counts = p.power_bins(scale, nbins)
    k_centers = 0.5 * (kbins[:-1] + kbins[1:])
    valid = counts > 0
    pk = np.zeros_like(k_centers)
    pk[valid] = psum[valid] / counts[valid]
    return k_centers[valid], pk[valid], counts[valid]


def power_spectrum_plotter(positions, boxsize: float, Nmesh: int, nbins: int = 100, grid_dtype=np.float64, clip: float = 1e-3,
                           device: int = 0):
    """(k_centers[valid], Pk_binned[valid]) — the estimator of the thesis' two plotting scripts after their h5py read:
    1LPTvs2LPT/1vs2tlptplotter.py:81-116 (float64 grid, 100 bin edges) and Different_Sizes/Plotter.py:148-202 (float32 grid:
    `grid_dtype=np.float32`, 300 edges): CIC deposit, delta = n / nbar - 1, fftn(delta) / clip(W_CIC, 1e-3), |.|^2 V / N^6, log-spaced
    bins from k_min to the corner mode. The float32 variant rounds the deposited grid to float32 once (the script accumulates in
    float32 particle by particle; the difference is ~1e-7 relative, below the estimator's own round-off)."""
    pos = np.ascontiguousarray(positions, np.float64)
    with FieldPlan(Nmesh, boxsize, device) as p:
        p.cic(pos)
        if np.dtype(grid_dtype) == np.float32:
            p.round_f32()
        p.affine(pos.shape[0] / Nmesh ** 3, 1.0, 1.0)          # delta / rho_mean - 1
        if np.dtype(grid_dtype) == np.float32:
            p.round_f32()
        p.deconvolve(clip)
        kbins, psum, counts = p.power_bins(boxsize ** 3 / Nmesh ** 6, nbins)
    k_centers = 0.5 * (kbins[:-1] + kbins[1:])
    valid = counts > 0
    return k_centers[valid], psum[valid] / counts[valid]


def compute_power_spectrum_from_snapshot(snapshot_path: str, L: float, Nmesh: int, deconvolve: bool = False,
                                         use_fac: bool = False):
    """Drop-in for SnapVsPK/Power_compare.py:135-200: returns (k_centers[valid], Pk_binned[valid])."""
    pos = _snap.read_positions(snapshot_path)
    k, p, _ = power_spectrum(pos, L, Nmesh, deconvolve, use_fac)
    return k, p

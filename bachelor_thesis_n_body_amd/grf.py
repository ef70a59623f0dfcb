"""Gaussian-random-field and Zel'dovich initial conditions on the GPU.

Host-side mirror of the reference's generators, same argument meaning:
    createR(N) / hermitian_noise(c_1, c_2)   TheFeedingPipeline/Cosmo data create (new).py:24-47 (variant "assign_self"),
                                             CiC/3D startbetingelser1.py:82-94 (variant "skip_self")
    find_delta(...)  -> delta_from_noise     TheFeedingPipeline/Cosmo data create (new).py:114-137
    displacement_fields / zeldovich_particles  CiC/3D startbetingelser1.py:98-112, 119-173
CLASS is not available, so the power spectrum enters as P(k) evaluated on the |k| grid (`P_k_grid`, e.g. interpolated
from a table with `tabulated_pk_grid`). The Gaussian draws stay on the host (NumPy RNG, like the reference); the
O(N^3) symmetrisation loop, the spectral products and the inverse FFTs run in HIP kernels + hipFFT.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import field as _field

_vp, _dbl, _int = C.c_void_p, C.c_double, C.c_int
GRF_SIGNATURES = {
    "nbf_noise_hermitian": (_int, [_vp, _vp, _vp, _int]),
    "nbf_get_noise": (_int, [_vp, _vp]),
    "nbf_set_amplitude": (_int, [_vp, _vp, _vp]),
    "nbf_grf_field": (_int, [_vp, _int]),
    "nbf_zeldovich": (_int, [_vp, _vp]),
}
_VARIANTS = {"assign_self": 0, "skip_self": 1}
_bound = False


def _bind(lib):
    global _bound
    if not _bound:
        for name, (res, args) in GRF_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _bound = True


def k_grid(N: int, L: float) -> Tuple[np.ndarray, np.ndarray]:
    """(k_vals along one axis, |k| on the grid) — CiC/3D startbetingelser1.py:67-69."""
    k_vals = 2 * np.pi * np.fft.fftfreq(N, d=L / N)
    kx, ky, kz = np.meshgrid(k_vals, k_vals, k_vals, indexing="ij")
    return k_vals, np.sqrt(kx ** 2 + ky ** 2 + kz ** 2)


def tabulated_pk_grid(k_table, p_table, N: int, L: float) -> np.ndarray:
    """P(k) on the |k| grid from a table, log-log interpolated and clipped to the table's range
    (the role of pk_interp / np.clip in CiC/3D startbetingelser1.py:72-77; the k = 0 mode gets P = 0)."""
    _, kmag = k_grid(N, L)
    kt, pt = np.asarray(k_table, np.float64), np.asarray(p_table, np.float64)
    kc = np.clip(kmag, kt.min(), kt.max())
    out = np.exp(np.interp(np.log(kc), np.log(kt), np.log(pt)))
    out[0, 0, 0] = 0.0
    return out


class GRFPlan(_field.FieldPlan):
    """FieldPlan + the noise / amplitude grids of the generator, all resident in HBM."""

    def __init__(self, ngrid: int, boxsize: float, device: int = 0):
        super().__init__(ngrid, boxsize, device)
        _bind(self._lib)

    def set_noise(self, c_1, c_2, variant: str = "assign_self") -> None:
        a = np.ascontiguousarray(c_1, np.float64)
        b = np.ascontiguousarray(c_2, np.float64)
        if a.shape != (self.ngrid,) * 3 or b.shape != a.shape:
            raise ValueError(f"c_1 and c_2 must have shape {(self.ngrid,) * 3}")
        self._chk(self._lib.nbf_noise_hermitian(self._p, _field._ptr(a), _field._ptr(b), _VARIANTS[variant]), "nbf_noise_hermitian")

    def noise(self) -> np.ndarray:
        out = np.empty((self.ngrid,) * 3, np.complex128)
        self._chk(self._lib.nbf_get_noise(self._p, _field._ptr(out)), "nbf_get_noise")
        return out

    def set_spectrum(self, P_k_grid) -> None:
        P = np.asarray(P_k_grid, np.float64)
        if P.shape != (self.ngrid,) * 3:
            raise ValueError(f"P_k_grid must have shape {(self.ngrid,) * 3}")
        fac = np.sqrt(2 * np.pi / (self.boxsize ** 3)) * self.ngrid ** 3      # Cosmo data create (new).py:131-132
        amp = np.ascontiguousarray(np.sqrt(P) * fac)
        k_vals = np.ascontiguousarray(2 * np.pi * np.fft.fftfreq(self.ngrid, d=self.boxsize / self.ngrid))
        self._chk(self._lib.nbf_set_amplitude(self._p, _field._ptr(amp), _field._ptr(k_vals)), "nbf_set_amplitude")

    def field(self, component: int = 0) -> np.ndarray:
        """0: delta(x); 1..3: psi_x, psi_y, psi_z."""
        self._chk(self._lib.nbf_grf_field(self._p, int(component)), "nbf_grf_field")
        return self.get()

    def zeldovich(self) -> np.ndarray:
        out = np.empty((self.ngrid ** 3, 3), np.float64)
        self._chk(self._lib.nbf_zeldovich(self._p, _field._ptr(out)), "nbf_zeldovich")
        return out


def hermitian_noise(c_1, c_2, variant: str = "assign_self", device: int = 0) -> np.ndarray:
    """The reference's createR symmetrisation of c_1 + 1j*c_2, on the GPU."""
    n = np.asarray(c_1).shape[0]
    with GRFPlan(n, 1.0, device) as p:
        p.set_noise(c_1, c_2, variant)
        return p.noise()


def createR(N: int, rng: Optional[np.random.Generator] = None, mu: float = 0.0, sig: float = 1 / np.sqrt(2),
            variant: str = "assign_self") -> np.ndarray:
    """createR(N) of Cosmo data create (new).py:24-47; draws come from `rng` (default: NumPy's global RNG, like the
    reference) in the reference's order (c_1 first, then c_2)."""
    size = [N] * 3
    if rng is None:
        c_1 = np.random.normal(mu, sig, size=size); c_2 = np.random.normal(mu, sig, size=size)
    else:
        c_1 = rng.normal(mu, sig, size=size); c_2 = rng.normal(mu, sig, size=size)
    return hermitian_noise(c_1, c_2, variant)


def delta_from_noise(P_k_grid, c_1, c_2, L: float, variant: str = "assign_self", device: int = 0) -> np.ndarray:
    n = np.asarray(c_1).shape[0]
    with GRFPlan(n, L, device) as p:
        p.set_noise(c_1, c_2, variant)
        p.set_spectrum(P_k_grid)
        return p.field(0)


def zeldovich_particles(P_k_grid, c_1, c_2, L: float, variant: str = "skip_self", device: int = 0) -> np.ndarray:
    """(N^3, 3) positions of a lattice displaced by psi = ifftn(i k/k^2 delta_k R), wrapped into [0, L)."""
    n = np.asarray(c_1).shape[0]
    with GRFPlan(n, L, device) as p:
        p.set_noise(c_1, c_2, variant)
        p.set_spectrum(P_k_grid)
        return p.zeldovich()

"""CPU oracle for periodic (Ewald) gravity and the comoving background — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED at the level of forces: the reference ran CONCEPT's `pp` gravity in a periodic comoving box but only the
resulting power spectrum survives (PPPMP3M/PP/powerspec/powerspec_a=1.00:11-26). This module states the textbook
Ewald sum in NumPy (real-space sum over the 27 nearest images, reciprocal sum over |n| <= kmax) for small N; it is pinned by
the splitting-parameter independence of the result, by the vanishing force on a perfect lattice, by the Newtonian limit at small
separation and by momentum conservation (tests/test_ewald_oracle.py). The end-to-end pin against the reference's P(k) table
is statistical (tests/test_gpu_cosmo.py).
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf


def ewald_accel(pos, mass, L: float, eps: float = 0.0, G: float = 1.0, alpha_L: float = 6.2, kmax: int = 7,
                images: int = 1) -> np.ndarray:
    pos = np.asarray(pos, np.float64)
    mass = np.asarray(mass, np.float64)
    n = pos.shape[0]
    alpha = alpha_L / L
    acc = np.zeros((n, 3))
    d0 = pos[None, :, :] - pos[:, None, :]
    d0 -= L * np.rint(d0 / L)                                  # minimum image
    rng = range(-images, images + 1)
    for ix in rng:
        for iy in rng:
            for iz in rng:
                d = d0 + L * np.array([ix, iy, iz], np.float64)
                r2 = (d * d).sum(-1)
                with np.errstate(divide="ignore", invalid="ignore"):
                    r = np.sqrt(r2)
                    u = alpha * r
                    lng = (erf(u) - 2 * u / np.sqrt(np.pi) * np.exp(-u * u)) / (r2 * r)
                    soft = (r2 + eps * eps) ** -1.5 if (ix, iy, iz) == (0, 0, 0) else 1.0 / (r2 * r)
                    f = np.where(r2 > 0, mass[None, :] * (soft - lng), 0.0)
                acc += (f[:, :, None] * d).sum(1)
    V = L ** 3
    ks = []
    for a in range(-kmax, kmax + 1):
        for b in range(-kmax, kmax + 1):
            for c in range(-kmax, kmax + 1):
                if 0 < a * a + b * b + c * c <= kmax * kmax:
                    ks.append((a, b, c))
    k = 2 * np.pi / L * np.array(ks, np.float64)               # all k (both half spaces)
    k2 = (k * k).sum(1)
    w = (4 * np.pi / V) * np.exp(-k2 / (4 * alpha * alpha)) / k2
    ph = pos @ k.T                                             # (n, nk)
    sc, ss = mass @ np.cos(ph), mass @ np.sin(ph)
    t = -w[None, :] * (sc[None, :] * np.sin(ph) - ss[None, :] * np.cos(ph))
    acc += t @ k
    return G * acc


# ---------------------------------------------------------------- flat LCDM background (radiation optional)
def hubble(a, H0, Om, Or=0.0):
    a = np.asarray(a, np.float64)
    return H0 * np.sqrt(Or / a ** 4 + Om / a ** 3 + (1.0 - Om - Or))


def growth(a, Om, Or=0.0, n=4000):
    """Linear growth factor D(a) (normalised D(1) = 1) and f = dlnD/dlna, from the integral solution for flat LCDM
    (exact without radiation; radiation enters the background only)."""
    def D_un(aa):
        x = np.linspace(1e-8, aa, n)
        E = np.sqrt(Or / x ** 4 + Om / x ** 3 + (1.0 - Om - Or))
        integ = np.trapezoid(1.0 / (x * E) ** 3, x)
        return np.sqrt(Or / aa ** 4 + Om / aa ** 3 + (1.0 - Om - Or)) * integ
    d1 = D_un(1.0)
    D = D_un(a) / d1
    h = 1e-4 * a
    f = (np.log(D_un(a + h)) - np.log(D_un(a - h))) / (np.log(a + h) - np.log(a - h))
    return D, f

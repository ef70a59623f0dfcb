"""CPU oracle for the PM / P3M solvers — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED at the level of forces: the reference selects CONCEPT's `'gravity': 'pm'` / `'p3m'`
(Different_Sizes/Giant.param:66-68, TheFeedingPipeline/NN.param:85-87, with the `potential_options` of NN.param:53-82) but
CONCEPT's source is not part of it and only power spectra of such runs survive (PPPMP3M/{PM,P3M}/powerspec/powerspec_a=1.00).
This module restates the textbook method in NumPy (Hockney & Eastwood; CONCEPT's documented short-range defaults
scale = 1.25 cells, range = 4.5 scale):

  mesh  : CIC deposit (optionally two interlaced lattices) -> FFT -> -4 pi G / k^2 [ * exp(-k^2 r_s^2) for P3M ],
          CIC window deconvolved twice, i k differentiation, Nyquist planes zeroed -> inverse FFT -> CIC interpolation
  short : sum over minimum-image pairs with r < r_cut of
          G m_j d [ (r^2+eps^2)^{-3/2} - (erf(r/2r_s) - r/(r_s sqrt(pi)) exp(-r^2/4r_s^2)) / r^3 ]

It is pinned by its agreement with the Ewald oracle (`ewald_oracle.py`), i.e. P3M -> exact periodic gravity
(tests/test_mesh_oracle.py), and the end-to-end pin against the reference is the ratio of its P3M and PP power spectra.
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf


def _cic_weights(pos, h, ngrid, shift):
    u = pos / h - shift
    f = np.floor(u)
    w1 = u - f
    i0 = np.mod(f.astype(np.int64), ngrid)
    i1 = np.mod(i0 + 1, ngrid)
    return i0, i1, 1.0 - w1, w1


def cic_deposit(pos, mass, L, ngrid, shift=0.0):
    i0, i1, w0, w1 = _cic_weights(np.asarray(pos, np.float64), L / ngrid, ngrid, shift)
    grid = np.zeros((ngrid,) * 3)
    for a, (ia, wa) in enumerate(((i0[:, 0], w0[:, 0]), (i1[:, 0], w1[:, 0]))):
        for b, (ib, wb) in enumerate(((i0[:, 1], w0[:, 1]), (i1[:, 1], w1[:, 1]))):
            for c, (ic, wc) in enumerate(((i0[:, 2], w0[:, 2]), (i1[:, 2], w1[:, 2]))):
                np.add.at(grid, (ia, ib, ic), mass * wa * wb * wc)
    return grid


def cic_gather(grid, pos, L, shift=0.0):
    ngrid = grid.shape[0]
    i0, i1, w0, w1 = _cic_weights(np.asarray(pos, np.float64), L / ngrid, ngrid, shift)
    out = np.zeros(pos.shape[0])
    for ia, wa in ((i0[:, 0], w0[:, 0]), (i1[:, 0], w1[:, 0])):
        for ib, wb in ((i0[:, 1], w0[:, 1]), (i1[:, 1], w1[:, 1])):
            for ic, wc in ((i0[:, 2], w0[:, 2]), (i1[:, 2], w1[:, 2])):
                out += wa * wb * wc * grid[ia, ib, ic]
    return out


def mesh_accel(pos, mass, L, ngrid, G=1.0, rs=None, interlace=True):
    """Mesh force on the particles; rs = None: plain PM, otherwise the Gaussian-damped long-range part of P3M."""
    pos = np.asarray(pos, np.float64); mass = np.asarray(mass, np.float64)
    h = L / ngrid
    kv = 2 * np.pi * np.fft.fftfreq(ngrid, d=h)
    if ngrid % 2 == 0:
        kv[ngrid // 2] = np.pi / h                       # sign convention of the +Nyquist index (planes are zeroed anyway)
    kx, ky, kz = np.meshgrid(kv, kv, kv, indexing="ij")
    theta = 0.5 * h * (kx + ky + kz)
    rho = np.fft.fftn(cic_deposit(pos, mass, L, ngrid, 0.0))
    if interlace:
        rho = 0.5 * (rho + np.exp(-1j * theta) * np.fft.fftn(cic_deposit(pos, mass, L, ngrid, 0.5)))
    k2 = kx ** 2 + ky ** 2 + kz ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        g = 4 * np.pi * G / L ** 3 / k2
        if rs is not None:
            g = g * np.exp(-k2 * rs * rs)
        w = np.sinc(kv * h / (2 * np.pi)) ** 2           # CIC window per axis
        W = w[:, None, None] * w[None, :, None] * w[None, None, :]
        g = g / W ** 2
    dead = np.zeros((ngrid,) * 3, bool)
    dead[0, 0, 0] = True
    if ngrid % 2 == 0:
        n2 = ngrid // 2
        dead[n2, :, :] = True; dead[:, n2, :] = True; dead[:, :, n2] = True
    g[dead] = 0.0
    acc = np.zeros((pos.shape[0], 3))
    ncells = ngrid ** 3
    for c, kc in enumerate((kx, ky, kz)):
        gk = 1j * kc * g * rho
        f1 = np.fft.ifftn(gk).real * ncells              # unnormalised inverse, as hipFFT
        a = cic_gather(f1, pos, L, 0.0)
        if interlace:
            f2 = np.fft.ifftn(gk * np.exp(1j * theta)).real * ncells
            a = 0.5 * (a + cic_gather(f2, pos, L, 0.5))
        acc[:, c] = a
    return acc


def short_range_accel(pos, mass, L, rs, rcut, eps=0.0, G=1.0, chunk=512, sinks=None):
    """Erfc-complement pair force inside rcut, minimum image, all pairs (O(N^2)). `sinks`: indices of the particles whose acceleration is wanted
    (default all) — lets a large configuration be checked on a sample."""
    pos = np.asarray(pos, np.float64); mass = np.asarray(mass, np.float64)
    pos = pos - L * np.floor(pos / L)
    n = pos.shape[0]
    alpha = 0.5 / rs
    snk = np.arange(n) if sinks is None else np.asarray(sinks)
    chunk = max(1, min(chunk, int(2e7 // n)))
    acc = np.zeros((len(snk), 3))
    for s in range(0, len(snk), chunk):
        d = pos[None, :, :] - pos[snk[s:s + chunk], None, :]
        d -= L * np.rint(d / L)
        r2 = (d * d).sum(-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.sqrt(r2)
            u = alpha * r
            lng = (erf(u) - 2 * u / np.sqrt(np.pi) * np.exp(-u * u)) / (r2 * r)
            f = np.where((r2 > 0) & (r2 < rcut * rcut), mass[None, :] * ((r2 + eps * eps) ** -1.5 - lng), 0.0)
        acc[s:s + chunk] = (f[:, :, None] * d).sum(1)
    return G * acc


def p3m_accel(pos, mass, L, ngrid, eps=0.0, G=1.0, scale_cells=1.25, range_scales=4.5, interlace=True):
    rs = scale_cells * L / ngrid
    return mesh_accel(pos, mass, L, ngrid, G, rs, interlace) + short_range_accel(pos, mass, L, rs, range_scales * rs, eps, G)

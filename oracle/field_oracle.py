"""CPU oracle for the snapshot -> density-field chain — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Unlike the gravity oracle this one IS pinned by the reference's own data: the restatement below reproduces
TheFeedingPipeline/Datafolder4/{Test data_CONCEPT,Training & val data_CONCEPT}/delta_*_id-*.npy from the CONCEPT
snapshots next to them (tests/golden/concept/, tests/test_field_oracle.py) to <= 1e-13.

Each function follows the reference line by line (same operation order in fp64):
  cic_density          TheFeedingPipeline/FraP_kTilDelta.py:93-123  (identical loops: SnapVsPK/Power_compare.py:100-132)
  compute_delta        TheFeedingPipeline/FraP_kTilDelta.py:126-127
  cic_window_ft        TheFeedingPipeline/FraP_kTilDelta.py:73-84   (zero guard) / SnapVsPK/Power_compare.py:80-90 (none)
  deconvolve_cic       TheFeedingPipeline/FraP_kTilDelta.py:87-91   (Power_compare.py:93-97 uses norm='ortho': same result)
  delta_field          TheFeedingPipeline/Feeder_pipe.py:174-195    (test: no mean subtraction; train: `delta -= delta.mean()`)
  power_spectrum       SnapVsPK/Power_compare.py:135-200
"""
from __future__ import annotations

import numpy as np


def cic_density(positions, boxsize: float, N: int) -> np.ndarray:
    """Cloud-in-cell deposit of unit weights. The reference loops over particles and over the 8 corners
    (di, dj, dk) in that nesting order; np.add.at applies the same additions in the same order."""
    p = np.asarray(positions, np.float64)
    cell_size = boxsize / N
    q = p / cell_size
    i0 = q.astype(np.int64) % N                       # int(p/cell) % N   (truncation, then Python modulo)
    d = np.mod(p, cell_size) / cell_size              # (p % cell) / cell (Python float modulo == np.mod)
    grid = np.zeros((N, N, N), np.float64)
    idx = np.empty((p.shape[0], 8), np.int64)
    wgt = np.empty((p.shape[0], 8), np.float64)
    c = 0
    for di in (0, 1):
        wx = d[:, 0] if di else 1 - d[:, 0]
        for dj in (0, 1):
            wy = d[:, 1] if dj else 1 - d[:, 1]
            for dk in (0, 1):
                wz = d[:, 2] if dk else 1 - d[:, 2]
                wgt[:, c] = wx * wy * wz
                idx[:, c] = (((i0[:, 0] + di) % N) * N + (i0[:, 1] + dj) % N) * N + (i0[:, 2] + dk) % N
                c += 1
    np.add.at(grid.reshape(-1), idx.reshape(-1), wgt.reshape(-1))   # particle-major, corner-minor: reference order
    return grid


def compute_delta(density: np.ndarray) -> np.ndarray:
    return density / density.mean() - 1.0


def window_1d(ngrid: int, boxsize: float) -> np.ndarray:
    """sinc(k/(2 k_Nyq)) along one axis, computed with exactly the reference's expressions."""
    d = boxsize / ngrid
    k_vals = 2 * np.pi * np.fft.fftfreq(ngrid, d=d)
    k_quist = np.pi * (d) ** (-1)
    return np.sinc(k_vals / (2 * k_quist))


def cic_window_ft(ngrid: int, boxsize: float, zero_guard: bool = True) -> np.ndarray:
    w = window_1d(ngrid, boxsize)
    Wx, Wy, Wz = np.meshgrid(w, w, w, indexing="ij")
    W = (Wx * Wy * Wz) ** 2
    if zero_guard:
        W[W == 0] = 1.0
    return W


def deconvolve_cic(delta_grid: np.ndarray, boxsize: float, ngrid: int) -> np.ndarray:
    delta_k = np.fft.fftn(delta_grid)
    delta_k /= cic_window_ft(ngrid, boxsize)
    return np.fft.ifftn(delta_k).real


def delta_field(positions, boxsize: float, ngrid: int, subtract_mean: bool) -> np.ndarray:
    dens = cic_density(positions, boxsize, ngrid)
    delta = compute_delta(dens).astype(np.float64)
    delta = deconvolve_cic(delta, boxsize, ngrid).astype(np.float64)
    if subtract_mean:
        delta -= delta.mean()
    return delta


def kbin_edges(L: float, Nmesh: int, nbins: int = 300):
    """k-grid values along one axis and the log-spaced bin edges of Power_compare.py:179-191."""
    k_vals = 2 * np.pi * np.fft.fftfreq(Nmesh, d=L / Nmesh)
    a = np.abs(k_vals).max()
    k_max = np.sqrt(a ** 2 + a ** 2 + a ** 2)          # == k_mag.max(): (kx**2 + ky**2) + kz**2, then sqrt
    k_min = 2 * np.pi / L
    return k_vals, np.logspace(np.log10(k_min), np.log10(k_max), nbins)


def power_spectrum(positions, L: float, Nmesh: int, deconvolve: bool = False, use_fac: bool = False, nbins: int = 300):
    """Returns (k_centers[valid], Pk_binned[valid], counts[valid])."""
    pos = np.asarray(positions, np.float64)
    delta = cic_density(pos, L, Nmesh)
    if not use_fac:
        cell_vol = (L / Nmesh) ** 3
        rho_cell = delta / cell_vol
        rho_mean = pos.shape[0] / (L ** 3)
        delta = (rho_cell / rho_mean) - 1.0
    else:
        delta -= 1
    if deconvolve:
        delta_k = np.fft.fftn(delta, norm="ortho")
        delta_k /= cic_window_ft(Nmesh, L, zero_guard=False)
        delta = np.fft.ifftn(delta_k, norm="ortho").real
    delta_k = np.fft.fftn(delta)
    V = L ** 3
    if use_fac:
        fac = Nmesh ** 3 * np.sqrt(2 * np.pi / V)
        Pk3d = np.abs(delta_k) ** 2 / (fac ** 2)
    else:
        Pk3d = np.abs(delta_k) ** 2 * V / (Nmesh ** 6)
    k_vals, kbins = kbin_edges(L, Nmesh, nbins)
    kx, ky, kz = np.meshgrid(k_vals, k_vals, k_vals, indexing="ij")
    k_mag = np.sqrt(kx ** 2 + ky ** 2 + kz ** 2).ravel()
    Pk_flat = Pk3d.ravel()
    mask = k_mag > 0
    k_mag, Pk_flat = k_mag[mask], Pk_flat[mask]
    k_centers = 0.5 * (kbins[:-1] + kbins[1:])
    counts, _ = np.histogram(k_mag, bins=kbins)
    Psum, _ = np.histogram(k_mag, bins=kbins, weights=Pk_flat)
    valid = counts > 0
    Pk_binned = np.zeros_like(k_centers)
    Pk_binned[valid] = Psum[valid] / counts[valid]
    return k_centers[valid], Pk_binned[valid], counts[valid]


def power_spectrum_plotter(positions, boxsize: float, Nmesh: int, nbins: int = 100, grid_dtype=np.float64, clip: float = 1e-3):
    """Restatement of the estimator of the thesis' plotting scripts, line by line after the h5py read:
    1LPTvs2LPT/1vs2tlptplotter.py:81-116 (`nbins=100`, float64 grid) and Different_Sizes/Plotter.py:148-202 (`nbins=300`,
    `grid_dtype=np.float32`: the density is accumulated in a float32 array, Plotter.py:167)."""
    coords = np.asarray(positions, np.float64)
    k_vals = 2 * np.pi * np.fft.fftfreq(Nmesh, d=boxsize / Nmesh)
    kx, ky, kz = np.meshgrid(k_vals, k_vals, k_vals, indexing="ij")
    k_mag = np.sqrt(kx ** 2 + ky ** 2 + kz ** 2)
    k_quist = np.pi * Nmesh / boxsize
    W = np.clip((np.sinc(kx / (2 * k_quist)) * np.sinc(ky / (2 * k_quist)) * np.sinc(kz / (2 * k_quist))) ** 2, clip, None)
    delta = cic_density(coords, boxsize, Nmesh).astype(grid_dtype)        # the scripts' cic_assign is the same deposit (weights 1)
    rho_mean = coords.shape[0] / Nmesh ** 3
    delta = delta / rho_mean - 1.0
    delta_k = np.fft.fftn(delta) / W
    Pk = np.abs(delta_k) ** 2 * boxsize ** 3 / (Nmesh ** 6)
    k_min, k_max = 2 * np.pi / boxsize, k_mag.max()
    bins = np.logspace(np.log10(k_min), np.log10(k_max), nbins)
    centers = 0.5 * (bins[:-1] + bins[1:])
    kf, Pf = k_mag.ravel(), Pk.ravel()
    mask = kf > 0
    counts, _ = np.histogram(kf[mask], bins=bins)
    sums, _ = np.histogram(kf[mask], bins=bins, weights=Pf[mask])
    nz = counts > 0
    return centers[nz], sums[nz] / counts[nz]


def power_modes(pos, boxsize: float, ngrid: int, interlace: bool = True):
    """Per-mode (|k|, P) of the CIC-deposited, window-deconvolved overdensity; interlace: two lattices half a cell apart combined in
    k-space — CONCEPT's estimator settings ('interpolation': 'CIC', 'deconvolve': True, 'interlace': True: TheFeedingPipeline/NN.param:30-41),
    the estimator behind the tables PPPMP3M/{PP,PM,P3M}/powerspec/powerspec_a=1.00. NumPy restatement (host FFTs) of what
    `cosmo.power_modes` / `nbf_power_modes` compute on the GPU."""
    h = boxsize / ngrid
    pos = np.asarray(pos, np.float64)
    d1 = np.fft.fftn(compute_delta(cic_density(pos, boxsize, ngrid)))
    kv = 2 * np.pi * np.fft.fftfreq(ngrid, d=h)
    kx, ky, kz = np.meshgrid(kv, kv, kv, indexing="ij")
    if interlace:
        d2 = np.fft.fftn(compute_delta(cic_density(np.mod(pos - 0.5 * h, boxsize), boxsize, ngrid)))
        d1 = 0.5 * (d1 + np.exp(-0.5j * h * (kx + ky + kz)) * d2)
    w = np.sinc(kv * h / (2 * np.pi))                      # CIC window per axis: sinc^2
    W = (w[:, None, None] * w[None, :, None] * w[None, None, :]) ** 2
    pk = np.abs(d1 / W) ** 2 * boxsize ** 3 / ngrid ** 6
    kmag = np.sqrt(kx ** 2 + ky ** 2 + kz ** 2)
    sel = kmag > 0
    return kmag[sel], pk[sel]

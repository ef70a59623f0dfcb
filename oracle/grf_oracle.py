"""CPU oracle for the Gaussian-random-field + Zel'dovich initial-condition generator —
TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, line by line, what the reference does after CLASS has produced P(k) on the |k| grid (CLASS is not
available here, so P(k) is an INPUT: any tabulated spectrum evaluated on the grid):
  hermitian_noise      TheFeedingPipeline/Cosmo data create (new).py:24-47   (variant "assign_self": self-conjugate
                       points are overwritten by their own conjugate)  and
                       CiC/3D startbetingelser1.py:82-94                     (variant "skip_self")
  delta_from_noise     TheFeedingPipeline/Cosmo data create (new).py:129-135 / CiC/3D startbetingelser1.py:98-102
  displacement_fields  CiC/3D startbetingelser1.py:105-112, 131-137
  zeldovich_particles  CiC/3D startbetingelser1.py:119-124, 160-173 (particles start ON the grid points, so the
                       reference's RegularGridInterpolator returns the node values)
Pinning: the reference draws its noise from an unseeded NumPy RNG (Cosmo data create (new).py:216,251), so no
fixture of it can be reproduced bit for bit; the shipped fields only pin statistics. The triple loop is restated
literally for small N (tests/test_grf_oracle.py) and by its closed form for speed.
"""
from __future__ import annotations

import numpy as np


def hermitian_noise_loop(c_1: np.ndarray, c_2: np.ndarray, variant: str = "assign_self") -> np.ndarray:
    """The reference's O(N^3) Python triple loop, verbatim (use for N <= 16)."""
    N = c_1.shape[0]
    R = c_1 + 1j * c_2
    R[0, 0, 0] = R[0, 0, 0].real
    for i in np.arange(N):
        for j in np.arange(N):
            for k in np.arange(N):
                ii, jj, kk = -i % N, -j % N, -k % N
                if variant == "skip_self" and (i, j, k) == (ii, jj, kk):
                    continue
                R[ii, jj, kk] = np.conj(R[i, j, k])
    return R


def hermitian_noise(c_1: np.ndarray, c_2: np.ndarray, variant: str = "assign_self") -> np.ndarray:
    """Closed form of the loop: in lexicographic order the FIRST visited member of each mirror pair keeps its value
    and the later one becomes its conjugate; a self-conjugate point is conjugated once ("assign_self") or kept
    ("skip_self"); R[0,0,0] is made real beforehand."""
    N = c_1.shape[0]
    R = c_1 + 1j * c_2
    R[0, 0, 0] = R[0, 0, 0].real
    idx = np.arange(N)
    I, J, K = np.meshgrid(idx, idx, idx, indexing="ij")
    MI, MJ, MK = (-I) % N, (-J) % N, (-K) % N
    lin, mlin = (I * N + J) * N + K, (MI * N + MJ) * N + MK
    out = R.copy()
    later = lin > mlin                       # visited after its mirror: overwritten by conj(original mirror)
    out[later] = np.conj(R[MI[later], MJ[later], MK[later]])
    if variant == "assign_self":
        self_pt = lin == mlin
        out[self_pt] = np.conj(R[self_pt])
    return out


def k_grids(N: int, L: float):
    k_vals = 2 * np.pi * np.fft.fftfreq(N, d=L / N)
    kx, ky, kz = np.meshgrid(k_vals, k_vals, k_vals, indexing="ij")
    k_mag = np.sqrt(kx ** 2 + ky ** 2 + kz ** 2)
    return k_vals, kx, ky, kz, k_mag


def delta_k_amplitude(P_k_grid: np.ndarray, L: float, N: int) -> np.ndarray:
    fac = np.sqrt(2 * np.pi / (L ** 3)) * N ** 3
    return np.sqrt(P_k_grid) * fac


def delta_from_noise(P_k_grid: np.ndarray, R: np.ndarray, L: float) -> np.ndarray:
    N = R.shape[0]
    return np.fft.ifftn(delta_k_amplitude(P_k_grid, L, N) * R).real.astype(np.float64)


def displacement_fields(P_k_grid: np.ndarray, R: np.ndarray, L: float):
    N = R.shape[0]
    _, kx, ky, kz, k_mag = k_grids(N, L)
    k_mag[0, 0, 0] = 1e-10
    delta_k = delta_k_amplitude(P_k_grid, L, N)
    out = []
    for kc in (kx, ky, kz):
        psi_k = 1j * kc / k_mag ** 2 * delta_k
        out.append(np.fft.ifftn(psi_k * R).real)
    return out


def zeldovich_particles(P_k_grid: np.ndarray, R: np.ndarray, L: float) -> np.ndarray:
    N = R.shape[0]
    g = np.linspace(0, L, N, endpoint=False)
    particles = np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T.copy()
    px, py, pz = displacement_fields(P_k_grid, R, L)
    particles += np.column_stack((px.reshape(-1), py.reshape(-1), pz.reshape(-1)))
    particles %= L
    return particles


def second_order_potential_gradient(psi, boxsize: float) -> np.ndarray:
    """grad(phi2) on the lattice from the first-order displacement psi = -grad(phi1) (array (3, n, n, n)), where
        laplace(phi2) = sum_{i<j} [ phi1,ii phi1,jj - (phi1,ij)^2 ]
    — the source of second-order Lagrangian perturbation theory (2LPT; Bouchet et al. 1995, Scoccimarro 1998), CONCEPT's
    `realization_options = {'LPT': 2}` (Different_Sizes/Giant.param:6,70-72 carries the switch, 1LPTvs2LPT/1vs2tlptplotter.py:121-123
    compares the two). All derivatives spectral: phi1,ij = -d_j psi_i. NumPy restatement of what `nbf_lpt2_gradient` computes with hipFFT."""
    psi = np.asarray(psi, np.float64)
    n = psi.shape[1]
    kfull = 2 * np.pi * np.fft.fftfreq(n, d=boxsize / n)
    kv = kfull.copy()
    if n % 2 == 0:
        kv[n // 2] = 0.0                                   # odd derivatives of a real field have no Nyquist component
    k = np.meshgrid(kv, kv, kv, indexing="ij")
    pk = [np.fft.fftn(psi[i]) for i in range(3)]
    d = lambda i, j: np.fft.ifftn(-1j * k[j] * pk[i]).real * -1.0      # phi1,ij = -d_j psi_i
    xx, yy, zz = d(0, 0), d(1, 1), d(2, 2)
    xy, xz, yz = d(0, 1), d(0, 2), d(1, 2)
    src = np.fft.fftn(xx * yy + xx * zz + yy * zz - xy ** 2 - xz ** 2 - yz ** 2)
    k2 = kfull[:, None, None] ** 2 + kfull[None, :, None] ** 2 + kfull[None, None, :] ** 2
    k2[0, 0, 0] = 1.0
    phi2 = -src / k2
    phi2[0, 0, 0] = 0.0
    return np.stack([np.fft.ifftn(1j * k[c] * phi2).real for c in range(3)])

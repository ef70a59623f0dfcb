"""Pins the GRF / Zel'dovich oracle (oracle/grf_oracle.py): the closed form of the reference's Hermitian-symmetry
triple loop equals the literal loop, the generated field is real with the requested spectrum, and the displaced
lattice reproduces the density field to linear order. (The reference's noise is unseeded — statistical pins only.)"""
import numpy as np
import pytest

from oracle import grf_oracle as go
from oracle import field_oracle as fo


@pytest.mark.parametrize("N", [1, 2, 4, 5, 8])
@pytest.mark.parametrize("variant", ["assign_self", "skip_self"])
def test_closed_form_equals_reference_triple_loop(N, variant):
    rng = np.random.default_rng(N)
    c1, c2 = rng.normal(0, 1 / np.sqrt(2), (N, N, N)), rng.normal(0, 1 / np.sqrt(2), (N, N, N))
    np.testing.assert_array_equal(go.hermitian_noise(c1, c2, variant), go.hermitian_noise_loop(c1, c2, variant))


def test_noise_is_hermitian_off_the_self_conjugate_points():
    N = 8
    rng = np.random.default_rng(0)
    R = go.hermitian_noise(rng.normal(size=(N,) * 3), rng.normal(size=(N,) * 3))
    idx = (-np.arange(N)) % N
    mirror = np.conj(R[np.ix_(idx, idx, idx)])
    self_pts = np.zeros((N,) * 3, bool)
    self_pts[np.ix_([0, N // 2], [0, N // 2], [0, N // 2])] = True
    np.testing.assert_array_equal(R[~self_pts], mirror[~self_pts])
    assert R[0, 0, 0].imag == 0.0
    # the 7 other self-conjugate points keep an imaginary part in the reference (its ifftn(...).real drops the residue)
    assert np.abs(R[self_pts].imag).max() > 0


def test_field_statistics_and_zeldovich_linear_limit():
    N, L = 32, 500.0
    rng = np.random.default_rng(42)
    _, _, _, _, kmag = go.k_grids(N, L)
    kk = kmag.copy(); kk[0, 0, 0] = 1.0
    P = 2.0e-4 * kk ** -1.5            # small amplitude power law (linear regime: |psi| << cell)
    P[0, 0, 0] = 0.0
    R = go.hermitian_noise(rng.normal(0, 1 / np.sqrt(2), (N,) * 3), rng.normal(0, 1 / np.sqrt(2), (N,) * 3))
    delta = go.delta_from_noise(P, R, L)
    assert delta.dtype == np.float64 and abs(delta.mean()) < 1e-12
    psi = go.displacement_fields(P, R, L)
    assert max(np.abs(p).max() for p in psi) < 0.25 * L / N
    # linear theory: delta = -div(psi); check in Fourier space
    kv, kx, ky, kz, _ = go.k_grids(N, L)
    div_k = 1j * (kx * np.fft.fftn(psi[0]) + ky * np.fft.fftn(psi[1]) + kz * np.fft.fftn(psi[2]))
    nyq = (np.abs(kx) == np.abs(kv).max()) | (np.abs(ky) == np.abs(kv).max()) | (np.abs(kz) == np.abs(kv).max())
    dk = np.fft.fftn(delta)
    np.testing.assert_allclose(-div_k[~nyq], dk[~nyq], rtol=1e-8, atol=1e-10 * np.abs(dk).max())
    # displaced lattice -> CIC density correlates with the target field at low k
    pos = go.zeldovich_particles(P, R, L)
    assert pos.shape == (N ** 3, 3) and pos.min() >= 0 and pos.max() < L
    dens = fo.compute_delta(fo.cic_density(pos, L, N))
    lo = (kmag > 0) & (kmag < 0.3 * np.abs(kv).max())
    a, b = np.fft.fftn(dens)[lo], dk[lo]
    corr = np.abs(np.vdot(a, b)) / np.sqrt(np.vdot(a, a).real * np.vdot(b, b).real)
    assert corr > 0.98

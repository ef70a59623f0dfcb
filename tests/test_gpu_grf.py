"""GPU parity of the GRF / Zel'dovich generator (include/nbody_field.h) against the oracle that
restates the reference's loops. The Hermitian symmetrisation only moves/negates values: bit-exact. Fields pass through
one hipFFT inverse transform: |difference| <= 1e-12 of the field's largest value."""
import numpy as np
import pytest

from bachelor_thesis_n_body_amd import grf
from oracle import grf_oracle as go

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [1, 2, 4, 5, 8, 33, 64])
@pytest.mark.parametrize("variant", ["assign_self", "skip_self"])
def test_hermitian_noise_bit_exact(N, variant):
    rng = np.random.default_rng(N)
    c1, c2 = rng.normal(0, 1 / np.sqrt(2), (N,) * 3), rng.normal(0, 1 / np.sqrt(2), (N,) * 3)
    got = grf.hermitian_noise(c1, c2, variant)
    want = go.hermitian_noise_loop(c1, c2, variant) if N <= 8 else go.hermitian_noise(c1, c2, variant)
    np.testing.assert_array_equal(got.real, want.real)
    np.testing.assert_array_equal(got.imag, want.imag)


def _setup(N, L, seed):
    rng = np.random.default_rng(seed)
    _, kmag = grf.k_grid(N, L)
    kk = kmag.copy(); kk[0, 0, 0] = 1.0
    P = 3.0e-3 * kk ** -1.2
    P[0, 0, 0] = 0.0
    return P, rng.normal(0, 1 / np.sqrt(2), (N,) * 3), rng.normal(0, 1 / np.sqrt(2), (N,) * 3)


@pytest.mark.parametrize("N,L", [(16, 100.0), (32, 500.0), (48, 250.0)])
def test_delta_and_displacements_match_oracle(N, L):
    P, c1, c2 = _setup(N, L, N)
    R = go.hermitian_noise(c1, c2, "assign_self")
    want = go.delta_from_noise(P, R, L)
    got = grf.delta_from_noise(P, c1, c2, L, "assign_self")
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * np.abs(want).max())
    with grf.GRFPlan(N, L) as p:
        p.set_noise(c1, c2, "assign_self")
        p.set_spectrum(P)
        psi = go.displacement_fields(P, R, L)
        for c in range(3):
            np.testing.assert_allclose(p.field(c + 1), psi[c], rtol=0, atol=1e-12 * np.abs(psi[c]).max())


def test_zeldovich_particles_match_oracle_and_feed_the_field_chain():
    from bachelor_thesis_n_body_amd import field
    N, L = 32, 500.0
    P, c1, c2 = _setup(N, L, 7)
    R = go.hermitian_noise(c1, c2, "skip_self")
    want = go.zeldovich_particles(P, R, L)
    got = grf.zeldovich_particles(P, c1, c2, L, "skip_self")
    assert got.shape == (N ** 3, 3) and got.min() >= 0 and got.max() < L
    d = np.abs(got - want)
    d = np.minimum(d, L - d)                 # a particle within round-off of the box edge may wrap differently
    assert d.max() < 1e-10
    # generated ICs -> CIC -> P(k): the loop the thesis closes in CiC/3D startbetingelser1.py:274-315
    k, pk, cnt = field.power_spectrum(got, L, N, deconvolve=True)
    assert np.all(np.isfinite(pk)) and cnt.sum() == N ** 3 - 1

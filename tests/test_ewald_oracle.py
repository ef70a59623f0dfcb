"""Pins the Ewald oracle: independence of the splitting parameter, zero force on a perfect lattice, Newtonian limit,
momentum conservation, periodicity; and the LCDM growth factor against its Einstein-de Sitter limit."""
import numpy as np
import pytest

from oracle import ewald_oracle as eo


def _rand(n, L, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(0, L, (n, 3)), rng.uniform(0.5, 2.0, n)


def test_result_does_not_depend_on_the_splitting():
    pos, m = _rand(40, 10.0, 0)
    a = eo.ewald_accel(pos, m, 10.0, alpha_L=5.0, kmax=9, images=2)
    b = eo.ewald_accel(pos, m, 10.0, alpha_L=7.5, kmax=13, images=1)
    assert np.abs(a - b).max() < 1e-7 * np.abs(a).max()
    # the product's parameters (alpha L = 6.2, kmax = 7, minimum image only) are good to ~1e-5
    c = eo.ewald_accel(pos, m, 10.0, alpha_L=6.2, kmax=7, images=0)
    assert np.abs(c - b).max() < 3e-5 * np.abs(b).max()


def test_perfect_lattice_feels_no_force_and_momentum_is_conserved():
    g = (np.arange(4) + 0.5) * 2.5
    pos = np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T
    a = eo.ewald_accel(pos, np.ones(64), 10.0)
    assert np.abs(a).max() < 1e-9
    pos, m = _rand(30, 7.0, 1)
    a = eo.ewald_accel(pos, m, 7.0)
    assert np.abs((m[:, None] * a).sum(0)).max() < 1e-9 * (m[:, None] * np.abs(a)).sum(0).max()


def test_newtonian_limit_and_periodicity():
    L = 100.0
    pos = np.array([[50.0, 50.0, 50.0], [50.3, 50.0, 50.0]])
    a = eo.ewald_accel(pos, np.array([1.0, 3.0]), L)
    assert a[0, 0] == pytest.approx(3.0 / 0.3 ** 2, rel=1e-5)          # periodic images matter at the (r/L)^3 level
    shifted = (pos + np.array([70.0, -130.0, 260.0])) % L
    np.testing.assert_allclose(eo.ewald_accel(shifted, np.array([1.0, 3.0]), L), a, rtol=1e-9, atol=1e-9)
    # a pair half a box apart along one axis feels no net force from each other + images
    pos = np.array([[10.0, 20.0, 30.0], [60.0, 20.0, 30.0]])
    assert np.abs(eo.ewald_accel(pos, np.ones(2), L)).max() < 1e-10


def test_growth_factor_limits():
    D, f = eo.growth(0.5, 1.0)                       # Einstein-de Sitter: D = a, f = 1
    assert D == pytest.approx(0.5, rel=1e-3) and f == pytest.approx(1.0, rel=1e-3)
    D, f = eo.growth(1.0, 0.31)
    assert D == pytest.approx(1.0) and f == pytest.approx(0.31 ** 0.55, rel=0.02)
    D, _ = eo.growth(0.02, 0.31)
    assert D == pytest.approx(0.02 / 0.787, rel=0.02)   # D(a)/a -> 1/g(Om=0.31) with g ~ 0.787


def test_product_growth_ode_matches_closed_form_without_radiation():
    """cosmo.Cosmology.growth solves the growth ODE (valid with radiation); without radiation it must equal the closed
    integral solution the oracle uses."""
    from bachelor_thesis_n_body_amd import cosmo
    c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12, radiation=False)
    for a in (0.02, 0.1, 0.5, 1.0):
        D, f = c.growth(a)
        D0, f0 = eo.growth(a, c.Om)
        assert D == pytest.approx(D0, rel=1e-6) and f == pytest.approx(f0, rel=1e-5)
    cr = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12, radiation=True)
    assert cr.growth(0.02)[0] == pytest.approx(1.0094 * c.growth(0.02)[0], rel=2e-3)   # radiation: D ~ a + 2/3 a_eq early on

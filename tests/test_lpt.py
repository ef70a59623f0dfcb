"""Second-order Lagrangian perturbation theory source: closed forms, for the NumPy restatement (oracle/grf_oracle.py, CPU) and for the
product (cosmo.second_order_potential_gradient -> nbf_lpt2_gradient, 13 hipFFT transforms; `-m gpu`), and the two against each other."""
import numpy as np
import pytest

from bachelor_thesis_n_body_amd import cosmo, params
from oracle import grf_oracle

IMPLS = [pytest.param(grf_oracle.second_order_potential_gradient, id="oracle"),
         pytest.param(cosmo.second_order_potential_gradient, id="gpu", marks=pytest.mark.gpu)]


def _grid(n, L):
    g = np.arange(n) * L / n
    return np.meshgrid(g, g, g, indexing="ij")


@pytest.mark.parametrize("grad", IMPLS)
@pytest.mark.parametrize("n", [16, 21])
def test_two_crossed_waves_closed_form(n, grad):
    """phi1 = A cos(kx x) + B cos(ky y)  ->  laplace(phi2) = phi1,xx phi1,yy = A B kx^2 ky^2 cos(kx x) cos(ky y)."""
    L = 100.0
    x, y, z = _grid(n, L)
    kx, ky = 2 * np.pi / L * 2, 2 * np.pi / L * 3
    A, B = 0.7, -0.4
    psi = np.stack([A * kx * np.sin(kx * x), B * ky * np.sin(ky * y), np.zeros_like(x)])       # psi = -grad(phi1)
    c = A * B * kx ** 2 * ky ** 2 / (kx ** 2 + ky ** 2)
    want = np.stack([c * kx * np.sin(kx * x) * np.cos(ky * y), c * ky * np.cos(kx * x) * np.sin(ky * y), np.zeros_like(x)])
    got = grad(psi, L)
    assert np.abs(got - want).max() < 1e-13 * np.abs(want).max()


@pytest.mark.parametrize("grad", IMPLS)
def test_plane_wave_has_no_second_order_term(grad):
    """The Zel'dovich approximation is exact in one dimension."""
    L, n = 50.0, 16
    x, y, z = _grid(n, L)
    psi = np.stack([np.zeros_like(x), 0.3 * np.sin(2 * np.pi / L * 3 * y), np.zeros_like(x)])
    assert np.abs(grad(psi, L)).max() < 1e-15


def _band_limited_displacement(n, L, seed=0):
    rng = np.random.default_rng(seed)
    kv = 2 * np.pi * np.fft.fftfreq(n, d=L / n)
    k = np.meshgrid(kv, kv, kv, indexing="ij")
    k2 = k[0] ** 2 + k[1] ** 2 + k[2] ** 2
    k2[0, 0, 0] = 1.0
    low = (np.sqrt(k2) < 0.6 * np.abs(kv).max())                      # band-limited: the product does not alias
    dk = np.fft.fftn(rng.normal(size=(n, n, n))) * low
    dk[0, 0, 0] = 0.0
    return np.stack([np.fft.ifftn(1j * k[c] / k2 * dk).real for c in range(3)]), k, kv


@pytest.mark.parametrize("grad", IMPLS)
def test_second_order_term_is_quadratic_and_curl_free(grad):
    L, n = 80.0, 16
    psi, k, kv = _band_limited_displacement(n, L)
    g1 = grad(psi, L)
    g2 = grad(2.0 * psi, L)
    np.testing.assert_allclose(g2, 4.0 * g1, atol=1e-12 * np.abs(g1).max())
    gk = [np.fft.fftn(g1[c]) for c in range(3)]
    curl_z = np.fft.ifftn(1j * k[0] * gk[1] - 1j * k[1] * gk[0]).real
    assert np.abs(curl_z).max() < 1e-10 * np.abs(g1).max() * np.abs(kv).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [16, 21, 32])
def test_gpu_gradient_matches_the_numpy_restatement(n):
    """A full-band random displacement (aliasing products and all): hipFFT against NumPy, even and odd grids."""
    L = 123.0
    psi = np.random.default_rng(n).normal(size=(3, n, n, n))
    got, ref = cosmo.second_order_potential_gradient(psi, L), grf_oracle.second_order_potential_gradient(psi, L)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()


def test_param_file_switch():
    assert params.settings({"initial_conditions": {"N": 8}, "boxsize": 1.0})["lpt"] == 1
    assert params.settings({"initial_conditions": {"N": 8}, "boxsize": 1.0, "realization_options": {"LPT": 2}})["lpt"] == 2

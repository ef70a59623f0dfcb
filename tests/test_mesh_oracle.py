"""The PM / P3M oracle against the Ewald oracle and closed forms (CPU only)."""
import numpy as np
import pytest

from oracle import ewald_oracle as eo
from oracle import mesh_oracle as mo


@pytest.fixture(scope="module")
def cloud():
    rng = np.random.default_rng(3)
    L, n = 100.0, 700
    return L, rng.uniform(0, L, (n, 3)), rng.uniform(0.5, 1.5, n)


@pytest.mark.parametrize("ngrid,interlace,tol", [(16, True, 5e-3), (32, True, 6e-3), (32, False, 2e-2)])
def test_p3m_converges_to_ewald(cloud, ngrid, interlace, tol):
    """mesh + short range = exact periodic gravity up to the mesh discretisation error (sub-percent with interlacing)."""
    L, pos, m = cloud
    ref = eo.ewald_accel(pos, m, L, 0.3)
    a = mo.p3m_accel(pos, m, L, ngrid, 0.3, interlace=interlace)
    e = np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert np.sqrt((e ** 2).mean()) < tol


def test_short_range_is_antisymmetric_and_cut_off(cloud):
    L, pos, m = cloud
    rs = 1.25 * L / 32
    a = mo.short_range_accel(pos, m, L, rs, 4.5 * rs, 0.3)
    p = (m[:, None] * a).sum(0)
    assert np.abs(p).max() < 1e-12 * np.abs(m[:, None] * a).sum()
    # two particles farther apart than the cut-off feel nothing
    two = np.array([[10.0, 10.0, 10.0], [10.0 + 4.6 * rs, 10.0, 10.0]])
    assert np.all(mo.short_range_accel(two, np.ones(2), L, rs, 4.5 * rs) == 0.0)
    # close pair: Newtonian minus a small long-range remainder
    two[1, 0] = 10.0 + 0.05 * rs
    a2 = mo.short_range_accel(two, np.ones(2), L, rs, 4.5 * rs)
    assert a2[0, 0] == pytest.approx(1.0 / (0.05 * rs) ** 2, rel=1e-3) and a2[1, 0] == -a2[0, 0]


def test_mesh_force_vanishes_on_a_lattice_and_is_periodic():
    L, n = 64.0, 8
    g = (np.arange(n) + 0.25) * (L / n)
    q = np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T
    a = mo.mesh_accel(q, np.ones(len(q)), L, 16)
    assert np.abs(a).max() < 1e-12
    rng = np.random.default_rng(0)
    pos = rng.uniform(0, L, (200, 3)); m = np.ones(200)
    a0 = mo.mesh_accel(pos, m, L, 16, rs=5.0)
    a1 = mo.mesh_accel(pos + np.array([L, -2 * L, 3 * L]), m, L, 16, rs=5.0)
    np.testing.assert_allclose(a1, a0, atol=1e-10 * np.abs(a0).max())


def test_pm_linear_plane_wave():
    """Lattice displaced by psi = A sin(k q) x: the PM acceleration is 4 pi G rho_mean psi (linear theory, k << k_Nyquist).
    The particle lattice sits a quarter cell off the mesh: with particles exactly on the vertices of one of the two
    interlaced lattices CIC is not linearisable (|psi| terms of relative size k h / 2)."""
    L, n = 100.0, 32
    g = (np.arange(n) + 0.25) * (L / n)
    q = np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T
    k = 2 * np.pi / L * 2
    A = 0.02
    psi = A * np.sin(k * q[:, 0])
    pos = q.copy(); pos[:, 0] += psi
    mass = np.full(len(q), 2.0)
    a = mo.mesh_accel(pos, mass, L, 32, G=1.5)
    rho = mass.sum() / L ** 3
    want = 4 * np.pi * 1.5 * rho * psi
    assert np.abs(a[:, 0] - want).max() < 2e-3 * np.abs(want).max()
    assert np.abs(a[:, 1:]).max() < 1e-9 * np.abs(want).max()


def test_cic_deposit_conserves_mass_and_gather_is_its_adjoint():
    rng = np.random.default_rng(1)
    L = 10.0
    pos = rng.uniform(-L, 2 * L, (300, 3)); m = rng.uniform(0.1, 1, 300)
    for shift in (0.0, 0.5):
        grid = mo.cic_deposit(pos, m, L, 12, shift)
        assert grid.sum() == pytest.approx(m.sum(), rel=1e-13)
        f = rng.normal(size=(12, 12, 12))
        assert (grid * f).sum() == pytest.approx((m * mo.cic_gather(f, pos, L, shift)).sum(), rel=1e-12)

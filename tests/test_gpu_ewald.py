"""GPU periodic (Ewald) forces and the comoving KDK step vs the NumPy oracle (oracle/ewald_oracle.py).
Tolerance: product parameters alpha L = 6.2, kmax = 7, minimum image only -> 3e-5 of the largest force vs the converged
oracle (27 images, kmax 13); vs the oracle run with the SAME parameters: 1e-9 (fp64, different summation order)."""
import numpy as np
import pytest

import bachelor_thesis_n_body_amd as nb
from bachelor_thesis_n_body_amd import cosmo, params
from oracle import ewald_oracle as eo

pytestmark = pytest.mark.gpu


def _accel(pos, mass, L, eps=0.0, G=1.0, alpha_L=6.2, kmax=7):
    with nb.Engine(len(mass), np.float64) as e:
        e.set_periodic(L, alpha_L, kmax)
        e.set_state(nb.State(pos, np.zeros_like(pos), mass))
        return e.accel(eps, G)


@pytest.mark.parametrize("n,L,eps", [(2, 100.0, 0.0), (37, 10.0, 0.0), (300, 50.0, 0.2), (1000, 500.0, 0.5)])
def test_ewald_accel_matches_oracle(n, L, eps):
    rng = np.random.default_rng(n)
    pos, m = rng.uniform(0, L, (n, 3)), rng.uniform(0.5, 2.0, n)
    a = _accel(pos, m, L, eps, G=2.0)
    same = eo.ewald_accel(pos, m, L, eps, G=2.0, alpha_L=6.2, kmax=7, images=0)
    assert np.abs(a - same).max() < 1e-9 * np.abs(same).max()
    conv = eo.ewald_accel(pos, m, L, eps, G=2.0, alpha_L=7.5, kmax=13, images=1)
    assert np.abs(a - conv).max() < 3e-5 * np.abs(conv).max()


def test_lattice_momentum_and_periodic_images():
    g = (np.arange(8) + 0.5) * 2.0
    pos = np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T
    # zero by symmetry; the product's truncation (minimum image only, alpha L = 6.2) leaves < 1e-4 of the nearest-neighbour force
    assert np.abs(_accel(pos, np.ones(512), 16.0)).max() < 1e-4 * (1.0 / 2.0 ** 2)
    assert np.abs(_accel(pos, np.ones(512), 16.0, alpha_L=7.5, kmax=12)).max() < 1e-5 * (1.0 / 2.0 ** 2)
    rng = np.random.default_rng(3)
    pos, m = rng.uniform(0, 16.0, (700, 3)), rng.uniform(0.5, 2.0, 700)
    a = _accel(pos, m, 16.0)
    assert np.abs((m[:, None] * a).sum(0)).max() < 1e-9 * (m[:, None] * np.abs(a)).sum(0).max()
    np.testing.assert_allclose(_accel(pos + np.array([16.0, -32.0, 48.0]), m, 16.0), a, rtol=1e-9, atol=1e-12)


def test_comoving_kdk_step_matches_oracle_and_wraps_positions():
    L, n = 50.0, 200
    rng = np.random.default_rng(5)
    pos, vel, m = rng.uniform(0, L, (n, 3)), rng.normal(0, 0.3, (n, 3)), np.full(n, 1.5)
    c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)
    a0, a1 = 0.5, 0.52
    am = np.sqrt(a0 * a1)
    k1, d, k2 = c.kick_factor(a0, am), c.drift_factor(a0, a1), c.kick_factor(am, a1)
    G, eps = params.G_NEWTON, 0.3
    with nb.Engine(n, np.float64) as e:
        e.set_periodic(L)
        e.set_state(nb.State(pos, vel, m))
        e.kdk_step(k1, d, k2, eps, G)
        st = e.get_state()
    acc = lambda p: eo.ewald_accel(p, m, L, eps, G, images=0)
    v = vel + acc(pos) * k1
    x = np.mod(pos + v * d, L)
    v = v + acc(x) * k2
    np.testing.assert_allclose(st.pos, x, rtol=0, atol=1e-10)
    np.testing.assert_allclose(st.vel, v, rtol=1e-9, atol=1e-12)
    assert st.pos.min() >= 0 and st.pos.max() < L


@pytest.mark.parametrize("n,L,eps", [(500, 50.0, 0.2), (4000, 500.0, 0.25)])
def test_ewald_fp32_kernel_matches_oracle(n, L, eps):
    """fp32 contexts: erfc by a 1.5e-7 rational approximation, fp32 sums flushed to fp64 — 2e-5 of the largest force
    (the Ewald truncation itself is 3e-5)."""
    rng = np.random.default_rng(n)
    pos, m = rng.uniform(0, L, (n, 3)).astype(np.float32), rng.uniform(0.5, 2.0, n).astype(np.float32)
    with nb.Engine(n, np.float32) as e:
        e.set_periodic(L)
        e.set_state(nb.State(pos, np.zeros_like(pos), m))
        a = e.accel(eps, 2.0)
    ref = eo.ewald_accel(pos.astype(np.float64), m.astype(np.float64), L, eps, G=2.0, alpha_L=6.2, kmax=7, images=0)
    assert a.dtype == np.float32
    assert np.abs(a - ref).max() < 2e-5 * np.abs(ref).max()
    assert np.median(np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)) < 2e-6


def test_linear_growth_of_a_zeldovich_field():
    """Physics check of the whole comoving loop: a low-amplitude Zel'dovich field evolved from a = 0.1 to 0.2 grows by
    D(0.2)/D(0.1) in amplitude (linear theory), measured on the lowest modes."""
    from bachelor_thesis_n_body_amd import field
    n_side, L = 16, 400.0
    c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12, radiation=False)
    k = np.logspace(-3, 1, 100)
    tab = np.column_stack([k, 1.0e3 * (k / 0.05) ** -1.0])             # linear P(k) at a = 1, small amplitude
    pos, vel = cosmo.zeldovich_ics(tab, n_side, L, c, 0.1, 11, 12)
    m = c.particle_mass(L, n_side ** 3)
    p1, _ = cosmo.run_pp(pos, vel, m, L, c, 0.1, 0.2, nsteps=40)
    def low_modes(p):
        d = np.fft.fftn(field.compute_delta(field.cic_density(p, L, n_side)))
        return np.array([d[1, 0, 0], d[0, 1, 0], d[0, 0, 1], d[1, 1, 0], d[1, 0, 1], d[0, 1, 1]])
    ratio = np.abs(low_modes(p1)) / np.abs(low_modes(pos))
    D0, _ = c.growth(0.1)
    D1, _ = c.growth(0.2)
    assert np.median(ratio) == pytest.approx(D1 / D0, rel=0.03)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("clustered", [False, True])
def test_ewald_cell_list_real_space_sum(clustered, dtype):
    """With alpha_L >= 9.9 the real-space part is summed from cell lists (cut-off at alpha r = 3.3) instead of over all
    minimum-image pairs: same forces as the converged oracle and as the all-pairs path, to the truncation / fp32 level."""
    rng = np.random.default_rng(11)
    n, L, eps = 5000, 80.0, 0.2
    pos = rng.uniform(0, L, (n, 3))
    if clustered:
        pos[: n // 2] = np.mod(40.0 + 2.0 * rng.normal(size=(n // 2, 3)), L)
    pos, m = pos.astype(dtype), rng.uniform(0.5, 2.0, n).astype(dtype)

    def run(aL, kmax):
        with nb.Engine(n, dtype) as e:
            e.set_periodic(L, aL, kmax)
            e.set_state(nb.State(pos, np.zeros_like(pos), m))
            return e.accel(eps, 2.0).astype(np.float64)
    cells, pairs = run(13.2, 15), run(6.2, 7)
    scale = np.sqrt((pairs ** 2).sum(1).mean())
    assert np.abs(cells - pairs).max() < 1e-4 * scale and np.sqrt(((cells - pairs) ** 2).sum(1).mean()) < 2e-5 * scale
    if dtype == np.float64 and not clustered:                            # and both against the converged CPU sum (27 images, kmax 13)
        k = 2500
        sub_pos, sub_m = pos[:k], m[:k]
        with nb.Engine(k, dtype) as e:
            e.set_periodic(L, 13.2, 15)
            e.set_state(nb.State(sub_pos, np.zeros_like(sub_pos), sub_m))
            got = e.accel(eps, 2.0)                                       # k < 4096: all-pairs path with the same splitting
        ref = eo.ewald_accel(sub_pos, sub_m, L, eps, G=2.0, alpha_L=7.5, kmax=13, images=1)
        assert np.abs(got - ref).max() < 3e-5 * np.abs(ref).max()
    with nb.Engine(n, np.float32) as e:                                  # kmax range of the API
        with pytest.raises(RuntimeError, match="kmax"):
            e.set_periodic(L, 13.2, 32)

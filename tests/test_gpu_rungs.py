"""nb_rung_step / cosmo.run_rungs — CONCEPT's time-loop structure: base steps from the reference's logs, the mesh
force once per base step, the short-range force through a rung hierarchy. Pins: (i) with a single level it IS the plain KDK of nb_kdk_step
over the same steps; (ii) the rung populations against the ones the reference's logs record; (iii) with the hierarchy the small-scale power
lost by CONCEPT's coarse base steps comes back."""
from pathlib import Path

import numpy as np
import pytest

from bachelor_thesis_n_body_amd import cosmo

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden" / "concept"


def _setup(n_side, seed):
    ref = np.loadtxt(GOLD / "PP_powerspec_a=1.00")
    k_ref, p_lin = ref[:, 0], ref[:, 4]
    slope = np.log(p_lin[-1] / p_lin[-3]) / np.log(k_ref[-1] / k_ref[-3])
    k_ext = np.geomspace(k_ref[-1] * 1.05, 1.0, 12)
    tab = np.column_stack([np.concatenate([k_ref, k_ext]), np.concatenate([p_lin, p_lin[-1] * (k_ext / k_ref[-1]) ** slope])])
    c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)
    pos, vel = cosmo.zeldovich_ics(tab, n_side, 500.0, c, 0.02, seed, seed + 50, lpt=1)
    return c, pos, vel


def test_one_level_is_the_plain_kdk_over_the_same_steps():
    c, pos, vel = _setup(16, 3)
    m = c.particle_mass(500.0, 16 ** 3)
    p1, v1 = cosmo.run_pp(pos, vel, m, 500.0, c, 0.02, 0.2, dtype=np.float64, gravity="p3m", ngrid=32, schedule="concept")
    p2, v2, pops = cosmo.run_rungs(pos, vel, m, 500.0, c, 0.02, 0.2, levels=1, dtype=np.float64, ngrid=32, return_populations=True)
    assert np.all(pops[:, 0] == 16 ** 3)
    d = np.abs(p1 - p2); d = np.minimum(d, 500.0 - d)
    # same scheme, two quadratures of the kick / drift integrals (65-point trapezoid per half step vs cumulative 9-point pieces): 1e-5 relative
    assert d.max() < 1e-3 and np.abs(v1 - v2).max() < 2e-4 * np.abs(v1).max()


def test_rung_populations_follow_the_reference_logs_and_restore_small_scale_power():
    """32^3 p3m farm job (NN.param) on CONCEPT's own base steps. The logs (concept_step_schedule.txt, column 7) record the mean fraction of particles above rung 0:
    0 until a ~ 0.25, 0.8 % at a = 0.46, 2.5 % at 0.57, 5.2 % at 0.70, 5.8 % at a = 1."""
    sched = np.loadtxt(GOLD / "concept_step_schedule.txt")
    frac_ref = sched[:, 6]
    c, pos, vel = _setup(32, 11)
    m = c.particle_mass(500.0, 32 ** 3)
    p_r, _, pops = cosmo.run_rungs(pos, vel, m, 500.0, c, 0.02, 1.0, levels=5, return_populations=True)
    assert pops.shape == (143, 5) and np.all(pops.sum(1) == 32 ** 3)
    frac = 1.0 - pops[:, 0] / 32 ** 3
    assert np.all(frac[:80] < 2e-3)                                              # everybody on rung 0 while the field is smooth
    # the logged curve (mean of 300 runs; one realisation here scatters by ~15 %): 0.8 % at step 112, 2.5 % at 120, 5.2 % at 128, 6.1 % at 136, 5.7 % at 141
    for k in (112, 120, 128, 136, 141):
        assert frac[k] == pytest.approx(frac_ref[k], rel=0.3), (k, frac[k], frac_ref[k])
    assert np.mean(frac[112:142] / frac_ref[112:142]) == pytest.approx(1.0, abs=0.15)
    # the split over the rungs late in the run: mean populations of the 300 logs (columns 9-13 of the fixture; e.g. step 138: 30 840 / 1 300 / 610 / 18)
    mean_pop = sched[130:142, 8:13].mean(0)
    ours = pops[130:142].mean(0)
    assert ours[1] == pytest.approx(mean_pop[1], rel=0.15) and ours[2] == pytest.approx(mean_pop[2], rel=0.3) and ours[3] < 4 * max(mean_pop[3], 10), (ours, mean_pop)
    assert np.all(np.diff(frac[96:128:8]) > 0)                                   # the higher rungs fill as structure forms
    # same base steps without the hierarchy vs with it vs the step-converged run: the hierarchy recovers the small-scale power
    p_0, _ = cosmo.run_pp(pos, vel, m, 500.0, c, 0.02, 1.0, dtype=np.float32, gravity="p3m", ngrid=32, schedule="concept")
    p_c, _ = cosmo.run_pp(pos, vel, m, 500.0, c, 0.02, 1.0, nsteps=600, dtype=np.float32, gravity="p3m", ngrid=32)
    top = lambda p: cosmo.power_modes(p, 500.0, 32, interlace=True)
    k, pk_c = top(p_c); _, pk_0 = top(p_0); _, pk_r = top(p_r)
    hi = k > 0.15
    r0, rr = pk_0[hi].mean() / pk_c[hi].mean(), pk_r[hi].mean() / pk_c[hi].mean()
    assert r0 < 0.985 and abs(rr - 1.0) < abs(r0 - 1.0) * 0.6, (r0, rr)


def test_rung_step_argument_checks_and_interplay_with_plain_steps():
    import bachelor_thesis_n_body_amd as nb
    rng = np.random.default_rng(0)
    n, L = 3000, 30.0
    s = nb.State(rng.uniform(0, L, (n, 3)).astype(np.float32), rng.normal(size=(n, 3)).astype(np.float32) * 0.1, np.full(n, 1.0, np.float32))
    kc, dc = np.linspace(0, 0.02, 5), np.linspace(0, 0.02, 5)
    with nb.Engine(n, np.float32) as e:
        e.set_state(s)
        with pytest.raises(RuntimeError, match="P3M"):
            e.rung_step(2, kc, dc, 0.02, 1e-4, 0.1)                 # direct solver: no short/long-range split
        e.set_mesh("p3m", L, 32)
        with pytest.raises(ValueError):
            e.rung_step(3, kc, dc, 0.02, 1e-4, 0.1)                 # tables of the wrong length for 3 levels
        pops = e.rung_step(2, kc, dc, 0.02, 1e-4, 0.1)
        assert sum(pops) == n
        a_after = e.accel(0.1)                                      # a plain force evaluation afterwards is the TOTAL force again
        st = e.get_state()
    with nb.Engine(n, np.float32) as e:
        e.set_mesh("p3m", L, 32)
        e.set_state(st)
        np.testing.assert_allclose(e.accel(0.1), a_after, rtol=0, atol=1e-5 * np.abs(a_after).max())
    # a huge criterion puts everybody on rung 0: the step is then one KDK step with these factors
    with nb.Engine(n, np.float32) as e:
        e.set_mesh("p3m", L, 32); e.set_state(s)
        assert e.rung_step(3, np.linspace(0, 0.02, 9), np.linspace(0, 0.02, 9), 0.02, 1e30, 0.1) == [n, 0, 0]
        r = e.get_state()
    with nb.Engine(n, np.float32) as e:
        e.set_mesh("p3m", L, 32); e.set_state(s)
        e.kdk_step(0.01, 0.02, 0.01, 0.1)
        k = e.get_state()
    d = np.abs(r.pos - k.pos); d = np.minimum(d, L - d)
    assert d.max() < 1e-4 and np.abs(r.vel - k.vel).max() < 1e-4 * np.abs(k.vel).max()

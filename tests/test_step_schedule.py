"""CONCEPT's time-step schedule and clock, as the reference's own artefacts record them:
tests/golden/concept/concept_step_schedule.txt is extracted from the 300 run logs under TheFeedingPipeline/Datafolder4 by
make_step_schedule.py (beside it). These CPU tests pin the host-side background integration (`Cosmology.cosmic_time`) and the
schedule reconstruction (`concept_time_edges`) against those numbers, and `sigma8_of_table` against the nine sigma_8 values in the
headers of the reference's PP / PM / P3M power-spectrum tables."""
import re
from pathlib import Path

import numpy as np
import pytest

from bachelor_thesis_n_body_amd import cosmo

GOLD = Path(__file__).resolve().parent / "golden" / "concept"
C = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)        # NN.param:87-89, Giant.param:75-80


def test_fixture_shape_and_internal_consistency():
    tab = np.loadtxt(GOLD / "concept_step_schedule.txt")
    step, a, t, dt = tab[:, 0], tab[:, 1], tab[:, 2], tab[:, 3]
    assert len(tab) == 143 and np.array_equal(step, np.arange(143)) and a[0] == 0.02 and a[-1] == pytest.approx(0.9999)
    # all 300 logs carry the same schedule: the base step is set by the background, not by the particles
    assert np.all(tab[:, 4] == tab[:, 5]) and np.all(tab[:, 7] == 300)
    # t_{n+1} = t_n + dt_n to the four digits the log prints
    assert np.all(np.abs(t[1:] - (t[:-1] + dt[:-1])) < 1e-3 * t[1:])
    # the step size is held for 8 base steps and then updated (until the late-time limiter takes over)
    for b in range(0, 128, 8):
        assert np.all(dt[b:b + 8] == dt[b]), b
    assert np.all(np.diff(dt[:128:8]) > 0)
    # every particle on rung 0 for the first ~90 steps, a few per cent on higher rungs at late times
    assert np.all(tab[:88, 6] == 0) and 0.03 < tab[-1, 6] < 0.08


def test_package_copy_of_the_schedule_is_the_fixture():
    pkg = Path(cosmo.__file__).resolve().parent / "data" / "concept_step_schedule.txt"
    assert pkg.read_bytes() == (GOLD / "concept_step_schedule.txt").read_bytes()


def test_cosmic_clock_matches_concepts_logs_and_table_header():
    a, t, _ = cosmo.load_concept_schedule()
    ours = np.array([C.cosmic_time(x) for x in a[::6]])
    np.testing.assert_allclose(ours, t[::6], rtol=5e-4)           # 4-5 printed digits
    # PPPMP3M/PP/powerspec/powerspec_a=1.00:1 — "at t = 13.814035 Gyr, a = 1"
    hdr = (GOLD / "PP_powerspec_a=1.00").read_text(encoding="utf8").splitlines()[0]
    t1 = float(re.search(r"t = ([0-9.]+) Gyr", hdr).group(1))
    assert C.cosmic_time(1.0) == pytest.approx(t1, rel=3e-7)
    assert C.scale_factor(C.cosmic_time(0.3)) == pytest.approx(0.3, rel=1e-10)


def test_schedule_reconstruction_reproduces_the_logged_scale_factors():
    a, t, dt = cosmo.load_concept_schedule()
    edges_t = cosmo.concept_time_edges(C, 0.02, 1.0, dt)
    assert len(edges_t) == 144                                     # 143 steps, the last one cut at a = 1
    a_edges = np.array([C.scale_factor(x) for x in edges_t])
    np.testing.assert_allclose(a_edges[:-1], a, rtol=6e-4)          # the log prints 4 digits of dt: 4e-4 accumulated
    assert a_edges[-1] == pytest.approx(1.0, abs=1e-7)                  # the schedule code interpolates a 40 000-point clock table (~1e-8)
    # a run that stops early takes the same steps and cuts the last one
    e2 = cosmo.concept_time_edges(C, 0.02, 0.5, dt)
    assert np.all(e2[:-1] == edges_t[:len(e2) - 1]) and C.scale_factor(e2[-1]) == pytest.approx(0.5, rel=1e-7)


@pytest.mark.parametrize("name", ["PP", "PM", "P3M"])
def test_sigma8_of_the_reference_tables(name):
    path = GOLD / f"{name}_powerspec_a=1.00"
    hdr = [l for l in path.read_text(encoding="utf8").splitlines() if "σ₈" in l][0]
    printed = [float(v) for v in re.findall(r"σ₈ = ([0-9.e+-]+)", hdr)]
    tab = np.loadtxt(path)
    for col, want in zip((2, 3, 4), printed):
        got = cosmo.sigma8_of_table(tab[:, 0], tab[:, col], 0.6736)
        assert abs(got - want) < 1.6e-3, (name, col, got, want)       # three printed digits, +- 1 in the last


def test_corrected_rebinning_removes_a_known_noise_modulation():
    rng = np.random.default_rng(0)
    k = np.sort(rng.uniform(0.01, 0.4, 4000))
    plin = 1e4 * (k / 0.02) ** -1.5
    r2 = rng.exponential(1.0, k.size)                               # |R_k|^2 of a Gaussian field
    p = 1.7 * plin * r2                                             # "evolved" spectrum = constant boost x realisation
    counts = [10, 40, 150, 400, 1400]
    kb, pb, pc = cosmo.rebin_corrected_like_concept(k, p, plin, r2, counts)
    kr, pr = cosmo.rebin_like_concept(k, p, counts)
    np.testing.assert_allclose(kb, kr); np.testing.assert_allclose(pb, pr)
    start = 0
    for i, c in enumerate(counts):
        np.testing.assert_allclose(pc[i], 1.7 * plin[start:start + 2 * c].mean(), rtol=1e-12)
        start += 2 * c

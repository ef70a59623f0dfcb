"""Oracle vs the committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py)."""
from pathlib import Path

import numpy as np
import pytest

from oracle import nbody_oracle as ora

G = Path(__file__).resolve().parent / "golden"
CASES = ["n2", "n3", "n64", "n257", "n1024"]


@pytest.fixture(scope="module")
def accel_golden():
    return np.load(G / "accel_golden.npz")


@pytest.mark.parametrize("name", CASES)
def test_numpy_oracle_matches_golden(accel_golden, name):
    g = accel_golden
    a = ora.compute_forces(g[f"{name}_pos"], g[f"{name}_mass"], float(g[f"{name}_eps"]), float(g[f"{name}_G"]))
    np.testing.assert_allclose(a, g[f"{name}_acc"], rtol=1e-13, atol=1e-15)
    assert ora.potential_energy(g[f"{name}_pos"], g[f"{name}_mass"], float(g[f"{name}_eps"]), float(g[f"{name}_G"])) == \
        pytest.approx(float(g[f"{name}_pot"]), rel=1e-13)
    assert ora.kinetic_energy(g[f"{name}_vel"], g[f"{name}_mass"]) == pytest.approx(float(g[f"{name}_kin"]), rel=1e-13)


@pytest.mark.parametrize("name", CASES)
def test_c_oracle_matches_golden(accel_golden, coracle, name):
    g = accel_golden
    for seq in (True, False):
        a = coracle.accel(g[f"{name}_pos"], g[f"{name}_mass"], float(g[f"{name}_eps"]), float(g[f"{name}_G"]), sequential=seq)
        np.testing.assert_allclose(a, g[f"{name}_acc"], rtol=1e-11, atol=1e-13)


def test_kdk_golden_trajectory():
    g = np.load(G / "kdk_golden.npz")
    s = ora.State(g["pos0"].copy(), g["vel0"].copy(), g["mass"].copy())
    ora.step(s, float(g["dt"]), 8, eps=float(g["eps"]))
    np.testing.assert_allclose(s.pos, g["pos8"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(s.vel, g["vel8"], rtol=1e-12, atol=1e-14)
    ora.step(s, float(g["dt"]), 24, eps=float(g["eps"]))
    np.testing.assert_allclose(s.pos, g["pos32"], rtol=1e-11, atol=1e-13)
    k, u = ora.energy(s, float(g["eps"]))
    assert k == pytest.approx(float(g["kin32"]), rel=1e-11) and u == pytest.approx(float(g["pot32"]), rel=1e-11)


# ---- the size class that takes the pair-symmetric kernels: sampled sinks of N = 40000 / 100001 (tests/golden/make_golden_large.py) ----------
LARGE = ["n40000_f32", "n40000_f64_general", "n100001_f32_general", "n100001_f64"]


def large_inputs(g, name):
    """Rebuild the inputs of a large golden case from (n, seed) and check them against the stored checksums."""
    import sys
    sys.path.insert(0, str(G))
    from make_golden_large import inputs
    dtype = np.float32 if "_f32" in name else np.float64
    pos, mass = inputs(int(g[f"{name}_n"]), int(g[f"{name}_seed"]), bool(g[f"{name}_unequal"]), dtype)
    np.testing.assert_allclose(pos.astype(np.float64).sum(0), g[f"{name}_sum_pos"], rtol=0, atol=1e-9)
    assert float(mass.astype(np.float64).sum()) == pytest.approx(float(g[f"{name}_sum_mass"]), rel=1e-13)
    return pos, mass


@pytest.mark.parametrize("name", LARGE)
def test_c_oracle_matches_large_golden(coracle, name):
    g = np.load(G / "accel_golden_large.npz")
    pos, mass = large_inputs(g, name)
    p64 = pos.astype(np.float64)
    a = coracle.accel(p64, mass.astype(np.float64), float(g[f"{name}_eps"]), float(g[f"{name}_G"]), sinks=p64[g[f"{name}_idx"]])
    np.testing.assert_allclose(a, g[f"{name}_acc"], rtol=1e-11, atol=1e-13)

"""The energy kernels AT THE SIZES WHERE THE LARGE-N VARIANTS RUN, directly against the C oracle (oracle/nbody_oracle.c) on the same inputs:

  nb_energy       n >= 8192 takes `potential_self_kernel` (csrc/potential.hip: packed self-potential, every unordered pair once),
  nb_step_energy  n >= 18432 (fp32) / 10240 (fp64) takes the POT variants of the pair-symmetric force kernels (accel_sym_*.hip),
  nb_accel_pair_dev with phi: the POT variants of the two-block form,

including eps = 0 with exactly coincident particles (the GUARD instantiations, which drop a pair with a zero denominator — the rule the
oracle's force and potential sums share). Below those sizes tests/test_gpu_parity.py and test_gpu_fused_potential.py cover the small-n paths.

Tolerances (relative error of the total potential energy U / of a per-sink potential phi_i against the fp64 oracle on the same rounded
inputs), within 5-7x of the measured errors (profiles/r04_parity_measured_errors.txt): U fp32 5e-8 (measured <= 8.3e-9: fp32 pair terms, partial
sums flushed to fp64 per tile, and the errors of 10^9 terms average out), fp64 5e-15 (measured <= 7.8e-16); phi_i fp32 5e-7 (measured <= 1.05e-7),
fp64 2e-13 (measured <= 3.6e-14)."""
import numpy as np
import pytest

import bachelor_thesis_n_body_amd as nb
from conftest import note

pytestmark = pytest.mark.gpu
DTYPES = [np.float32, np.float64]
UTOL = {np.float32: 5e-8, np.float64: 5e-15}
PTOL = {np.float32: 5e-7, np.float64: 2e-13}


def _cloud(n, dtype, general, seed=31):
    s = nb.plummer_sphere(n, seed, dtype)
    if general:
        rng = np.random.default_rng(seed + 1)
        s = nb.State(s.pos, s.vel, (s.mass * rng.uniform(0.25, 4.0, n)).astype(dtype))
    return s


def _f64(s):
    return s.pos.astype(np.float64), s.vel.astype(np.float64), s.mass.astype(np.float64)


def _coincide(s, n):
    """Exactly coincident particles: inside one set of the kernels, in different sets, next to the ragged end, and a triple."""
    pos = s.pos.copy()
    pos[17] = pos[3]
    pos[n // 2 + 5] = pos[40]
    pos[n - 1] = pos[n - 2]
    pos[900] = pos[n - 7]
    pos[901] = pos[n - 7]
    return nb.State(pos, s.vel, s.mass)


# 262144: one case per dtype (the oracle's O(N^2) fp64 sum takes ~10 s there)
CASES = [(8192, False), (8192, True), (12001, True), (40000, False), (40000, True), (100003, False)]
BIG = [(np.float32, 262144, False), (np.float64, 262144, True)]


@pytest.mark.parametrize("dtype,n,general", [(d, n, g) for d in DTYPES for n, g in CASES] + BIG)
def test_nb_energy_matches_oracle_at_large_n(dtype, n, general, coracle):
    s = _cloud(n, dtype, general)
    p, v, m = _f64(s)
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        k, u = e.energy(0.01, 1.7)
    assert k == pytest.approx(coracle.kinetic(v, m), rel=1e-6 if dtype == np.float32 else 1e-13)
    ref = coracle.potential(p, m, 0.01, 1.7)
    note(f"nb_energy U n={n} {np.dtype(dtype).name} {'general' if general else 'uniform'}", abs(u / ref - 1), UTOL[dtype])
    assert abs(u / ref - 1) < UTOL[dtype], f"U {u!r} oracle {ref!r} rel {abs(u / ref - 1):.3e}"


@pytest.mark.parametrize("dtype,n,general", [(d, n, g) for d in DTYPES for n, g in CASES] + BIG)
def test_nb_step_energy_matches_oracle_at_large_n(dtype, n, general, coracle):
    """One fused step: K and U come out of the force evaluation at the NEW positions; the oracle evaluates them at the state the engine
    hands back (same rounded numbers)."""
    s = _cloud(n, dtype, general, seed=32)
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        K, U = e.step_with_energy(1.0 / 256, 1, 0.01)
        sym = e.last_accel_symmetric()
        st = e.get_state()
    assert sym == (n >= (18432 if dtype == np.float32 else 10240))
    p, v, m = _f64(st)
    assert K[0] == pytest.approx(coracle.kinetic(v, m), rel=1e-6 if dtype == np.float32 else 1e-13)
    ref = coracle.potential(p, m, 0.01)
    note(f"nb_step_energy U n={n} {np.dtype(dtype).name} {'general' if general else 'uniform'}", abs(U[0] / ref - 1), UTOL[dtype])
    assert abs(U[0] / ref - 1) < UTOL[dtype], f"U {U[0]!r} oracle {ref!r} rel {abs(U[0] / ref - 1):.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("n", [8192, 20000, 40000, 70001])
def test_zero_softening_with_coincident_particles_at_large_n(dtype, general, n, coracle):
    """eps = 0: the GUARD instantiations of potential_self_kernel (nb_energy) and of the POT pair-symmetric kernels (nb_step_energy, dt = 0
    so that the state — and its coincidences — stay put). Finite, and equal to the oracle's sums with the coincident pairs dropped."""
    s = _coincide(_cloud(n, dtype, general, seed=33), n)
    p, v, m = _f64(s)
    ref = coracle.potential(p, m, 0.0)
    assert np.isfinite(ref)
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        k, u = e.energy(0.0)
        K, U = e.step_with_energy(0.0, 1, 0.0)
        a = e.cached_acc()
        sym = e.last_accel_symmetric()
    assert sym == (n >= (18432 if dtype == np.float32 else 10240))
    assert np.isfinite(u) and np.isfinite(U[0]) and np.isfinite(a).all()
    tol = UTOL[dtype]
    note(f"eps=0 nb_energy U n={n} {np.dtype(dtype).name} {'general' if general else 'uniform'}", abs(u / ref - 1), tol)
    note(f"eps=0 nb_step_energy U n={n} {np.dtype(dtype).name} {'general' if general else 'uniform'}", abs(U[0] / ref - 1), tol)
    assert abs(u / ref - 1) < tol, f"nb_energy: rel {abs(u / ref - 1):.3e}"
    assert abs(U[0] / ref - 1) < tol, f"nb_step_energy: rel {abs(U[0] / ref - 1):.3e}"
    # the coincident particles feel the same force (to rounding: they sit in different lanes / sets, i.e. other summation orders)
    ftol = 1e-6 if dtype == np.float32 else 1e-13
    af = a.astype(np.float64)
    assert np.linalg.norm(af[17] - af[3]) < ftol * np.linalg.norm(af[3]) and np.linalg.norm(af[900] - af[901]) < ftol * np.linalg.norm(af[901])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("na,nb_,eps", [(40000, 33001, 0.01), (20000, 50000, 0.0), (131072, 131072, 0.01)])
def test_two_block_fused_potential_matches_oracle(dtype, general, na, nb_, eps, coracle):
    """nb_accel_pair_dev with phi at sizes that take the POT two-block kernels unforced; eps = 0: a particle of B coincides with one of A."""
    import torch
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    s = _cloud(na + nb_, dtype, general, seed=34)
    pos = s.pos.copy()
    if eps == 0.0:
        pos[na + 11] = pos[7]
        pos[na + nb_ - 1] = pos[na - 1]
    if not general:
        s = nb.State(pos, s.vel, np.full(na + nb_, 0.41 / (na + nb_), dtype))
    else:
        s = nb.State(pos, s.vel, s.mass)
    k = nb.HipKernels(0)
    blk = lambda lo, hi: torch.from_numpy(np.vstack([s.pos[lo:hi].T, s.mass[None, lo:hi]])).contiguous().cuda()
    A, B = blk(0, na), blk(na, na + nb_)
    acc_a, acc_b = torch.zeros((3, na), dtype=tdt, device="cuda"), torch.zeros((3, nb_), dtype=tdt, device="cuda")
    phi_a = torch.zeros(na, dtype=torch.float64, device="cuda")
    phi_b = torch.full((nb_,), 9.0, dtype=torch.float64, device="cuda")
    k.accel_pair(A, na, B, nb_, eps, 1.0, acc_a, acc_b, False, uniform_mass=not general, phi_a=phi_a, phi_b=phi_b)
    assert k.last_accel_symmetric()
    torch.cuda.synchronize()
    p64, m64 = s.pos.astype(np.float64), s.mass.astype(np.float64)
    rng = np.random.default_rng(na)
    ia = np.unique(np.concatenate([rng.integers(0, na, 200), [0, 7, na - 1]]))
    ib = np.unique(np.concatenate([rng.integers(0, nb_, 200), [0, 11, nb_ - 1]]))
    ra = coracle.phi(p64[na:], m64[na:], eps, p64[:na][ia])
    rb = coracle.phi(p64[:na], m64[:na], eps, p64[na:][ib])
    ga, gb = phi_a.cpu().numpy()[ia], phi_b.cpu().numpy()[ib]
    assert np.isfinite(ga).all() and np.isfinite(gb).all()
    tol = PTOL[dtype]
    note(f"two-block phi na={na} nb={nb_} eps={eps} {np.dtype(dtype).name} {'general' if general else 'uniform'}", max(np.abs(ga / ra - 1).max(), np.abs(gb / rb - 1).max()), tol)
    assert np.abs(ga / ra - 1).max() < tol and np.abs(gb / rb - 1).max() < tol, (np.abs(ga / ra - 1).max(), np.abs(gb / rb - 1).max())
    k.close()

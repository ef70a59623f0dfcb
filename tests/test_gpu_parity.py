"""GPU parity tests proper: the HIP path, called through the C-ABI (ctypes -> libnbody_amd.so), against the CPU
oracle on the same seeded inputs, against the committed golden vectors, and — at BASELINE.json's full sizes —
through size-independent properties (Newton's third law, permutation / translation invariance, linearity in
mass and G, time reversibility, source-split / chunk-accumulate consistency).

Tolerances (written here, SURVEY.md §8d "Parity gates"):
  fp64 kernel vs fp64 oracle : per-particle |da|/|a| <= 2e-13   (north_star bar: 1e-6; measured 1e-14 .. 2.3e-14 — the first-order
  rsqrt refinement of accel_f64.hip; 1e-13 is its documented contract)
  fp32 kernel vs fp64 oracle on the SAME fp32-rounded inputs : <= 1e-5, and <= 1e-6 from n = 32768 on (measured up to 2.3e-6 at
  n = 1500 .. 16384, 8e-8 .. 2e-7 at large n thanks to the per-tile fp64 flush of the fp32 partial sums; plain fp32 accumulation would be
  ~1e-4 at these N) — within 5x of the measured errors, so losing the flush fails
"""
from pathlib import Path

import numpy as np
import pytest

import bachelor_thesis_n_body_amd as nb
from oracle import nbody_oracle as ora
from conftest import rel_err

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
TOL = {np.float32: 1e-5, np.float64: 2e-13}
TOL_LARGE = {np.float32: 1e-6, np.float64: 2e-13}      # n >= 32768
DTYPES = [np.float32, np.float64]


def _loaded_native():
    """Fail loudly if the HIP extension is not the thing being exercised."""
    from bachelor_thesis_n_body_amd import _lib
    lib = _lib.load()
    assert lib.nb_device_count() >= 1
    assert _lib.LIB_PATH.exists()


def _ref(s, eps, G=1.0, coracle=None):
    p, m = s.pos.astype(np.float64), s.mass.astype(np.float64)
    return coracle.accel(p, m, eps, G) if coracle is not None else ora.compute_forces(p, m, eps, G)


def test_native_library_is_loaded():
    _loaded_native()


# ------------------------------------------------------------------ accelerations vs oracle
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 257, 1000, 1024, 4097])
def test_accel_matches_oracle_small(dtype, n):
    s = nb.plummer_sphere(n, 100 + n, dtype)
    a = nb.compute_forces(s.pos, s.mass, 0.01)
    assert a.dtype == dtype and a.shape == (n, 3)
    if n == 1:
        assert np.all(a == 0)
        return
    assert rel_err(a, _ref(s, 0.01)).max() < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [20000, 65536])
def test_accel_matches_oracle_config1(dtype, n, coracle):
    """configs[1]: N = 65536 (C oracle, OpenMP: ~seconds)."""
    s = nb.plummer_sphere(n, 1, dtype)
    a = nb.compute_forces(s.pos, s.mass, 0.01)
    ref = _ref(s, 0.01, coracle=coracle)
    e = rel_err(a, ref)
    if not e.max() < (TOL_LARGE if n >= 32768 else TOL)[dtype]:
        # say WHICH side is off: a second GPU evaluation and an independent NumPy-oracle subset
        a2 = nb.compute_forces(s.pos, s.mass, 0.01)
        idx = np.arange(0, n, max(1, n // 64))
        sub = ora.compute_forces(s.pos.astype(np.float64), s.mass.astype(np.float64), 0.01, sinks=s.pos[idx].astype(np.float64))
        pytest.fail(f"max {e.max():.3e} median {np.median(e):.3e}; gpu run1 vs run2 {rel_err(a2, a).max():.3e}; "
                    f"gpu vs numpy-oracle subset {rel_err(a[idx], sub).max():.3e}; C-oracle vs numpy-oracle subset "
                    f"{rel_err(ref[idx], sub).max():.3e}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_accel_unequal_masses_G_and_eps(dtype):
    rng = np.random.default_rng(3)
    s = nb.plummer_sphere(3000, 7, dtype)
    s.mass = (s.mass * rng.uniform(0.1, 5.0, s.n)).astype(dtype)
    s.mass[::7] = 0            # massless tracers feel forces but exert none
    for eps, G in ((0.0, 1.0), (1e-3, 4.30091e-3), (0.3, 6.674e-11), (50.0, 1.0)):
        a = nb.compute_forces(s.pos, s.mass, eps, G)
        assert rel_err(a, _ref(s, eps, G)).max() < TOL[dtype], (eps, G)


@pytest.mark.parametrize("dtype", DTYPES)
def test_coincident_particles_and_zero_softening(dtype):
    """eps = 0 with exactly coincident particles: the pair contributes nothing, everything stays finite."""
    s = nb.plummer_sphere(600, 9, dtype)
    s.pos[10] = s.pos[500]
    s.pos[11] = s.pos[500]
    s.pos[0] = 0
    a = nb.compute_forces(s.pos, s.mass, 0.0)
    assert np.all(np.isfinite(a))
    assert rel_err(a, _ref(s, 0.0)).max() < TOL[dtype]
    np.testing.assert_array_equal(a[10], a[11])


@pytest.mark.parametrize("name", ["n2", "n3", "n64", "n257", "n1024"])
def test_accel_matches_golden_vectors(name):
    g = np.load(GOLD / "accel_golden.npz")
    pos, mass, eps, G = g[f"{name}_pos"], g[f"{name}_mass"], float(g[f"{name}_eps"]), float(g[f"{name}_G"])
    a64 = nb.compute_forces(pos, mass, eps, G)
    assert rel_err(a64, g[f"{name}_acc"]).max() < 1e-10
    a32 = nb.compute_forces(pos.astype(np.float32), mass.astype(np.float32), eps, G)
    ref32 = ora.compute_forces(pos.astype(np.float32).astype(np.float64), mass.astype(np.float32).astype(np.float64), eps, G)
    assert rel_err(a32, ref32).max() < TOL[np.float32]
    with nb.Engine(len(mass), np.float64) as e:
        e.set_state(nb.State(pos, g[f"{name}_vel"], mass))
        k, u = e.energy(eps, G)
    assert k == pytest.approx(float(g[f"{name}_kin"]), rel=1e-12)
    assert u == pytest.approx(float(g[f"{name}_pot"]), rel=1e-11)


# ------------------------------------------------------------------ integrator and energies
@pytest.mark.parametrize("dtype", DTYPES)
def test_kdk_steps_match_oracle(dtype):
    s = nb.plummer_sphere(512, 11, dtype)
    o = ora.State(s.pos.astype(np.float64), s.vel.astype(np.float64), s.mass.astype(np.float64))
    ora.step(o, 1 / 256, 10, eps=0.01)
    g = nb.step(s.copy(), 1 / 256, 10, eps=0.01)
    tol = 1e-11 if dtype == np.float64 else 3e-5
    assert np.abs(g.pos - o.pos).max() < tol and np.abs(g.vel - o.vel).max() < tol
    assert rel_err(g.acc, o.acc).max() < TOL[dtype] * 10


def test_kdk_golden_trajectory_fp64():
    g = np.load(GOLD / "kdk_golden.npz")
    with nb.Engine(len(g["mass"]), np.float64) as e:
        e.set_state(nb.State(g["pos0"], g["vel0"], g["mass"]))
        e.step(float(g["dt"]), 8, float(g["eps"]))
        st = e.get_state()
        np.testing.assert_allclose(st.pos, g["pos8"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(st.vel, g["vel8"], rtol=0, atol=1e-12)
        e.step(float(g["dt"]), 24, float(g["eps"]))   # continues from the cached acceleration
        st = e.get_state()
        np.testing.assert_allclose(st.pos, g["pos32"], rtol=0, atol=1e-11)
        np.testing.assert_allclose(e.cached_acc(), g["acc32"], rtol=1e-9, atol=1e-11)
        k, u = e.energy(float(g["eps"]))
    assert k == pytest.approx(float(g["kin32"]), rel=1e-10) and u == pytest.approx(float(g["pot32"]), rel=1e-10)


def test_one_step_equals_n_single_steps_fp64():
    """step(n) fuses closing+opening half-kicks into one kernel; arithmetic must equal n separate steps."""
    s = nb.plummer_sphere(300, 12)
    a = nb.step(s.copy(), 1 / 128, 6, eps=0.02)
    b = s.copy()
    for _ in range(6):
        nb.step(b, 1 / 128, 1, eps=0.02)
    np.testing.assert_array_equal(a.pos, b.pos)
    np.testing.assert_array_equal(a.vel, b.vel)


@pytest.mark.parametrize("dtype", DTYPES)
def test_time_reversibility(dtype):
    s0 = nb.plummer_sphere(256, 13, dtype)
    with nb.Engine(s0.n, dtype) as e:
        e.set_state(s0)
        e.step(1 / 128, 30, 0.05)
        st = e.get_state()
        st.vel = -st.vel
        e.set_state(st)
        e.step(1 / 128, 30, 0.05)
        st = e.get_state()
    tol = 1e-11 if dtype == np.float64 else 2e-4
    assert np.abs(st.pos - s0.pos).max() < tol and np.abs(-st.vel - s0.vel).max() < tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_energy_matches_oracle_and_drift_tracks(dtype, coracle):
    s = nb.plummer_sphere(1024, 1, dtype)
    o = ora.State(s.pos.astype(np.float64), s.vel.astype(np.float64), s.mass.astype(np.float64))
    eps, dt = 0.01, 1 / 256
    with nb.Engine(s.n, dtype) as e:
        e.set_state(s)
        k, u = e.energy(eps)
        assert k == pytest.approx(coracle.kinetic(o.vel, o.mass), rel=1e-6 if dtype == np.float32 else 1e-13)
        assert u == pytest.approx(coracle.potential(o.pos, o.mass, eps), rel=2e-6 if dtype == np.float32 else 1e-12)
        e0 = k + u
        e.step(dt, 200, eps)
        k1, u1 = e.energy(eps)
    ora.step(o, dt, 200, eps=eps)
    ko, uo = ora.energy(o, eps)
    drift_gpu, drift_ref = (k1 + u1 - e0) / abs(e0), (ko + uo - (coracle.kinetic(s.vel.astype(np.float64), o.mass) + coracle.potential(s.pos.astype(np.float64), o.mass, eps))) / abs(e0)
    assert abs(drift_gpu) < 1e-3
    if dtype == np.float64:   # energy drift tracks the oracle's curve (within 10 %: SURVEY.md §8d)
        assert drift_gpu == pytest.approx(drift_ref, rel=0.1, abs=1e-12)


# ------------------------------------------------------------------ device-pointer surface (torch tensors)
@pytest.mark.parametrize("dtype", DTYPES)
def test_device_pointer_api_chunks_accumulate(dtype):
    """nb_accel_dev on torch memory: sinks != sources, ragged chunk sizes, accumulate over chunks = one shot."""
    import torch
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    s = nb.plummer_sphere(5000, 17, dtype)
    k = nb.HipKernels(0)
    pm = torch.from_numpy(np.vstack([s.pos.T, s.mass[None, :]])).contiguous().cuda()
    snk = pm[:, :1777].contiguous()
    acc = torch.full((3, 1777), 7.0, dtype=tdt, device="cuda")
    bounds = [0, 1, 513, 2048, 4999, 5000]
    for i, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
        k.accel(snk, 1777, pm[:, a:b].contiguous(), b - a, 0.01, 1.0, acc, i > 0)
    one = torch.empty_like(acc)
    k.accel(snk, 1777, pm, 5000, 0.01, 1.0, one, False)
    torch.cuda.synchronize()
    ref = ora.compute_forces(s.pos.astype(np.float64), s.mass.astype(np.float64), 0.01, sinks=s.pos[:1777].astype(np.float64))
    assert rel_err(one.cpu().numpy().T, ref).max() < TOL[dtype]
    assert rel_err(acc.cpu().numpy().T, ref).max() < TOL[dtype] * 2
    # empty source set: zeros (or untouched when accumulating)
    k.accel(snk, 1777, pm, 0, 0.01, 1.0, one, False)
    torch.cuda.synchronize()
    assert float(one.abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_sharded_engine_world1_matches_engine(dtype):
    s = nb.plummer_sphere(3000, 19, dtype)
    eng = nb.ShardedEngine(s.n, dtype)
    eng.set_state(s)
    eng.step(1 / 256, 3, 0.01)
    k1, u1 = eng.energy(0.01)
    st = eng.gather_state()
    with nb.Engine(s.n, dtype) as e:
        e.set_state(s)
        e.step(1 / 256, 3, 0.01)
        ref = e.get_state()
        k0, u0 = e.energy(0.01)
    np.testing.assert_array_equal(st.pos, ref.pos)
    np.testing.assert_array_equal(st.vel, ref.vel)
    assert k1 == pytest.approx(k0, rel=1e-12) and u1 == pytest.approx(u0, rel=1e-12)


# ------------------------------------------------------------------ full-size properties (BASELINE configs)
@pytest.mark.parametrize("dtype,n", [(np.float32, 65536), (np.float32, 1 << 20), (np.float64, 1 << 18)])
def test_full_size_properties(dtype, n):
    """At sizes the oracle cannot reach in seconds: Newton's third law, linearity in G and mass scale, translation
    invariance, and agreement of a sampled subset of sinks with the oracle."""
    s = nb.plummer_sphere(n, 0, dtype)
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        a = e.accel(0.01)
        assert np.all(np.isfinite(a))
        m64, a64 = s.mass.astype(np.float64), a.astype(np.float64)
        net = np.abs((m64[:, None] * a64).sum(0)).max()
        scale = (m64[:, None] * np.abs(a64)).sum(0).max()
        assert net < (3e-7 if dtype == np.float32 else 1e-12) * scale
        a2 = e.accel(0.01, G=2.0)
        np.testing.assert_allclose(a2, 2 * a, rtol=2e-7 if dtype == np.float32 else 1e-15, atol=0)
    # sampled sinks against the oracle (all N sources)
    idx = np.random.default_rng(0).choice(n, 64, replace=False)
    ref = ora.compute_forces(s.pos.astype(np.float64), m64, 0.01, sinks=s.pos[idx].astype(np.float64))
    assert rel_err(a[idx], ref).max() < TOL_LARGE[dtype]
    # permutation invariance: reversed particle order gives the reversed result (another summation order: twice the rounding)
    ar = nb.compute_forces(s.pos[::-1].copy(), s.mass[::-1].copy(), 0.01)[::-1]
    assert rel_err(ar, a).max() < (2e-6 if dtype == np.float32 else 1e-11)


# ------------------------------------------------------------------ hipGraph replay of the step loop
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [2000, 40000])
def test_graph_replayed_steps_equal_eager_steps(dtype, n, monkeypatch):
    """nb_step replays a captured [kick_drift, accel] graph for steps 1..n-1 when N <= 65536 and nsteps >= 4;
    the state must be bit-identical to stepping one at a time (eager launches), also across calls that
    interleave an energy evaluation (which may reallocate the scratch buffer the graph points into).
    n = 40000: the captured force evaluation is the pair-symmetric kernel + its slab reduction."""
    s = nb.plummer_sphere(n, 23, dtype)
    with nb.Engine(s.n, dtype) as e:
        e.set_state(s)
        e.step(1 / 256, 7, 0.01)            # eager step 0 + 6 graph replays
        e.energy(0.01)
        e.step(1 / 256, 5, 0.01)            # reuses (or rebuilds) the graph
        g = e.get_state()
    with nb.Engine(s.n, dtype) as e:
        e.set_state(s)
        for _ in range(12):
            e.step(1 / 256, 1, 0.01)        # nsteps < 4: always eager
        ref = e.get_state()
    np.testing.assert_array_equal(g.pos, ref.pos)
    np.testing.assert_array_equal(g.vel, ref.vel)


# ------------------------------------------------------------------ uniform-mass fast path
def test_uniform_mass_flag_matches_general_path():
    """NB_UNIFORM_MASS drops the per-source mass multiply; with equal masses it must agree with the general kernel
    (and the oracle), for ragged source counts (padding sources sit at 1e18), accumulation and eps = 0."""
    import torch
    s = nb.plummer_sphere(4321, 29, np.float32)
    k = nb.HipKernels(0)
    pm = torch.from_numpy(np.vstack([s.pos.T, s.mass[None, :]])).contiguous().cuda()
    for eps in (0.01, 0.0):
        gen = torch.empty(3, s.n, device="cuda")
        uni = torch.empty(3, s.n, device="cuda")
        k.accel(pm, s.n, pm, s.n, eps, 2.5, gen, False, False)
        k.accel(pm, s.n, pm[:, :1000].contiguous(), 1000, eps, 2.5, uni, False, True)
        k.accel(pm, s.n, pm[:, 1000:].contiguous(), s.n - 1000, eps, 2.5, uni, True, True)
        torch.cuda.synchronize()
        ref = ora.compute_forces(s.pos.astype(np.float64), s.mass.astype(np.float64), eps, 2.5)
        assert rel_err(uni.cpu().numpy().T, ref).max() < TOL[np.float32]
        assert rel_err(gen.cpu().numpy().T, ref).max() < TOL[np.float32]


def test_engine_detects_uniform_and_nonuniform_masses():
    s = nb.plummer_sphere(3000, 31, np.float32)
    a_uni = nb.compute_forces(s.pos, s.mass, 0.01)                    # equal masses -> uniform kernel
    m2 = s.mass.copy(); m2[1234] *= 1.0000001                         # one different mass -> general kernel
    a_gen = nb.compute_forces(s.pos, m2, 0.01)
    assert rel_err(a_uni, ora.compute_forces(s.pos.astype(np.float64), s.mass.astype(np.float64), 0.01)).max() < TOL[np.float32]
    assert rel_err(a_gen, ora.compute_forces(s.pos.astype(np.float64), m2.astype(np.float64), 0.01)).max() < TOL[np.float32]


# ------------------------------------------------------------------ energy conservation at scale (configs[3] in small)
@pytest.mark.parametrize("dtype,tol", [(np.float64, 2e-5), (np.float32, 5e-5)])
def test_energy_conservation_large_n(dtype, tol):
    """N = 2^18 Plummer sphere, 20 KDK steps of dt = 1/256: the relative energy error stays at the integrator's level
    (the fp64 run is the scaled-down version of BASELINE configs[3]; fp32 adds only round-off thanks to the fp64 flush)."""
    n = 1 << 18
    s = nb.plummer_sphere(n, 2, dtype)
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        k0, u0 = e.energy(0.01)
        e.step(1 / 256, 20, 0.01)
        k1, u1 = e.energy(0.01)
    e0, e1 = k0 + u0, k1 + u1
    assert abs(2 * k0 / abs(u0) - 1) < 0.02 and abs(e0 + 0.25) < 0.01       # virial, Hénon units
    assert abs(e1 - e0) / abs(e0) < tol, f"dE/E = {(e1 - e0) / abs(e0):.3e}"


# ------------------------------------------------------------------ regression: stale step graph after nb_set_state
@pytest.mark.parametrize("dtype", DTYPES)
def test_set_state_with_new_masses_invalidates_the_captured_step_graph(dtype):
    """A graph captured for equal masses bakes in the UNIFORM kernel variant (masses ignored, m[0] applied at the end); a second
    nb_set_state with unequal masses on the SAME context must not replay it."""
    n = 3000
    s1 = nb.plummer_sphere(n, 41, dtype)
    s2 = nb.plummer_sphere(n, 42, dtype)
    s2.mass = (s2.mass * np.random.default_rng(3).uniform(0.25, 4.0, n)).astype(dtype)
    with nb.Engine(n, dtype) as e:
        e.set_state(s1)
        e.step(1 / 256, 8, 0.01)            # eager step 0 + 7 replays of the equal-mass graph
        e.set_state(s2)
        e.step(1 / 256, 8, 0.01)            # same dt / eps / G: the stale graph would be reused
        got = e.get_state()
    with nb.Engine(n, dtype) as e:
        e.set_state(s2)
        e.step(1 / 256, 8, 0.01)
        ref = e.get_state()
    np.testing.assert_array_equal(got.pos, ref.pos)
    np.testing.assert_array_equal(got.vel, ref.vel)
    o = ora.State(s2.pos.astype(np.float64), s2.vel.astype(np.float64), s2.mass.astype(np.float64))
    ora.step(o, 1 / 256, 8, eps=0.01)
    assert np.abs(got.pos.astype(np.float64) - o.pos).max() < (2e-5 if dtype == np.float32 else 1e-11)


@pytest.mark.parametrize("eps", [1e-12, 3e-11])
def test_fused_potential_self_term_follows_the_kernels_guard_fp32(eps):
    """fp32 force kernels drop coincident pairs for eps^2 < 1e-20 (GUARD variant): the O(N) self-term removal must use the same
    predicate, or it subtracts m_i * rsq(eps^2) ~ 1e10 m_i that was never added."""
    s = nb.plummer_sphere(2048, 43, np.float32)
    with nb.Engine(s.n, np.float32) as e:
        e.set_state(s)
        k_f, u_f = e.step_with_energy(1 / 1024, 1, eps)
        k_s, u_s = e.energy(eps)
    assert np.isfinite(u_f[0]) and u_f[0] == pytest.approx(u_s, rel=5e-6) and k_f[0] == pytest.approx(k_s, rel=1e-6)
    assert -1.0 < u_f[0] < -0.3


# ------------------------------------------------------------------ BASELINE configs[3] at full size on one GPU
def test_config3_n4M_fp64_full_size(coracle):
    """configs[3]: N = 2^22 = 4 194 304, fp64, direct summation. One force evaluation is 1.76e13 interactions (~9.4 s on one MI355X):
    64 sampled sinks against the C oracle over ALL sources at 1e-10, Newton's third law, then 4 energy-monitored KDK steps
    (fused potential) with |dE/E| at the integrator's level."""
    n = 1 << 22
    s = nb.plummer_sphere(n, 0, np.float64)
    with nb.Engine(n, np.float64) as e:
        e.set_state(s)
        a = e.accel(0.01)
        assert np.all(np.isfinite(a))
        net = np.abs((s.mass[:, None] * a).sum(0)).max()
        assert net < 1e-12 * (s.mass[:, None] * np.abs(a)).sum(0).max()
        idx = np.random.default_rng(0).choice(n, 64, replace=False)
        ref = coracle.accel(s.pos, s.mass, 0.01, sinks=s.pos[idx])
        assert rel_err(a[idx], ref).max() < 1e-10
        k0, u0 = e.energy(0.01)
        K, U = e.step_with_energy(1 / 256, 4, 0.01)
    e0 = k0 + u0
    assert abs(2 * k0 / abs(u0) - 1) < 0.02 and abs(e0 + 0.25) < 0.01
    drift = np.abs((K + U) - e0).max() / abs(e0)
    assert drift < 1e-7, f"max |dE/E| over 4 steps = {drift:.3e}"


# ------------------------------------------------------------------ degenerate sizes
@pytest.mark.parametrize("dtype", DTYPES)
def test_single_particle_and_two_coincident_particles(dtype):
    """N = 1: no force, free drift, zero potential. N = 2 at the same position with eps = 0: the coincident pair contributes nothing (GUARD variant)."""
    s = nb.State(np.array([[0.1, 0.2, 0.3]], dtype), np.array([[1.0, 0.0, -1.0]], dtype), np.array([2.0], dtype))
    with nb.Engine(1, dtype) as e:
        e.set_state(s)
        assert np.all(e.accel(0.01) == 0.0)
        e.step(0.5, 2, 0.01)
        st = e.get_state()
        k, u = e.energy(0.01)
    np.testing.assert_allclose(st.pos, [[1.1, 0.2, -0.7]], rtol=2e-7)
    assert u == 0.0 and k == pytest.approx(2.0, rel=1e-6)
    s2 = nb.State(np.zeros((2, 3), dtype), np.zeros((2, 3), dtype), np.ones(2, dtype))
    with nb.Engine(2, dtype) as e:
        e.set_state(s2)
        assert np.all(e.accel(0.0) == 0.0)
        k, u = e.energy(0.0)
        assert u == 0.0
    with nb.Engine(3, dtype, devices=[0, 0, 0]) as e:           # one particle per device
        s3 = nb.plummer_sphere(3, 1, dtype)
        e.set_state(s3)
        a = e.accel(0.05)
    ref = ora.compute_forces(s3.pos.astype(np.float64), s3.mass.astype(np.float64), 0.05)
    assert rel_err(a, ref).max() < TOL[dtype]

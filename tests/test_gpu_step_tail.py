"""The leapfrog tail applied by the force launch (include/nbody_amd.h: nb_accel_kdk_dev, nb_kick_drift_dev; csrc/accel_sym_*.hip: the reduce
kernel completes the accelerations of a particle, finishes the tiles a block boundary cut, and applies  v += a*kick_close [; v += a*kick_open;
x += v*drift]) must be BIT-identical to the separate launches it replaces (nb_accel_dev, nb_kick_dev, nb_drift_dev) — the integrator's
products and sums are rounded one by one in both (csrc/integrate.hip), so there is no tolerance in this file. Sizes on both sides of the
pair-symmetric crossover (below it the entry points fall back to the plain kernels), ragged last sets, both dtypes, equal and per-particle
masses, several slab passes (the tail rides on the last pass only)."""
import numpy as np
import pytest

import bachelor_thesis_n_body_amd as nb

pytestmark = pytest.mark.gpu
DTYPES = [np.float32, np.float64]


def _blocks(n, dtype, general, seed=41):
    import torch
    s = nb.plummer_sphere(n, seed, dtype)
    mass = s.mass
    if general:
        mass = (mass * np.random.default_rng(seed).uniform(0.25, 4.0, n)).astype(dtype)
    pm = torch.from_numpy(np.vstack([s.pos.T, mass[None, :]])).contiguous().cuda()
    vel = torch.from_numpy(np.ascontiguousarray(s.vel.T)).cuda()
    return pm, vel


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("n", [5000, 20000, 40001, 70001, 131072])
@pytest.mark.parametrize("tail", [(1 / 512, 0.0, 0.0), (1 / 512, 1 / 512, 1 / 256), (0.003, 0.001, 0.007)])
def test_accel_kdk_equals_separate_launches(dtype, general, n, tail):
    import torch
    k = nb.HipKernels(0)
    pm, vel = _blocks(n, dtype, general)
    acc = torch.zeros_like(vel)
    pm2, vel2, acc2 = pm.clone(), vel.clone(), torch.zeros_like(vel)
    close, open_, drift = tail
    k.accel_kdk(pm, n, pm, n, 0.01, 1.0, acc, False, vel, close, open_, drift, uniform_mass=not general)
    fused_sym = k.last_accel_symmetric()
    k.accel(pm2, n, pm2, n, 0.01, 1.0, acc2, False, uniform_mass=not general)
    assert k.last_accel_symmetric() == fused_sym == (n >= (18432 if dtype == np.float32 else 10240))
    k.kick(vel2, acc2, n, close)
    if open_ != 0.0 or drift != 0.0:
        k.kick(vel2, acc2, n, open_)
        k.drift(pm2, vel2, n, drift)
    torch.cuda.synchronize()
    assert torch.equal(acc, acc2) and torch.equal(vel, vel2) and torch.equal(pm, pm2)
    assert bool(torch.isfinite(pm).all()) and bool((vel != 0).any())
    k.close()


@pytest.mark.parametrize("dtype", DTYPES)
def test_tail_rides_on_the_last_slab_pass_and_on_accumulate(dtype):
    """A slab budget of one slot forces several passes (the velocities must be kicked once, by the pass that writes the accelerations);
    NB_ACCUMULATE: the kick uses the accumulated acceleration."""
    import torch
    n = 60000
    k = nb.HipKernels(0)
    k.set_option("sym_slab_bytes", 1)
    pm, vel = _blocks(n, dtype, False)
    base = torch.full_like(vel, 0.125)
    acc, acc2 = base.clone(), base.clone()
    pm2, vel2 = pm.clone(), vel.clone()
    k.accel_kdk(pm, n, pm, n, 0.02, 1.0, acc, True, vel, 0.01, 0.01, 0.02, uniform_mass=True)
    assert k.last_accel_symmetric()
    k.accel(pm2, n, pm2, n, 0.02, 1.0, acc2, True, uniform_mass=True)
    k.kick(vel2, acc2, n, 0.01)
    k.kick_drift(pm2, vel2, acc2, n, 0.01, 0.02)
    torch.cuda.synchronize()
    assert torch.equal(acc, acc2) and torch.equal(vel, vel2) and torch.equal(pm, pm2)
    k.close()


@pytest.mark.parametrize("dtype", DTYPES)
def test_tail_on_different_sink_and_source_blocks_falls_back(dtype):
    """Sinks != sources: one-sided kernel, the tail runs as its own launches; only the sinks move."""
    import torch
    n = 30000
    k = nb.HipKernels(0)
    pm, vel = _blocks(n, dtype, True)
    src = pm.clone()
    src0 = src.clone()
    acc = torch.zeros_like(vel)
    pm2, vel2, acc2 = pm.clone(), vel.clone(), torch.zeros_like(vel)
    k.accel_kdk(pm, n, src, n, 0.01, 1.0, acc, False, vel, 0.004, 0.004, 0.008)
    assert not k.last_accel_symmetric()
    k.accel(pm2, n, src, n, 0.01, 1.0, acc2, False)
    k.kick(vel2, acc2, n, 0.004)
    k.kick_drift(pm2, vel2, acc2, n, 0.004, 0.008)
    torch.cuda.synchronize()
    assert torch.equal(acc, acc2) and torch.equal(vel, vel2) and torch.equal(pm, pm2) and torch.equal(src, src0)
    k.close()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [3000, 40000])
def test_sharded_engine_one_call_of_k_steps_equals_k_calls_and_the_context_path(dtype, n):
    """ShardedEngine.step(dt, K) on one rank fuses the opening half of step k + 1 into the force launch of step k; K calls of one step
    cannot (the state must be whole when a call returns); nb_step of a context does the same inside the library. All three: same bits."""
    s = nb.plummer_sphere(n, 43, dtype)
    a = nb.ShardedEngine(n, dtype); a.set_state(s)
    a.step(1 / 256, 5, 0.01)
    b = nb.ShardedEngine(n, dtype); b.set_state(s)
    for _ in range(5):
        b.step(1 / 256, 1, 0.01)
    sa, sb = a.gather_state(), b.gather_state()
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        e.step(1 / 256, 5, 0.01)
        sc = e.get_state()
    for x in (sb, sc):
        np.testing.assert_array_equal(sa.pos, x.pos)
        np.testing.assert_array_equal(sa.vel, x.vel)
    assert not np.array_equal(sa.pos, s.pos)


def test_new_entry_points_reject_bad_arguments():
    import ctypes as C
    from bachelor_thesis_n_body_amd import _lib
    lib = _lib.load()
    k = nb.HipKernels(0)
    z = C.c_void_p(0)
    assert lib.nb_kick_drift_dev(k._ctx, 7, z, z, z, z, z, z, z, z, z, 0, 0.1, 0.1) == -1
    assert lib.nb_kick_drift_dev(k._ctx, _lib.NB_F32, z, z, z, z, z, z, z, z, z, 5, 0.1, 0.1) == -1
    assert lib.nb_kick_drift_dev(k._ctx, _lib.NB_F32, z, z, z, z, z, z, z, z, z, 0, 0.1, 0.1) == _lib.NB_OK
    assert lib.nb_accel_kdk_dev(k._ctx, _lib.NB_F32, z, z, z, 5, z, z, z, z, 5, 0.01, 1.0, z, z, z, 0, z, z, z, 0.1, 0.0, 0.0) == -1
    k.close()

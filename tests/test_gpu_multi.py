"""nb_create_multi — the multi-device context of the C-ABI (include/nbody_amd.h; SURVEY.md §8b "ndev, dev_ids", §8e): one process
drives P devices, each owning a contiguous sink range and pulling the other owners' source blocks by peer copies overlapped with
its force kernel. The test box has ONE GPU, so the P-device schedule is rehearsed with the same device listed P times (own streams,
buffers and events per shard — everything but the xGMI link itself). Results must equal the single-device context's up to the
summation order of the chunks, and the oracle's."""
import ctypes as C

import numpy as np
import pytest

import bachelor_thesis_n_body_amd as nb
from bachelor_thesis_n_body_amd import _lib
from oracle import nbody_oracle as ora
from conftest import rel_err

pytestmark = pytest.mark.gpu


def _general(s, dtype, seed=1):
    s.mass = (s.mass * np.random.default_rng(seed).uniform(0.5, 2.0, s.n)).astype(dtype)
    return s


@pytest.mark.parametrize("dtype,P,n,uniform", [(np.float32, 2, 6001, True), (np.float32, 3, 8191, False), (np.float64, 2, 4097, False),
                                               (np.float64, 4, 5000, True), (np.float32, 8, 20003, True)])
def test_multi_device_context_matches_single_device_and_oracle(dtype, P, n, uniform):
    s = nb.plummer_sphere(n, 11, dtype)
    if not uniform:
        _general(s, dtype)
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        a1 = e.accel(0.01)
        k1, u1 = e.energy(0.01)
        e.step(1 / 256, 5, 0.01)
        ref = e.get_state()
        k1b, u1b = e.energy(0.01)
    with nb.Engine(n, dtype, devices=[0] * P) as e:
        cnt, off = e.layout()
        assert len(cnt) == P and sum(cnt) == n and off[0] == 0 and max(cnt) - min(cnt) <= 1
        e.set_state(s)
        aP = e.accel(0.01)
        kP, uP = e.energy(0.01)
        e.step(1 / 256, 5, 0.01)
        got = e.get_state()
        kPb, uPb = e.energy(0.01)
        np.testing.assert_array_equal(e.cached_acc(), e.cached_acc())
    tol = 3e-6 if dtype == np.float32 else 1e-12
    assert np.abs(aP - a1).max() < tol * np.abs(a1).max()
    assert rel_err(aP, ora.compute_forces(s.pos.astype(np.float64), s.mass.astype(np.float64), 0.01)).max() < (1e-5 if dtype == np.float32 else 1e-10)
    assert np.abs(got.pos - ref.pos).max() < 10 * tol and np.abs(got.vel - ref.vel).max() < 10 * tol
    np.testing.assert_array_equal(got.mass, ref.mass)
    assert kP == pytest.approx(k1, rel=1e-12) and uP == pytest.approx(u1, rel=3e-6 if dtype == np.float32 else 1e-12)
    assert kPb == pytest.approx(k1b, rel=1e-5 if dtype == np.float32 else 1e-11) and uPb == pytest.approx(u1b, rel=1e-5 if dtype == np.float32 else 1e-11)


def test_multi_context_many_steps_with_buffer_reuse_fp64_is_deterministic_and_tracks_the_oracle():
    """40 steps on 3 shards: every receive buffer is overwritten 40 times while earlier kernels may still read it — a missing event
    would show as a wrong trajectory. Same run twice: bit-identical; vs the NumPy oracle: round-off."""
    n = 1500
    s = _general(nb.plummer_sphere(n, 5, np.float64), np.float64, 2)
    runs = []
    for _ in range(2):
        with nb.Engine(n, np.float64, devices=[0, 0, 0]) as e:
            e.set_state(s)
            e.step(1 / 512, 40, 0.02)
            runs.append(e.get_state())
    np.testing.assert_array_equal(runs[0].pos, runs[1].pos)
    np.testing.assert_array_equal(runs[0].vel, runs[1].vel)
    o = ora.State(s.pos.copy(), s.vel.copy(), s.mass.copy())
    ora.step(o, 1 / 512, 40, eps=0.02)
    assert np.abs(runs[0].pos - o.pos).max() < 1e-11 and np.abs(runs[0].vel - o.vel).max() < 1e-10


@pytest.mark.parametrize("dtype,uniform", [(np.float32, True), (np.float64, False)])
def test_multi_context_step_with_energy(dtype, uniform):
    """nb_step_energy on P devices: per-device (K, W) logs from the fused force+potential kernels, phi accumulated over the chunks, added on
    the host — equal to the single-device log and to the separate energy pass at the final state."""
    n = 5003
    s = nb.plummer_sphere(n, 21, dtype)
    if not uniform:
        _general(s, dtype, 3)
    with nb.Engine(n, dtype) as e:
        e.set_state(s)
        K1, U1 = e.step_with_energy(1 / 256, 4, 0.01)
    with nb.Engine(n, dtype, devices=[0, 0, 0]) as e:
        e.set_state(s)
        K3, U3 = e.step_with_energy(1 / 256, 4, 0.01)
        k, u = e.energy(0.01)
    np.testing.assert_allclose(K3, K1, rtol=1e-5 if dtype == np.float32 else 1e-11)
    np.testing.assert_allclose(U3, U1, rtol=3e-6 if dtype == np.float32 else 1e-11)
    assert K3[-1] == pytest.approx(k, rel=1e-12) and U3[-1] == pytest.approx(u, rel=3e-6 if dtype == np.float32 else 1e-11)


def test_multi_context_generalised_kdk_and_state_replacement():
    n = 3000
    s1 = nb.plummer_sphere(n, 3, np.float32)
    s2 = _general(nb.plummer_sphere(n, 4, np.float32), np.float32)
    k1, d, k2 = [0.001, 0.002, 0.0015], [0.002, 0.004, 0.003], [0.001, 0.002, 0.0015]
    with nb.Engine(n, np.float32) as e:
        e.set_state(s2); e.kdk_run(k1, d, k2, 0.01); ref = e.get_state()
    with nb.Engine(n, np.float32, devices=[0, 0]) as e:
        e.set_state(s1); e.step(1 / 256, 3, 0.01)          # equal masses first (uniform kernel) ...
        e.set_state(s2); e.kdk_run(k1, d, k2, 0.01)         # ... then unequal ones on the same context
        got = e.get_state()
    assert np.abs(got.pos - ref.pos).max() < 3e-5 and np.abs(got.vel - ref.vel).max() < 3e-5


def test_multi_context_rejects_what_it_does_not_support_and_bad_arguments():
    lib = _lib.load()
    ctx = C.c_void_p()
    ids = (C.c_int * 2)(0, 0)
    assert lib.nb_create_multi(C.byref(ctx), 1, _lib.NB_F32, 2, ids) == -1 and b"fewer particles" in lib.nb_last_error(None)
    assert lib.nb_create_multi(C.byref(ctx), 100, _lib.NB_F32, 0, ids) == -1
    bad = (C.c_int * 2)(0, 99)
    assert lib.nb_create_multi(C.byref(ctx), 100, _lib.NB_F32, 2, bad) == -3 and not ctx
    with nb.Engine(1000, np.float32, devices=[0, 0]) as e:
        with pytest.raises(RuntimeError):
            e.cached_acc()                                   # no state yet
        e.set_state(nb.plummer_sphere(1000, 1, np.float32))
        with pytest.raises(RuntimeError, match="multi-device"):
            e.set_solver("tree", 0.5)
        with pytest.raises(RuntimeError, match="multi-device"):
            e.set_periodic(10.0)
        with pytest.raises(RuntimeError, match="multi-device"):
            e.set_stream(0)
        with pytest.raises(RuntimeError, match="no cached acceleration"):
            e.cached_acc()                                   # state, but no force evaluation yet
    with nb.Engine(64, np.float32) as e:                     # a plain context reports one device
        assert e.layout() == ([64], [0])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_multi_context_reports_schedule_and_own_chunk_kernel_separately(dtype):
    """nb_get_option(NB_QUERY_MULTI_SCHEDULE) is the schedule the context RUNS (its option); nb_last_accel_symmetric is the own-chunk launch
    of device 0 — in the one-sided schedule the later cross-chunk launches (one-sided by construction) must not overwrite it, and in the
    pair-symmetric schedule small chunks whose two-block hop falls back to one-sided launches must not either."""
    n = 3 * 40000
    s = nb.plummer_sphere(n, 3, dtype)
    with nb.Engine(n, dtype, devices=[0, 0, 0]) as e:
        e.set_state(s)
        assert e.get_option("multi_schedule") == 1 and e.get_option("pair_symmetry") == -1
        a_pair = e.accel(0.01)
        assert e.last_accel_symmetric() and e.get_option("own_chunk_symmetric") == 1        # chunks of 40 000: the pair-symmetric kernel
        e.set_option("pair_symmetry", 0)                      # one-sided schedule AND one-sided kernels
        assert e.get_option("multi_schedule") == 0
        a_one = e.accel(0.01)
        assert not e.last_accel_symmetric()
        e.set_option("pair_symmetry", -1)
        assert e.get_option("multi_schedule") == 1
    assert rel_err(a_pair, a_one.astype(np.float64)).max() < (1e-6 if dtype == np.float32 else 2e-13)
    with nb.Engine(40000, dtype) as e:                        # a single-device context runs no multi-device schedule
        e.set_state(nb.plummer_sphere(40000, 3, dtype))
        e.accel(0.01, fetch=False)
        assert e.get_option("multi_schedule") == 0 and e.get_option("own_chunk_symmetric") == 1
    n = 3 * 9000                                              # own chunks of 9000 < sym_min_n: own chunk one-sided, hops fall back too
    with nb.Engine(n, dtype, devices=[0, 0, 0]) as e:
        e.set_state(nb.plummer_sphere(n, 4, dtype))
        e.accel(0.01, fetch=False)
        assert e.get_option("multi_schedule") == 1 and not e.last_accel_symmetric()

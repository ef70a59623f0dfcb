"""PM / P3M on the GPU (csrc/mesh.hip, through nb_set_mesh + nb_accel) against the NumPy oracle, part by part, and against the
Ewald solver (the physics check: P3M -> exact periodic gravity)."""
import numpy as np
import pytest

import bachelor_thesis_n_body_amd as nb
from oracle import ewald_oracle as eo
from oracle import mesh_oracle as mo

pytestmark = pytest.mark.gpu
DTYPES = [np.float32, np.float64]


def _accel(pos, m, dtype, L, solver, ngrid, eps=0.0, G=1.0, **kw):
    s = nb.State(pos.astype(dtype), np.zeros_like(pos, dtype=dtype), m.astype(dtype))
    with nb.Engine(s.n, dtype) as e:
        e.set_mesh(solver, L, ngrid, **kw)
        e.set_state(s)
        return e.accel(eps, G).astype(np.float64)


def _cloud(n, L, seed=0, clustered=False):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0, L, (n, 3))
    if clustered:                     # two thirds of the particles in one tight blob: many empty cells, one very full one
        k = 2 * n // 3
        pos[:k] = np.mod(0.5 * L + 0.03 * L * rng.normal(size=(k, 3)), L)
    return pos, rng.uniform(0.5, 1.5, n)


def _scale(a):
    return np.sqrt((a * a).sum(1).mean())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ngrid,interlace", [(16, True), (32, True), (32, False), (27, True)])
def test_pm_matches_oracle(dtype, ngrid, interlace):
    L = 50.0
    pos, m = _cloud(3001, L, 1)
    got = _accel(pos, m, dtype, L, "pm", ngrid, G=2.0, interlace=interlace)
    p = pos.astype(dtype).astype(np.float64); mm = m.astype(dtype).astype(np.float64)
    ref = mo.mesh_accel(p, mm, L, ngrid, 2.0, None, interlace)
    assert np.abs(got - ref).max() < (1e-10 if dtype == np.float64 else 3e-6) * _scale(ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("clustered", [False, True])
def test_p3m_parts_match_oracle(dtype, clustered):
    L, ngrid, eps = 80.0, 32, 0.2
    pos, m = _cloud(4000, L, 2, clustered)
    p = pos.astype(dtype).astype(np.float64); mm = m.astype(dtype).astype(np.float64)
    rs = 1.25 * L / ngrid
    ref_l = mo.mesh_accel(p, mm, L, ngrid, 1.0, rs, True)
    ref_s = mo.short_range_accel(p, mm, L, rs, 4.5 * rs, eps)
    got_l = _accel(pos, m, dtype, L, "p3m", ngrid, eps, part="long")
    got_s = _accel(pos, m, dtype, L, "p3m", ngrid, eps, part="short")
    got = _accel(pos, m, dtype, L, "p3m", ngrid, eps)
    tol = 1e-10 if dtype == np.float64 else 2e-5
    assert np.abs(got_l - ref_l).max() < tol * _scale(ref_l)
    assert np.abs(got_s - ref_s).max() < tol * _scale(ref_s)
    assert np.abs(got - (ref_l + ref_s)).max() < tol * _scale(ref_s)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("eps", [0.2, 0.0])
@pytest.mark.parametrize("n", [4000, 300000])
def test_short_range_variants_equal_masses_and_zero_softening(dtype, eps, n):
    """The four instantiations of the pair kernel: equal masses (mass multiply hoisted out of the loop) x eps > 0 (no r = 0 guard in the loop),
    on few chunks (one wave per chunk, neighbouring row and z-part, fixed-order slab reduction) and on many (one wave per chunk) — against
    the NumPy oracle on sampled sinks."""
    if n > 4000 and eps == 0.0:
        pytest.skip("covered at n = 4000")
    L, ngrid = 80.0, 32 if n == 4000 else 96
    pos, _ = _cloud(n, L, 5)
    m = np.full(n, 0.75)
    p = pos.astype(dtype).astype(np.float64)
    rs = 1.25 * L / ngrid
    got = _accel(pos, m, dtype, L, "p3m", ngrid, eps, part="short")
    idx = np.arange(n) if n == 4000 else np.random.default_rng(1).choice(n, 256, replace=False)
    ref = mo.short_range_accel(p, m, L, rs, 4.5 * rs, eps, sinks=idx) if n > 4000 else mo.short_range_accel(p, m, L, rs, 4.5 * rs, eps)
    tol = 1e-10 if dtype == np.float64 else 2e-5
    assert np.abs(got[idx] - ref).max() < tol * _scale(ref)


def test_p3m_agrees_with_ewald_solver():
    """The two periodic solvers of the library against each other and against the Ewald oracle."""
    L, n, eps = 100.0, 3200, 0.3
    pos, m = _cloud(n, L, 3)
    s = nb.State(pos, np.zeros_like(pos), m)
    with nb.Engine(n, np.float64) as e:
        e.set_periodic(L)
        e.set_state(s)
        ew = e.accel(eps)
    p3 = _accel(pos, m, np.float64, L, "p3m", 32, eps)
    err = np.linalg.norm(p3 - ew, axis=1) / np.linalg.norm(ew, axis=1)
    assert np.sqrt((err ** 2).mean()) < 6e-3
    sub = slice(0, 300)
    ref = eo.ewald_accel(pos, m, L, eps)[sub]
    err = np.linalg.norm(p3[sub] - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert np.sqrt((err ** 2).mean()) < 6e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_positions_outside_the_box_and_on_cell_edges(dtype):
    L, ngrid = 40.0, 24
    pos, m = _cloud(1000, L, 4)
    pos[:50] = np.floor(pos[:50] / (L / ngrid)) * (L / ngrid)          # exactly on mesh nodes
    pos[50] = [0.0, 0.0, 0.0]; pos[51] = [L, L, L]; pos[52] = [L - 1e-9, 0.0, L / 2]
    shifted = pos + np.array([L, -L, 2 * L]) * (np.arange(1000) % 2)[:, None]
    a0 = _accel(pos, m, dtype, L, "p3m", ngrid, 0.1)
    a1 = _accel(shifted, m, dtype, L, "p3m", ngrid, 0.1)
    assert np.isfinite(a0).all()
    tol = 1e-9 if dtype == np.float64 else 2e-4       # fp32: the shifted coordinates themselves lose 3 digits
    assert np.abs(a1 - a0).max() < tol * _scale(a0)


def test_momentum_is_conserved_by_both_parts():
    L = 60.0
    pos, m = _cloud(5000, L, 5, clustered=True)
    for part in ("long", "short", "both"):
        a = _accel(pos, m, np.float64, L, "p3m", 32, 0.2, part=part)
        assert np.abs((m[:, None] * a).sum(0)).max() < 1e-9 * np.abs(m[:, None] * a).sum()


def test_mesh_argument_errors():
    s = nb.plummer_sphere(256, 1, np.float32)
    with nb.Engine(s.n, np.float32) as e:
        with pytest.raises(RuntimeError, match="boxsize"):
            e.set_mesh("pm", 0.0, 16)
        with pytest.raises(RuntimeError, match="grid size"):
            e.set_mesh("pm", 1.0, 2)
        with pytest.raises(RuntimeError, match="too coarse"):
            e.set_mesh("p3m", 1.0, 16)             # cut-off 5.6 cells: fewer than 3 short-range cells per side
        e.set_mesh("pm", 1.0, 16)                  # fine for PM


@pytest.mark.parametrize("dtype", DTYPES)
def test_steps_wrap_positions_and_track_the_ewald_trajectory(dtype):
    L, n = 30.0, 2000
    pos, m = _cloud(n, L, 6)
    rng = np.random.default_rng(7)
    vel = rng.normal(size=(n, 3))
    s = nb.State(pos.astype(dtype), vel.astype(dtype), m.astype(dtype))
    out = {}
    for name in ("p3m", "ewald"):
        with nb.Engine(n, dtype) as e:
            if name == "p3m":
                e.set_mesh("p3m", L, 32)
            else:
                e.set_periodic(L)
            e.set_state(s)
            e.step(0.05, 20, 0.1)
            out[name] = e.get_state()
    assert out["p3m"].pos.min() >= 0.0 and out["p3m"].pos.max() <= L
    wrap = lambda d: d - L * np.rint(d / L)
    d = wrap(out["p3m"].pos.astype(np.float64) - out["ewald"].pos.astype(np.float64))
    by_force = wrap(out["ewald"].pos.astype(np.float64) - (pos + vel * (0.05 * 20)))      # what gravity did to the orbits
    rms = lambda a: np.sqrt((a * a).sum(1).mean())
    assert rms(d) < 0.05 * rms(by_force)          # same trajectories up to the sub-percent force error of the mesh


@pytest.mark.parametrize("ngrid,interlace,dtype", [(27, True, np.float64), (50, False, np.float64), (64, True, np.float32), (9, True, np.float64),
                                                   (32, True, np.float64), (40, False, np.float64), (8, True, np.float64), (16, False, np.float32)])
def test_long_range_through_lds_bricks_matches_oracle(ngrid, interlace, dtype):
    """n >= 1e5 takes the brick-sorted, LDS-privatised deposit and gather: mesh sizes that are not multiples of the 8-cell brick,
    a mesh smaller than one brick tile, particles far outside the box, clustered (one brick holds a third of the particles). Meshes of whole
    bricks (64, 32, 40, 16, and 8 = a single brick that is its own neighbour) take the staged deposit — tiles to one slot per chunk, summed by
    brick_assemble_kernel — the others (and meshes with fewer than 32 particles per brick) the atomic flush."""
    L, n = 75.0, 120_000
    rng = np.random.default_rng(ngrid)
    pos = rng.uniform(-L, 2 * L, (n, 3))
    pos[: n // 3] = 0.4 * L + 0.02 * L * rng.normal(size=(n // 3, 3))
    m = rng.uniform(0.5, 1.5, n)
    rs = 1.25 * L / ngrid
    p = pos.astype(dtype).astype(np.float64); mm = m.astype(dtype).astype(np.float64)
    ref = mo.mesh_accel(p, mm, L, ngrid, 1.0, rs if ngrid >= 27 else None, interlace)
    if ngrid >= 27:
        got = _accel(pos, m, dtype, L, "p3m", ngrid, 0.1, part="long", interlace=interlace)
    else:
        got = _accel(pos, m, dtype, L, "pm", ngrid, interlace=interlace)
    assert np.abs(got - ref).max() < (1e-9 if dtype == np.float64 else 5e-6) * _scale(ref)


@pytest.mark.parametrize("eps", [0.15, 0.0])
@pytest.mark.parametrize("equal_masses", [True, False])
@pytest.mark.parametrize("clustered", [False, True])
def test_short_range_tile_kernel_matches_oracle(eps, equal_masses, clustered):
    """fp32, >= 16 384 particles and >= 4 per cell of edge r_cut/3: the pair sum runs on 16-sink tiles (candidate segments from the blob's
    bounding box, sources culled against it and compacted in LDS). Sampled sinks — among them the ones next to the box faces, whose
    partners are periodic images — against the NumPy oracle, for the four instantiations of the kernel."""
    n, L, ngrid = 300000, 80.0, 48
    pos, mr = _cloud(n, L, 11, clustered=clustered)
    m = np.full(n, 0.75) if equal_masses else mr
    p = pos.astype(np.float32).astype(np.float64); mm = m.astype(np.float32).astype(np.float64)
    rs = 1.25 * L / ngrid
    got = _accel(pos, m, np.float32, L, "p3m", ngrid, eps, part="short")
    near_face = np.argsort(np.minimum(p, L - p).min(1))[:64]
    idx = np.unique(np.concatenate([np.random.default_rng(2).choice(n, 192, replace=False), near_face]))
    ref = mo.short_range_accel(p, mm, L, rs, 4.5 * rs, eps, sinks=idx)
    assert np.isfinite(got).all()
    assert np.abs(got[idx] - ref).max() < 3e-5 * _scale(ref)


@pytest.mark.parametrize("seed", range(7))
def test_short_range_tile_kernel_random_configurations(seed):
    """The tile kernel over its parameter space: 8- and 16-sink blobs (n below / above 100 000), cells of r_cut/3 and r_cut/2 (>= 4 particles
    per cell decides), few cells per side (every neighbouring row a periodic image), box sizes, softenings, equal and unequal masses,
    uniform and clustered clouds — sampled sinks against the NumPy oracle."""
    rng = np.random.default_rng(500 + seed)
    n = int(rng.choice([20000, 60000, 150000]))
    L = float(rng.choice([10.0, 333.0]))
    ngrid = int(rng.choice([18, 24, 32, 40]))     # 3 ... 7 cells of edge r_cut per side; always >= 4 particles per r_cut/2 cell
    eps = float(rng.choice([0.0, 0.02 * L / ngrid, 0.4 * L / ngrid]))
    if seed == 6:
        n, L, ngrid, eps = 20000, 100.0, 40, 0.05     # 2.2 particles per r_cut/3 cell, 7.3 per r_cut/2 cell: the coarser cells
    pos, mr = _cloud(n, L, 700 + seed, clustered=bool(seed % 2))
    m = mr if seed % 3 else np.full(n, 1.25)
    p = pos.astype(np.float32).astype(np.float64); mm = m.astype(np.float32).astype(np.float64)
    rs = 1.25 * L / ngrid
    got = _accel(pos, m, np.float32, L, "p3m", ngrid, eps, part="short")
    near_face = np.argsort(np.minimum(p, L - p).min(1))[:32]
    idx = np.unique(np.concatenate([rng.choice(n, 96, replace=False), near_face]))
    ref = mo.short_range_accel(p, mm, L, rs, 4.5 * rs, eps, sinks=idx)
    assert np.isfinite(got).all()
    assert np.abs(got[idx] - ref).max() < 3e-5 * _scale(ref), (n, L, ngrid, eps)


def test_close_pair_across_a_box_face_keeps_fp32_precision():
    """Two particles 0.08 apart on either side of the face z = L, in a box of L = 80 (fp32 ulp(L) = 8e-6): the pair kernels work in
    coordinates relative to the sink blob, so the image shift is a small number and the separation keeps its precision — with a plain +-L
    shift the force of this pair was off by 4.3e-5 of its size."""
    L, ngrid, eps = 80.0, 48, 0.15
    rs = 1.25 * L / ngrid
    for n in (2000, 30000, 300000):               # the 64-sink kernel, the 8-sink and the 16-sink tile kernel
        pos, _ = _cloud(n, L, 21)
        pos[0] = (64.74076843, 31.28139877, 79.99965668)
        pos[1] = (64.66902160, 31.25299840, 0.02958335)
        m = np.full(n, 0.75)
        p = pos.astype(np.float32).astype(np.float64)
        got = _accel(pos, m, np.float32, L, "p3m", ngrid, eps, part="short")
        ref = mo.short_range_accel(p, m, L, rs, 4.5 * rs, eps, sinks=np.array([0, 1]))
        # the 64-sink kernel subtracts L from the coordinate next to the far face (exact); the tile kernels round the source once at the
        # scale of the blob's extent (a few units here, not L)
        assert np.abs(got[:2] - ref).max() < (4e-6 if n == 2000 else 1e-5) * np.abs(ref).max(), n


@pytest.mark.parametrize("seed", range(8))
def test_short_range_random_configurations(seed):
    """Small random set-ups of the cell-list pair sum: few particles (all lanes of a block idle but a few), the minimum of 3 cells
    per side, zero softening, every particle in one cell, unequal masses, both precisions."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.choice([3, 50, 129, 700, 2500]))
    ngrid = int(rng.choice([17, 24, 33, 48]))
    L = float(rng.choice([10.0, 333.0]))
    eps = float(rng.choice([0.0, 0.01 * L / ngrid, 0.5 * L / ngrid]))
    dtype = DTYPES[seed % 2]
    pos = rng.uniform(0, L, (n, 3))
    if seed % 3 == 0:
        pos = np.mod(0.3 * L + 0.3 * (L / ngrid) * rng.normal(size=(n, 3)), L)        # one cell
    m = rng.uniform(0.1, 3.0, n)
    p = pos.astype(dtype).astype(np.float64); mm = m.astype(dtype).astype(np.float64)
    rs = 1.25 * L / ngrid
    ref = mo.short_range_accel(p, mm, L, rs, 4.5 * rs, eps)
    got = _accel(pos, m, dtype, L, "p3m", ngrid, eps, part="short")
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < (1e-10 if dtype == np.float64 else 3e-5) * max(_scale(ref), 1e-300)


@pytest.mark.parametrize("solver", ["p3m", "ewald"])
def test_kdk_run_equals_single_steps(solver):
    """nb_kdk_run enqueues all steps back to back (one host synchronisation); same state as step-by-step calls, bit for bit."""
    L, n = 30.0, 5000
    pos, m = _cloud(n, L, 8)
    vel = np.random.default_rng(9).normal(size=(n, 3))
    s = nb.State(pos.astype(np.float32), vel.astype(np.float32), m.astype(np.float32))
    k1 = np.linspace(0.01, 0.02, 6); d = np.linspace(0.03, 0.02, 6); k2 = np.linspace(0.015, 0.005, 6)
    out = []
    for batched in (True, False):
        with nb.Engine(n, np.float32) as e:
            if solver == "p3m":
                e.set_mesh("p3m", L, 32)
            else:
                e.set_periodic(L, 13.2, 15)
            e.set_state(s)
            if batched:
                e.kdk_run(k1, d, k2, 0.1)
            else:
                for a, b, c in zip(k1, d, k2):
                    e.kdk_step(a, b, c, 0.1)
            out.append(e.get_state())
    np.testing.assert_array_equal(out[0].pos, out[1].pos)
    np.testing.assert_array_equal(out[0].vel, out[1].vel)
    with nb.Engine(n, np.float32) as e:
        e.set_state(s)
        with pytest.raises(ValueError):
            e.kdk_run([0.1, 0.2], [0.1], [0.1, 0.2], 0.1)
        e.kdk_run([], [], [], 0.1)                                          # zero steps: nothing happens


@pytest.mark.parametrize("solver,dtype", [("p3m", np.float32), ("pm", np.float64), ("ewald", np.float32), ("direct", np.float64)])
def test_graph_replayed_run_equals_eager_steps(solver, dtype):
    """>= 8 steps on a small system: nb_kdk_run captures ONE step (factors read from a device table by a device-side step counter) into a
    hipGraph and replays it — for the mesh solvers that is rocPRIM sorts, hipFFT transforms and ~20 kernels per step. The state must be bit
    for bit the one of step-by-step eager calls, and the context must report that the graph path was really taken."""
    L, n, steps = 30.0, 4000, 12
    pos, m = _cloud(n, L, 18)
    vel = np.random.default_rng(19).normal(size=(n, 3))
    s = nb.State(pos.astype(dtype), vel.astype(dtype), m.astype(dtype))
    k1 = np.linspace(0.01, 0.02, steps); d = np.linspace(0.03, 0.02, steps); k2 = np.linspace(0.015, 0.005, steps)
    out = []
    for batched in (True, False):
        with nb.Engine(n, dtype) as e:
            if solver in ("p3m", "pm"):
                e.set_mesh(solver, L, 32)
            elif solver == "ewald":
                e.set_periodic(L, 13.2, 15)
            e.set_state(s)
            if batched:
                e.kdk_run(k1, d, k2, 0.1)
                assert e.last_run_used_graph(), "the graph path was not taken (capture refused?)"
                e.kdk_run(k1[:3], d[:3], k2[:3], 0.1)                      # < 8 steps: eager, continues from the replayed state
                assert not e.last_run_used_graph()
            else:
                for a, b, c in zip(k1, d, k2):
                    e.kdk_step(a, b, c, 0.1)
                for a, b, c in zip(k1[:3], d[:3], k2[:3]):
                    e.kdk_step(a, b, c, 0.1)
            out.append(e.get_state())
    if solver == "pm":          # small-n PM deposits with fp64 global atomics: the summation order differs from run to run (last bits)
        np.testing.assert_allclose(out[0].pos, out[1].pos, rtol=0, atol=1e-11)
        np.testing.assert_allclose(out[0].vel, out[1].vel, rtol=0, atol=1e-10)
    else:
        np.testing.assert_array_equal(out[0].pos, out[1].pos)
        np.testing.assert_array_equal(out[0].vel, out[1].vel)


def test_contexts_release_all_device_memory():
    """A farm runs many jobs in one process: creating, using and destroying contexts of every periodic solver must not leak
    device memory (mesh grids, FFT plans, cell lists, tables, structure factors)."""
    import torch
    L, n = 40.0, 20000
    pos, m = _cloud(n, L, 12)
    s = nb.State(pos.astype(np.float32), np.zeros((n, 3), np.float32), m.astype(np.float32))

    def cycle():
        for setup in (lambda e: e.set_mesh("pm", L, 32), lambda e: e.set_mesh("p3m", L, 32), lambda e: e.set_periodic(L, 16.5, 19),
                      lambda e: e.set_periodic(L), lambda e: e.set_solver("tree", 0.5), None):
            with nb.Engine(n, np.float32) as e:
                if setup is not None:
                    setup(e)
                e.set_state(s)
                e.accel(0.1, fetch=False)
                if setup is None:
                    e.step_with_energy(0.01, 2, 0.1)            # direct solver: the phi / energy-log buffers too
        with nb.Engine(n, np.float32, devices=[0, 0, 0]) as e:  # multi-device context: shards, receive buffers, copy streams, events
            e.set_state(s)
            e.step_with_energy(0.01, 2, 0.1)
        with nb.Engine(n, np.float32) as e:                     # graph-replayed run + rung hierarchy: factor tables, graphs, rung arrays
            e.set_mesh("p3m", L, 32)
            e.set_state(s)
            e.kdk_run(np.full(9, 0.01), np.full(9, 0.01), np.full(9, 0.01), 0.1)
            e.rung_step(3, np.linspace(0, 0.02, 9), np.linspace(0, 0.02, 9), 0.02, 1e-4, 0.1)
    cycle()                                   # first use: library-level caches (hipFFT kernels, rocPRIM) are allocated once
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, (free0, free1)          # nothing accumulates (allow allocator granularity)

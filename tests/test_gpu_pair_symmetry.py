"""The pair-symmetric all-pairs kernels (csrc/accel_sym_f32.hip, accel_sym_f64.hip: a particle set on itself, every unordered pair evaluated
once) against the C oracle on the same inputs, against the one-sided kernels they replace, and through their own invariants: Newton's third
law, bit-reproducibility, independence of the number of slab passes. All calls go through the C-ABI (ctypes -> libnbody_amd.so).

Tolerances (per-particle |da| / |a| against the fp64 oracle on the same rounded inputs): fp64 <= 2e-13; fp32 <= 1e-6 from n = 32768 on
(measured 8e-8 .. 2e-7, profiles/r03_sym_kernel_development_log.txt: fp32 pair terms, partial sums flushed to fp64 per tile / set) and
<= 1e-5 below (measured up to 2.3e-6 at n = 1500 .. 16384 — profiles/r04_parity_measured_errors.txt — where a few central particles have
|a| far below the typical pair term); fp64 measured <= 4.2e-14. All within 5x of what is measured, so a lost flush or a mis-added slab fails."""
import numpy as np
import pytest

import bachelor_thesis_n_body_amd as nb
from conftest import rel_err, note

pytestmark = pytest.mark.gpu
TOL = {np.float32: 1e-5, np.float64: 2e-13}


def tol(dtype, n):
    return 1e-6 if (dtype == np.float32 and n >= 32768) else TOL[dtype]
DTYPES = [np.float32, np.float64]


def _cloud(n, dtype, general, seed=11):
    s = nb.plummer_sphere(n, seed, dtype)
    if general:
        rng = np.random.default_rng(seed + 1)
        s = nb.State(s.pos, s.vel, (s.mass * rng.uniform(0.25, 4.0, n)).astype(dtype))
    return s


def _engine(n, dtype, state, force_sym=True):
    e = nb.Engine(n, dtype)
    e.set_state(state)
    if force_sym:
        e.set_option("sym_min_n", 0)          # take the pair-symmetric kernel at every size it can run at (>= two sets)
    return e


def _sample(n, k=160):
    rng = np.random.default_rng(n)
    return np.unique(np.concatenate([rng.integers(0, n, k), [0, 1, n // 2, n - 2, n - 1]]))


# set sizes: fp32 1024 / 2048 / 3072 particles (4 / 8 / 12 sinks per lane), fp64 512 / 1024 / 2048 — chosen from n (accel_sym_*.hip);
# the sizes below hit two sets (one-sided tiles only), three sets (one two-sided offset), ragged last sets incl. a last set of ONE particle,
# odd and even set counts, and every sinks-per-lane variant
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("n", [1500, 2049, 3073, 5000, 12289, 40000, 65536, 100001, 150001, 230000])
def test_symmetric_kernel_matches_oracle(dtype, general, n, coracle):
    s = _cloud(n, dtype, general)
    with _engine(n, dtype, s) as e:
        a = e.accel(0.01)
        assert e.last_accel_symmetric(), "the pair-symmetric kernel did not run"
    idx = np.arange(n) if n <= 20000 else _sample(n)
    ref = coracle.accel(s.pos.astype(np.float64), s.mass.astype(np.float64), 0.01, sinks=s.pos[idx].astype(np.float64))
    err = rel_err(a[idx], ref)
    note(f"sym accel vs oracle n={n} {np.dtype(dtype).name} {'general' if general else 'uniform'}", err.max(), tol(dtype, n))
    assert np.isfinite(a).all() and err.max() < tol(dtype, n), f"max {err.max():.3e} median {np.median(err):.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("n", [3000, 50000, 262144, 1 << 20])
def test_symmetric_and_one_sided_kernels_agree_to_rounding(dtype, general, n):
    """EVERY sink of the set, up to the headline size: the two kernels share no partial-sum structure (sets x offsets x tiles with slabs
    against sink blocks x source splits), so a mis-added slab or a dropped tile of either shows up here."""
    s = _cloud(n, dtype, general, seed=5)
    with _engine(n, dtype, s) as e:
        a_sym = e.accel(0.02, G=1.7)
        assert e.last_accel_symmetric()
        e.set_option("pair_symmetry", 0)
        a_one = e.accel(0.02, G=1.7)
        assert not e.last_accel_symmetric()
        e.set_option("pair_symmetry", -1)     # back to the default
        a_again = e.accel(0.02, G=1.7)
        assert e.last_accel_symmetric()
    # both sum the same pair terms, in different orders, with fp32 partials flushed to fp64 (fp32) / plain fp64
    d = rel_err(a_sym, a_one.astype(np.float64))
    note(f"sym vs one-sided n={n} {np.dtype(dtype).name} {'general' if general else 'uniform'}", d.max())
    # (two fp32 results against each other: twice the rounding of one against the oracle — measured up to 1.1e-6 at n = 50 000)
    assert d.max() < ((4e-6 if n < 32768 else 2e-6) if dtype == np.float32 else 2e-13), f"{d.max():.3e}"
    assert np.array_equal(a_sym, a_again), "the pair-symmetric kernel is not bit-reproducible"


@pytest.mark.parametrize("dtype", DTYPES)
def test_newtons_third_law_holds_term_by_term(dtype):
    """Each pair term is computed once and added to one particle, subtracted from the other: sum_i m_i a_i vanishes to the rounding of the
    partial sums — orders of magnitude below the one-sided kernels, whose two copies of a pair term round independently."""
    n = 60000
    s = _cloud(n, dtype, True, seed=2)
    with _engine(n, dtype, s) as e:
        a = e.accel(0.01).astype(np.float64)
    m = s.mass.astype(np.float64)[:, None]
    assert np.abs((m * a).sum(0)).max() / np.abs(m * a).sum(0).max() < (2e-9 if dtype == np.float32 else 1e-15)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("general", [False, True])
def test_result_does_not_depend_on_the_number_of_slab_passes(dtype, general):
    """A scratch budget of 1 MiB forces one offset per pass (dozens of passes, partial sums collected in fp64 between them); the default
    budget takes one pass. Same pair terms, same per-slab sums: equal up to the fp64 addition order of the slabs."""
    n = 30000
    s = _cloud(n, dtype, general, seed=8)
    with _engine(n, dtype, s) as e:
        a1 = e.accel(0.01)
        e.set_option("sym_slab_bytes", 1 << 20)
        a2 = e.accel(0.01)
        assert e.last_accel_symmetric()
    d = rel_err(a2, a1.astype(np.float64))
    # (fp32: the passes cut the work into other block shares — a tile walked by two blocks adds two fp32 partial sums instead of one)
    # measured up to 9.5e-7 at this n (30 000: the class where a single evaluation is within 2.3e-6 of the oracle)
    assert d.max() < (2e-6 if dtype == np.float32 else 1e-14), f"{d.max():.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_scratch_budget_extremes_and_allocator_refusal(dtype):
    """A slab budget meant as 'unlimited' (INT64_MAX) must not wrap in the slots-per-pass arithmetic; a scratch limit below the default
    request makes the launch halve its slots per pass until it fits (still pair-symmetric, same result to the fp64 order of the slabs); a
    limit below even ONE slot hands the evaluation to the one-sided kernel instead of failing it."""
    n = 60000
    s = _cloud(n, dtype, False, seed=9)
    with _engine(n, dtype, s) as e:
        a0 = e.accel(0.01)
        assert e.last_accel_symmetric() and e.get_option("sym_scratch_limit") == -1
        e.set_option("sym_slab_bytes", (1 << 63) - 1)
        a1 = e.accel(0.01)
        assert e.last_accel_symmetric() and np.array_equal(a1, a0)
    # The limit is on what is ASKED of the allocator (scratch a context already holds is used): fresh contexts, the limit halved from
    # 1 GiB down to 256 KiB. Generous limits leave the launch as it is; below the one-pass request it takes more passes; below what even one
    # slab slot needs the one-sided kernel runs — and at no limit does the evaluation fail or change beyond the summation order.
    flags, results = [], []
    for shift in range(30, 17, -1):
        with _engine(n, dtype, s) as e:
            e.set_option("sym_scratch_limit", 1 << shift)
            assert e.get_option("sym_scratch_limit") == 1 << shift
            results.append(e.accel(0.01))
            flags.append(e.last_accel_symmetric())
            if shift == 18:
                e.set_option("sym_scratch_limit", -1)
                a4 = e.accel(0.01)
                assert e.last_accel_symmetric() and np.array_equal(a4, a0)
    assert flags[0] and not flags[-1] and flags == sorted(flags, reverse=True), flags        # symmetric ... symmetric, one-sided ... one-sided
    assert np.array_equal(results[0], a0)
    t = 2e-6 if dtype == np.float32 else 2e-13               # two fp32 kernels against each other: twice the rounding of one
    for f, r in zip(flags, results):
        assert rel_err(r, a0.astype(np.float64)).max() < ((1e-6 if dtype == np.float32 else 1e-13) if f else t)
    # the two-block entry under the same limits
    import torch
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    na = 30000
    blk = lambda lo, hi: torch.from_numpy(np.vstack([s.pos[lo:hi].T, s.mass[None, lo:hi]])).contiguous().cuda()
    A, B = blk(0, na), blk(na, n)
    res = []
    for limit, sym in ((-1, True), (1 << 18, False)):
        k = nb.HipKernels(0)
        k.set_option("sym_scratch_limit", limit)
        acc_a, acc_b = torch.zeros((3, na), dtype=tdt, device="cuda"), torch.zeros((3, n - na), dtype=tdt, device="cuda")
        k.accel_pair(A, na, B, n - na, 0.01, 1.0, acc_a, acc_b, False, uniform_mass=True)
        assert k.last_accel_symmetric() == sym, limit
        torch.cuda.synchronize()
        res.append((acc_a.cpu().numpy().T.astype(np.float64), acc_b.cpu().numpy().T.astype(np.float64)))
        k.close()
    for ra, rb in res[1:]:
        assert rel_err(ra, res[0][0]).max() < t and rel_err(rb, res[0][1]).max() < t


@pytest.mark.parametrize("dtype", DTYPES)
def test_zero_softening_drops_coincident_pairs(dtype, coracle):
    """eps = 0 with truly coincident particles — in the same set, in different sets, and next to the padding of a ragged last set."""
    n = 4200
    s = _cloud(n, dtype, True, seed=4)
    pos = s.pos.copy()
    pos[17] = pos[3]; pos[2500] = pos[40]; pos[n - 1] = pos[n - 2]
    s = nb.State(pos, s.vel, s.mass)
    with _engine(n, dtype, s) as e:
        a = e.accel(0.0)
        assert e.last_accel_symmetric()
    assert np.isfinite(a).all()
    p64, m64 = pos.astype(np.float64), s.mass.astype(np.float64)
    d = p64[None, :, :] - p64[:400, None, :]
    r2 = (d * d).sum(-1)
    w = np.where(r2 > 0, m64[None, :] / np.where(r2 > 0, r2, 1.0) ** 1.5, 0.0)
    ref = (w[:, :, None] * d).sum(1)
    assert rel_err(a[:400], ref).max() < (5e-5 if dtype == np.float32 else 1e-12)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("budget", [-1, 1 << 21])
def test_fused_potential_of_the_symmetric_kernel(dtype, general, budget):
    """nb_step_energy on a self-interacting set: the potential rides along in the pair-symmetric kernel (both particles of a pair get
    m / r); compared with the separate potential kernel (nb_energy) and with the energies of the one-sided path. budget = 2 MiB: the
    evaluation takes several slab passes (accelerations AND potential collected in fp64 between them)."""
    n = 40000
    s = _cloud(n, dtype, general, seed=6)
    with _engine(n, dtype, s) as e:
        e.set_option("sym_slab_bytes", budget)
        k1, u1 = e.step_with_energy(1.0 / 256, 3, 0.01)
        kk, uu = e.energy(0.01)
        st1 = e.get_state()
    with _engine(n, dtype, s, force_sym=False) as e:
        e.set_option("pair_symmetry", 0)
        k0, u0 = e.step_with_energy(1.0 / 256, 3, 0.01)
        st0 = e.get_state()
    tol = 3e-6 if dtype == np.float32 else 1e-12
    assert abs(u1[-1] / uu - 1) < tol and abs(k1[-1] / kk - 1) < tol
    assert np.abs(np.asarray(u1) / np.asarray(u0) - 1).max() < tol and np.abs(np.asarray(k1) / np.asarray(k0) - 1).max() < tol
    assert np.abs(st1.pos - st0.pos).max() < (1e-6 if dtype == np.float32 else 1e-13)


@pytest.mark.parametrize("dtype", DTYPES)
def test_accumulate_flag_and_device_pointer_entry(dtype):
    """nb_accel_dev with identical sink and source blocks takes the pair-symmetric kernel too (the own-chunk launch of a sharded run);
    NB_ACCUMULATE adds to what the arrays hold."""
    import torch
    n = 50000
    s = _cloud(n, dtype, False, seed=3)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    k = nb.HipKernels(0)
    pm = torch.from_numpy(np.vstack([s.pos.T, s.mass[None, :]])).contiguous().cuda()
    acc = torch.zeros((3, n), dtype=tdt, device="cuda")
    k.accel(pm, n, pm, n, 0.01, 1.0, acc, False, uniform_mass=True)
    assert k.last_accel_symmetric()
    a1 = acc.clone()
    k.accel(pm, n, pm, n, 0.01, 1.0, acc, True, uniform_mass=True)
    torch.cuda.synchronize()
    assert torch.allclose(acc, 2 * a1, rtol=1e-6 if dtype == np.float32 else 1e-15, atol=0)
    other = pm.clone()                                   # same values, different arrays: sinks != sources -> one-sided kernel
    acc2 = torch.zeros_like(acc)
    k.accel(pm, n, other, n, 0.01, 1.0, acc2, False, uniform_mass=True)
    assert not k.last_accel_symmetric()
    torch.cuda.synchronize()
    d = (acc2.double() - a1.double()).norm(dim=0) / a1.double().norm(dim=0)
    assert float(d.max()) < (3e-6 if dtype == np.float32 else 2e-13)
    k.close()


def test_headline_size_fp32_sampled(coracle):
    """BASELINE configs[2]'s size on one GPU through the pair-symmetric kernel (12 sinks per lane, 342 sets): sampled sinks against the
    C oracle over all 2^20 sources, Newton's third law, and the option switch."""
    n = 1 << 20
    s = nb.plummer_sphere(n, 0, np.float32)
    with nb.Engine(n, np.float32) as e:
        e.set_state(s)
        a = e.accel(0.01)
        assert e.last_accel_symmetric()
    idx = _sample(n, 3000)              # ~3000 sinks x 2^20 sources: half a second of the C oracle
    ref = coracle.accel(s.pos.astype(np.float64), s.mass.astype(np.float64), 0.01, sinks=s.pos[idx].astype(np.float64))
    assert rel_err(a[idx], ref).max() < 1e-6
    m = s.mass.astype(np.float64)[:, None]
    assert np.abs((m * a).sum(0)).max() / np.abs(m * a).sum(0).max() < 1e-9


# ---- two different blocks, every A-B pair once (nb_accel_pair_dev): the hop of the pair-symmetric multi-GPU schedule ------------------------
def _phi_ref(snk, src, m, eps):
    out = np.empty(len(snk))
    for a in range(0, len(snk), 512):
        d = src[None, :, :] - snk[a:a + 512, None, :]
        out[a:a + 512] = (m[None, :] / np.sqrt((d * d).sum(-1) + eps * eps)).sum(1)
    return out


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("na,nb_,force", [(40000, 33001, False), (3000, 7001, True), (1, 5000, True), (20000, 1025, True), (131072, 131072, False),
                                          (3000, 7001, False), (30000, 26001, "passes")])
def test_two_block_entry_matches_oracle(dtype, general, na, nb_, force, coracle):
    """acc_a (+)= force of B on A, acc_b = force of A on B, from ONE evaluation of every A-B pair; ragged and very unequal block sizes, the
    accumulate flag of A, B overwritten; small blocks forced through the symmetric kernel with sym_min_n = 0, or — not forced — through the
    fallback (the one-sided kernel in both directions); "passes": a 1 MiB slab budget, i.e. several passes over ranges of A's sets with B's
    sums accumulated between them."""
    import torch
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    s = _cloud(na + nb_, dtype, general, seed=21)
    if not general:
        s = nb.State(s.pos, s.vel, np.full(na + nb_, 0.37 / (na + nb_), dtype))
    k = nb.HipKernels(0)
    if force:
        k.set_option("sym_min_n", 0)
    if force == "passes":
        k.set_option("sym_slab_bytes", 1 << 20)
    blk = lambda lo, hi: torch.from_numpy(np.vstack([s.pos[lo:hi].T, s.mass[None, lo:hi]])).contiguous().cuda()
    A, B = blk(0, na), blk(na, na + nb_)
    acc_a = torch.full((3, na), 7.0, dtype=tdt, device="cuda")
    acc_b = torch.full((3, nb_), -3.0, dtype=tdt, device="cuda")
    k.accel_pair(A, na, B, nb_, 0.01, 1.3, acc_a, acc_b, False, uniform_mass=not general)
    took_sym = k.last_accel_symmetric()
    assert took_sym == (min(na, nb_) >= 1 and (bool(force) or 2 * min(na, nb_) >= (18432 if dtype == np.float32 else 10240)))
    first = acc_a.clone()
    k.accel_pair(A, na, B, nb_, 0.01, 1.3, acc_a, acc_b, True, uniform_mass=not general)      # A accumulates, B is overwritten
    torch.cuda.synchronize()
    p64, m64 = s.pos.astype(np.float64), s.mass.astype(np.float64)
    ia, ib = (np.arange(na) if na <= 4000 else _sample(na, 128)), (np.arange(nb_) if nb_ <= 4000 else _sample(nb_, 128))
    ref_a = coracle.accel(p64[na:], m64[na:], 0.01, 1.3, sinks=p64[:na][ia])
    ref_b = coracle.accel(p64[:na], m64[:na], 0.01, 1.3, sinks=p64[na:][ib])
    ga, gb = first.cpu().numpy().T[ia], acc_b.cpu().numpy().T[ib]
    assert rel_err(ga, ref_a).max() < tol(dtype, min(na, nb_)) and rel_err(gb, ref_b).max() < tol(dtype, min(na, nb_))
    assert torch.allclose(acc_a, 2 * first, rtol=1e-6 if dtype == np.float32 else 1e-15, atol=0)
    # momentum exchanged between the blocks cancels: sum m_a a_a = - sum m_b a_b (exactly the same pair terms on both sides)
    pa = (m64[:na, None] * first.cpu().numpy().T.astype(np.float64)).sum(0)
    pb = (m64[na:, None] * acc_b.cpu().numpy().T.astype(np.float64)).sum(0)
    scale = np.abs(m64[:na, None] * first.cpu().numpy().T).sum(0).max()
    # (fp32: up to the final rounding of the stored accelerations — 6e-8 of a single value when a block is one particle)
    assert np.abs(pa + pb).max() / scale < (2e-7 if dtype == np.float32 else 1e-15) * (1 if took_sym else 50)
    k.close()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("budget", [-1, 1 << 20])
def test_two_block_entry_fused_potential(dtype, general, budget):
    """budget = 1 MiB: several passes over ranges of A's sets, B's sums (accelerations and potential) accumulated between them."""
    import torch
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    na, nb_ = 9000, 12001
    s = _cloud(na + nb_, dtype, general, seed=23)
    k = nb.HipKernels(0)
    k.set_option("sym_min_n", 0)
    k.set_option("sym_slab_bytes", budget)
    blk = lambda lo, hi: torch.from_numpy(np.vstack([s.pos[lo:hi].T, s.mass[None, lo:hi]])).contiguous().cuda()
    A, B = blk(0, na), blk(na, na + nb_)
    acc_a, acc_b = torch.zeros((3, na), dtype=tdt, device="cuda"), torch.zeros((3, nb_), dtype=tdt, device="cuda")
    phi_a = torch.full((na,), 5.0, dtype=torch.float64, device="cuda")
    phi_b = torch.full((nb_,), 9.0, dtype=torch.float64, device="cuda")
    k.accel_pair(A, na, B, nb_, 0.02, 1.0, acc_a, acc_b, True, uniform_mass=not general, phi_a=phi_a, phi_b=phi_b)
    assert k.last_accel_symmetric()
    acc_a0, acc_b0 = torch.zeros_like(acc_a), torch.zeros_like(acc_b)
    k.accel_pair(A, na, B, nb_, 0.02, 1.0, acc_a0, acc_b0, False, uniform_mass=not general)
    torch.cuda.synchronize()
    p64, m64 = s.pos.astype(np.float64), s.mass.astype(np.float64)
    tol = 2e-6 if dtype == np.float32 else 1e-12
    assert np.abs((phi_a.cpu().numpy() - 5.0) / _phi_ref(p64[:na], p64[na:], m64[na:], 0.02) - 1).max() < tol      # accumulated onto the 5.0
    assert np.abs(phi_b.cpu().numpy() / _phi_ref(p64[na:], p64[:na], m64[:na], 0.02) - 1).max() < tol             # overwritten
    # the forces of the fused variant: the same pair terms in another summation order (other tile and set sizes)
    ftol = 3e-6 if dtype == np.float32 else 2e-13
    assert rel_err(acc_a.cpu().numpy().T, acc_a0.cpu().numpy().T).max() < ftol and rel_err(acc_b.cpu().numpy().T, acc_b0.cpu().numpy().T).max() < ftol
    k.close()


def test_two_block_entry_rejects_bad_arguments():
    import torch
    k = nb.HipKernels(0)
    A = torch.zeros((4, 100), device="cuda")
    acc = torch.zeros((3, 100), device="cuda")
    with pytest.raises(RuntimeError, match="different blocks"):
        k.accel_pair(A, 100, A, 100, 0.01, 1.0, acc, acc.clone(), False)
    with pytest.raises(RuntimeError, match="go together"):
        k.accel_pair(A, 100, A.clone(), 100, 0.01, 1.0, acc, acc.clone(), False, phi_a=torch.zeros(100, dtype=torch.float64, device="cuda"))
    k.close()


@pytest.mark.parametrize("name", ["n40000_f32", "n40000_f64_general", "n100001_f32_general", "n100001_f64"])
def test_symmetric_kernel_matches_committed_golden_vectors(name):
    """The committed fixtures of this size class (tests/golden/accel_golden_large.npz: NumPy oracle, cross-checked against the C restatement when they
    were generated) — no oracle at run time."""
    from pathlib import Path
    from test_golden import large_inputs
    g = np.load(Path(__file__).resolve().parent / "golden" / "accel_golden_large.npz")
    pos, mass = large_inputs(g, name)
    n, dtype = len(mass), pos.dtype.type
    with nb.Engine(n, dtype) as e:
        e.set_state(nb.State(pos, np.zeros_like(pos), mass))
        a = e.accel(float(g[f"{name}_eps"]), float(g[f"{name}_G"]))
        assert e.last_accel_symmetric()
    err = rel_err(a[g[f"{name}_idx"]], g[f"{name}_acc"])
    assert err.max() < tol(dtype, n), f"{name}: {err.max():.3e}"


BLOCKS_CODE = """
import sys, numpy as np
sys.path.insert(0, %r)
import bachelor_thesis_n_body_amd as nb
for dtype in (np.float32, np.float64):
    for n, general in ((40000, False), (70001, True)):
        s = nb.plummer_sphere(n, 13, dtype)
        if general:
            s = nb.State(s.pos, s.vel, (s.mass * np.random.default_rng(3).uniform(0.5, 2.0, n)).astype(dtype))
        with nb.Engine(n, dtype) as e:
            e.set_state(s)
            a = e.accel(0.01)
            assert e.last_accel_symmetric()
            k, u = e.step_with_energy(0.0, 1, 0.01)
        np.save(sys.argv[1] + "/a_%%s_%%d.npy" %% (np.dtype(dtype).name, n), a)
        np.save(sys.argv[1] + "/u_%%s_%%d.npy" %% (np.dtype(dtype).name, n), np.array([u[0]]))
print("done")
"""


@pytest.mark.parametrize("blocks", [1, 7, 333, 1000, 5001])
def test_any_block_count_gives_the_same_sums(blocks, tmp_path):
    """The equal-share deal of steps to blocks (accel_sym_*.hip) for block counts the built-in rule never picks — one block, a prime, counts that
    cut almost every tile, more blocks than tiles have steps to spare: the same pair terms, so the same accelerations and potential to the
    rounding of another summation order. (NB_F32_SYM_BLOCKS / NB_F64_SYM_BLOCKS are read once per process: child processes.)"""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = str(Path(__file__).resolve().parent.parent)
    res = {}
    for tag, env_blocks in (("default", "0"), ("forced", str(blocks))):
        d = tmp_path / tag
        d.mkdir()
        env = dict(os.environ, NB_F32_SYM_BLOCKS=env_blocks, NB_F64_SYM_BLOCKS=env_blocks)
        out = subprocess.run([sys.executable, "-c", BLOCKS_CODE % root, str(d)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "done" in out.stdout, out.stderr[-2000:]
        res[tag] = {f.name: np.load(f) for f in d.glob("*.npy")}
    for name, ref in res["default"].items():
        got = res["forced"][name]
        f32 = "float32" in name
        if name.startswith("a_"):
            assert rel_err(got, ref.astype(np.float64)).max() < (1e-6 if f32 else 1e-13), name
        else:
            assert abs(got[0] / ref[0] - 1) < (1e-7 if f32 else 1e-14), name

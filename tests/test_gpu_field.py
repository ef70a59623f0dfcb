"""GPU parity of the field chain (CIC deposit, overdensity, CIC-window deconvolution, binned P(k)) through
include/nbody_field.h against (1) the reference's own delta fields (tests/golden/concept) and (2) the pinned oracle.

Tolerances: the grid values are fp64 sums of <= a few hundred O(1) weights per cell accumulated with global atomics
(order not fixed) and pass through two 32^3 FFTs: |difference| <= 1e-12 absolute on O(1) values. P(k) bins: rtol 1e-9 + 1e-12 of the
largest bin (FFT round-off scales with the largest mode). Histogram COUNTS are
integers and must be bit-exact (|k| and the bin edges are computed with the reference's exact arithmetic)."""
from pathlib import Path

import numpy as np
import pytest

import bachelor_thesis_n_body_amd as nb
from bachelor_thesis_n_body_amd import field, snapshot as sn
from oracle import field_oracle as fo

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden" / "concept"


@pytest.mark.parametrize("snap,npy,sub", [("test_id0200_snapshot_a=1.00.hdf5", "delta_test_id-1.npy", False),
                                          ("train_id0000_snapshot_a=1.00.hdf5", "delta_train_id-1.npy", True)])
def test_delta_field_reproduces_reference_outputs(snap, npy, sub):
    """The thesis pipeline's per-snapshot body (Feeder_pipe.py:174-195) on the GPU vs the .npy it wrote."""
    pos, box = sn.read_positions(str(G / snap)), sn.read_boxsize(str(G / snap))
    got = field.delta_field(pos, box, 32, subtract_mean=sub)
    want = np.load(G / npy)
    assert got.shape == want.shape and got.dtype == np.float64
    assert np.abs(got - want).max() <= 1e-12


@pytest.mark.parametrize("n,ngrid,box", [(0, 8, 10.0), (1, 4, 1.0), (1000, 16, 100.0), (50000, 32, 500.0), (200000, 64, 250.0)])
def test_cic_density_matches_oracle(n, ngrid, box):
    rng = np.random.default_rng(n + ngrid)
    pos = rng.uniform(0, box, (n, 3))
    if n > 10:
        pos[0] = [0.0, 0.0, 0.0]                      # on a grid point
        pos[1] = [box * (1 - 1e-12)] * 3              # wraps around the upper edge
        pos[2] = [box / ngrid * 3, box / ngrid * 5.5, box * 0.999999]
    got = field.cic_density(pos, box, ngrid)
    want = fo.cic_density(pos, box, ngrid)
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, want.max())
    assert got.sum() == pytest.approx(float(n), rel=1e-13, abs=1e-13)


def test_cic_out_of_box_positions_follow_python_semantics():
    """Negative / beyond-box coordinates: int() truncates toward zero, % wraps like Python's (FraP_kTilDelta.py:100-109)."""
    pos = np.array([[-0.3, 10.7, 25.0], [-9.99, -0.01, 0.5], [31.4, -15.2, 100.0]])
    got = field.cic_density(pos, 10.0, 8)
    np.testing.assert_allclose(got, fo.cic_density(pos, 10.0, 8), rtol=0, atol=1e-14)


def test_compute_delta_and_deconvolve_match_oracle():
    rng = np.random.default_rng(5)
    dens = rng.uniform(0.0, 3.0, (24, 24, 24))
    d = field.compute_delta(dens)
    np.testing.assert_allclose(d, fo.compute_delta(dens), rtol=0, atol=1e-14)
    dec = field.deconvolve_cic(d, 300.0, 24)
    np.testing.assert_allclose(dec, fo.deconvolve_cic(fo.compute_delta(dens), 300.0, 24), rtol=0, atol=1e-12)
    # deconvolving a constant field leaves it unchanged (W(k=0) = 1)
    np.testing.assert_allclose(field.deconvolve_cic(np.full((8, 8, 8), 2.5), 10.0, 8), 2.5, rtol=0, atol=1e-14)


@pytest.mark.parametrize("deconv", [False, True])
@pytest.mark.parametrize("use_fac", [False, True])
def test_power_spectrum_matches_oracle_on_reference_snapshot(deconv, use_fac):
    """The four variants of Power_compare.py:211-235 on a CONCEPT snapshot of the reference (32^3 particles, N = 32)."""
    f = str(G / "test_id0200_snapshot_a=1.00.hdf5")
    pos, L = sn.read_positions(f), sn.read_boxsize(f)
    k, p, c = field.power_spectrum(pos, L, 32, deconvolve=deconv, use_fac=use_fac)
    k0, p0, c0 = fo.power_spectrum(pos, L, 32, deconvolve=deconv, use_fac=use_fac)
    np.testing.assert_array_equal(c, c0)               # mode counts per bin: bit-exact
    np.testing.assert_array_equal(k, k0)
    np.testing.assert_allclose(p, p0, rtol=1e-9, atol=1e-12 * p0.max())   # FFT round-off scales with the largest mode
    assert c.sum() == 32 ** 3 - 1
    k2, p2 = field.compute_power_spectrum_from_snapshot(f, L, 32, deconvolve=deconv, use_fac=use_fac)
    np.testing.assert_array_equal(k2, k) and np.testing.assert_allclose(p2, p, rtol=1e-12)


def test_power_spectrum_larger_mesh_counts_exact():
    rng = np.random.default_rng(11)
    pos = rng.uniform(0, 1000.0, (300000, 3))
    k, p, c = field.power_spectrum(pos, 1000.0, 96, deconvolve=True)
    k0, p0, c0 = fo.power_spectrum(pos, 1000.0, 96, deconvolve=True)
    np.testing.assert_array_equal(c, c0)
    np.testing.assert_allclose(p, p0, rtol=1e-8, atol=1e-12 * p0.max())


def test_plan_reuse_and_engine_state_roundtrip(tmp_path):
    """A plan is reusable across snapshots, and engine state written by write_snapshot feeds the chain."""
    s = nb.plummer_sphere(20000, 3)
    box = 40.0
    pos = (s.pos + box / 2) % box
    out = tmp_path / "engine_snapshot.hdf5"
    sn.write_snapshot(str(out), pos, s.vel * s.mass[:, None], box)
    with field.FieldPlan(32, box) as plan:
        a = plan.delta_field(sn.read_positions(str(out)))
        b = plan.delta_field(pos, subtract_mean=True)
        c = plan.delta_field(sn.read_positions(str(out)))
    np.testing.assert_allclose(a, fo.delta_field(pos, box, 32, False), rtol=1e-11, atol=1e-11)   # peak delta ~ 40 here
    np.testing.assert_allclose(b, fo.delta_field(pos, box, 32, True), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(a, c, rtol=1e-11, atol=1e-11)   # atomics: order-dependent only at round-off (x FFT gain)
    assert abs(b.mean()) < 1e-14


@pytest.mark.parametrize("nbins,grid_dtype,tol", [(100, np.float64, 1e-10), (300, np.float32, 2e-5)])
def test_plotting_script_estimator_variants(nbins, grid_dtype, tol):
    """The P(k) estimator of the thesis' two plotting scripts (1LPTvs2LPT/1vs2tlptplotter.py:81-116: float64 grid, 100 bin edges;
    Different_Sizes/Plotter.py:148-202: float32 grid, 300 edges): window clipped at 1e-3, log bins up to the corner mode — on a reference
    snapshot, GPU chain vs the line-by-line restatement."""
    pos = sn.read_positions(str(G / "test_id0200_snapshot_a=1.00.hdf5"))
    for nmesh in (32, 48):
        k, p = field.power_spectrum_plotter(pos, 500.0, nmesh, nbins, grid_dtype)
        k0, p0 = fo.power_spectrum_plotter(pos, 500.0, nmesh, nbins, grid_dtype)
        np.testing.assert_allclose(k, k0, rtol=1e-14)
        np.testing.assert_allclose(p, p0, rtol=tol)
    # the clip only matters next to the Nyquist planes: away from them this is the deconvolved estimator of Power_compare.py
    k1, p1, _ = field.power_spectrum(pos, 500.0, 32, deconvolve=True, nbins=100)
    k2, p2 = field.power_spectrum_plotter(pos, 500.0, 32, 100, np.float64, clip=1e-30)
    np.testing.assert_allclose(k1, k2, rtol=1e-14)
    np.testing.assert_allclose(p1, p2, rtol=1e-9)


@pytest.mark.parametrize("ngrid,interlace", [(32, True), (32, False), (27, True), (48, True)])
def test_power_modes_match_the_oracle(ngrid, interlace):
    """CONCEPT's estimator settings (CIC, deconvolved, interlaced) per mode: both lattices through one complex FFT on the GPU against the NumPy
    restatement with two host FFTs, on a reference snapshot. Odd grid sizes (27) have no Nyquist plane; 48 != the particle lattice."""
    from bachelor_thesis_n_body_amd import cosmo
    pos, box = sn.read_positions(str(G / "test_id0200_snapshot_a=1.00.hdf5")), sn.read_boxsize(str(G / "test_id0200_snapshot_a=1.00.hdf5"))
    k_got, p_got = cosmo.power_modes(pos, box, ngrid, interlace=interlace)
    k_ref, p_ref = fo.power_modes(pos, box, ngrid, interlace=interlace)
    assert k_got.shape == k_ref.shape == (ngrid ** 3 - 1,)
    np.testing.assert_allclose(k_got, k_ref, rtol=1e-14)
    assert np.abs(p_got - p_ref).max() <= 1e-10 * p_ref.max()
    big = p_ref > 1e-6 * p_ref.max()
    np.testing.assert_allclose(p_got[big], p_ref[big], rtol=1e-7)

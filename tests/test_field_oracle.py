"""Pins the field-chain oracle (oracle/field_oracle.py) against the reference's OWN data: CONCEPT snapshots and the
delta fields the thesis pipeline produced from them (tests/golden/concept/, copied from
TheFeedingPipeline/Datafolder4 — see the README there). Also covers the HDF5 snapshot reader/writer."""
import glob
import os
from pathlib import Path

import numpy as np
import pytest

from oracle import field_oracle as fo
from bachelor_thesis_n_body_amd import snapshot as sn

G = Path(__file__).resolve().parent / "golden" / "concept"
REF = Path("/root/reference/TheFeedingPipeline/Datafolder4")
NGRID = 32     # BoxParams.txt: "0 500 32"


@pytest.mark.parametrize("snap,npy,sub", [("test_id0200_snapshot_a=1.00.hdf5", "delta_test_id-1.npy", False),
                                          ("train_id0000_snapshot_a=1.00.hdf5", "delta_train_id-1.npy", True)])
def test_oracle_reproduces_reference_delta_fields(snap, npy, sub):
    pos = sn.read_positions(str(G / snap))
    box = sn.read_boxsize(str(G / snap))
    assert pos.shape == (32768, 3) and pos.dtype == np.float64 and box == 500.0
    got = fo.delta_field(pos, box, NGRID, subtract_mean=sub)
    want = np.load(G / npy)
    assert got.shape == want.shape == (32, 32, 32)
    assert np.abs(got - want).max() <= 1e-13


@pytest.mark.skipif(not REF.exists(), reason="reference tree only exists in the build container")
def test_oracle_reproduces_more_reference_pairs():
    """Spot-check further pairs straight from the reference tree (build container only): ids 2-6 and the last."""
    for root, out, prefix, sub in ((REF / "Test_Data_concept", REF / "Test data_CONCEPT", "delta_test_id-", False),
                                  (REF / "TrainingVal_Concept", REF / "Training & val data_CONCEPT", "delta_train_id-", True)):
        files = sorted(glob.glob(str(root / "**" / "*.hdf5"), recursive=True))
        for idx in (2, 3, 4, 5, 6, len(files)):
            pos = sn.read_positions(files[idx - 1])
            got = fo.delta_field(pos, sn.read_boxsize(files[idx - 1]), NGRID, subtract_mean=sub)
            assert np.abs(got - np.load(out / f"{prefix}{idx}.npy")).max() <= 1e-13, (prefix, idx)


def test_cic_density_properties():
    rng = np.random.default_rng(0)
    pos = rng.uniform(0, 100.0, (5000, 3))
    g = fo.cic_density(pos, 100.0, 16)
    assert g.sum() == pytest.approx(5000.0, rel=1e-13)          # weights of one particle sum to 1
    assert g.min() >= 0
    # a particle exactly on a grid point deposits everything into that cell; periodic wrap at the upper edge
    one = fo.cic_density(np.array([[25.0, 50.0, 93.75]]), 100.0, 16)
    assert one[4, 8, 15] == 1.0 and one.sum() == 1.0
    edge = fo.cic_density(np.array([[99.0, 99.0, 99.0]]), 100.0, 16)
    assert edge[0, 0, 0] > 0 and edge[15, 15, 15] > 0 and edge.sum() == pytest.approx(1.0)
    # the scalar-loop statement of the reference (FraP_kTilDelta.py:93-123), literally
    N, L = 8, 10.0
    p = rng.uniform(0, L, (200, 3))
    ref = np.zeros((N, N, N))
    cs = L / N
    for q in p:
        i, j, k = int(q[0] / cs) % N, int(q[1] / cs) % N, int(q[2] / cs) % N
        dx, dy, dz = (q[0] % cs) / cs, (q[1] % cs) / cs, (q[2] % cs) / cs
        for di in (0, 1):
            for dj in (0, 1):
                for dk in (0, 1):
                    w = ((1 - dx) if di == 0 else dx) * ((1 - dy) if dj == 0 else dy) * ((1 - dz) if dk == 0 else dz)
                    ref[(i + di) % N, (j + dj) % N, (k + dk) % N] += w
    np.testing.assert_array_equal(fo.cic_density(p, L, N), ref)   # bit-identical: same additions in the same order


def test_power_spectrum_shape_and_white_noise_level():
    """Poisson-distributed particles: P(k) ~ V/N_p (shot noise) at all k; binning keeps every k>0 mode once."""
    rng = np.random.default_rng(1)
    L, n, npart = 200.0, 16, 16 ** 3        # one particle per cell: the `use_fac` branch (delta -= 1) is then an overdensity too
    pos = rng.uniform(0, L, (npart, 3))
    k, p, c = fo.power_spectrum(pos, L, n, deconvolve=True)
    assert c.sum() == n ** 3 - 1
    assert np.all(np.diff(k) > 0)
    lo = k < 0.5 * np.pi * n / L
    assert np.average(p[lo], weights=c[lo]) == pytest.approx(L ** 3 / npart, rel=0.3)
    # use_fac variant then differs only by normalisation: fac^2 = N^6 * 2*pi / V  vs  N^6 / V (Power_compare.py:170-176)
    k2, p2, _ = fo.power_spectrum(pos, L, n, deconvolve=True, use_fac=True)
    np.testing.assert_allclose(p2 * 2 * np.pi, p, rtol=1e-10)


def test_snapshot_reader_and_writer_roundtrip(tmp_path):
    src = G / "test_id0200_snapshot_a=1.00.hdf5"
    raw = src.read_bytes()
    pos = sn.read_positions(str(src))
    # pos is a contiguous little-endian f64 (32768,3) block at byte offset 10240, mom follows
    np.testing.assert_array_equal(pos, np.frombuffer(raw, "<f8", 32768 * 3, 10240).reshape(-1, 3))
    np.testing.assert_array_equal(sn.read_momenta(str(src)), np.frombuffer(raw, "<f8", 32768 * 3, 10240 + 786432).reshape(-1, 3))
    a = sn.read_attrs(str(src))
    assert a["boxsize"] == 500.0 and a["a"] == 1.0
    assert sn.read_attrs(str(src), "components/matter")["N"] == 32768
    out = tmp_path / "snap.hdf5"
    sn.write_snapshot(str(out), pos[:100], -pos[:100], 123.5)
    np.testing.assert_array_equal(sn.read_positions(str(out)), pos[:100])
    np.testing.assert_array_equal(sn.read_momenta(str(out)), -pos[:100])
    assert sn.read_boxsize(str(out)) == 123.5 and sn.read_attrs(str(out))["unit length"] == "Mpc"
    with pytest.raises(sn.HDF5Error):
        bad = tmp_path / "x.hdf5"
        bad.write_bytes(b"not hdf5 at all")
        sn.read_positions(str(bad))
    with pytest.raises(KeyError):
        sn.read_dataset(str(src), "components/nope/pos")


def test_writer_reproduces_the_byte_layout_of_a_reference_snapshot(tmp_path):
    """A reference snapshot pushed through write_snapshot comes out with the reference's own layout: same file size
    (1 583 104 B for 32^3), `pos` at byte 10 240, `mom` at 10 240 + 786 432, both bit-identical to the source file's data blocks —
    the writer is pinned by the reference's bytes, not only by this package's reader. The attributes the thesis scripts and a resumed
    CONCEPT-style run need (boxsize, a, H0, units, mass, N) survive too."""
    src = G / "test_id0200_snapshot_a=1.00.hdf5"
    raw = src.read_bytes()
    attrs, comp = sn.read_attrs(str(src)), sn.read_attrs(str(src), "components/matter")
    out = tmp_path / "again.hdf5"
    sn.write_snapshot(str(out), sn.read_positions(str(src)), sn.read_momenta(str(src)), attrs["boxsize"], mass=comp["mass"],
                      extra_attrs={"a": attrs["a"], "H0": attrs["H0"]})
    new = out.read_bytes()
    assert len(new) == len(raw) == 1583104
    assert new[10240:] == raw[10240:]                         # pos block + mom block, byte for byte
    assert new[:8] == raw[:8] == b"\x89HDF\r\n\x1a\n"
    a2, c2 = sn.read_attrs(str(out)), sn.read_attrs(str(out), "components/matter")
    assert a2["boxsize"] == attrs["boxsize"] and a2["a"] == attrs["a"] and a2["H0"] == attrs["H0"]
    assert c2["mass"] == comp["mass"] and c2["N"] == comp["N"]
    np.testing.assert_array_equal(sn.read_positions(str(out)), sn.read_positions(str(src)))
    # a different particle number keeps the layout rule: metadata first, arrays behind, nothing after
    small = tmp_path / "small.hdf5"
    sn.write_snapshot(str(small), np.zeros((10, 3)), np.ones((10, 3)), 1.0)
    assert small.stat().st_size == 10240 + 2 * 240 and small.read_bytes()[10240 + 240:] == np.ones((10, 3)).tobytes()


def test_particle_mass_and_hubble_rate_match_the_reference_snapshot_attributes():
    """The reference's snapshots carry CONCEPT's own numbers for the unit system: root attribute H0 [1/Gyr] and components/matter `mass`
    [1e10 M_sun] — the mean matter density of the box divided over the particles. `Cosmology` reproduces both from the parameter file's
    H0 = 67.36 km/s/Mpc, omega_b, omega_cdm (NN.param:87-89) — G, the unit conversions and Omega_m in one number."""
    from bachelor_thesis_n_body_amd import cosmo
    src = G / "test_id0200_snapshot_a=1.00.hdf5"
    c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)
    a = sn.read_attrs(str(src))
    assert c.H0 == pytest.approx(a["H0"], rel=1e-9)
    assert c.Om == pytest.approx(a["Ωb"] + a["Ωcdm"], rel=1e-12)
    assert c.particle_mass(500.0, 32 ** 3) == pytest.approx(sn.read_attrs(str(src), "components/matter")["mass"], rel=2e-7)


def test_momentum_fixture_obeys_linear_continuity():
    """p3m_farm_test100_momentum.txt (reference `mom` datasets): at the lowest k the divergence spectrum of the momentum density is
    (a^2 H f)^2 P_delta — the reference's momenta ARE a^2 m dx/dt in Mpc / Gyr, the variable this library integrates."""
    from bachelor_thesis_n_body_amd import cosmo
    c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)
    pdiv = np.loadtxt(G / "p3m_farm_test100_momentum.txt")[:, 4]
    pdel = np.loadtxt(G / "p3m_farm_test100_pk.txt")[:, 4]
    lin = (float(c.H(1.0)) * c.growth(1.0)[1]) ** 2
    assert pdiv[0] / pdel[0] == pytest.approx(lin, rel=0.03) and pdiv[1] / pdel[1] == pytest.approx(lin, rel=0.06)


def test_plotter_estimator_restatement_against_the_power_compare_one():
    """oracle.power_spectrum_plotter (1vs2tlptplotter.py:81-116 / Plotter.py:148-202) vs oracle.power_spectrum (Power_compare.py): same
    deposit, same bins for nbins = 300; they differ only where the clipped window differs from the plain one (|k| next to the Nyquist planes:
    W >= 1e-3 everywhere for CIC on this grid except the corner region)."""
    pos = sn.read_positions(str(G / "test_id0200_snapshot_a=1.00.hdf5"))
    k1, p1, _ = fo.power_spectrum(pos, 500.0, 32, deconvolve=True, nbins=300)
    k2, p2 = fo.power_spectrum_plotter(pos, 500.0, 32, 300, np.float64, clip=1e-30)
    np.testing.assert_allclose(k1, k2, rtol=1e-14)
    np.testing.assert_allclose(p1, p2, rtol=1e-9)
    k3, p3 = fo.power_spectrum_plotter(pos, 500.0, 32, 300, np.float32)          # float32 grid: 1e-7-level differences only
    np.testing.assert_allclose(p3[: len(p3) // 2], p2[: len(p3) // 2], rtol=1e-4)

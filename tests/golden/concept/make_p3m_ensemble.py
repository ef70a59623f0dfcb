"""Ensemble power spectrum of the reference's P3M farm output — generator of p3m_farm_ensemble_pk.txt (runs only where
/root/reference exists). The reference holds 300 independent 32^3 `p3m` runs (TheFeedingPipeline/Datafolder4/
{Test_Data_concept,TrainingVal_Concept}/id*/NNPM32_As*/snapshots/snapshot_a=1.00.hdf5; NN.param; seeds and A_s drawn per job in
NeuNetværk_test.py:100-137). For every snapshot: CIC-deposited, deconvolved, interlaced P(k) per mode on a 32^3 grid (the
estimator settings of NN.param:28-45), averaged in 8 linear |k| bins, divided by the job's A_s and multiplied by
A_ref = 2.105e-9 (NN.param:92). Output columns: k_lo k_hi modes mean(k) mean(P) std(P) n_snapshots.
The estimator is the NumPy oracle (oracle/mesh_oracle.cic_deposit); the reader is the package's host-only HDF5 reader."""
import glob, os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from bachelor_thesis_n_body_amd import snapshot
from oracle import mesh_oracle as mo

REF = "/root/reference/TheFeedingPipeline/Datafolder4"
NG, L, A_REF = 32, 500.0, 2.105e-9
EDGES = np.linspace(2 * np.pi / L * 0.5, np.pi * NG / L, 9)


def power_modes(pos):
    h = L / NG
    ones = np.ones(len(pos))
    d1 = np.fft.fftn(mo.cic_deposit(pos, ones, L, NG, 0.0) / (len(pos) / NG ** 3) - 1.0)
    d2 = np.fft.fftn(mo.cic_deposit(pos, ones, L, NG, 0.5) / (len(pos) / NG ** 3) - 1.0)
    kv = 2 * np.pi * np.fft.fftfreq(NG, d=h)
    kx, ky, kz = np.meshgrid(kv, kv, kv, indexing="ij")
    d = 0.5 * (d1 + np.exp(-0.5j * h * (kx + ky + kz)) * d2)
    w = np.sinc(kv * h / (2 * np.pi)) ** 2
    W = w[:, None, None] * w[None, :, None] * w[None, None, :]
    pk = np.abs(d / W) ** 2 * L ** 3 / NG ** 6
    return np.sqrt(kx ** 2 + ky ** 2 + kz ** 2).ravel(), pk.ravel()


def binned(pos):
    k, p = power_modes(pos)
    which = np.digitize(k, EDGES) - 1
    ok = (which >= 0) & (which < len(EDGES) - 1)
    cnt = np.bincount(which[ok], minlength=len(EDGES) - 1)
    return (np.bincount(which[ok], weights=k[ok], minlength=len(EDGES) - 1) / cnt,
            np.bincount(which[ok], weights=p[ok], minlength=len(EDGES) - 1) / cnt, cnt)


if __name__ == "__main__":
    files = sorted(glob.glob(os.path.join(REF, "*oncept", "id*", "NNPM32_As*", "snapshots", "snapshot_a=1.00.hdf5")))
    rows = []
    for f in files:
        a_s = float(re.search(r"A_s([0-9.e+-]+)", f).group(1))
        kb, pb, cnt = binned(snapshot.read_positions(f))
        rows.append(pb * (A_REF / a_s))
    rows = np.array(rows)
    out = np.column_stack([EDGES[:-1], EDGES[1:], cnt, kb, rows.mean(0), rows.std(0, ddof=1), np.full(len(kb), len(rows))])
    here = os.path.dirname(os.path.abspath(__file__))
    np.savetxt(os.path.join(here, "p3m_farm_ensemble_pk.txt"), out,
               header="k_lo k_hi modes mean_k mean_P[Mpc^3] std_P n_snapshots   (P scaled to A_s = 2.105e-9; see make_p3m_ensemble.py)")
    print(out)
    # the 100 test jobs alone, UNSCALED (the non-linear spectrum is not proportional to A_s), with their A_s values in the same order
    test = [(float(re.search(r"A_s([0-9.e+-]+)", f).group(1)), f) for f in files if "Test_Data_concept" in f]
    rows = np.array([binned(snapshot.read_positions(f))[1] for _, f in test])
    np.savetxt(os.path.join(here, "p3m_farm_test100_pk.txt"),
               np.column_stack([EDGES[:-1], EDGES[1:], cnt, kb, rows.mean(0), rows.std(0, ddof=1), np.full(len(kb), len(rows))]),
               header="k_lo k_hi modes mean_k mean_P[Mpc^3] std_P n_snapshots   (the 100 snapshots of Test_Data_concept, unscaled; A_s per job in p3m_farm_test100_As.txt)")
    np.savetxt(os.path.join(here, "p3m_farm_test100_As.txt"), np.array([a for a, _ in test]), header="A_s of the 100 Test_Data_concept jobs, sorted() path order")

"""Momentum statistics of the reference's 100 P3M TEST snapshots — generator of p3m_farm_test100_momentum.txt (runs only where
/root/reference exists). The thesis scripts read `components/matter/mom` next to `pos` (SnapVsPK/Power_compare.py:143-144,
TheFeedingPipeline/FraP_kTilDelta.py:59-60) but never use it, so nothing in the reference compares velocities; these two statistics do,
and with them the comoving kick/drift convention and the unit system of the integrator (positions alone do not pin them):
  * v_rms = sqrt(<|mom / m|^2>) per snapshot   [Mpc/Gyr; CONCEPT's mom = a^2 m dx/dt, m = the `mass` attribute of components/matter]
  * P_div(k): power spectrum of the divergence of the CIC-deposited momentum density  q(x) = sum_i (mom_i / m) W(x - x_i) / nbar
    (window-deconvolved, not interlaced), per mode |i k . q_k|^2 L^3 / N^6, averaged in the 8 linear |k| bins of p3m_farm_test100_pk.txt.
    By continuity d(delta)/dt = -div(q)/a^2, so on linear scales P_div = (a^2 H f)^2 P_delta: a wrong power of a or a wrong time unit in
    the velocities moves it by an O(1) factor.
Both UNSCALED (A_s per job in p3m_farm_test100_As.txt, same order). Estimator: the NumPy oracle's CIC deposit with weights."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from bachelor_thesis_n_body_amd import snapshot
from oracle import mesh_oracle as mo

REF = "/root/reference/TheFeedingPipeline/Datafolder4"
NG, L = 32, 500.0
EDGES = np.linspace(2 * np.pi / L * 0.5, np.pi * NG / L, 9)


def momentum_stats(pos, vel):
    """(v_rms, binned P_div[8]) of particles at `pos` with velocity variable `vel` = mom / m = a^2 dx/dt."""
    h = L / NG
    nbar = len(pos) / NG ** 3
    kv = 2 * np.pi * np.fft.fftfreq(NG, d=h)
    kx, ky, kz = np.meshgrid(kv, kv, kv, indexing="ij")
    w = np.sinc(kv * h / (2 * np.pi)) ** 2
    W = w[:, None, None] * w[None, :, None] * w[None, None, :]
    div = sum(kc * np.fft.fftn(mo.cic_deposit(pos, vel[:, c], L, NG, 0.0) / nbar) for c, kc in enumerate((kx, ky, kz))) / W
    p = (np.abs(div) ** 2 * L ** 3 / NG ** 6).ravel()
    k = np.sqrt(kx ** 2 + ky ** 2 + kz ** 2).ravel()
    which = np.digitize(k, EDGES) - 1
    ok = (which >= 0) & (which < len(EDGES) - 1)
    cnt = np.bincount(which[ok], minlength=len(EDGES) - 1)
    return float(np.sqrt((vel ** 2).sum(1).mean())), np.bincount(which[ok], weights=p[ok], minlength=len(EDGES) - 1) / cnt


if __name__ == "__main__":
    files = sorted(f for f in glob.glob(os.path.join(REF, "*oncept", "id*", "NNPM32_As*", "snapshots", "snapshot_a=1.00.hdf5")) if "Test_Data_concept" in f)
    vr, rows = [], []
    for f in files:
        m = snapshot.read_attrs(f, "components/matter")["mass"]
        v, p = momentum_stats(snapshot.read_positions(f), snapshot.read_momenta(f) / m)
        vr.append(v); rows.append(p)
    vr, rows = np.array(vr), np.array(rows)
    pk = np.loadtxt(os.path.join(os.path.dirname(os.path.abspath(__file__)), "p3m_farm_test100_pk.txt"))
    out = np.column_stack([EDGES[:-1], EDGES[1:], pk[:, 2], pk[:, 3], rows.mean(0), rows.std(0, ddof=1), np.full(8, len(rows))])
    here = os.path.dirname(os.path.abspath(__file__))
    np.savetxt(os.path.join(here, "p3m_farm_test100_momentum.txt"), out,
               header="momentum statistics of the 100 snapshots of Test_Data_concept (make_p3m_momentum.py), unscaled\n"
                      f"v_rms = sqrt(<|mom/m|^2>) [Mpc/Gyr]: mean {vr.mean():.6f} std {vr.std(ddof=1):.6f} over {len(vr)} snapshots\n"
                      "k_lo k_hi modes mean_k mean_Pdiv std_Pdiv n_snapshots")
    print(f"v_rms {vr.mean():.5f} +- {vr.std(ddof=1):.5f}"); print(out)
    # linear-theory check of the convention: P_div / P_delta -> (a^2 H f)^2 at low k
    print("P_div / P_delta:", out[:, 4] / pk[:, 4])

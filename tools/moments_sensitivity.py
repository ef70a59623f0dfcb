"""Sensitivity of the moments of the 32^3 density contrast of the farm job (NN.param) to softening, mesh size and LPT order."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bachelor_thesis_n_body_amd import cosmo, field
gold = os.path.join(ROOT, "tests", "golden", "concept")
pp = np.loadtxt(os.path.join(gold, "PP_powerspec_a=1.00"))
k_ref, p_lin = pp[:, 0], pp[:, 4]
slope = np.log(p_lin[-1] / p_lin[-3]) / np.log(k_ref[-1] / k_ref[-3])
k_ext = np.geomspace(k_ref[-1] * 1.05, 1.0, 12)
tab = np.column_stack([np.concatenate([k_ref, k_ext]), np.concatenate([p_lin, p_lin[-1] * (k_ext / k_ref[-1]) ** slope])])
n_side, L = 32, 500.0
c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)
ref = np.loadtxt(os.path.join(gold, "p3m_farm_ensemble_moments.txt"))
print(f"reference: var {ref[1]:.4f}  S3 {ref[3]:.3f}  S4 {ref[5]:.2f}")
for label, kw in (("eps 0.03 d, mesh 32 (NN.param)", dict(eps=None, ngrid=32, gravity="p3m")), ("eps 0.01 d", dict(eps=0.01 * L / n_side, ngrid=32, gravity="p3m")),
                  ("eps 0.1 d", dict(eps=0.1 * L / n_side, ngrid=32, gravity="p3m")), ("mesh 64", dict(eps=None, ngrid=64, gravity="p3m")),
                  ("Ewald pp", dict(eps=None, gravity="pp"))):
    mom = []
    for seed in range(16):
        pos, vel = cosmo.zeldovich_ics(tab, n_side, L, c, 0.02, 100 + seed, 5000 + seed, lpt=2)
        pos, _ = cosmo.run_pp(pos, vel, c.particle_mass(L, n_side ** 3), L, c, 0.02, 1.0, 400, dtype=np.float32, **kw)
        d = field.delta_field(pos, L, n_side, subtract_mean=True)
        v = (d ** 2).mean()
        mom.append([v, (d ** 3).mean() / v ** 2, ((d ** 4).mean() - 3 * v * v) / v ** 3])
    m, e = np.mean(mom, 0), np.std(mom, 0, ddof=1) / 4
    print(f"{label:34s} var {m[0]:.4f}+-{e[0]:.4f}  S3 {m[1]:.3f}+-{e[1]:.3f}  S4 {m[2]:.2f}+-{e[2]:.2f}", flush=True)

# the reference's own A_s values, job by job (the moments are non-linear in the amplitude)
a_s = np.loadtxt(os.path.join(gold, "p3m_farm_test_As.txt"))
for label, kw in (("100 jobs, the reference's A_s, mesh 32", dict(ngrid=32, gravity="p3m")), ("same, Ewald pp", dict(gravity="pp"))):
    mom = []
    for i, A in enumerate(a_s):
        t = tab.copy(); t[:, 1] *= A / 2.105e-9
        pos, vel = cosmo.zeldovich_ics(t, n_side, L, c, 0.02, 300 + i, 9000 + i, lpt=2)
        pos, _ = cosmo.run_pp(pos, vel, c.particle_mass(L, n_side ** 3), L, c, 0.02, 1.0, 400, dtype=np.float32, **kw)
        d = field.delta_field(pos, L, n_side, subtract_mean=True)
        v = (d ** 2).mean()
        mom.append([v / (A / 2.105e-9), (d ** 3).mean() / v ** 2, ((d ** 4).mean() - 3 * v * v) / v ** 3])
    m, e = np.mean(mom, 0), np.std(mom, 0, ddof=1) / np.sqrt(len(mom))
    print(f"{label:40s} var*A_ref/A_s {m[0]:.4f}+-{e[0]:.4f}  S3 {m[1]:.3f}+-{e[1]:.3f}  S4 {m[2]:.2f}+-{e[2]:.2f}", flush=True)

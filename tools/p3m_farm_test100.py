"""The reference's 100 test jobs of the farm (TheFeedingPipeline/NN.param with the A_s values of Datafolder4/TestParams_CONCEPT.txt:
32^3 particles, 500 Mpc, 'p3m' on a 32^3 interlaced mesh, a = 0.02 -> 1) re-run here job by job with the SAME A_s values (own
seeds), against the ensemble of the reference's 100 snapshots: binned P(k) (tests/golden/concept/p3m_farm_test100_pk.txt) and the
moments of the CIC-deposited, deconvolved density contrast the CNN is fed (p3m_farm_ensemble_moments.txt). No amplitude scaling
is applied to either side: the non-linear spectrum is not proportional to A_s. usage: p3m_farm_test100.py [nsteps|concept|rungs] [lpt] [pm|p3m|pp]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bachelor_thesis_n_body_amd import cosmo, field
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "concept"))
from make_p3m_momentum import momentum_stats          # the estimator the momentum fixture was made with (NumPy oracle deposit)

schedule = "concept" if (len(sys.argv) > 1 and sys.argv[1] in ("concept", "rungs")) else None      # the reference farm's own logged base steps
rungs = len(sys.argv) > 1 and sys.argv[1] == "rungs"                                                 # ... with the rung hierarchy for the short-range force
nsteps = 143 if schedule else (int(sys.argv[1]) if len(sys.argv) > 1 else 400)
lpt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
grav = sys.argv[3] if len(sys.argv) > 3 else "p3m"
gold = os.path.join(ROOT, "tests", "golden", "concept")
ens = np.loadtxt(os.path.join(gold, "p3m_farm_test100_pk.txt"))
a_s = np.loadtxt(os.path.join(gold, "p3m_farm_test100_As.txt"))
ref_m = np.loadtxt(os.path.join(gold, "p3m_farm_ensemble_moments.txt"))
edges = np.concatenate([ens[:, 0], ens[-1:, 1]])
pp = np.loadtxt(os.path.join(gold, "PP_powerspec_a=1.00"))
k_ref, p_lin = pp[:, 0], pp[:, 4]
slope = np.log(p_lin[-1] / p_lin[-3]) / np.log(k_ref[-1] / k_ref[-3])
k_ext = np.geomspace(k_ref[-1] * 1.05, 1.0, 12)
tab = np.column_stack([np.concatenate([k_ref, k_ext]), np.concatenate([p_lin, p_lin[-1] * (k_ext / k_ref[-1]) ** slope])])
n_side, L = 32, 500.0
c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)
rows, mom, vrms, pdivs = [], [], [], []
t0 = time.perf_counter()
for i, A in enumerate(a_s):
    t = tab.copy(); t[:, 1] *= A / 2.105e-9                      # the table is CLASS' linear spectrum for A_s = 2.105e-9 (NN.param:92)
    pos, vel = cosmo.zeldovich_ics(t, n_side, L, c, 0.02, 300 + i, 9000 + i, lpt=lpt)
    if rungs:
        pos, vel = cosmo.run_rungs(pos, vel, c.particle_mass(L, n_side ** 3), L, c, 0.02, 1.0, ngrid=32)
    else:
        pos, vel = cosmo.run_pp(pos, vel, c.particle_mass(L, n_side ** 3), L, c, 0.02, 1.0, nsteps, dtype=np.float32, gravity=grav, ngrid=32, schedule=schedule)
    k, p = cosmo.power_modes(pos, L, n_side, interlace=True)
    which = np.digitize(k, edges) - 1
    ok = (which >= 0) & (which < len(edges) - 1)
    rows.append(np.bincount(which[ok], weights=p[ok], minlength=len(edges) - 1) / np.bincount(which[ok], minlength=len(edges) - 1))
    d = field.delta_field(pos, L, n_side, subtract_mean=True)
    v = (d ** 2).mean()
    mom.append([v, (d ** 3).mean() / v ** 2, ((d ** 4).mean() - 3 * v * v) / v ** 3])
    vr, pdiv = momentum_stats(pos, vel)
    vrms.append(vr); pdivs.append(pdiv)
wall = time.perf_counter() - t0
rows, mom = np.array(rows), np.array(mom)
n = len(a_s)
mean, err = rows.mean(0), rows.std(0, ddof=1) / np.sqrt(n)
ref_err = ens[:, 5] / np.sqrt(ens[:, 6])
steps_txt = "CONCEPT's logged 143-step schedule + rung hierarchy (eta = 0.018)" if rungs else "CONCEPT's logged 143-step schedule" if schedule else f"{nsteps} steps uniform in ln a"
print(f"# 100 jobs with the reference's A_s values, {lpt}LPT, {steps_txt}, gravity {grav!r}: {wall:.1f} s ({wall/n:.2f} s per job incl. ICs, P(k), delta field)")
print("#  mean_k    modes   P reference     +-        P here          +-      ratio    +-")
ratio, rerr = mean / ens[:, 4], np.hypot(err / ens[:, 4], mean * ref_err / ens[:, 4] ** 2)
for i in range(len(mean)):
    print(f"{ens[i,3]:.4e} {int(ens[i,2]):6d}   {ens[i,4]:.4e}  {ref_err[i]:.1e}   {mean[i]:.4e}  {err[i]:.1e}   {ratio[i]:.3f}  {rerr[i]:.3f}")
mm, me = mom.mean(0), mom.std(0, ddof=1) / np.sqrt(n)
# the reference moments file stores variance scaled by A_ref/A_s; undo nothing here: compare S3, S4 (dimensionless) and the
# variance of each side scaled the same way
var_scaled = (mom[:, 0] * 2.105e-9 / a_s)
print("# moments of the 32^3 density contrast (CIC, deconvolved, mean removed): variance x A_ref/A_s, S3 = <d^3>/<d^2>^2, S4")
print(f"#   reference (100 fields): {ref_m[1]:.4f} +- {ref_m[2]/10:.4f}   {ref_m[3]:.3f} +- {ref_m[4]/10:.3f}   {ref_m[5]:.2f} +- {ref_m[6]/10:.2f}")
print(f"#   here (100 jobs):        {var_scaled.mean():.4f} +- {var_scaled.std(ddof=1)/10:.4f}   {mm[1]:.3f} +- {me[1]:.3f}   {mm[2]:.2f} +- {me[2]:.2f}")
# momenta: v = mom/m = a^2 dx/dt — pins the comoving kick/drift convention and the unit system (tests/golden/concept/make_p3m_momentum.py)
mfix = np.loadtxt(os.path.join(gold, "p3m_farm_test100_momentum.txt"))
hdr = [l for l in open(os.path.join(gold, "p3m_farm_test100_momentum.txt")) if "v_rms" in l][0].split()
v_ref, v_ref_std = float(hdr[hdr.index("mean") + 1]), float(hdr[hdr.index("std") + 1])
vrms, pdivs = np.array(vrms), np.array(pdivs)
print(f"# v_rms = sqrt(<|mom/m|^2>) [Mpc/Gyr]: reference {v_ref:.5f} +- {v_ref_std/10:.5f}   here {vrms.mean():.5f} +- {vrms.std(ddof=1)/10:.5f}   ratio {vrms.mean()/v_ref:.4f}")
dratio = pdivs.mean(0) / mfix[:, 4]
derr = np.hypot(pdivs.std(0, ddof=1) / 10 / mfix[:, 4], pdivs.mean(0) * mfix[:, 5] / 10 / mfix[:, 4] ** 2)
print("# P_div(k) (divergence of the momentum density) here / reference per bin: " + " ".join(f"{r:.3f}+-{e:.3f}" for r, e in zip(dratio, derr)))
print(json.dumps({"vrms_ratio": float(vrms.mean() / v_ref), "pdiv_ratio": [float(x) for x in dratio], "nsteps": nsteps, "lpt": lpt, "gravity": grav, "wall_s": wall, "ratio": [float(x) for x in ratio], "ratio_err": [float(x) for x in rerr],
                  "var_scaled": float(var_scaled.mean()), "S3": float(mm[1]), "S4": float(mm[2])}))

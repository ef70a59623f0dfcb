"""The reference's farm job (TheFeedingPipeline/NN.param: 32^3 particles, 500 Mpc, a = 0.02 -> 1, 'p3m' on a 32^3 mesh,
interlaced, A_s = 2.105e-9) run here for several seeds; the ensemble-mean P(k) next to the mean over the reference's 300 farm
snapshots (tests/golden/concept/p3m_farm_ensemble_pk.txt). usage: p3m_farm_ensemble.py [nseeds] [nsteps] [lpt order]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bachelor_thesis_n_body_amd import cosmo, field

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lpt = int(sys.argv[3]) if len(sys.argv) > 3 else 1
gold = os.path.join(ROOT, "tests", "golden", "concept")
ens = np.loadtxt(os.path.join(gold, "p3m_farm_ensemble_pk.txt"))
edges = np.concatenate([ens[:, 0], ens[-1:, 1]])
pp = np.loadtxt(os.path.join(gold, "PP_powerspec_a=1.00"))
k_ref, p_lin = pp[:, 0], pp[:, 4]
slope = np.log(p_lin[-1] / p_lin[-3]) / np.log(k_ref[-1] / k_ref[-3])
k_ext = np.geomspace(k_ref[-1] * 1.05, 1.0, 12)
tab = np.column_stack([np.concatenate([k_ref, k_ext]), np.concatenate([p_lin, p_lin[-1] * (k_ext / k_ref[-1]) ** slope])])
n_side, L = 32, 500.0
c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)
rows, ic_rows, moments = [], [], []
D_ini = c.growth(0.02)[0]
t0 = time.perf_counter()
for seed in range(nseeds):
    pos, vel = cosmo.zeldovich_ics(tab, n_side, L, c, 0.02, 100 + seed, 5000 + seed, lpt=lpt)
    k, p = cosmo.power_modes(pos, L, n_side, interlace=True)       # diagnostic: the initial conditions against D^2 P_lin
    want = D_ini ** 2 * np.exp(np.interp(np.log(k), np.log(tab[:, 0]), np.log(tab[:, 1])))
    which = np.digitize(k, edges) - 1
    ok = (which >= 0) & (which < len(edges) - 1)
    ic_rows.append(np.bincount(which[ok], weights=p[ok], minlength=len(edges) - 1) / np.bincount(which[ok], weights=want[ok], minlength=len(edges) - 1))
    pos, _ = cosmo.run_pp(pos, vel, c.particle_mass(L, n_side ** 3), L, c, 0.02, 1.0, nsteps, dtype=np.float32, gravity="p3m", ngrid=32,
                          interlace=True)
    d = field.delta_field(pos, L, n_side, subtract_mean=True)           # the farm's per-snapshot chain (Feeder_pipe.py:174-195)
    v = (d ** 2).mean()
    moments.append([v, (d ** 3).mean() / v ** 2, ((d ** 4).mean() - 3 * v * v) / v ** 3])
    k, p = cosmo.power_modes(pos, L, n_side, interlace=True)
    which = np.digitize(k, edges) - 1
    ok = (which >= 0) & (which < len(edges) - 1)
    rows.append(np.bincount(which[ok], weights=p[ok], minlength=len(edges) - 1) / np.bincount(which[ok], minlength=len(edges) - 1))
wall = time.perf_counter() - t0
rows = np.array(rows)
mean, err = rows.mean(0), rows.std(0, ddof=1) / np.sqrt(nseeds)
print(f"# {lpt}LPT initial conditions; {nseeds} seeds x {nsteps} steps of the NN.param job in {wall:.1f} s ({wall/nseeds:.2f} s per job incl. ICs and P(k))")
print("#  mean_k    modes   P reference(300 jobs)   P here        +-      ratio")
for i in range(len(mean)):
    print(f"{ens[i,3]:.4e} {int(ens[i,2]):6d}   {ens[i,4]:.4e}             {mean[i]:.4e}  {err[i]:.1e}  {mean[i]/ens[i,4]:.3f}")
mom = np.array(moments)
ref_m = np.loadtxt(os.path.join(gold, "p3m_farm_ensemble_moments.txt"))
print("# moments of the CIC-deposited, deconvolved 32^3 density contrast (the CNN's input): variance, S3 = <d^3>/<d^2>^2, S4")
print(f"#   reference (100 fields): {ref_m[1]:.4f} +- {ref_m[2]/10:.4f}   {ref_m[3]:.3f} +- {ref_m[4]/10:.3f}   {ref_m[5]:.2f} +- {ref_m[6]/10:.2f}")
print("#   here (%d seeds):        %.4f +- %.4f   %.3f +- %.3f   %.2f +- %.2f" % ((nseeds,) + tuple(x for m, e in zip(mom.mean(0), mom.std(0, ddof=1) / np.sqrt(nseeds)) for x in (m, e))))
ic = np.array(ic_rows)
print("# initial conditions at a = 0.02: <P>/<D^2 P_lin> per bin, mean over seeds +- error: " +
      " ".join(f"{m:.3f}+-{e:.3f}" for m, e in zip(ic.mean(0), ic.std(0, ddof=1) / np.sqrt(nseeds))))
print(json.dumps({"nseeds": nseeds, "nsteps": nsteps, "wall_s": wall, "ratio": [float(x) for x in mean / ens[:, 4]],
                  "stat_err": [float(x) for x in err / ens[:, 4]]}))

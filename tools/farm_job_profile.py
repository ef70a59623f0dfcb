"""Where does one 32^3 P3M farm job spend its wall time? ('32^3 farm job under 0.25 s'.) cProfile of job #2 after a warm-up job."""
import cProfile, io, os, pstats, sys, time
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
n_side, L = (int(sys.argv[2]) if len(sys.argv) > 2 else 32), 500.0
c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 400


def job(i, times):
    t0 = time.perf_counter()
    pos, vel = cosmo.zeldovich_ics(tab, n_side, L, c, 0.02, 300 + i, 9000 + i, lpt=2)
    t1 = time.perf_counter()
    pos, vel = cosmo.run_pp(pos, vel, c.particle_mass(L, n_side ** 3), L, c, 0.02, 1.0, nsteps, dtype=np.float32, gravity="p3m", ngrid=n_side)
    t2 = time.perf_counter()
    k, p = cosmo.power_modes(pos, L, n_side, interlace=True)
    d = field.delta_field(pos, L, n_side, subtract_mean=True)
    t3 = time.perf_counter()
    times.append((t1 - t0, t2 - t1, t3 - t2))


tm = []
job(0, tm)
pr = cProfile.Profile(); pr.enable()
for i in range(1, 6):
    job(i, tm)
pr.disable()
for t in tm:
    print("ICs %.3f s   run %.3f s   P(k)+delta %.3f s   total %.3f s" % (t + (sum(t),)))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])

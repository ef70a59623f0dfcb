"""The thesis' solver comparison (PPPMP3M/pppmp3mplotter.py:11-22): the SAME 64^3 realisation in a 500 Mpc box evolved from
a = 0.02 to 1 with 'pp', 'pm' and 'p3m' gravity; CONCEPT took 2 d 23:27:37, 12.1 s and 4:18 min. The reference's three tables
share their initial conditions, so the bin-by-bin ratios P_PM/P_PP and P_P3M/P_PP are free of cosmic variance; this tool
evolves one realisation of its own with the three solvers of this library and prints the same ratios next to the reference's.
usage: pp_pm_p3m_reference.py [nsteps] [n_side] [seed] [mesh gridsize]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bachelor_thesis_n_body_amd import cosmo

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 64
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ngrid = int(sys.argv[4]) if len(sys.argv) > 4 else n_side
L, a_begin = 500.0, 0.02
gold = os.path.join(ROOT, "tests", "golden", "concept")
ref = {g: np.loadtxt(os.path.join(gold, f"{g}_powerspec_a=1.00")) for g in ("PP", "PM", "P3M")}
k_ref, modes, p_lin = ref["PP"][:, 0], ref["PP"][:, 1].astype(int), ref["PP"][:, 4]
slope = np.log(p_lin[-1] / p_lin[-3]) / np.log(k_ref[-1] / k_ref[-3])
k_ext = np.geomspace(k_ref[-1] * 1.05, 1.0, 12)
tab = np.column_stack([np.concatenate([k_ref, k_ext]), np.concatenate([p_lin, p_lin[-1] * (k_ext / k_ref[-1]) ** slope])])
c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)
pos0, vel0 = cosmo.zeldovich_ics(tab, n_side, L, c, a_begin, seed, seed + 1000)
mass = c.particle_mass(L, n_side ** 3)
use = modes if n_side == 64 else modes[: max(2, len(modes) // 2)]
out, wall = {}, {}
for grav in ("pp", "pm", "p3m"):
    t0 = time.perf_counter()
    pos, _ = cosmo.run_pp(pos0, vel0, mass, L, c, a_begin, 1.0, nsteps, dtype=np.float32, gravity=grav, ngrid=ngrid, interlace=True)
    wall[grav] = time.perf_counter() - t0
    kmag, pk = cosmo.power_modes(pos, L, n_side, interlace=True)
    out[grav] = cosmo.rebin_like_concept(kmag, pk, use)[1]
nb = len(out["pp"])
print(f"# {n_side}^3 particles, {nsteps} steps, mesh {ngrid}^3 (interlaced, fp32 particles); wall pp {wall['pp']:.1f} s, pm {wall['pm']:.1f} s, "
      f"p3m {wall['p3m']:.1f} s   (CONCEPT: 257257 s, 12.1 s, 258 s)")
print("#  k          modes   PM/PP ref  PM/PP here   P3M/PP ref  P3M/PP here   PP here/ref")
rows = []
for i in range(nb):
    r = [ref["PM"][i, 2] / ref["PP"][i, 2], out["pm"][i] / out["pp"][i], ref["P3M"][i, 2] / ref["PP"][i, 2], out["p3m"][i] / out["pp"][i],
         out["pp"][i] / ref["PP"][i, 2]]
    rows.append(r)
    print(f"{k_ref[i]:.4e} {modes[i]:6d}   {r[0]:7.4f}    {r[1]:7.4f}      {r[2]:7.4f}     {r[3]:7.4f}      {r[4]:7.4f}")
rows = np.array(rows)
print(json.dumps({"n_side": n_side, "nsteps": nsteps, "ngrid": ngrid, "wall_s": wall,
                  "p3m_over_pp_max_dev_here": float(np.abs(rows[:, 3] - 1).max()), "p3m_over_pp_max_dev_ref": float(np.abs(rows[:, 2] - 1).max()),
                  "pm_over_pp_last_bin_here": float(rows[-1, 1]), "pm_over_pp_last_bin_ref": float(rows[-1, 0])}))

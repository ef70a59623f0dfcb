"""Growth of the power spectrum between a = 1/6 and a = 1 in the reference's SnapVsPK run (SnapVsPK/Uncertain.param: 64^3
particles, 1000 Mpc box, 'gravity': 'pm' on a 64^3 interlaced mesh — Uncertain.param:66-95; SnapVsPK/powerspec/powerspec_a={0.17,1.00}). Both tables come from ONE
run, so P(a=1)/P(a=1/6) per bin depends on the realisation only through mode coupling. The same run here for several seeds.
usage: growth_ratio_reference.py [nseeds] [nsteps] [pm|p3m]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bachelor_thesis_n_body_amd import cosmo

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
grav = sys.argv[3] if len(sys.argv) > 3 else "pm"
gold = os.path.join(ROOT, "tests", "golden", "concept")
early, late = (np.loadtxt(os.path.join(gold, f"SnapVsPK_powerspec_a={a}")) for a in ("0.17", "1.00"))
modes = early[:, 1].astype(int)
ref_ratio = late[:, 2] / early[:, 2]
lin = np.column_stack([late[:, 0], late[:, 4]])                     # the table's own linear column at a = 1
slope = np.log(lin[-1, 1] / lin[-3, 1]) / np.log(lin[-1, 0] / lin[-3, 0])
k_ext = np.geomspace(lin[-1, 0] * 1.05, 1.0, 12)
tab = np.vstack([lin, np.column_stack([k_ext, lin[-1, 1] * (k_ext / lin[-1, 0]) ** slope])])
n_side, L = 64, 1000.0
c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)
a_mid = 1.0 / 6.0
ratios = []
t0 = time.perf_counter()
for seed in range(nseeds):
    pos, vel = cosmo.zeldovich_ics(tab, n_side, L, c, 0.02, 200 + seed, 7000 + seed)
    mass = c.particle_mass(L, n_side ** 3)
    n1 = round(nsteps * np.log(a_mid / 0.02) / np.log(1 / 0.02))
    pos, vel = cosmo.run_pp(pos, vel, mass, L, c, 0.02, a_mid, n1, dtype=np.float32, gravity=grav, ngrid=64)
    p_early = cosmo.rebin_like_concept(*cosmo.power_modes(pos, L, n_side, interlace=True), modes)[1]
    pos, vel = cosmo.run_pp(pos, vel, mass, L, c, a_mid, 1.0, nsteps - n1, dtype=np.float32, gravity=grav, ngrid=64)
    p_late = cosmo.rebin_like_concept(*cosmo.power_modes(pos, L, n_side, interlace=True), modes)[1]
    ratios.append(p_late / p_early)
ratios = np.array(ratios)
mean, sd = ratios.mean(0), ratios.std(0, ddof=1)
lin_ratio = 1.0 / c.growth(a_mid)[0] ** 2
print(f"# gravity {grav!r}: {nseeds} seeds, {nsteps} steps each, {time.perf_counter()-t0:.1f} s; linear growth ratio here {lin_ratio:.3f}, CLASS (table) {late[0,4]/early[0,4]:.3f}")
print("#  k          modes   P(1)/P(1/6) reference   here: mean   seed scatter   (ref - mean)/scatter")
for i in range(len(mean)):
    print(f"{late[i,0]:.4e} {modes[i]:6d}   {ref_ratio[i]:8.3f}               {mean[i]:8.3f}     {sd[i]:6.3f}        {(ref_ratio[i]-mean[i])/sd[i]:+5.2f}")
print(json.dumps({"nseeds": nseeds, "ref_ratio": [float(x) for x in ref_ratio], "mean": [float(x) for x in mean], "scatter": [float(x) for x in sd]}))

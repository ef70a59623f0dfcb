"""The thesis' one direct-summation configuration on the MI355X: 64^3 particles, 500 Mpc periodic box, a = 0.02 -> 1,
H0 = 67.36, omega_b = 0.02237, omega_cdm = 0.12 (PPPMP3M/pppmp3mplotter.py:11-22, Different_Sizes/Giant.param:75-85), which
took CONCEPT 2 days 23:27:37 on the author's desktop. Initial conditions: Zel'dovich from the `P linear` column of the
reference's own table (tests/golden/concept/PP_powerspec_a=1.00); output: P(k) at a = 1 in CONCEPT's 16 bins next to the
reference's `P` column, the realisation-corrected column next to the reference's `P corrected`, and sigma_8 of both next to the values
in the table's header. Prints a text table + one JSON summary line. usage: run_pp_reference.py [nsteps|concept|rungs] [n_side] [seed] [f32|f64] [lpt] [pp|pm|p3m]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bachelor_thesis_n_body_amd import cosmo, field

schedule = "concept" if (len(sys.argv) > 1 and sys.argv[1] in ("concept", "rungs")) else None      # CONCEPT's own base steps (cosmo.load_concept_schedule)
rungs = len(sys.argv) > 1 and sys.argv[1] == "rungs"                                                 # ... with the rung hierarchy (cosmo.run_rungs; p3m only)
nsteps = 143 if schedule else (int(sys.argv[1]) if len(sys.argv) > 1 else 400)
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 64
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dtype = np.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else np.float64
lpt = int(sys.argv[5]) if len(sys.argv) > 5 else 1
grav = sys.argv[6] if len(sys.argv) > 6 else "pp"
L, a_begin = 500.0, 0.02
ref = np.loadtxt(os.path.join(ROOT, "tests", "golden", "concept", f"{grav.upper()}_powerspec_a=1.00"))
k_ref, modes, p_ref, pc_ref, p_lin = ref[:, 0], ref[:, 1].astype(int), ref[:, 2], ref[:, 3], ref[:, 4]
# linear input spectrum: the table's linear column, power-law extrapolated to the grid's corner modes
slope = np.log(p_lin[-1] / p_lin[-3]) / np.log(k_ref[-1] / k_ref[-3])
k_ext = np.geomspace(k_ref[-1] * 1.05, 1.0, 12)
tab = np.column_stack([np.concatenate([k_ref, k_ext]), np.concatenate([p_lin, p_lin[-1] * (k_ext / k_ref[-1]) ** slope])])
c = cosmo.Cosmology.from_h_omegas(0.6736, 0.02237, 0.12)
t0 = time.perf_counter()
pos, vel, noise = cosmo.zeldovich_ics(tab, n_side, L, c, a_begin, seed, seed + 1000, lpt=lpt, return_noise_power=True)
mass = c.particle_mass(L, n_side ** 3)
if rungs:
    pos, vel, pops = cosmo.run_rungs(pos, vel, mass, L, c, a_begin, 1.0, dtype=dtype, ngrid=n_side, return_populations=True)
else:
    pos, vel = cosmo.run_pp(pos, vel, mass, L, c, a_begin, 1.0, nsteps, dtype=dtype, gravity=grav, ngrid=n_side, schedule=schedule)
t_run = time.perf_counter() - t0
kmag, pk = cosmo.power_modes(pos, L, n_side, interlace=True)     # CONCEPT's estimator settings: CIC, deconvolved, interlaced
kv = 2 * np.pi * np.fft.fftfreq(n_side, d=L / n_side)
kx, ky, kz = np.meshgrid(kv, kv, kv, indexing="ij")
r2 = np.asarray(noise, np.float64)[np.sqrt(kx ** 2 + ky ** 2 + kz ** 2) > 0]                  # |R_k|^2 in power_modes' mode order
plin_modes = np.exp(np.interp(np.log(kmag), np.log(tab[:, 0]), np.log(tab[:, 1])))
counts = modes if n_side == 64 else modes[: max(2, len(modes) // 2)]
k_b, p_b, pc_b = cosmo.rebin_corrected_like_concept(kmag, pk, plin_modes, r2, counts)
steps_txt = ("CONCEPT's logged schedule (143 base steps) + rung hierarchy for the short-range force, final populations " + str(pops[-2].tolist())) if rungs else "CONCEPT's logged schedule (143 base steps, constant in cosmic time over 8 steps)" if schedule else f"{nsteps} steps uniform in ln a"
print(f"# gravity {grav!r}; {lpt}LPT ICs; {n_side}^3 particles, {steps_txt}, {np.dtype(dtype).name} forces, wall {t_run:.1f} s (CONCEPT pp: 257257 s)")
print("#  k_ref      k_here     modes   P_ref        P_here       ratio  | Pcorr_ref    Pcorr_here   ratio  | P_ref/P_lin  P_here/P_lin")
for i in range(len(k_b)):
    print(f"{k_ref[i]:.4e} {k_b[i]:.4e} {modes[i]:6d}  {p_ref[i]:.4e}  {p_b[i]:.4e}  {p_b[i]/p_ref[i]:6.3f} | {pc_ref[i]:.4e}  {pc_b[i]:.4e}  {pc_b[i]/pc_ref[i]:6.3f} |  {p_ref[i]/p_lin[i]:6.3f}      {p_b[i]/p_lin[i]:6.3f}")
r, rc = p_b / p_ref[: len(p_b)], pc_b / pc_ref[: len(pc_b)]
s8 = [cosmo.sigma8_of_table(k_b, v, 0.6736) for v in (p_b, pc_b)]
s8_ref = [cosmo.sigma8_of_table(k_ref, v, 0.6736) for v in (p_ref, pc_ref, p_lin)]
print(f"# sigma_8 of the tabulated range: here P {s8[0]:.4f}, corrected {s8[1]:.4f}; reference table P {s8_ref[0]:.4f}, corrected {s8_ref[1]:.4f}, linear {s8_ref[2]:.4f}")
print(json.dumps({"gravity": grav, "n_side": n_side, "nsteps": nsteps, "schedule": schedule or "uniform ln a", "dtype": np.dtype(dtype).name, "wall_s": t_run,
                  "ratio_mean_bins_4_16": float(r[3:].mean()), "ratio_min": float(r.min()), "ratio_max": float(r.max()),
                  "corrected_ratio": [round(float(x), 4) for x in rc], "corrected_ratio_mean_bins_4_16": float(rc[3:].mean()),
                  "corrected_ratio_rms_dev_bins_4_16": float(np.sqrt(((rc[3:] - 1) ** 2).mean())),
                  "sigma8_here": s8, "sigma8_ref": s8_ref, "reference_wall_s": 257257}))

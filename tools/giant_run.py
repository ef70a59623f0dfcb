"""The reference's largest configuration through the `concept` command line: Different_Sizes/Giant.param — N = 256^3 particles,
1000 Mpc box, 'gravity': 'pm' on a cbrt(N) mesh, interlace/deconvolve (True, True), 1LPT, a = 0.02 -> 1, snapshot + power
spectrum at a = 1. The parameter file below is written here with the same keys and values (it is not a copy of the file);
the linear P(k) at a = 1 comes from the reference's PP table (tests/golden/concept/PP_powerspec_a=1.00), power-law extrapolated.
usage: giant_run.py [size=256] [pm|p3m] [steps=400]"""
import json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bachelor_thesis_n_body_amd import concept, snapshot

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
grav = sys.argv[2] if len(sys.argv) > 2 else "pm"
steps = sys.argv[3] if len(sys.argv) > 3 else "400"
pp = np.loadtxt(os.path.join(ROOT, "tests", "golden", "concept", "PP_powerspec_a=1.00"))
k_ref, p_lin = pp[:, 0], pp[:, 4]
lo = np.geomspace(1e-3, k_ref[0] * 0.95, 8)
hi = np.geomspace(k_ref[-1] * 1.05, 3.0, 16)
s_lo = np.log(p_lin[2] / p_lin[0]) / np.log(k_ref[2] / k_ref[0])
s_hi = np.log(p_lin[-1] / p_lin[-3]) / np.log(k_ref[-1] / k_ref[-3])
tab = np.column_stack([np.concatenate([lo, k_ref, hi]),
                       np.concatenate([p_lin[0] * (lo / k_ref[0]) ** s_lo, p_lin, p_lin[-1] * (hi / k_ref[-1]) ** s_hi])])
with tempfile.TemporaryDirectory() as d:
    np.savetxt(os.path.join(d, "pk.txt"), tab)
    par = os.path.join(d, "giant.param")
    with open(par, "w", encoding="utf-8") as fh:
        fh.write(f"""_size = {size}
_lpt = 1
initial_conditions = {{'species': 'matter', 'N': _size**3}}
output_dirs = {{'snapshot': '{d}/out/snapshots', 'powerspec': '{d}/out/powerspec'}}
output_times = {{'snapshot': [1.0], 'powerspec': [1.0]}}
boxsize = 1000*Mpc
potential_options = {{
    'gridsize': {{'global': {{_size}}, 'particles': {{'gravity': {{'pm': ['cbrt(N)', 'cbrt(N)'], 'p3m': ['cbrt(N)', 'cbrt(N)']}}}}}},
    'interlace': {{'gravity': {{'pm': (True, True), 'p3m': (True, True)}}}},
    'deconvolve': {{'gravity': {{'pm': (True, True), 'p3m': (True, True)}}}},
}}
select_forces = {{'matter': {{'gravity': '{grav}'}}}}
realization_options = {{'LPT': _lpt}}
H0 = 67.36*km/(s*Mpc)
Ωb = 0.02237/h**2
Ωcdm = 0.12/h**2
a_begin = 0.02
primordial_spectrum = {{'A_s': 2.1e-9, 'n_s': 0.9649}}
random_seeds = {{'general': 0, 'primordial amplitudes': 1, 'primordial phases': 2}}
""")
    t0 = time.perf_counter()
    rc = concept.main(["-p", par, "-n", "1", "--local", "--cosmological", "--pk-table", os.path.join(d, "pk.txt"), "--steps", steps, "--dtype", "f32"])
    wall = time.perf_counter() - t0
    assert rc == 0
    snap = os.path.join(d, "out", "snapshots", "snapshot_a=1.00.hdf5")
    pos = snapshot.read_positions(snap)
    t = np.loadtxt(os.path.join(d, "out", "powerspec", "powerspec_a=1.00"))
    print(f"# Giant.param configuration: {size}^3 particles, 1000 Mpc, '{grav}', {steps} steps: {wall:.1f} s wall in total "
          f"(ICs + run + snapshot {os.path.getsize(snap)/2**20:.0f} MiB + P(k)); positions in [{pos.min():.3f}, {pos.max():.3f}]")
    print("#  k          modes     P           P_corrected   P_linear    P_corr/P_lin")
    for r in t[:: max(1, len(t) // 16)]:
        print(f"{r[0]:.4e} {int(r[1]):8d}  {r[2]:.4e}  {r[3]:.4e}    {r[4]:.4e}  {r[3]/r[4]:.3f}")
    print(json.dumps({"size": size, "gravity": grav, "steps": int(steps), "wall_s_total": wall, "bins": int(len(t))}))

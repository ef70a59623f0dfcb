"""Generates tests/golden/accel_golden.npz and kdk_golden.npz.

The reference repository has no fixture for pairwise gravity ("parity unpinned"), so these
vectors are produced by the fp64 NumPy oracle (oracle/nbody_oracle.py) in the build container and
cross-checked here against the independent sequential C restatement (oracle/nbody_oracle.c) before being
written. They are DATA (inputs + expected outputs); re-run with `python tests/golden/make_golden.py`.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from oracle import nbody_oracle as ora                      # noqa: E402
from bachelor_thesis_n_body_amd import plummer_sphere      # noqa: E402
from conftest import COracle                                # noqa: E402

co = COracle()
out = {}
# (name, n, seed, eps, G, unequal masses?)
cases = [("n2", 2, 11, 0.0, 1.0, True), ("n3", 3, 12, 0.05, 2.5, True), ("n64", 64, 13, 0.01, 1.0, False),
         ("n257", 257, 14, 0.01, 1.0, True), ("n1024", 1024, 1, 0.01, 1.0, False)]
for name, n, seed, eps, G, unequal in cases:
    s = plummer_sphere(n, seed)
    mass = s.mass.copy()
    if unequal:
        mass *= np.random.default_rng(seed).uniform(0.2, 3.0, n)
    acc = ora.compute_forces(s.pos, mass, eps, G)
    chk = co.accel(s.pos, mass, eps, G, sequential=True)
    assert np.allclose(acc, chk, rtol=1e-11, atol=1e-13), name
    pot = ora.potential_energy(s.pos, mass, eps, G)
    assert abs(pot - co.potential(s.pos, mass, eps, G)) <= 1e-12 * abs(pot)
    out.update({f"{name}_pos": s.pos, f"{name}_vel": s.vel, f"{name}_mass": mass, f"{name}_eps": eps,
                f"{name}_G": G, f"{name}_acc": acc, f"{name}_pot": pot, f"{name}_kin": ora.kinetic_energy(s.vel, mass)})
np.savez_compressed(ROOT / "tests" / "golden" / "accel_golden.npz", **out)

# KDK trajectory: N=256 Plummer sphere, dt = 1/256, eps = 0.01, state after 0/8/32 steps + energies
s = plummer_sphere(256, 21)
st = ora.State(s.pos.copy(), s.vel.copy(), s.mass.copy())
traj = {"pos0": st.pos.copy(), "vel0": st.vel.copy(), "mass": st.mass.copy(), "dt": 1 / 256, "eps": 0.01}
done = 0
for upto in (8, 32):
    ora.step(st, 1 / 256, upto - done, eps=0.01)
    done = upto
    k, u = ora.energy(st, 0.01)
    traj.update({f"pos{upto}": st.pos.copy(), f"vel{upto}": st.vel.copy(), f"acc{upto}": st.acc.copy(),
                 f"kin{upto}": k, f"pot{upto}": u})
np.savez_compressed(ROOT / "tests" / "golden" / "kdk_golden.npz", **traj)
print("golden vectors written")

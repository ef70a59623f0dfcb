"""Generates tests/golden/accel_golden_large.npz: sampled accelerations of N = 40000 and N = 100001 Plummer spheres — sizes whose force
evaluation takes the pair-symmetric kernels (csrc/accel_sym_*.hip) — from the fp64 NumPy oracle, cross-checked against the independent C
restatement before being written. Inputs are regenerated from (n, seed) by plummer_sphere (seeded PCG64, deterministic); the file holds the
sampled sink indices, the expected accelerations and a checksum of the inputs. DATA only; re-run with `python tests/golden/make_golden_large.py`."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from oracle import nbody_oracle as ora                      # noqa: E402
from bachelor_thesis_n_body_amd import plummer_sphere      # noqa: E402
from conftest import COracle                                # noqa: E402


def inputs(n, seed, unequal, dtype):
    """The inputs a test rebuilds: positions / masses in `dtype` (the kernels see exactly these values)."""
    s = plummer_sphere(n, seed, dtype)
    mass = s.mass.copy()
    if unequal:
        mass = (mass * np.random.default_rng(seed + 100).uniform(0.25, 4.0, n)).astype(dtype)
    return s.pos, mass


if __name__ == "__main__":
    co = COracle()
    out = {}
    for name, n, seed, eps, G, unequal, dtype in (("n40000_f32", 40000, 31, 0.01, 1.0, False, np.float32), ("n40000_f64_general", 40000, 32, 0.02, 1.5, True, np.float64),
                                                  ("n100001_f32_general", 100001, 33, 0.01, 1.0, True, np.float32), ("n100001_f64", 100001, 34, 0.01, 1.0, False, np.float64)):
        pos, mass = inputs(n, seed, unequal, dtype)
        p64, m64 = pos.astype(np.float64), mass.astype(np.float64)
        idx = np.unique(np.concatenate([np.random.default_rng(seed).integers(0, n, 250), [0, 1, n // 2, n - 2, n - 1]]))
        acc = ora.compute_forces(p64, m64, eps, G, sinks=p64[idx])
        chk = co.accel(p64, m64, eps, G, sinks=p64[idx])
        assert np.allclose(acc, chk, rtol=1e-11, atol=1e-13), name
        out.update({f"{name}_n": n, f"{name}_seed": seed, f"{name}_eps": eps, f"{name}_G": G, f"{name}_unequal": unequal,
                    f"{name}_idx": idx, f"{name}_acc": acc, f"{name}_sum_pos": p64.sum(0), f"{name}_sum_mass": m64.sum()})
    np.savez_compressed(ROOT / "tests" / "golden" / "accel_golden_large.npz", **out)
    print("large golden vectors written")

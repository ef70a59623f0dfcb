"""Measurement of the field chain on one MI355X: CIC deposit, deconvolution (2 FFTs + window) and
binned P(k), at the reference's size (32^3 particles, 32^3 mesh) and at production sizes, with the NumPy oracle timed
beside it on a bounded sample. Prints one JSON line per configuration.

CIC algorithmic bytes: 24 B read per particle + 8 fp64 atomic adds (64 B) per particle."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bachelor_thesis_n_body_amd import field
from oracle import field_oracle as fo


def timeit(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


def lattice_positions(n_side, box, rng, sigma):
    g = (np.arange(n_side) + 0.5) * (box / n_side)
    p = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    return (p + rng.normal(0, sigma, p.shape)) % box


for n_side, ngrid, order in ((32, 32, "lattice"), (128, 128, "lattice"), (256, 256, "lattice"), (256, 256, "random"), (256, 512, "lattice")):
    rng = np.random.default_rng(0)
    box = 500.0
    pos = lattice_positions(n_side, box, rng, box / n_side) if order == "lattice" else rng.uniform(0, box, (n_side ** 3, 3))
    npart = pos.shape[0]
    with field.FieldPlan(ngrid, box) as plan:
        t_cic = timeit(lambda: plan.cic(pos))
        t_dec = timeit(lambda: (plan.deconvolve(), plan.mean()))
        t_pk = timeit(lambda: plan.power_bins(1.0))
        t_chain = timeit(lambda: plan.delta_field(pos))
    out = {"config": f"{n_side}^3 particles ({order} order), {ngrid}^3 mesh", "cic_ms": t_cic * 1e3, "cic_particles_per_s": npart / t_cic,
           "cic_incl_h2d_GBps": npart * 24 / t_cic / 1e9, "deconvolve_ms": t_dec * 1e3, "power_bins_ms": t_pk * 1e3,
           "delta_field_chain_ms": t_chain * 1e3}
    if n_side <= 128:
        t0 = time.perf_counter(); fo.delta_field(pos, box, ngrid, False); out["numpy_oracle_chain_ms"] = (time.perf_counter() - t0) * 1e3
    print(json.dumps(out), flush=True)

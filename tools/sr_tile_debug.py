"""Debug aid: which pair does the tile pair kernel miss or count twice? (run on the GPU box)"""
import os, sys
import numpy as np
from scipy.special import erf
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bachelor_thesis_n_body_amd as nb
from oracle import mesh_oracle as mo

n, L, ngrid, eps = 300000, 80.0, 48, 0.15
rng = np.random.default_rng(11)
pos = rng.uniform(0, L, (n, 3)); rng.uniform(0.5, 1.5, n)
m = np.full(n, 0.75)
p = pos.astype(np.float32).astype(np.float64)
rs = 1.25 * L / ngrid; rcut = 4.5 * rs
s = nb.State(pos.astype(np.float32), np.zeros((n, 3), np.float32), m.astype(np.float32))
with nb.Engine(n, np.float32) as e:
    e.set_mesh("p3m", L, ngrid, part="short"); e.set_state(s); got = e.accel(eps, 1.0).astype(np.float64)
near_face = np.argsort(np.minimum(p, L - p).min(1))[:64]
idx = np.unique(np.concatenate([np.random.default_rng(2).choice(n, 192, replace=False), near_face]))
ref = mo.short_range_accel(p, m, L, rs, rcut, eps, sinks=idx)
err = got[idx] - ref
bad = np.argsort(-np.abs(err).max(1))[:4]
nf = int(np.floor(3 * L / rcut)); c = L / nf
print("rcut", rcut, "nf", nf, "c", c)
alpha = 0.5 / rs
for b in bad:
    i = idx[b]
    d = p - p[i]; d -= L * np.rint(d / L); r2 = (d * d).sum(1)
    r = np.sqrt(r2); u = alpha * r
    with np.errstate(all="ignore"):
        f = np.where((r2 > 0) & (r2 < rcut * rcut), 0.75 * ((r2 + eps * eps) ** -1.5 - (erf(u) - 2 * u / np.sqrt(np.pi) * np.exp(-u * u)) / (r2 * r)), 0.0)
    contrib = f[:, None] * d
    # which single pair explains the error (missing: err = -contrib; duplicate: err = +contrib)
    for sign, name in ((-1, "missing"), (1, "duplicate")):
        j = np.argmin(np.abs(err[b] - sign * contrib).sum(1))
        res = np.abs(err[b] - sign * contrib[j]).max()
        print(f"sink {i} pos {p[i]} cell {np.floor(p[i]/c)} err {err[b]} | {name} partner {j} pos {p[j]} cell {np.floor(p[j]/c)} d {d[j]} r {r[j]:.4f} residual {res:.2e}")

import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bachelor_thesis_n_body_amd as nb
from oracle import mesh_oracle as mo
def cloud(n, L, seed, clustered):
    rng = np.random.default_rng(seed); pos = rng.uniform(0, L, (n, 3))
    if clustered:
        k = 2 * n // 3; pos[:k] = np.mod(0.5 * L + 0.03 * L * rng.normal(size=(k, 3)), L)
    return pos, rng.uniform(0.5, 1.5, n)
n, L, ngrid = 300000, 80.0, 48
for clustered in (False, True):
  for equal in (True, False):
    for eps in (0.15, 0.0):
        pos, mr = cloud(n, L, 11, clustered); m = np.full(n, 0.75) if equal else mr
        p = pos.astype(np.float32).astype(np.float64); mm = m.astype(np.float32).astype(np.float64)
        rs = 1.25 * L / ngrid
        s = nb.State(pos.astype(np.float32), np.zeros((n, 3), np.float32), m.astype(np.float32))
        with nb.Engine(n, np.float32) as e:
            e.set_mesh("p3m", L, ngrid, part="short"); e.set_state(s); got = e.accel(eps, 1.0).astype(np.float64)
        idx = np.random.default_rng(2).choice(n, 192, replace=False)
        ref = mo.short_range_accel(p, mm, L, rs, 4.5 * rs, eps, sinks=idx)
        sc = np.sqrt((ref * ref).sum(1).mean())
        print(clustered, equal, eps, "max err / scale = %.2e" % (np.abs(got[idx] - ref).max() / sc), " rms err / scale = %.2e" % (np.sqrt(((got[idx]-ref)**2).mean()) / sc), flush=True)

"""NumPy stand-in for `HipKernels` used ONLY by the CPU rehearsal of the multi-rank path (test_dist_gloo.py):
same method signatures, torch CPU tensors in/out, arithmetic delegated to the oracle. It lets the ring schedule,
partitioning and buffer rotation of ShardedEngine run under gloo in a container without a GPU."""
import numpy as np
import torch

from oracle import nbody_oracle as ora


class OracleKernels:
    def accel(self, snk, n_snk, src, n_src, eps, G, acc, accumulate, uniform_mass=False):
        s, q = snk.numpy(), src.numpy()
        a = ora.compute_forces(q[:3, :n_src].T, q[3, :n_src], eps, G, sinks=s[:3, :n_snk].T).T
        out = acc.numpy()
        if accumulate:
            out[:, :n_snk] += a.astype(out.dtype)
        else:
            out[:, :n_snk] = a.astype(out.dtype)

    def accel_pot(self, snk, n_snk, src, n_src, eps, G, acc, phi, accumulate, uniform_mass=False):
        self.accel(snk, n_snk, src, n_src, eps, G, acc, accumulate, uniform_mass)
        s, q = snk.numpy().astype(np.float64), src.numpy().astype(np.float64)
        d = q[None, :3, :n_src].transpose(0, 2, 1) - s[:3, :n_snk].T[:, None, :]
        r2 = (d * d).sum(-1) + eps * eps
        with np.errstate(divide="ignore"):
            w = np.where(r2 > 0, q[3, :n_src][None, :] / np.sqrt(r2), 0.0)       # no pair skipped except coincident ones at eps = 0
        p = phi.numpy()
        p[:n_snk] = (p[:n_snk] if accumulate else 0.0) + w.sum(1)

    def accel_pair(self, blk_a, n_a, blk_b, n_b, eps, G, acc_a, acc_b, accumulate, uniform_mass=False, phi_a=None, phi_b=None):
        """Two different blocks: acc_a (+)= force of B on A, acc_b = force of A on B (the oracle, one direction at a time)."""
        if phi_a is not None:
            self.accel_pot(blk_a, n_a, blk_b, n_b, eps, G, acc_a, phi_a, accumulate, uniform_mass)
            self.accel_pot(blk_b, n_b, blk_a, n_a, eps, G, acc_b, phi_b, False, uniform_mass)
        else:
            self.accel(blk_a, n_a, blk_b, n_b, eps, G, acc_a, accumulate, uniform_mass)
            self.accel(blk_b, n_b, blk_a, n_a, eps, G, acc_b, False, uniform_mass)

    def pot_energy(self, phi, mass_row, n, eps, out, uniform_mass=False):
        m = mass_row.numpy().astype(np.float64)[:n]
        self_term = m / eps if eps > 0 else 0.0
        out.numpy()[0] = float((m * (phi.numpy()[:n] - self_term)).sum())

    def potential(self, snk, n_snk, snk_g0, src, n_src, src_g0, eps, out, accumulate):
        s, q = snk.numpy().astype(np.float64), src.numpy().astype(np.float64)
        d = q[None, :3, :n_src].transpose(0, 2, 1) - s[:3, :n_snk].T[:, None, :]
        r2 = (d * d).sum(-1) + eps * eps
        gi = snk_g0 + np.arange(n_snk)[:, None]
        gj = src_g0 + np.arange(n_src)[None, :]
        with np.errstate(divide="ignore"):
            w = np.where((gi != gj) & (r2 > 0), q[3, :n_src][None, :] / np.sqrt(r2), 0.0)
        val = float((s[3, :n_snk] * w.sum(1)).sum())
        o = out.numpy()
        o[0] = (o[0] if accumulate else 0.0) + val

    def kinetic(self, vel, mass_row, n, out):
        v, m = vel.numpy().astype(np.float64), mass_row.numpy().astype(np.float64)
        out.numpy()[0] = 0.5 * float((m[:n] * (v[:, :n] ** 2).sum(0)).sum())

    def kick(self, vel, acc, n, h):
        v, a = vel.numpy(), acc.numpy()
        v[:, :n] = v[:, :n] + a[:, :n] * v.dtype.type(h)

    def drift(self, pm, vel, n, dt):
        p, v = pm.numpy(), vel.numpy()
        p[:3, :n] = p[:3, :n] + v[:, :n] * p.dtype.type(dt)

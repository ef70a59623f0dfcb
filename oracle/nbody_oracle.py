"""CPU oracle for the direct-summation N-body path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
module. The product package (`bachelor_thesis_n_body_amd`) never does: it fails loudly when the HIP
library is missing.

PARITY UNPINNED against the reference: HusseinYoud/Bachelor-Thesis-N-Body contains no force kernel or
integrator of its own — its direct-summation runs went through the external CONCEPT
code (github.com/jmd-dk/concept, version not pinned anywhere in the repo; call sites
SnapVsPK/Ucertain.py:89-97, TheFeedingPipeline/NeuNetværk_test.py:155-171), whose solver source is
not vendored and whose only surviving `pp` artefact is a 16-row power spectrum of a periodic comoving
run (PPPMP3M/PP/powerspec/powerspec_a=1.00:11-26). There is therefore no reference file:line for these
functions to follow. They restate the definitions fixed in SURVEY.md §7.2 (rows ⊕1-⊕3 of §8a), in plain
fp64 NumPy, and are pinned by the analytic known-answer tests of SURVEY.md §8(c) items 1-7
(tests/test_oracle_kat.py) and by committed golden vectors cross-checked against the independent
scalar-loop C restatement in oracle/nbody_oracle.c (tests/golden/).

Definitions:
  a_i = G * sum_j m_j (r_j - r_i) / (|r_j - r_i|^2 + eps^2)^{3/2}     (pairs with zero denominator skipped)
  U   = -G/2 * sum_i sum_{j != i} m_i m_j / sqrt(|r_j - r_i|^2 + eps^2)            (pairs with zero denominator skipped)
  KDK : v += a*dt/2 ; x += v*dt ; a = compute_forces(x) ; v += a*dt/2     (a cached between steps)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

__all__ = ["State", "compute_forces", "potential_energy", "kinetic_energy", "energy", "step"]

_CHUNK_ELEMS = 1 << 22  # sink-chunk x N pair matrix elements held at once (~32 MiB per fp64 temporary)


@dataclass
class State:
    """Particle state, array-of-vectors like the CONCEPT snapshots (pos/mom are (N,3) f64:
    SnapVsPK/Power_compare.py:143-144) but with velocities instead of momenta."""

    pos: np.ndarray   # (N, 3)
    vel: np.ndarray   # (N, 3)
    mass: np.ndarray  # (N,)
    acc: Optional[np.ndarray] = None  # cached a(pos) for (eps, G) of the last step, or None

    def copy(self) -> "State":
        return State(self.pos.copy(), self.vel.copy(), self.mass.copy(),
                     None if self.acc is None else self.acc.copy())


def _as_f64(a, shape_tail=()):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape_tail and a.shape[1:] != shape_tail:
        raise ValueError(f"expected shape (N,{','.join(map(str, shape_tail))}), got {a.shape}")
    return a


def compute_forces(pos, mass, eps: float, G: float = 1.0, sinks=None) -> np.ndarray:
    """Softened all-pairs acceleration in fp64. `sinks` (M,3) defaults to `pos` (self-gravity)."""
    pos = _as_f64(pos, (3,))
    mass = _as_f64(mass)
    snk = pos if sinks is None else _as_f64(sinks, (3,))
    n_src, n_snk = pos.shape[0], snk.shape[0]
    acc = np.zeros((n_snk, 3), dtype=np.float64)
    if n_src == 0 or n_snk == 0:
        return acc
    eps2 = float(eps) * float(eps)
    chunk = max(1, _CHUNK_ELEMS // max(n_src, 1))
    for i0 in range(0, n_snk, chunk):
        i1 = min(n_snk, i0 + chunk)
        d = pos[None, :, :] - snk[i0:i1, None, :]            # (c, N, 3): r_j - r_i
        r2 = np.einsum("ijk,ijk->ij", d, d) + eps2           # (c, N)
        with np.errstate(divide="ignore", invalid="ignore"):
            w = np.where(r2 > 0.0, mass[None, :] / (r2 * np.sqrt(r2)), 0.0)
        acc[i0:i1] = np.einsum("ij,ijk->ik", w, d)
    return G * acc


def potential_energy(pos, mass, eps: float, G: float = 1.0) -> float:
    pos = _as_f64(pos, (3,))
    mass = _as_f64(mass)
    n = pos.shape[0]
    if n == 0:
        return 0.0
    eps2 = float(eps) * float(eps)
    chunk = max(1, _CHUNK_ELEMS // n)
    total = 0.0
    for i0 in range(0, n, chunk):
        i1 = min(n, i0 + chunk)
        d = pos[None, :, :] - pos[i0:i1, None, :]
        r2 = np.einsum("ijk,ijk->ij", d, d) + eps2
        with np.errstate(divide="ignore"):
            w = np.where(r2 > 0.0, mass[None, :] / np.sqrt(r2), 0.0)     # a pair with a zero denominator is skipped, as in the force sum
        w[np.arange(i1 - i0), np.arange(i0, i1)] = 0.0      # j != i
        total += float(np.dot(mass[i0:i1], w.sum(axis=1)))
    return -0.5 * G * total


def kinetic_energy(vel, mass) -> float:
    vel = _as_f64(vel, (3,))
    mass = _as_f64(mass)
    return 0.5 * float(np.dot(mass, np.einsum("ij,ij->i", vel, vel)))


def energy(state: State, eps: float, G: float = 1.0) -> Tuple[float, float]:
    return kinetic_energy(state.vel, state.mass), potential_energy(state.pos, state.mass, eps, G)


def step(state: State, dt: float, nsteps: int = 1, eps: float = 0.0, G: float = 1.0) -> State:
    """Kick-drift-kick leapfrog, in place; returns `state`. One force evaluation per step."""
    pos = state.pos = _as_f64(state.pos, (3,))
    vel = state.vel = _as_f64(state.vel, (3,))
    if state.acc is None:
        state.acc = compute_forces(pos, state.mass, eps, G)
    h = 0.5 * dt
    for _ in range(int(nsteps)):
        vel += state.acc * h
        pos += vel * dt
        state.acc = compute_forces(pos, state.mass, eps, G)
        vel += state.acc * h
    return state

"""Initial conditions: seeded Plummer sphere in Hénon units (SURVEY.md §8a row ⊕4, §8d "Synthetic input").

Aarseth-Hénon-Wielen sampling: G = M = 1, Plummer scale a = 3*pi/16 (total energy E = -1/4), equal masses
1/N, radii by inverse CDF r = a (u^{-2/3} - 1)^{-1/2} (redrawn while r > 10 a), isotropic directions, speeds by
von-Neumann rejection on g(q) = q^2 (1 - q^2)^{7/2} with v = q * v_esc(r); centre-of-mass position and velocity
removed. The reference itself only ever used CONCEPT's cosmological lattice ICs (Different_Sizes/Giant.param:6,
70-72); the Plummer sphere is the north-star benchmark's own input.
"""
from __future__ import annotations

import numpy as np

from .state import State

PLUMMER_A = 3.0 * np.pi / 16.0


def _isotropic(rng: np.random.Generator, n: int) -> np.ndarray:
    cos_t = rng.uniform(-1.0, 1.0, n)
    sin_t = np.sqrt(np.maximum(0.0, 1.0 - cos_t * cos_t))
    phi = rng.uniform(0.0, 2.0 * np.pi, n)
    return np.stack([sin_t * np.cos(phi), sin_t * np.sin(phi), cos_t], axis=1)


def plummer_sphere(n: int, seed: int = 0, dtype=np.float64) -> State:
    """N equal-mass particles drawn from a Plummer model; generated in fp64, cast to `dtype` at the end."""
    if n <= 0:
        raise ValueError("n must be > 0")
    rng = np.random.Generator(np.random.PCG64(seed))
    a = PLUMMER_A
    # radii
    r = np.empty(n)
    todo = np.arange(n)
    while todo.size:
        u = rng.uniform(0.0, 1.0, todo.size)
        u = np.where(u <= 0.0, 0.5, u)
        rr = a / np.sqrt(u ** (-2.0 / 3.0) - 1.0)
        ok = rr <= 10.0 * a
        r[todo[ok]] = rr[ok]
        todo = todo[~ok]
    pos = r[:, None] * _isotropic(rng, n)
    # speeds: q = v / v_esc with pdf ~ q^2 (1-q^2)^{7/2}, max of g is below 0.1
    q = np.empty(n)
    todo = np.arange(n)
    while todo.size:
        x = rng.uniform(0.0, 1.0, todo.size)
        y = rng.uniform(0.0, 0.1, todo.size)
        ok = y < x * x * (1.0 - x * x) ** 3.5
        q[todo[ok]] = x[ok]
        todo = todo[~ok]
    v_esc = np.sqrt(2.0) * (r * r + a * a) ** -0.25   # sqrt(-2 phi), phi = -GM / sqrt(r^2 + a^2)
    vel = (q * v_esc)[:, None] * _isotropic(rng, n)
    mass = np.full(n, 1.0 / n)
    pos -= pos.mean(axis=0)
    vel -= vel.mean(axis=0)
    return State(pos.astype(dtype), vel.astype(dtype), mass.astype(dtype))

"""Particle state exchanged with callers: array-of-vectors, like the CONCEPT snapshots the reference's
scripts read (`components/matter/pos` (N,3) f64 — SnapVsPK/Power_compare.py:143-144), with velocities in
place of momenta. The engine itself keeps structure-of-arrays on the device ("HBM layout")."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class State:
    pos: np.ndarray   # (N, 3)
    vel: np.ndarray   # (N, 3)
    mass: np.ndarray  # (N,)
    acc: Optional[np.ndarray] = None  # cached a(pos) after step(), or None

    @property
    def n(self) -> int:
        return int(self.pos.shape[0])

    def copy(self) -> "State":
        return State(self.pos.copy(), self.vel.copy(), self.mass.copy(),
                     None if self.acc is None else self.acc.copy())

    def astype(self, dtype) -> "State":
        return State(self.pos.astype(dtype), self.vel.astype(dtype), self.mass.astype(dtype),
                     None if self.acc is None else self.acc.astype(dtype))

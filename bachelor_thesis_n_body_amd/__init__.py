"""bachelor_thesis_n_body_amd — MI355X-native direct-summation N-body engine (hot path only).

The directory is named with underscores because Python identifiers cannot contain hyphens; it is the
package the task calls `bachelor-thesis-n-body_amd/`.

Public surface (mirrors oracle/nbody_oracle.py so parity tests read the same on both sides):
    compute_forces(pos, mass, eps, G=1.0)         SURVEY.md §8a ⊕1
    step(state, dt, nsteps=1, eps=..., G=1.0)     ⊕2 (kick-drift-kick)
    energy(state, eps, G=1.0) -> (K, U)           ⊕3
    plummer_sphere(n, seed)                       ⊕4
    Engine                                         resident single-GPU context (nb_ctx)
    ShardedEngine, partition                       sink decomposition over the GPUs of a node (§8e)
    field, snapshot                                §8f-1/2: CIC density / deconvolution / P(k) on the GPU, CONCEPT snapshot I/O
Everything computes on the GPU through libnbody_amd.so (include/nbody_amd.h); there is no CPU fallback.
"""
from .state import State
from .ics import plummer_sphere, PLUMMER_A
from .api import Engine, ClockProbe, compute_forces, step, energy
from .dist import ShardedEngine, HipKernels, partition
from . import field, grf, snapshot

__all__ = ["State", "plummer_sphere", "PLUMMER_A", "Engine", "ClockProbe", "compute_forces", "step", "energy",
           "ShardedEngine", "HipKernels", "partition", "field", "grf", "snapshot"]
__version__ = "0.1.0"

"""ctypes binding of libnbody_amd.so (C-ABI: include/nbody_amd.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load, importing the compute
entry points raises. The CPU oracle under oracle/ is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

NB_F32, NB_F64 = 0, 1
NB_OK = 0
NB_ACCUMULATE, NB_UNIFORM_MASS = 1, 2
NB_SOLVER_DIRECT, NB_SOLVER_TREE, NB_SOLVER_EWALD, NB_SOLVER_PM, NB_SOLVER_P3M = 0, 1, 2, 3, 4
NB_MESH_NO_INTERLACE, NB_MESH_LONG_ONLY, NB_MESH_SHORT_ONLY = 1, 2, 4

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libnbody_amd.so"

# NB_OPT_* of include/nbody_amd.h (nb_set_option): the pair-symmetric all-pairs kernels
OPTIONS = {"pair_symmetry": 1, "sym_min_n": 2, "sym_slab_bytes": 3, "sym_scratch_limit": 4}
QUERIES = {"multi_schedule": 100, "own_chunk_symmetric": 101}

# every symbol include/nbody_amd.h declares: (name, restype, argtypes)
_vp, _i64, _dbl, _int = C.c_void_p, C.c_int64, C.c_double, C.c_int
SIGNATURES = {
    "nb_version": (_int, []),
    "nb_device_count": (_int, []),
    "nb_last_error": (C.c_char_p, [_vp]),
    "nb_create": (_int, [C.POINTER(_vp), _i64, _int, _int]),
    "nb_create_multi": (_int, [C.POINTER(_vp), _i64, _int, _int, C.POINTER(_int)]),
    "nb_multi_layout": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), _int]),
    "nb_destroy": (_int, [_vp]),
    "nb_set_stream": (_int, [_vp, _vp]),
    "nb_synchronize": (_int, [_vp]),
    "nb_set_solver": (_int, [_vp, _int, _dbl]),
    "nb_set_periodic": (_int, [_vp, _dbl, _dbl, _int]),
    "nb_set_mesh": (_int, [_vp, _int, _dbl, _int, _dbl, _dbl, _int]),
    "nb_kdk_step": (_int, [_vp, _dbl, _dbl, _dbl, _dbl, _dbl]),
    "nb_kdk_run": (_int, [_vp, _i64, _vp, _vp, _vp, _dbl, _dbl]),
    "nb_rung_step": (_int, [_vp, _int, _vp, _vp, _dbl, _dbl, _dbl, _dbl, C.POINTER(_i64)]),
    "nb_last_run_used_graph": (_int, [_vp]),
    "nb_set_option": (_int, [_vp, _int, _i64]),
    "nb_last_accel_symmetric": (_int, [_vp]),
    "nb_get_option": (_int, [_vp, _int, C.POINTER(_i64)]),
    "nb_clock_probe_begin": (_int, [_int, _dbl, C.POINTER(_vp)]),
    "nb_clock_probe_stop": (_int, [_vp, _vp]),
    "nb_clock_probe_end": (_int, [_vp, C.POINTER(_dbl), C.POINTER(_dbl)]),
    "nb_clock_probe_end_xcd": (_int, [_vp, C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_dbl), _int]),
    "nb_set_state": (_int, [_vp] + [_vp] * 7),
    "nb_get_state": (_int, [_vp] + [_vp] * 7),
    "nb_accel": (_int, [_vp, _dbl, _dbl, _vp, _vp, _vp]),
    "nb_get_acc": (_int, [_vp, _vp, _vp, _vp]),
    "nb_step": (_int, [_vp, _dbl, _i64, _dbl, _dbl]),
    "nb_energy": (_int, [_vp, _dbl, _dbl, C.POINTER(_dbl), C.POINTER(_dbl)]),
    "nb_step_energy": (_int, [_vp, _dbl, _i64, _dbl, _dbl, _vp, _vp]),
    "nb_accel_dev": (_int, [_vp, _int, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _int]),
    "nb_accel_pot_dev": (_int, [_vp, _int, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _vp, _int]),
    "nb_accel_pair_dev": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int]),
    "nb_pot_energy_dev": (_int, [_vp, _int, _vp, _vp, _i64, _dbl, _int, _vp]),
    "nb_potential_dev": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _dbl, _vp, _int]),
    "nb_kinetic_dev": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _i64, _vp]),
    "nb_kick_dev": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl]),
    "nb_drift_dev": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl]),
    "nb_kick_drift_dev": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl]),
    "nb_accel_kdk_dev": (_int, [_vp, _int, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp, _vp, _vp, _int,
                                _vp, _vp, _vp, _dbl, _dbl, _dbl]),
}

_lib = None


def load() -> C.CDLL:
    """Load libnbody_amd.so (once). Raises RuntimeError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("NBODY_AMD_LIB", LIB_PATH))
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C bachelor_thesis_n_body_amd/csrc`) — there is no CPU fallback.")
    try:
        lib = C.CDLL(str(path))
    except OSError as e:  # e.g. libamdhip64 missing
        raise RuntimeError(f"cannot load {path}: {e} — there is no CPU fallback") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means header and library disagree
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(ctx, rc: int, what: str) -> None:
    if rc != NB_OK:
        msg = load().nb_last_error(ctx)
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else 'unknown error'}")

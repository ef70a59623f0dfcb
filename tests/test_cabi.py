"""The C-ABI library loads and exports exactly what include/nbody_amd.h declares; without a GPU the product
fails loudly (no CPU fallback). No compute call is made here."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "nbody_amd.h"
FIELD_HEADER = ROOT / "include" / "nbody_field.h"


def declared_functions(header=HEADER, prefix="nb_"):
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    names = declared_functions()
    for must in ("nb_create", "nb_destroy", "nb_set_state", "nb_get_state", "nb_accel", "nb_step", "nb_energy",
                 "nb_last_error", "nb_device_count", "nb_accel_dev", "nb_potential_dev", "nb_kick_dev", "nb_drift_dev"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from bachelor_thesis_n_body_amd import _lib
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/nbody_amd.h but not exported"
    # and the ctypes table binds exactly the header's functions
    assert sorted(_lib.SIGNATURES) == declared_functions()
    assert lib.nb_version() >= 100
    from bachelor_thesis_n_body_amd import field
    names = declared_functions(FIELD_HEADER, "nbf_")
    assert len(names) >= 10
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/nbody_field.h but not exported"
    from bachelor_thesis_n_body_amd import grf
    assert sorted(list(field.FIELD_SIGNATURES) + list(grf.GRF_SIGNATURES)) == names


def test_abi_is_plain_c():
    # no torch / C++ types in the signatures: only void*, const void*, ints, doubles and the opaque nb_ctx
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    assert 'extern "C"' in text
    for bad in ("std::", "torch", "at::", "hipStream_t", "template", "&"):
        assert bad not in text, bad


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from bachelor_thesis_n_body_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("NBODY_AMD_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_has_no_cpu_path(has_gpu):
    """Without a device every compute entry point raises; it never silently computes on the host."""
    if has_gpu:
        pytest.skip("GPU present")
    import bachelor_thesis_n_body_amd as nb
    s = nb.plummer_sphere(16, 0)
    with pytest.raises(RuntimeError, match="nb_create failed"):
        nb.compute_forces(s.pos, s.mass, 0.01)
    with pytest.raises(RuntimeError):
        nb.step(s, 0.01, 1, eps=0.01)
    with pytest.raises(RuntimeError):
        nb.energy(s, 0.01)


def test_bad_arguments_return_error_codes():
    from bachelor_thesis_n_body_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.nb_create(C.byref(ctx), 0, 0, 0) == -1          # NB_ERR_ARG: n <= 0
    assert b"n must be > 0" in lib.nb_last_error(None)
    assert lib.nb_create(C.byref(ctx), 10, 7, 0) == -1         # unknown dtype
    assert lib.nb_create(None, 10, 0, 0) == -1
    assert lib.nb_destroy(None) == 0


def test_product_never_imports_the_oracle():
    """Static check of the tier rule: nothing under the package (Python or HIP) refers to oracle/."""
    pkg = ROOT / "bachelor_thesis_n_body_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        src = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
        assert "libnbody_oracle" not in src and "nbody_oracle" not in src.replace("oracle/nbody_oracle", ""), f

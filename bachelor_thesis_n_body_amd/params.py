"""Reader for the subset of CONCEPT `.param` files the thesis uses.

The reference's parameter files are executable Python with unit symbols, `_user` variables and a `path` object
(`Different_Sizes/Giant.param:1-85`, `TheFeedingPipeline/NN.param:1-95`, `SnapVsPK/Uncertain.param`), templated by
line-rewriting in `SnapVsPK/Ucertain.py:22-67` and `TheFeedingPipeline/NeuNetværk_test.py:79-120` (which prepend
`path.output_dir = ...`, `path.job_dir = ...` and a `random_seeds = {...}` block). This module evaluates such a file
in a namespace that provides those symbols and returns the handful of keys the engine understands:

    initial_conditions['N'], boxsize, output_dirs['snapshot'], output_times['snapshot'],
    select_forces[species]['gravity'], random_seeds, primordial_spectrum, a_begin, H0, Ωb, Ωcdm

Unit system: Mpc = Gyr = 1, mass unit 1e10 m_sun — the units CONCEPT records in its snapshots
(`unit length`, `unit time`, `unit mass` root attributes).  Forward references (NN.param uses `primordial_spectrum`
in `_tag` 90 lines before defining it) are resolved by evaluating the statements repeatedly until no new one succeeds.
"""
from __future__ import annotations

import ast
import math
from pathlib import Path
from types import SimpleNamespace
from typing import Any, Dict

# units: Mpc, Gyr, 1e10 m_sun
_KM = 1.0 / 3.0856775814913673e19          # km in Mpc
_S = 1.0 / 3.15576e16                      # s in Gyr (Julian year)
UNITS = {
    "Mpc": 1.0, "kpc": 1e-3, "pc": 1e-6, "Gpc": 1e3, "km": _KM, "m": _KM * 1e-3,
    "Gyr": 1.0, "Myr": 1e-3, "kyr": 1e-6, "yr": 1e-9, "s": _S, "minutes": 60 * _S, "hr": 3600 * _S, "day": 86400 * _S,
    "m_sun": 1e-10, "km_s": _KM / _S,
    "pi": math.pi, "π": math.pi, "sqrt": math.sqrt, "cbrt": lambda x: x ** (1.0 / 3.0), "inf": math.inf,
    "exp": math.exp, "log": math.log, "sin": math.sin, "cos": math.cos,
}
G_NEWTON = 4.300917270e-9 * (1e10) * (_KM / _S) ** 2   # G in Mpc^3 / (1e10 m_sun) / Gyr^2  (4.3009e-9 Mpc km^2 s^-2 / m_sun)


class ParamError(ValueError):
    pass


class _Namespace(dict):
    """Globals of the parameter file: `h` is derived lazily from H0 (Giant.param:78-80 writes `0.02237 /h**2`)."""

    def __missing__(self, key):
        if key == "h":
            if "H0" not in self:
                raise NameError("h")
            return self["H0"] / (100 * UNITS["km"] / (UNITS["s"] * UNITS["Mpc"]))
        raise KeyError(key)


def parse(path, overrides: Dict[str, Any] | None = None) -> Dict[str, Any]:
    """Evaluate a `.param` file and return its user-visible variables (a plain dict, plus 'path')."""
    text = Path(path).read_text(encoding="utf-8")
    try:
        tree = ast.parse(text, filename=str(path))
    except SyntaxError as e:
        raise ParamError(f"{path}: {e}") from e
    ns = _Namespace(UNITS)
    ns["__builtins__"] = {"range": range, "len": len, "int": int, "float": float, "str": str, "min": min, "max": max,
                          "abs": abs, "round": round, "sum": sum, "NameError": NameError, "True": True, "False": False,
                          "None": None, "dict": dict, "list": list, "tuple": tuple, "set": set, "print": print}
    base = Path(path).resolve().parent
    ns["path"] = SimpleNamespace(output_dir=str(base / "output"), ic_dir=str(base / "ic"), job_dir=str(base / "job"),
                                 param_dir=str(base), concept_dir=str(base))
    for k, v in (overrides or {}).items():
        ns[k] = v
    pending = list(tree.body)
    errors: Dict[int, Exception] = {}
    while pending:
        still = []
        for node in pending:
            code = compile(ast.Module(body=[node], type_ignores=[]), str(path), "exec")
            try:
                exec(code, ns)
            except (NameError, KeyError) as e:     # forward reference: try again after the rest
                errors[id(node)] = e
                still.append(node)
            except Exception as e:
                raise ParamError(f"{path}:{getattr(node, 'lineno', '?')}: {type(e).__name__}: {e}") from e
        if len(still) == len(pending):
            node = still[0]
            raise ParamError(f"{path}:{getattr(node, 'lineno', '?')}: unresolved name: {errors[id(node)]}")
        pending = still
    out = {k: v for k, v in ns.items() if k not in UNITS and k != "__builtins__"}
    return out


def _gridsize_value(v, n: int) -> int:
    """An entry of `potential_options['gridsize']`: a number, or an expression in N such as 'cbrt(N)' / '2*cbrt(N)'
    (Different_Sizes/Giant.param:33-36); pairs (upstream, downstream) are given as 2-sequences."""
    if isinstance(v, (list, tuple)):
        v = v[0]
    if isinstance(v, (set, frozenset)):                      # Giant.param:31 writes {'global': {_size}}
        v = next(iter(v))
    if isinstance(v, str):
        cb = round(n ** (1.0 / 3.0))
        v = eval(v.replace("Ñ", "N"), {"__builtins__": {}}, {"N": n, "cbrt": lambda x: round(x ** (1.0 / 3.0)), "cbrtN": cb})
    return int(round(float(v)))


def mesh_options(p: Dict[str, Any], gravity, n: int) -> Dict[str, Any]:
    """`potential_options` reduced to what the PM / P3M solvers use (TheFeedingPipeline/NN.param:53-82,
    Different_Sizes/Giant.param:28-63): mesh size, interlacing, and the P3M short-range scale / range in CONCEPT's
    `shortrange_params` convention. Unspecified values take CONCEPT's documented defaults: gridsize cbrt(N) for pm and
    2 cbrt(N) for p3m, no interlacing, scale 1.25 cells, range 4.5 scales. Differentiation is always done in Fourier space
    (the choice NN.param:55-61 makes); a finite-difference order in the file is accepted and noted by the CLI."""
    out = {"gridsize": None, "interlace": False, "scale": 1.25, "range": 4.5, "differentiation": "Fourier"}
    if gravity not in ("pm", "p3m") or n <= 0:
        return out
    cb = round(n ** (1.0 / 3.0))
    out["gridsize"] = cb if gravity == "pm" else 2 * cb
    po = p.get("potential_options", {})
    if isinstance(po, dict):
        g = po.get("gridsize")
        if isinstance(g, dict):
            sub = g.get("particles", {})
            sub = sub.get("gravity", {}) if isinstance(sub, dict) else {}
            if isinstance(sub, dict) and gravity in sub:
                out["gridsize"] = _gridsize_value(sub[gravity], n)
            elif g.get("global"):
                out["gridsize"] = _gridsize_value(g["global"], n)
        elif g is not None:
            out["gridsize"] = _gridsize_value(g, n)
        il = po.get("interlace")
        if isinstance(il, dict):
            il = il.get("gravity", il)
            il = il.get(gravity, il.get("default", False)) if isinstance(il, dict) else il
        if il is not None and not isinstance(il, dict):
            out["interlace"] = bool(any(il)) if isinstance(il, (list, tuple)) else bool(il)
        df = po.get("differentiation")
        while isinstance(df, dict):
            df = df.get(gravity, df.get("gravity", df.get("all", df.get("default", next(iter(df.values()), None)))))
        if df is not None:
            out["differentiation"] = df
    sr = p.get("shortrange_params", {})
    if isinstance(sr, dict):
        sr = sr.get("gravity", sr)
        if isinstance(sr, dict):
            for key in ("scale", "range"):
                v = sr.get(key)
                if isinstance(v, (int, float)):
                    out[key] = float(v)
    return out


def _pick(d, component: str = "matter", default=None):
    """A per-component option dict of CONCEPT ({'default': v} / {'matter': v} / {'all': v} / a bare value)."""
    if not isinstance(d, dict):
        return default if d is None else d
    for key in (component, "all", "default"):
        if key in d:
            return d[key]
    return default


def powerspec_options(p: Dict[str, Any], n: int, boxsize: float) -> Dict[str, Any]:
    """`powerspec_options` of the reference's parameter files (TheFeedingPipeline/NN.param:28-45; SnapVsPK/Uncertain.param:29-60, whose
    comments cite CONCEPT's documentation for the defaults), reduced to what the estimator here acts on:

      'upstream gridsize' {'particles': 'cbrt(Ñ)'}      -> gridsize of the CIC mesh (an expression in N / Ñ, or a number)
      'interpolation' {'default': 'CIC'}                -> only 'CIC' is implemented (the thesis changed CONCEPT's 'PCS' to 'CIC')
      'deconvolve', 'interlace', 'realization correction' -> booleans
      'k_max' {'default': 'nyquist'}                    -> 'nyquist' (= pi gridsize / boxsize), a number, or an expression in nyquist / k_min / k_max
      'bins per decade' {'default': {'  4*k_min': 4, '100*k_min': 40}} -> [(k, bins per decade)] sorted by k; keys are expressions in
                          k_min = 2 pi / boxsize, k_max, nyquist and gridsize; None when the key is absent (the CLI then keeps its
                          linear bins of width 2 k_min)
      'tophat' {'default': 8*Mpc/h}                     -> radius of the sigma printed in the table header
      'significant figures' {'default': 8}              -> digits of the table

    Values are validated here so that a parameter file the estimator cannot honour fails before the run."""
    po = p.get("powerspec_options", {})
    if not isinstance(po, dict):
        raise ParamError("powerspec_options must be a dict")
    cb = round(n ** (1.0 / 3.0)) if n > 0 else 0
    up = po.get("upstream gridsize", {})
    up = up.get("particles", up.get("default", None)) if isinstance(up, dict) else up
    grid = _gridsize_value(up, n) if up is not None and n > 0 else cb
    interp = str(_pick(po.get("interpolation"), default="CIC")).upper()
    if interp != "CIC":
        raise ParamError(f"powerspec_options['interpolation'] = {interp!r}: only 'CIC' is implemented")
    k_min = 2.0 * math.pi / float(boxsize)
    nyq = math.pi * grid / float(boxsize) if grid else math.inf
    env = {"k_min": k_min, "nyquist": nyq, "gridsize": grid, "sqrt": math.sqrt, "pi": math.pi, "π": math.pi, "inf": math.inf}

    def number(v, what, extra=None):
        if isinstance(v, (int, float)):
            return float(v)
        try:
            return float(eval(str(v).strip(), {"__builtins__": {}}, dict(env, **(extra or {}))))
        except Exception as e:
            raise ParamError(f"powerspec_options[{what!r}]: cannot evaluate {v!r}: {e}") from e
    k_max = number(_pick(po.get("k_max"), default="nyquist"), "k_max")
    bpd = _pick(po.get("bins per decade"), default=None)
    table = None
    if bpd is not None:
        if isinstance(bpd, (int, float)):
            table = [(k_min, float(bpd))]
        elif isinstance(bpd, dict) and bpd:
            table = sorted((number(k, "bins per decade", {"k_max": k_max}), float(v)) for k, v in bpd.items())
        else:
            raise ParamError("powerspec_options['bins per decade'] must be a number or a non-empty {k: bins} dict")
        if any(not (k > 0 and b > 0) for k, b in table):
            raise ParamError("powerspec_options['bins per decade']: wavenumbers and bin densities must be positive")
    return {"gridsize": int(grid), "interpolation": interp,
            "deconvolve": bool(_pick(po.get("deconvolve"), default=True)), "interlace": bool(_pick(po.get("interlace"), default=True)),
            "realization_correction": bool(_pick(po.get("realization correction"), default=True)),
            "k_max": k_max, "bins_per_decade": table,
            "tophat": number(_pick(po.get("tophat"), default=8.0 / (p["h"] if "h" in p else 0.6736)), "tophat") if po.get("tophat") is not None or True else None,
            "significant_figures": int(_pick(po.get("significant figures"), default=8))}


def settings(p: Dict[str, Any]) -> Dict[str, Any]:
    """The keys the engine acts on, validated (SURVEY.md §8f-4 list)."""
    ic = p.get("initial_conditions")
    if isinstance(ic, dict):
        n = int(ic.get("N", 0))
        species = ic.get("species", "matter")
    elif isinstance(ic, str):
        n, species = 0, "matter"
    else:
        raise ParamError("initial_conditions missing")
    if "boxsize" not in p:
        raise ParamError("boxsize missing")
    forces = p.get("select_forces", {})
    gravity = None
    for comp, d in forces.items():
        if isinstance(d, dict) and "gravity" in d:
            gravity = d["gravity"]
    mesh = mesh_options(p, gravity, n)
    out_dirs = p.get("output_dirs", {})
    times = p.get("output_times", {})
    snap_times = times.get("snapshot", times.get("a", {}).get("snapshot", [])) if isinstance(times, dict) else []
    if not isinstance(snap_times, (list, tuple)):
        snap_times = [snap_times]
    return {
        "N": n, "species": species, "ic_file": ic if isinstance(ic, str) else None,
        "boxsize": float(p["boxsize"]),
        "snapshot_dir": out_dirs.get("snapshot"), "powerspec_dir": out_dirs.get("powerspec"),
        "snapshot_times": [float(t) for t in snap_times],
        "powerspec_times": [float(t) for t in (times.get("powerspec", []) if isinstance(times, dict) else [])],
        "gravity": gravity, "mesh": mesh, "powerspec": powerspec_options(p, n, float(p["boxsize"])),
        "lpt": int(p.get("realization_options", {}).get("LPT", 1)) if isinstance(p.get("realization_options", {}), dict) else 1,
        "random_seeds": dict(p.get("random_seeds", {})),
        "primordial_spectrum": dict(p.get("primordial_spectrum", {})),
        "a_begin": float(p.get("a_begin", 1.0)),
        "H0": p.get("H0"), "Ωb": p.get("Ωb"), "Ωcdm": p.get("Ωcdm"),
    }

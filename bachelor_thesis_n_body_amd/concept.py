"""`concept`-style command line for the engine (first slice).

The thesis drives its solver as a subprocess:
    [<run_dir>/bin/concept, "-p", <param file>, "-n", "1", "--local"]          SnapVsPK/Ucertain.py:89-97,
                                                                               TheFeedingPipeline/NeuNetværk_test.py:155-171
reads nothing from it but the exit status, and later opens `<output_dirs['snapshot']>/snapshot_a=1.00.hdf5`
(`FraP_kTilDelta.py:129-146`). This module accepts that command line and that parameter-file subset
(`params.py`) and leaves the same artefacts behind, so a farm script can point `concept` at
`python -m bachelor_thesis_n_body_amd.concept`.

Two modes. Default: the ISOLATED NEWTONIAN problem of SURVEY.md §0 item 2 (only `'gravity': 'pp'`; the scale factors in
`output_times` are file labels, the run length is `--steps` fixed KDK steps of `--dt` Gyr). With `--cosmological`: what
CONCEPT itself integrates — comoving coordinates in an expanding flat LCDM background, periodic box, Zel'dovich initial
conditions at `a_begin` — with `'gravity': 'pp'` (Ewald summation), `'pm'` or `'p3m'` (mesh size, interlacing and
short-range parameters from `potential_options` / `shortrange_params`, `params.mesh_options`). Initial conditions: a CONCEPT snapshot (`--ic`, or
`initial_conditions = '<path>'`), or a Zel'dovich-displaced lattice of N = n^3 particles at rest from a tabulated P(k)
(`--pk-table`, two columns k [1/Mpc], P [Mpc^3]) seeded by `random_seeds`.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

from . import params as _params


def _write_powerspec(path: Path, k, pk, counts, a: float, ngrid: int, corrected=None, linear=None, digits: int = 8) -> None:
    """Text table in the layout of CONCEPT's powerspec files — k, modes, P, P corrected, P linear
    (PPPMP3M/PP/powerspec/powerspec_a=1.00:1-26) — so that the thesis' loaders, which read columns (0, 2, 3, 4)
    (PPPMP3M/pppmp3mplotter.py:123-134, SnapVsPK/Power_compare.py:625-629), open it unchanged. Unknown columns hold nan."""
    nan = np.full(len(k), np.nan)
    corrected = nan if corrected is None else corrected
    linear = nan if linear is None else linear
    with open(path, "w", encoding="utf-8") as fh:
        fh.write(f"# Power spectrum from bachelor_thesis_n_body_amd at a = {a:g}\n#\n# Components:\n#   0: matter  (upstream grid size {ngrid})\n#\n")
        fh.write(f"#                          grid size {ngrid}\n#                      component 0    (corrected)      (linear)\n")
        fh.write("#  k [Mpc⁻¹]   modes     P [Mpc³]       P [Mpc³]       P [Mpc³]\n")
        d = max(1, int(digits) - 1)          # powerspec_options['significant figures'] (default 8, as the reference's tables)
        for ki, ci, pi, pc, pl in zip(k, counts, pk, corrected, linear):
            fh.write(f"{ki:.{d}e} {int(ci):6d}  {pi:.{d}e}  {pc:.{d}e}  {pl:.{d}e}\n")


def _load_pk_table(args, s) -> np.ndarray:
    tab = np.loadtxt(args.pk_table)[:, :2].astype(np.float64)
    a_s = s["primordial_spectrum"].get("A_s")
    if args.pk_table_As is not None and a_s is not None:
        tab[:, 1] *= float(a_s) / float(args.pk_table_As)
    return tab


def _run_cosmological(args, p, s) -> int:
    """CONCEPT-like run: Zel'dovich / 2LPT ICs at a_begin from the linear P(k) table (at a = 1) — or a snapshot to resume from —
    comoving KDK with periodic 'pp' / 'pm' / 'p3m' forces up to the last requested scale factor, snapshot and power spectrum at
    their requested output times."""
    from . import cosmo, snapshot
    L = s["boxsize"]
    if not p.get("H0"):
        print("concept: --cosmological needs H0/Ωb/Ωcdm in the parameter file", file=sys.stderr)
        return 2
    h = float(p["H0"]) / (100 * _params.UNITS["km"] / _params.UNITS["s"])
    c = cosmo.Cosmology(float(p["H0"]), float(p["Ωb"]) + float(p["Ωcdm"]), (2.473e-5 * (1.0 + 0.2271 * 3.046)) / h ** 2)
    seeds = s["random_seeds"]
    ic_file = args.ic or s["ic_file"]
    table = _load_pk_table(args, s) if args.pk_table else None
    if ic_file:
        # resume from a CONCEPT-layout snapshot (`initial_conditions = '<path>'`): positions, comoving momenta a^2 m dx/dt and
        # the scale factor it was written at (root attribute 'a', as in the reference's own snapshots)
        pos = snapshot.read_positions(ic_file)
        attrs, comp = snapshot.read_attrs(ic_file), snapshot.read_attrs(ic_file, "components/matter")
        vel = snapshot.read_momenta(ic_file) / float(comp["mass"])
        a0 = float(attrs.get("a", s["a_begin"]))
        n_side = round(pos.shape[0] ** (1.0 / 3.0))
        noise_power = None
        if float(attrs.get("boxsize", L)) != L or n_side ** 3 != pos.shape[0]:
            print("concept: the snapshot's boxsize / particle number does not fit the parameter file", file=sys.stderr)
            return 2
    else:
        n_side = round(s["N"] ** (1.0 / 3.0))
        if n_side ** 3 != s["N"] or table is None:
            print("concept: --cosmological needs N = n^3 and --pk-table (linear P(k) at a = 1) to generate initial conditions", file=sys.stderr)
            return 2
        a0 = s["a_begin"]
        if s["lpt"] not in (1, 2):
            print(f"concept: realization_options['LPT'] = {s['lpt']}: orders 1 and 2 are implemented", file=sys.stderr)
            return 3
        pos, vel, noise_power = cosmo.zeldovich_ics(table, n_side, L, c, a0, int(seeds.get("primordial amplitudes", 0)),
                                                    int(seeds.get("primordial phases", 1)), lpt=s["lpt"], return_noise_power=True)
    mass = float(comp["mass"]) if ic_file else c.particle_mass(L, n_side ** 3)
    snap_t = set(t for t in (s["snapshot_times"] or [1.0]) if t > a0)
    # power spectra: at `output_times['powerspec']` when given (SnapVsPK/Uncertain.param:13-16), else with every snapshot
    ps_t = set(t for t in (s["powerspec_times"] or (snap_t if s["powerspec_dir"] else [])) if t > a0)
    outs = sorted(snap_t | ps_t)
    if not outs:
        print(f"concept: nothing to do: every output time is at or before the starting scale factor a = {a0:g}", file=sys.stderr)
        return 0
    out = Path(s["snapshot_dir"])
    out.mkdir(parents=True, exist_ok=True)
    nsteps = args.steps if args.steps > 0 else 400
    total = np.log(outs[-1] / a0)
    for a1 in outs:
        k = max(1, round(nsteps * np.log(a1 / a0) / total))
        if args.schedule == "rungs":
            if s["gravity"] != "p3m":
                print("concept: --schedule rungs needs select_forces gravity 'p3m' (the rungs belong to its short-range force)", file=sys.stderr)
                return 2
            pos, vel = cosmo.run_rungs(pos, vel, mass, L, c, a0, a1, eps=args.eps, dtype=np.float32 if args.dtype == "f32" else np.float64,
                                       ngrid=s["mesh"]["gridsize"], interlace=s["mesh"]["interlace"], scale_cells=s["mesh"]["scale"],
                                       range_scales=s["mesh"]["range"])
        else:
            pos, vel = cosmo.run_pp(pos, vel, mass, L, c, a0, a1, nsteps=k, eps=args.eps, gravity=s["gravity"],
                                    dtype=np.float32 if args.dtype == "f32" else np.float64,
                                    ngrid=s["mesh"]["gridsize"], interlace=s["mesh"]["interlace"],
                                    scale_cells=s["mesh"]["scale"], range_scales=s["mesh"]["range"],
                                    schedule="concept" if args.schedule == "concept" else None)
        if a1 in snap_t:
            snapshot.write_snapshot(str(out / f"snapshot_a={a1:.2f}.hdf5"), pos, vel * mass, L, mass=mass,
                                    extra_attrs={"a": float(a1), "H0": float(p["H0"]), "Ωb": float(p["Ωb"]), "Ωcdm": float(p["Ωcdm"])})
        if a1 in ps_t:
            pdir = Path(s["powerspec_dir"] or out.parent / "powerspec"); pdir.mkdir(parents=True, exist_ok=True)
            po = s["powerspec"]              # powerspec_options of the parameter file (params.powerspec_options)
            grid = po["gridsize"] or n_side
            kk, cnt, pk, pcorr, plin = cosmo.concept_power_table(pos, L, grid, table, c.growth(a1)[0], noise_power if grid == n_side else None,
                                                                 bins_per_decade=po["bins_per_decade"], k_max=min(po["k_max"], np.pi * grid / L),
                                                                 interlace=po["interlace"], deconvolve=po["deconvolve"])
            if not po["realization_correction"]:
                pcorr = np.full_like(pk, np.nan)
            _write_powerspec(pdir / f"powerspec_a={a1:.2f}", kk, pk, cnt, a1, grid, pcorr, plin, po["significant_figures"])
        a0 = a1
    print(f"concept: cosmological {s['gravity']} run finished at a = {a0:g}; {len(snap_t)} snapshot(s) in {out}")
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="concept", description=__doc__.split("\n\n")[0])
    ap.add_argument("-p", "--param", required=True)
    ap.add_argument("-n", "--nprocs", default="1", help="accepted for compatibility (one process per GPU)")
    ap.add_argument("--local", action="store_true", help="accepted for compatibility")
    ap.add_argument("-c", "--command-line-param", action="append", default=[], help="extra `name = value` lines")
    ap.add_argument("--ic", help="CONCEPT-layout snapshot to start from")
    ap.add_argument("--pk-table", help="two-column text file k [1/Mpc], P(k) [Mpc^3] for generated initial conditions")
    ap.add_argument("--pk-table-As", type=float, default=None,
                    help="primordial amplitude A_s the --pk-table was computed for: the table is rescaled by "
                         "primordial_spectrum['A_s'] / this value (linear P(k) is proportional to A_s), so one table serves a farm "
                         "that sweeps A_s per job (TheFeedingPipeline/NeuNetværk_test.py:100-137)")
    ap.add_argument("--dt", type=float, default=0.01, help="KDK step in Gyr")
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--eps", type=float, default=None, help="Plummer softening in Mpc (default 0.03 mean separations)")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f64")
    ap.add_argument("--solver", choices=["direct", "tree"], default="direct")
    ap.add_argument("--theta", type=float, default=0.5)
    ap.add_argument("--schedule", choices=["lna", "concept", "rungs"], default="lna",
                    help="time steps of a --cosmological run: 'lna' = --steps steps uniform in ln a (default); 'concept' = CONCEPT's own base steps "
                         "as the reference's run logs record them (143 from a = 0.02 to 1; cosmo.load_concept_schedule); 'rungs' = those base steps with a rung "
                         "hierarchy for the short-range force ('gravity': 'p3m' only; cosmo.run_rungs — CONCEPT's time-loop structure)")
    ap.add_argument("--cosmological", action="store_true",
                    help="comoving integration from a_begin to the largest output time in a periodic box (Ewald forces, fp64): "
                         "the physics of a CONCEPT `pp` run; --steps counts steps uniform in ln a (default 400)")
    args = ap.parse_args(argv)

    try:
        p = _params.parse(args.param)
        for line in args.command_line_param:
            exec(line, {**_params.UNITS}, p)
        s = _params.settings(p)
    except _params.ParamError as e:
        print(f"concept: {e}", file=sys.stderr)
        return 2
    if s["gravity"] not in ("pp", "pm", "p3m"):
        print(f"concept: select_forces gravity = {s['gravity']!r}: this engine implements 'pp', 'pm' and 'p3m'", file=sys.stderr)
        return 3
    if s["gravity"] != "pp" and not args.cosmological:
        print(f"concept: 'gravity': {s['gravity']!r} is a periodic-box solver: run it with --cosmological "
              "(the isolated Newtonian mode implements 'pp' only)", file=sys.stderr)
        return 3
    if s["gravity"] != "pp" and s["mesh"]["differentiation"] != "Fourier":
        print(f"concept: note: potential differentiation {s['mesh']['differentiation']!r} requested; this engine differentiates "
              "in Fourier space", file=sys.stderr)
    if not s["snapshot_dir"]:
        print("concept: output_dirs['snapshot'] missing", file=sys.stderr)
        return 2

    from . import Engine, State, field, grf, snapshot   # needs the HIP library from here on

    L = s["boxsize"]
    if args.cosmological:
        return _run_cosmological(args, p, s)
    ic_file = args.ic or s["ic_file"]
    if ic_file:
        pos = snapshot.read_positions(ic_file)
        mom = snapshot.read_momenta(ic_file)
        attrs = snapshot.read_attrs(ic_file, "components/matter")
        mass = float(attrs.get("mass", 1.0))
        vel = mom / mass
        n = pos.shape[0]
    else:
        n_side = round(s["N"] ** (1.0 / 3.0))
        if n_side ** 3 != s["N"]:
            print(f"concept: initial_conditions['N'] = {s['N']} is not a cube", file=sys.stderr)
            return 2
        if not args.pk_table:
            print("concept: generated initial conditions need --pk-table (CLASS is not available to this engine)", file=sys.stderr)
            return 2
        tab = _load_pk_table(args, s)
        P = grf.tabulated_pk_grid(tab[:, 0], tab[:, 1], n_side, L)
        seeds = s["random_seeds"]
        c_1 = np.random.default_rng(int(seeds.get("primordial amplitudes", 0))).normal(0, 1 / np.sqrt(2), (n_side,) * 3)
        c_2 = np.random.default_rng(int(seeds.get("primordial phases", 1))).normal(0, 1 / np.sqrt(2), (n_side,) * 3)
        pos = grf.zeldovich_particles(P, c_1, c_2, L, "skip_self")
        vel = np.zeros_like(pos)
        n = pos.shape[0]
        rho_m = (float(p.get("Ωb", 0.049)) + float(p.get("Ωcdm", 0.264))) * 3 * float(p["H0"]) ** 2 / (8 * np.pi * _params.G_NEWTON) \
            if p.get("H0") else 1.0
        mass = rho_m * L ** 3 / n
    eps = args.eps if args.eps is not None else 0.03 * L / round(n ** (1.0 / 3.0))
    dtype = np.float32 if args.dtype == "f32" else np.float64
    state = State(pos.astype(dtype), vel.astype(dtype), np.full(n, mass, dtype))
    if args.steps > 0:
        with Engine(n, dtype) as e:
            if args.solver == "tree":
                e.set_solver("tree", args.theta)
            e.set_state(state)
            e.step(args.dt, args.steps, eps, _params.G_NEWTON)
            state = e.get_state()
    out = Path(s["snapshot_dir"])
    out.mkdir(parents=True, exist_ok=True)
    labels = s["snapshot_times"] or [1.0]
    for a in labels:
        snapshot.write_snapshot(str(out / f"snapshot_a={a:.2f}.hdf5"), np.mod(state.pos.astype(np.float64), L),
                                state.vel.astype(np.float64) * mass, L, mass=mass,
                                extra_attrs={"a": float(a), "H0": float(p["H0"]) if p.get("H0") else 0.0})
    if s["powerspec_dir"] or s["powerspec_times"]:
        pdir = Path(s["powerspec_dir"] or out.parent / "powerspec")
        pdir.mkdir(parents=True, exist_ok=True)
        ngrid = max(2, round(n ** (1.0 / 3.0)))
        k, pk, cnt = field.power_spectrum(np.mod(state.pos.astype(np.float64), L), L, ngrid, deconvolve=True)
        for a in (s["powerspec_times"] or labels):
            _write_powerspec(pdir / f"powerspec_a={a:.2f}", k, pk, cnt, a, ngrid)
    print(f"concept: wrote {len(labels)} snapshot(s) to {out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

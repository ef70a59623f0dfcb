"""Cosmological (comoving, periodic) direct-summation runs — the physics of the reference's one `pp` run
(SURVEY.md §0 item 2, §8f-4): 64^3 particles in a 500 Mpc periodic box from a = 0.02 to 1
(`PPPMP3M/pppmp3mplotter.py:11-22`), parameters as in `Different_Sizes/Giant.param:75-85`.

Host side (NumPy): flat LCDM background, growth factor, kick/drift integrals, Zel'dovich initial conditions from a
tabulated linear P(k) (CLASS is not available: the spectrum is an input — e.g. the `P linear` column of the reference's own
`powerspec_a=1.00`). Device side: periodic Ewald forces and the generalised KDK step of the C-ABI
(`nb_set_periodic`, `nb_kdk_step`), GRF synthesis (`grf.py`), P(k) estimator (`field.py`).

Equations (comoving positions x, v := a^2 dx/dt = canonical momentum per unit mass, as in CONCEPT / GADGET):
    dx/dt = v / a^2,   dv/dt = g(x) / a,   g = G sum_j m_j d_ij / |d_ij|^3  (periodic, mean density subtracted)
    KDK over [a0, a1]:  v += g K(a0, am);  x += v D(a0, a1);  g = g(x);  v += g K(am, a1)
    K(a, b) = int_a^b da' / (a'^2 H(a')),   D(a, b) = int_a^b da' / (a'^3 H(a')),   am = sqrt(a0 a1)
Units: Mpc, Gyr, 1e10 M_sun (`params.py`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

from . import params as _params


@dataclass
class Cosmology:
    H0: float            # 1/Gyr
    Om: float            # matter (baryons + CDM)
    Or: float = 0.0      # radiation (photons + massless neutrinos)

    @classmethod
    def from_h_omegas(cls, h: float, omega_b: float, omega_cdm: float, radiation: bool = True) -> "Cosmology":
        """h, omega_b = Omega_b h^2, omega_cdm as in Giant.param:75-80 (`H0 = 67.36*km/(s*Mpc)`, `Ωb = 0.02237/h**2`)."""
        H0 = 100.0 * h * _params.UNITS["km"] / _params.UNITS["s"]
        Orr = (2.473e-5 * (1.0 + 0.2271 * 3.046)) / h ** 2 if radiation else 0.0
        return cls(H0, (omega_b + omega_cdm) / h ** 2, Orr)

    def E(self, a):
        a = np.asarray(a, np.float64)
        return np.sqrt(self.Or / a ** 4 + self.Om / a ** 3 + (1.0 - self.Om - self.Or))

    def H(self, a):
        return self.H0 * self.E(a)

    def kick_factor(self, a0: float, a1: float, n: int = 65) -> float:
        a = np.linspace(a0, a1, n)
        return float(np.trapezoid(1.0 / (a ** 2 * self.H(a)), a))

    def drift_factor(self, a0: float, a1: float, n: int = 65) -> float:
        a = np.linspace(a0, a1, n)
        return float(np.trapezoid(1.0 / (a ** 3 * self.H(a)), a))

    def cosmic_time(self, a: float) -> float:
        """t(a) = int_0^a da' / (a' H(a')) [Gyr] — CONCEPT's clock: its logs print `Cosmic time` and `Step size` per base step
        (…/params_id0200_*.log:49-52) and its power-spectrum tables the time of the dump (`t = 13.814035 Gyr, a = 1`,
        PPPMP3M/PP/powerspec/powerspec_a=1.00:1), both reproduced by this integral (tests/test_step_schedule.py)."""
        from scipy.integrate import quad
        return float(quad(lambda x: 1.0 / (x * float(self.H(x))), 0.0, float(a), epsabs=0, epsrel=1e-11)[0])

    def scale_factor(self, t: float) -> float:
        """a(t), inverse of cosmic_time (bracketing root find; flat LCDM with radiation)."""
        from scipy.optimize import brentq
        return float(brentq(lambda a: self.cosmic_time(a) - float(t), 1e-8, 4.0, xtol=1e-15, rtol=1e-13))

    def _clock_table(self):
        """(ln a, t) on 40 000 points from a = 1e-5 to 3: cumulative Simpson-like integral of dt = dln a / H, anchored by one exact quadrature.
        Interpolating it gives t(a) and a(t) to ~1e-9 relative — what the schedule code calls hundreds of times per run."""
        tab = self.__dict__.get("_clock")
        if tab is None:
            la = np.linspace(np.log(1e-5), np.log(3.0), 40001)
            f = 1.0 / self.H(np.exp(la))
            mid = 1.0 / self.H(np.exp(0.5 * (la[:-1] + la[1:])))
            t = self.cosmic_time(1e-5) + np.concatenate([[0.0], np.cumsum((f[:-1] + 4.0 * mid + f[1:]) * (la[1] - la[0]) / 6.0)])
            tab = self.__dict__["_clock"] = (la, t)
        return tab

    def scale_factor_fast(self, t):
        la, tt = self._clock_table()
        return np.exp(np.interp(t, tt, la))

    def cosmic_time_fast(self, a):
        la, tt = self._clock_table()
        return np.interp(np.log(a), la, tt)

    def growth(self, a: float):
        """(D(a) with D(1) = 1, f = dlnD/dlna) of sub-horizon matter perturbations, from the linear growth ODE
            D'' + (2 + dlnH/dlna) D' - (3/2) Omega_m(a) D = 0      (' = d/dlna)
        started on the growing mode D = a + (2/3) a_eq deep in the radiation era. (The closed integral E(a) int da/(aE)^3 is
        only a solution without radiation; with Omega_r in E(a) it under-estimates D(0.02) by 3 %, i.e. 6 % in power.)"""
        from scipy.integrate import solve_ivp
        Om, Or = self.Om, self.Or
        OL = 1.0 - Om - Or

        def rhs(x, y):
            aa = np.exp(x)
            E2 = Or / aa ** 4 + Om / aa ** 3 + OL
            dlnH = 0.5 * (-4 * Or / aa ** 4 - 3 * Om / aa ** 3) / E2
            return [y[1], -(2.0 + dlnH) * y[1] + 1.5 * (Om / aa ** 3 / E2) * y[0]]

        a_eq = Or / Om if Or > 0 else 0.0
        a0 = 1e-5 if Or > 0 else 1e-4
        y0 = [a0 + 2.0 / 3.0 * a_eq, a0]
        sol = solve_ivp(rhs, [np.log(a0), 0.0], y0, rtol=1e-10, atol=1e-14, dense_output=True)
        d1 = sol.sol(0.0)[0]
        d, dp = sol.sol(np.log(a))
        return float(d / d1), float(dp / d)

    def particle_mass(self, boxsize: float, n: int, G: float = _params.G_NEWTON) -> float:
        rho_m = self.Om * 3.0 * self.H0 ** 2 / (8.0 * np.pi * G)
        return rho_m * boxsize ** 3 / n


def second_order_potential_gradient(psi, boxsize: float) -> np.ndarray:
    """grad(phi2) on the lattice from the first-order displacement psi = -grad(phi1) (array (3, n, n, n)), where
        laplace(phi2) = sum_{i<j} [ phi1,ii phi1,jj - (phi1,ij)^2 ]
    — the source of second-order Lagrangian perturbation theory (2LPT; Bouchet et al. 1995, Scoccimarro 1998), CONCEPT's
    `realization_options = {'LPT': 2}` (Different_Sizes/Giant.param:6,70-72 carries the switch, 1LPTvs2LPT/1vs2tlptplotter.py:121-123
    compares the two). All derivatives spectral: phi1,ij = -d_j psi_i; the 13 transforms run through hipFFT (`nbf_lpt2_gradient`);
    oracle: oracle/grf_oracle.py::second_order_potential_gradient."""
    from . import field
    psi = np.asarray(psi, np.float64)
    with field.FieldPlan(psi.shape[1], boxsize) as plan:
        return plan.lpt2_gradient(psi)


def zeldovich_ics(pk_table, n_side: int, boxsize: float, cosmo: Cosmology, a_begin: float, seed_amp: int = 0,
                  seed_phase: int = 1, lpt: int = 1, return_noise_power: bool = False):
    """Lattice + growing-mode displacement at a_begin from the LINEAR spectrum at a = 1 (two columns k, P):
    returns comoving positions (N,3) in [0, L) and v = a^2 dx/dt (N,3). lpt = 2 adds the second-order displacement
    D2 grad(phi2) with D2 = -(3/7) D1^2 Omega_m(a)^(-1/143), f2 = 2 Omega_m(a)^(6/11)."""
    from . import grf
    tab = np.asarray(pk_table, np.float64)
    D, f = cosmo.growth(a_begin)
    # grf's amplitude carries the thesis' `fac` = sqrt(2 pi / V) N^3; the standard convention is sqrt(P / V) N^3
    P = grf.tabulated_pk_grid(tab[:, 0], tab[:, 1], n_side, boxsize) * D * D / (2.0 * np.pi)
    c_1 = np.random.default_rng(int(seed_amp)).normal(0, 1 / np.sqrt(2), (n_side,) * 3)
    c_2 = np.random.default_rng(int(seed_phase)).normal(0, 1 / np.sqrt(2), (n_side,) * 3)
    with grf.GRFPlan(n_side, boxsize) as plan:
        plan.set_noise(c_1, c_2, "skip_self")
        plan.set_spectrum(P)
        psi_grid = np.stack([plan.field(c + 1) for c in range(3)])
        g2 = plan.lpt2_gradient(psi_grid).reshape(3, -1).T if lpt == 2 else None       # same plan: no second hipFFT plan for a 32^3 job
        psi = psi_grid.reshape(3, -1).T
        noise_power = np.abs(plan.noise()) ** 2 if return_noise_power else None    # |R_k|^2: the realisation's power / expectation
    g = np.linspace(0, boxsize, n_side, endpoint=False)
    q = np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T
    aH = a_begin ** 2 * float(cosmo.H(a_begin))
    vel = (aH * f) * psi                                                # a^2 dx/dt with x = q + D(a) psi_1
    if lpt == 2:
        om_a = cosmo.Om / a_begin ** 3 / float(cosmo.E(a_begin)) ** 2
        psi2 = (-3.0 / 7.0) * om_a ** (-1.0 / 143.0) * g2              # psi already carries D1, so grad(phi2) carries D1^2
        vel = vel + (aH * 2.0 * om_a ** (6.0 / 11.0)) * psi2
        psi = psi + psi2
    elif lpt != 1:
        raise ValueError("lpt must be 1 or 2")
    pos = np.mod(q + psi, boxsize)
    if return_noise_power:
        return pos, vel, noise_power
    return pos, vel


def bins_per_decade_edges(table, k_min: float, k_max: float) -> np.ndarray:
    """Bin edges for `powerspec_options['bins per decade']` (SnapVsPK/Uncertain.param:50-55: {'4*k_min': 4, '100*k_min': 40}, the default
    CONCEPT's documentation gives): the local bin density d(k) is interpolated linearly in log k between the given (k, bins per decade)
    points and held constant outside them; bin CENTRES start at k_min and follow c' = c * 10^(1 / d(c)); edges sit at the geometric means
    of neighbouring centres (the first one half a bin below k_min), the last edge is k_max, a remainder narrower than half a bin joins
    the bin before it. CONCEPT's own routine is not in the repository: for the reference's 64^3 / 500 Mpc tables this rule gives the
    same 16 bins, the first three mode counts (3, 25, 124) exactly and the others to <= 10 % (PPPMP3M/PP/powerspec/powerspec_a=1.00:11-26;
    tests/test_params_cli.py)."""
    pts = sorted((float(k), float(b)) for k, b in table)
    lk, nb = np.log([k for k, _ in pts]), [b for _, b in pts]
    dens = lambda k: float(np.interp(np.log(k), lk, nb))
    c = [float(k_min)]
    while c[-1] < k_max:
        c.append(c[-1] * 10.0 ** (1.0 / dens(c[-1])))
    e = [c[0] / np.sqrt(c[1] / c[0])] + [np.sqrt(c[i] * c[i + 1]) for i in range(len(c) - 1)]
    e = [x for x in e if x < k_max]
    if len(e) >= 2 and (k_max / e[-1]) < np.sqrt(e[-1] / e[-2]):      # a remainder narrower than half the last bin
        e.pop()
    return np.array(e + [float(k_max)])


def concept_power_table(pos, boxsize: float, ngrid: int, pk_table=None, growth: float = 1.0, noise_power=None, nbins: Optional[int] = None,
                        bins_per_decade=None, k_max: Optional[float] = None, interlace: bool = True, deconvolve: bool = True):
    """The columns of a CONCEPT power-spectrum file (PPPMP3M/PP/powerspec/powerspec_a=1.00:10-26; read back by
    pppmp3mplotter.py:123-134 and Power_compare.py:625-629 with usecols (0, 2, 3, 4)):
        k (mean |k| of the bin), modes (independent half-space modes), P, P corrected, P linear
    P from the CIC-deposited, deconvolved, interlaced field (`power_modes`, the settings of NN.param:28-45) in linear |k| bins of
    width 2 k_fundamental up to the Nyquist frequency — or, with `bins_per_decade` (the parsed powerspec_options['bins per decade']), in
    the logarithmic bins of `bins_per_decade_edges` up to `k_max`. `P linear` = growth^2 x the tabulated linear spectrum averaged over the bin's
    modes; `P corrected` = P x sum(P_lin) / sum(P_lin |R_k|^2) — CONCEPT's 'realization correction', which removes the cosmic
    variance of the initial noise R from the estimate (nan columns when no table / noise is given)."""
    k, p = power_modes(pos, boxsize, ngrid, interlace=interlace, deconvolve=deconvolve)
    kf = 2 * np.pi / boxsize
    kny = np.pi * ngrid / boxsize if k_max is None else float(k_max)
    if bins_per_decade is not None:          # powerspec_options['bins per decade'] of the parameter file
        edges = bins_per_decade_edges(bins_per_decade, kf, kny)
        edges[-1] *= 1.0 + 1e-12             # the modes AT k_max belong to the last bin
        nb = len(edges) - 1
    else:
        nb = int(nbins) if nbins else max(1, int(np.ceil((kny - 0.5 * kf) / (2 * kf))))
        edges = np.linspace(0.5 * kf, kny, nb + 1)
    which = np.digitize(k, edges) - 1
    ok = (which >= 0) & (which < nb)
    cnt = np.bincount(which[ok], minlength=nb)
    keep = cnt > 0
    mean = lambda w: (np.bincount(which[ok], weights=w[ok], minlength=nb)[keep] / cnt[keep])
    out = [mean(k), cnt[keep] // 2, mean(p)]
    if pk_table is not None:
        tab = np.asarray(pk_table, np.float64)
        plin = growth ** 2 * np.exp(np.interp(np.log(k), np.log(tab[:, 0]), np.log(tab[:, 1])))
        lin = mean(plin)
        if noise_power is not None and np.asarray(noise_power).size == ngrid ** 3:
            kv = 2 * np.pi * np.fft.fftfreq(ngrid, d=boxsize / ngrid)
            kx, ky, kz = np.meshgrid(kv, kv, kv, indexing="ij")
            r2 = np.asarray(noise_power, np.float64)[np.sqrt(kx ** 2 + ky ** 2 + kz ** 2) > 0]       # same mode order as power_modes
            out += [out[2] * lin / mean(plin * r2), lin]
        else:
            out += [np.full_like(lin, np.nan), lin]
    else:
        out += [np.full(keep.sum(), np.nan), np.full(keep.sum(), np.nan)]
    return tuple(out)


def load_concept_schedule(path=None):
    """CONCEPT's base time steps as its logs record them: the table extracted from the reference's 300 run
    logs by tests/golden/concept/make_step_schedule.py — identical in all 300 (the base step is set by the background: constant
    in cosmic time for 8 steps, then updated; 143 steps from a = 0.02 to 1 for the 32^3 `p3m` farm jobs of NN.param).
    Returns (a_start[n], t_start[n] in Gyr, dt[n] in Gyr)."""
    from pathlib import Path
    if path is None:
        path = Path(__file__).resolve().parent / "data" / "concept_step_schedule.txt"
    tab = np.loadtxt(path)
    return tab[:, 1].copy(), tab[:, 2].copy(), tab[:, 3].copy()


def concept_time_edges(cosmo: Cosmology, a_begin: float, a_end: float, dt_steps) -> np.ndarray:
    """Step boundaries in COSMIC TIME for a run from a_begin to a_end that takes CONCEPT's logged step sizes `dt_steps` [Gyr] in
    order (the last step is cut at t(a_end), as CONCEPT cuts its last step at an output time; steps beyond a_end are dropped; if the
    sizes run out before a_end the last one is repeated)."""
    t0, t1 = float(cosmo.cosmic_time_fast(a_begin)), float(cosmo.cosmic_time_fast(a_end))
    edges, k = [t0], 0
    dts = [float(v) for v in dt_steps]
    while edges[-1] < t1 * (1.0 - 1e-12):
        dt = dts[min(k, len(dts) - 1)]
        nxt = edges[-1] + dt
        edges.append(t1 if nxt >= t1 * (1.0 - 1e-9) else nxt)
        k += 1
    return np.array(edges)


def run_pp(pos, vel, mass: float, boxsize: float, cosmo: Cosmology, a_begin: float, a_end: float = 1.0, nsteps: int = 400,
           eps: Optional[float] = None, alpha_L: Optional[float] = None, kmax: Optional[int] = None,
           callback: Optional[Callable[[float, "object"], None]] = None, device: int = 0, dtype=np.float64,
           gravity: str = "pp", ngrid: Optional[int] = None, interlace: bool = True, scale_cells: float = 1.25,
           range_scales: float = 4.5, schedule=None):
    """Comoving periodic run on the GPU with `nsteps` steps uniform in ln a. Returns (pos, vel) at a_end.
    gravity = 'pp' (Ewald summation; dtype float32 selects the fp32 kernels, forces good to ~1e-5. Default splitting
    alpha_L = 14.9, kmax = 17 (fp32; 16.5, 19 in fp64): the real-space terms have died out within a quarter of the box and are summed from cell lists;
    alpha_L = 6.2, kmax = 7 selects the all-pairs minimum-image real-space sum, `nb_set_periodic`),
    'pm' or 'p3m' (mesh of ngrid^3 cells, default cbrt(N) / 2 cbrt(N) as in CONCEPT) — the three solvers of
    PPPMP3M/pppmp3mplotter.py:14.
    schedule: None = `nsteps` steps uniform in ln a, kicks split at the geometric mean of the step's scale factors;
    "concept" = CONCEPT's own base steps (`load_concept_schedule`: the step sizes its logs record, constant in cosmic time over
    8 steps), or an explicit array of step sizes dt [Gyr] — then the kick halves meet at the step's mid-TIME, as in a leapfrog
    advanced in t. `nsteps` is ignored when a schedule is given."""
    from .api import Engine
    from .state import State
    n = pos.shape[0]
    if eps is None:
        eps = 0.03 * boxsize / round(n ** (1.0 / 3.0))
    if schedule is None:
        edges = np.exp(np.linspace(np.log(a_begin), np.log(a_end), nsteps + 1))
        mids = np.sqrt(edges[:-1] * edges[1:])
    else:
        if isinstance(schedule, str) and schedule == "concept":
            a_fix, _, dts = load_concept_schedule()
            # a run (or a leg of one, after an output time) that starts inside CONCEPT's k-th base step continues with that step's size
            dts = dts[max(0, int(np.searchsorted(a_fix, a_begin * (1.0 + 1e-4), side="right")) - 1):]
        else:
            dts = np.asarray(schedule, np.float64)
        t_edges = concept_time_edges(cosmo, a_begin, a_end, dts)
        edges = np.asarray(cosmo.scale_factor_fast(t_edges), np.float64).copy()
        edges[0], edges[-1] = a_begin, a_end
        mids = np.asarray(cosmo.scale_factor_fast(0.5 * (t_edges[:-1] + t_edges[1:])), np.float64)
    if gravity not in ("pp", "pm", "p3m"):
        raise ValueError("gravity must be 'pp', 'pm' or 'p3m'")
    with Engine(n, dtype, device) as e:
        if gravity == "pp":
            # the balance between the cell-list real-space sum and the reciprocal sum (tools/ewald_alpha_sweep.py, 64^3): fp32 — tile pair
            # kernel — 14.9 / 17: 6.0 ms per evaluation (16.5 / 19: 6.8); fp64 — exact erf/exp pair factor — 16.5 / 19
            f32 = np.dtype(dtype) == np.float32
            e.set_periodic(boxsize, alpha_L if alpha_L else (14.9 if f32 else 16.5), kmax if kmax else (17 if f32 else 19))
        else:
            cb = round(n ** (1.0 / 3.0))
            e.set_mesh(gravity, boxsize, int(ngrid) if ngrid else (cb if gravity == "pm" else 2 * cb), scale_cells, range_scales,
                       interlace)
        e.set_state(State(np.asarray(pos, dtype), np.asarray(vel, dtype), np.full(n, mass, dtype)))
        if callback is None:                       # the whole run as one call: no host round trip between the steps
            e.kdk_run([cosmo.kick_factor(a0, am) for a0, am in zip(edges[:-1], mids)],
                      [cosmo.drift_factor(a0, a1) for a0, a1 in zip(edges[:-1], edges[1:])],
                      [cosmo.kick_factor(am, a1) for am, a1 in zip(mids, edges[1:])], eps, _params.G_NEWTON)
        else:
            for a0, a1, am in zip(edges[:-1], edges[1:], mids):
                e.kdk_step(cosmo.kick_factor(a0, am), cosmo.drift_factor(a0, a1), cosmo.kick_factor(am, a1), eps, _params.G_NEWTON)
                callback(float(a1), e)
        st = e.get_state()
    return st.pos.astype(np.float64), st.vel.astype(np.float64)


run_nbody = run_pp          # the name that fits all three solvers


def run_rungs(pos, vel, mass: float, boxsize: float, cosmo: Cosmology, a_begin: float, a_end: float = 1.0, schedule="concept",
              eta: float = 0.018, levels: int = 5, eps: Optional[float] = None, ngrid: Optional[int] = None, interlace: bool = True,
              scale_cells: float = 1.25, range_scales: float = 4.5, device: int = 0, dtype=np.float32, return_populations: bool = False):
    """Comoving P3M run with CONCEPT's time-loop structure: base steps from `schedule` ("concept" = the steps its logs
    record, or an array of step sizes in Gyr), the mesh force kicking every particle once per base step, the short-range force through a
    rung hierarchy — a particle on rung l takes short-range kick-drift-kick steps of dt_base / 2^l (`nb_rung_step`). Rung limiter: the usual
    dt_i = sqrt(2 eta eps_phys / |a_phys|) with physical softening a*eps and the peculiar acceleration g / a^2 of the SHORT-RANGE force alone (CONCEPT determines
    its rung populations from the short-range interaction: …/params_id0200_*.log:42-46), i.e. dt_i^2 = 2 eta eps a^3 / |g_sr| in the code's variables; `eta` = 0.018 is calibrated against the rung populations the reference's logs record: with it the fraction of particles above rung 0 follows the logged curve
    (0.8 % at a = 0.46, 2.5 % at 0.57, 5.2 % at 0.70, 6.1 % at 0.87) to a few per cent, and the mean populations of rungs 1 / 2 / 3 over steps 130-141 read 1 276 / 532 / 21 against
    the logs' 1 266 / 572 / 17 (tests/test_gpu_rungs.py). Returns (pos, vel[, populations per step])."""
    from .api import Engine
    from .state import State
    n = pos.shape[0]
    if eps is None:
        eps = 0.03 * boxsize / round(n ** (1.0 / 3.0))
    if isinstance(schedule, str) and schedule == "concept":
        a_fix, _, dts = load_concept_schedule()
        dts = dts[max(0, int(np.searchsorted(a_fix, a_begin * (1.0 + 1e-4), side="right")) - 1):]
    else:
        dts = np.asarray(schedule, np.float64)
    t_edges = concept_time_edges(cosmo, a_begin, a_end, dts)
    half = 2 * (1 << (levels - 1))
    pops = []
    with Engine(n, dtype, device) as e:
        cb = round(n ** (1.0 / 3.0))
        e.set_mesh("p3m", boxsize, int(ngrid) if ngrid else cb, scale_cells, range_scales, interlace)
        e.set_state(State(np.asarray(pos, dtype), np.asarray(vel, dtype), np.full(n, mass, dtype)))
        for t0, t1 in zip(t_edges[:-1], t_edges[1:]):
            ab = np.asarray(cosmo.scale_factor_fast(np.linspace(t0, t1, half + 1)), np.float64)
            kc = np.concatenate([[0.0], np.cumsum([cosmo.kick_factor(x, y, 9) for x, y in zip(ab[:-1], ab[1:])])])
            dc = np.concatenate([[0.0], np.cumsum([cosmo.drift_factor(x, y, 9) for x, y in zip(ab[:-1], ab[1:])])])
            a_mid = ab[half // 2]
            pops.append(e.rung_step(levels, kc, dc, t1 - t0, 2.0 * eta * eps * a_mid ** 3, eps, _params.G_NEWTON))
        st = e.get_state()
    out = (st.pos.astype(np.float64), st.vel.astype(np.float64))
    return out + (np.array(pops),) if return_populations else out


def power_modes(pos, boxsize: float, ngrid: int, interlace: bool = True, deconvolve: bool = True):
    """Per-mode (|k|, P) of the CIC-deposited, window-deconvolved overdensity, optionally INTERLACED (two grids half a
    cell apart combined in k-space, which cancels the odd aliases) — CONCEPT's estimator settings in NN.param:30-41
    ('interpolation': 'CIC', 'deconvolve': True, 'interlace': True). The thesis' own estimator (Power_compare.py) does not
    interlace; `field.power_spectrum` mirrors that one. Deposits, FFT and deconvolution on the GPU (`nbf_power_modes`: both lattices
    through one complex FFT); oracle: oracle/field_oracle.py::power_modes."""
    from . import field
    with field.FieldPlan(ngrid, boxsize) as plan:
        pk = plan.power_modes(pos, interlace)
    kv = 2 * np.pi * np.fft.fftfreq(ngrid, d=boxsize / ngrid)
    k2 = kv * kv
    kmag = np.sqrt(k2[:, None, None] + k2[None, :, None] + k2[None, None, :])
    if not deconvolve:                       # powerspec_options['deconvolve'] = False: put the CIC window (sinc^2 per axis, squared for power) back
        w = np.sinc(kv * (boxsize / ngrid) / (2 * np.pi)) ** 2
        pk = pk * (w[:, None, None] * w[None, :, None] * w[None, None, :]) ** 2
    sel = kmag > 0
    return kmag[sel], pk[sel]


def rebin_like_concept(k_modes, p_modes, half_mode_counts):
    """Mean k and mean P of consecutive |k|-ordered groups of modes whose sizes are twice `half_mode_counts` — the binning
    of a CONCEPT power-spectrum table, which lists independent (half-space) mode counts per bin
    (PPPMP3M/PP/powerspec/powerspec_a=1.00:10-26)."""
    order = np.argsort(k_modes, kind="stable")
    k, p = np.asarray(k_modes)[order], np.asarray(p_modes)[order]
    out_k, out_p, start = [], [], 0
    for c in half_mode_counts:
        stop = min(len(k), start + 2 * int(c))
        if stop <= start:
            break
        out_k.append(k[start:stop].mean()); out_p.append(p[start:stop].mean())
        start = stop
    return np.array(out_k), np.array(out_p)


def rebin_corrected_like_concept(k_modes, p_modes, plin_modes, noise_power_modes, half_mode_counts):
    """`P corrected` of a CONCEPT table in the table's own bins (consecutive |k|-ordered groups of 2 x `half_mode_counts` modes):
    P_bin x sum(P_lin) / sum(P_lin |R_k|^2) — the realisation correction of `concept_power_table`, which divides out the power the
    run's own primordial noise R put into the bin, so that two runs with different seeds compare without cosmic variance
    (column 4 of PPPMP3M/PP/powerspec/powerspec_a=1.00:11-26)."""
    order = np.argsort(k_modes, kind="stable")
    k, p = np.asarray(k_modes)[order], np.asarray(p_modes)[order]
    pl, r2 = np.asarray(plin_modes)[order], np.asarray(noise_power_modes)[order]
    out_k, out_p, out_c, start = [], [], [], 0
    for c in half_mode_counts:
        stop = min(len(k), start + 2 * int(c))
        if stop <= start:
            break
        sl = slice(start, stop)
        out_k.append(k[sl].mean()); out_p.append(p[sl].mean())
        out_c.append(p[sl].mean() * pl[sl].sum() / (pl[sl] * r2[sl]).sum())
        start = stop
    return np.array(out_k), np.array(out_p), np.array(out_c)


def sigma8_of_table(k, p, h: float, radius: float = 8.0) -> float:
    """The sigma_8 a CONCEPT power-spectrum table prints in its header (`σ₈ = 8.77e-01`, PPPMP3M/PP/powerspec/powerspec_a=1.00:9):
    rms top-hat (R = 8 Mpc/h) overdensity of the TABULATED spectrum only, i.e. over the k range the box resolves —
        sigma^2 = 1/(2 pi^2) int k^2 P(k) W(kR)^2 dk,   W(x) = 3 (sin x - x cos x) / x^3,
    trapezoidal in k over the table's bins plus the piece below the first bin with P held at its first value. Reproduces all nine
    header values of the reference's PP / PM / P3M tables to their three printed digits +- 1 (tests/test_step_schedule.py)."""
    k, p = np.asarray(k, np.float64), np.asarray(p, np.float64)
    x = k * (radius / h)
    w = 3.0 * (np.sin(x) - x * np.cos(x)) / x ** 3
    f = k ** 2 * p * w ** 2
    return float(np.sqrt((np.trapezoid(f, k) + p[0] * k[0] ** 3 / 3.0) / (2.0 * np.pi ** 2)))

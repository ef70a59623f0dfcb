/* nbody_amd.h — C-ABI of libnbody_amd.so: MI355X (gfx950) direct-summation N-body engine.
 *
 * What this boundary replaces in the reference (HusseinYoud/Bachelor-Thesis-N-Body):
 * the reference has NO in-process force/integrator interface. Its only
 * boundary to a direct-summation solver is process-level — a CONCEPT `pp` run spawned as
 *     subprocess.check_call([concept, "-p", param, "-n", "1", "--local"], ...)
 * at SnapVsPK/Ucertain.py:89-97 and TheFeedingPipeline/NeuNetværk_test.py:155-171, selected by
 * `select_forces = {'matter': {'gravity': 'pp'}}` (pattern: Different_Sizes/Giant.param:66-68),
 * with particle state exchanged as HDF5 `components/matter/{pos,mom}` (readers:
 * SnapVsPK/Power_compare.py:143-144, TheFeedingPipeline/FraP_kTilDelta.py:55-68).
 * The entry points below are the in-process surface that SURVEY.md §8(a) rows ⊕1-⊕3 define
 * for that path (compute_forces / step / energy) — what a ctypes binding on the reference's
 * Python side would call instead of spawning the external solver.
 *
 * Conventions
 *  - plain C, no exceptions cross the ABI; every function returns 0 on success, <0 on error
 *    (NB_ERR_*); nb_last_error(ctx) returns the message of the last failure.
 *  - dtype: NB_F32 or NB_F64 selects the element type of every `void*` array of a context.
 *  - arrays are structure-of-arrays: separate x, y, z, (m) pointers, `n` contiguous elements.
 *  - a_i = G * sum_j m_j (r_j - r_i) / (|r_j - r_i|^2 + eps^2)^{3/2}   (Plummer softening;
 *    pairs with |r_j - r_i|^2 + eps^2 == 0 contribute nothing), SURVEY.md §7.2.
 *  - step = kick-drift-kick leapfrog, one force evaluation per step, acceleration cached.
 *  - a context is NOT thread-safe (one context per host thread); nb_create binds it to one GPU, nb_create_multi spreads its particles
 *    over several GPUs of the node (sink decomposition, peer pulls of the source blocks).
 *  - "_dev" functions take DEVICE pointers (e.g. torch tensors' data_ptr()) and enqueue on the
 *    context's stream without synchronising; the others take HOST pointers and copy.
 */
#ifndef NBODY_AMD_H
#define NBODY_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nb_ctx nb_ctx;

enum { NB_F32 = 0, NB_F64 = 1 };
/* force solvers selectable per context (nb_set_solver) */
enum { NB_SOLVER_DIRECT = 0, NB_SOLVER_TREE = 1, NB_SOLVER_EWALD = 2, NB_SOLVER_PM = 3, NB_SOLVER_P3M = 4 };
/* flags of nb_set_mesh */
enum {
  NB_MESH_NO_INTERLACE = 1, /* one lattice instead of two half-cell-shifted ones */
  NB_MESH_LONG_ONLY = 2,    /* diagnostics: only the mesh part of a P3M force */
  NB_MESH_SHORT_ONLY = 4    /* diagnostics: only the cell-list part of a P3M force */
};
/* flags of nb_accel_dev */
enum {
  NB_ACCUMULATE = 1,   /* add to the existing ax/ay/az instead of overwriting (source chunks arriving hop by hop) */
  NB_UNIFORM_MASS = 2  /* the caller guarantees all n_src source masses equal ms[0]: enables the 11-op inner loop */
};
enum {
  NB_OK = 0,
  NB_ERR_ARG = -1,      /* bad argument (null pointer, n <= 0, unknown dtype ...) */
  NB_ERR_HIP = -2,      /* a HIP runtime call or kernel launch failed */
  NB_ERR_NODEVICE = -3, /* no gfx950 device visible / bad device index */
  NB_ERR_STATE = -4     /* call order violated (e.g. nb_step before nb_set_state) */
};

/* library / device discovery */
int nb_version(void);                       /* 10000*major + 100*minor + patch */
int nb_device_count(void);                  /* number of visible HIP devices, <0 on error */
const char* nb_last_error(const nb_ctx* ctx); /* ctx may be NULL: error of the last failed nb_create */

/* lifetime: `n` particles owned (integrated) by this context, on HIP device `device` */
int nb_create(nb_ctx** out, int64_t n, int dtype, int device);
/* the same kind of context spread over `ndev` GPUs of one node: device dev_ids[g] owns the g-th contiguous range of
 * sinks (the first n % ndev devices one particle more) and integrates only those; at every force evaluation each device pulls the
 * (x,y,z,m) block of every other owner over xGMI (peer copies on a copy stream, overlapped with the kernel on the block before), sums
 * chunk by chunk, no reduction of accelerations. One host thread drives all devices; nothing synchronises inside nb_step. Supported on
 * such a context: nb_set_state / nb_get_state / nb_accel / nb_get_acc / nb_step / nb_step_energy / nb_kdk_step / nb_kdk_run /
 * nb_energy / nb_synchronize / nb_destroy with GLOBAL host arrays of n elements — results equal the single-device context's up to the summation
 * order of the chunks. Everything else (tree / periodic / mesh solvers, nb_set_stream, the _dev surface) returns
 * NB_ERR_ARG. A device may be listed more than once (rehearsal of the P-device schedule on one GPU). */
int nb_create_multi(nb_ctx** out, int64_t n, int dtype, int ndev, const int* dev_ids);
/* sink ranges of a context: fills counts[g], offsets[g] for g < min(cap, ndev) (either may be NULL), returns ndev (1 for nb_create) */
int nb_multi_layout(const nb_ctx* ctx, int64_t* counts, int64_t* offsets, int cap);
int nb_destroy(nb_ctx* ctx);
/* run all subsequent work on the caller's hipStream_t (e.g. torch's current stream) instead of the private
 * stream created by nb_create; the handle is used as given, NULL being HIP's default (null) stream. The handle is BORROWED: the
 * context keeps enqueueing on and synchronising it, so it must stay alive until the next nb_set_stream or nb_destroy. On the NULL
 * stream nb_step cannot use hipGraph replay for small N (HIP's legacy stream cannot be captured): it then launches every step eagerly,
 * same results, more launch overhead. */
int nb_set_stream(nb_ctx* ctx, void* hip_stream);
int nb_synchronize(nb_ctx* ctx);
/* force solver used by nb_accel / nb_step of this context: NB_SOLVER_DIRECT (O(N^2), default) or NB_SOLVER_TREE =
 * Barnes-Hut with opening angle theta (BASELINE.json configs[4]; fp32 contexts only; theta = 0 degenerates to the
 * direct sum) */
int nb_set_solver(nb_ctx* ctx, int solver, double theta);
/* periodic cubic box of edge `boxsize`: nb_accel / nb_step / nb_kdk_step use Ewald summation (minimum-image real-space sum +
 * reciprocal sum over |n| <= kmax, splitting alpha = alpha_L / boxsize; uniform background subtracted) — the force law of the
 * CONCEPT `pp` run behind PPPMP3M/PP/powerspec/powerspec_a=1.00. fp64 contexts use erf/exp in fp64; fp32 contexts a 1.5e-7
 * erfc approximation with fp64-flushed fp32 sums (~5x faster). Positions are wrapped into [0, boxsize)
 * after every drift. boxsize <= 0 returns to the isolated direct sum. Suggested alpha_L = 6.2, kmax = 7 (error ~1e-5). With
 * alpha_L >= 9.9 (real-space terms below 7e-5 of the pair force within a third of the box) and n >= 4096 the real-space part is summed
 * from cell lists instead of over all minimum-image pairs: alpha_L = 16.5, kmax = 19 gives the same accuracy almost four times
 * faster (kmax <= 31; the reciprocal sum walks rows of k-vectors by complex rotation, one sincos per row and particle). */
int nb_set_periodic(nb_ctx* ctx, double boxsize, double alpha_L, int kmax);
/* periodic cubic box solved on a mesh — the solvers the reference's parameter files select with
 * `select_forces = {'matter': {'gravity': 'pm' | 'p3m'}}` (Different_Sizes/Giant.param:66-68, TheFeedingPipeline/NN.param:85-87):
 *   NB_SOLVER_PM : CIC deposit -> FFT -> -4 pi G / k^2, CIC deconvolved twice, i k differentiation -> CIC interpolation
 *   NB_SOLVER_P3M: the same mesh force damped by exp(-k^2 r_s^2), r_s = scale_cells * boxsize / ngrid, plus the
 *                  erfc-complement pair force inside range_scales * r_s from cell lists (CONCEPT's documented defaults:
 *                  scale_cells = 1.25, range_scales = 4.5; potential_options of NN.param:53-82: gridsize = cbrt(N),
 *                  'Fourier' differentiation, CIC, deconvolve (True, True), interlace (True, True))
 * Two interlaced lattices unless NB_MESH_NO_INTERLACE. Positions are wrapped into [0, boxsize) after every drift. */
int nb_set_mesh(nb_ctx* ctx, int solver, double boxsize, int ngrid, double scale_cells, double range_scales, int flags);
/* one generalised KDK step: v += a*kick1; x += v*drift; a = a(x); v += a*kick2 — with kick = integral dt/a and
 * drift = integral dt/a^2 this is the comoving leapfrog of a cosmological run (v = canonical momentum per unit mass) */
int nb_kdk_step(nb_ctx* ctx, double kick1, double drift, double kick2, double eps, double G);
/* nsteps generalised KDK steps with per-step factors (host arrays of nsteps doubles), enqueued back to back: one host
 * synchronisation for the whole run — a comoving run from a_begin to an output time is one call */
int nb_kdk_run(nb_ctx* ctx, int64_t nsteps, const double* kick1, const double* drift, const double* kick2, double eps, double G);
/* One BASE step of a comoving P3M run with a rung hierarchy for the short-range force — the structure of CONCEPT's time loop as its logs show it
 * (…/params_id0200_*.log:56-96): the mesh force kicks every particle over the base step (half at its start, half at its end); a
 * particle on rung l receives its short-range kicks in steps of dt_base / 2^l (kick-drift-kick per rung step); all particles drift with the finest
 * populated step and the short-range force is re-evaluated after every drift. Rungs are assigned at the start of the step from the short-range acceleration
 * g = a_sr (as CONCEPT determines its rung populations: …/params_id0200_*.log:42-46): the lowest l < levels with dt_base / 2^l <= sqrt(crit / |g|)  (crit = 2 eta eps a^3 for the usual sqrt(2 eta eps_phys / |a_phys|) limiter in
 * the comoving variables of nb_kdk_step). kc / dc: the cumulative kick integral int dt/a and drift integral int dt/a^2 from the start of the base step to each
 * of the 2 * 2^(levels-1) + 1 boundaries of its finest half-sub-steps (host arrays). populations[l] (may be NULL) returns the rung populations. P3M contexts only. */
int nb_rung_step(nb_ctx* ctx, int levels, const double* kc, const double* dc, double dt_base, double crit, double eps, double G,
                 int64_t* populations);
/* 1 if the last nb_kdk_run replayed a captured hipGraph of one step (n <= 300 000 particles and >= 8 steps: the launch-bound regime of the
 * thesis' 32^3 / 64^3 farm jobs — the step's kick/drift factors are read from a device table indexed by a device-side step counter, so every
 * step is the same launch sequence), 0 if it launched every step eagerly (large n, NB_NO_GRAPH set, the NULL stream, or a capture the
 * runtime refused). Results are bit-identical either way. */
int nb_last_run_used_graph(const nb_ctx* ctx);

/* ---- the pair-symmetric all-pairs kernels (direct solver, a particle set on itself: nb_accel / nb_step / nb_energy paths and nb_accel_dev
 * with identical sink and source arrays) evaluate every unordered pair once and give both particles their term (csrc/accel_sym_f32.hip,
 * accel_sym_f64.hip); results agree with the one-sided kernels to rounding (different summation order). Options of a context:
 *   NB_OPT_PAIR_SYMMETRY : 1 = use them where they apply (default; process-wide default from NB_F32_SYM / NB_F64_SYM), 0 = one-sided kernels only
 *   NB_OPT_SYM_MIN_N     : smallest n that takes them (default 18432 fp32 / 10240 fp64: below, the one-sided kernel is faster — profiles/r04_sym_crossover_to_one_sided_bench.txt)
 *   NB_OPT_SYM_SLAB_BYTES: scratch budget for the per-offset partial-sum slabs (default 4 GiB); larger problems run in several passes.
 *                          The budget covers the slabs only: the per-group sink sums (up to ~100 MB at N = 2^20), the fp64 totals of a
 *                          multi-pass launch and 25 % allocation headroom come on top (N = 2^20 fp32: ~3 GB per context in all)
 *   NB_OPT_SYM_SCRATCH_LIMIT: the most scratch (bytes, everything included) these kernels may ask the allocator for (default: no limit).
 *                          When a request is above the limit or the allocator refuses it, the launch halves its slab slots per pass until
 *                          it fits, and if even one slot does not fit the one-sided kernel (megabytes of scratch) runs instead —
 *                          an out-of-memory condition never fails a force evaluation that the one-sided kernel could do
 * value < 0 restores the default. No counterpart in the reference (CONCEPT's pp solver is not in the repository). */
#define NB_OPT_PAIR_SYMMETRY 1
#define NB_OPT_SYM_MIN_N 2
#define NB_OPT_SYM_SLAB_BYTES 3
#define NB_OPT_SYM_SCRATCH_LIMIT 4
int nb_set_option(nb_ctx* ctx, int option, int64_t value);
/* 1 if the last direct-summation force launch of this context ran the pair-symmetric kernel, 0 if the one-sided one. A multi-device context
 * (nb_create_multi) answers for the OWN-CHUNK launch of its first device in the last force evaluation — the later launches of an evaluation
 * say nothing about it (one-sided by construction in the one-sided schedule; two one-sided launches instead of a two-block one for very small
 * chunks in the pair-symmetric schedule); which SCHEDULE the context runs is NB_QUERY_MULTI_SCHEDULE below. */
int nb_last_accel_symmetric(const nb_ctx* ctx);
/* Read back an option as set (-1 = default in force), or one of the read-only queries:
 *   NB_QUERY_MULTI_SCHEDULE      : 1 = a multi-device context that runs the pair-symmetric schedule (each pair of chunks worked on by one of its
 *                                  owners, partial sums pushed to the other), 0 = the one-sided schedule or a single-device context
 *   NB_QUERY_OWN_CHUNK_SYMMETRIC : what nb_last_accel_symmetric returns */
#define NB_QUERY_MULTI_SCHEDULE 100
#define NB_QUERY_OWN_CHUNK_SYMMETRIC 101
int nb_get_option(const nb_ctx* ctx, int option, int64_t* value);

/* ---- measurement aid ("reported absolute and as a fraction of the vector-FP32 roofline") ----
 * Sustained shader clock of `device` over the next `window_ms` milliseconds at most (clamped to 5 s), measured by one sleeping wave on a private
 * stream that reads the shader-cycle counter and the constant-rate counter at both ends of the window — start it, run the kernels of
 * interest, nb_clock_probe_stop on their stream, then nb_clock_probe_end (blocks until the window has ended; frees the handle) returns cycles per microsecond (= MHz) and the
 * window actually covered. bench.py reports it next to roofline.frac: boxes of this pool differ by several per cent in the clock they
 * hold under the all-pairs kernel, and cycles per interaction is the box-independent figure. No counterpart in the reference. */
typedef struct nb_clock_probe nb_clock_probe;
int nb_clock_probe_begin(int device, double window_ms, nb_clock_probe** probe);
/* enqueue the end of the window on `stream` (a hipStream_t; NULL = the default stream): the probe stops when the work enqueued on that stream
 * before this call has finished, or after window_ms, whichever comes first */
int nb_clock_probe_stop(nb_clock_probe* probe, void* stream);
int nb_clock_probe_end(nb_clock_probe* probe, double* mhz, double* window_ms);
/* the same, and the clock of every XCD separately: the probe runs one wave per XCD (each reports the XCC it ran on); `mhz` is their mean,
 * xcd_mhz[i] the clock of XCD i (0 where no wave landed). Returns the number of entries written (<= cap), < 0 on error. */
int nb_clock_probe_end_xcd(nb_clock_probe* probe, double* mhz, double* window_ms, double* xcd_mhz, int cap);

/* ---- host-pointer surface: the drop-in compute_forces()/step()/energy() (SURVEY §8a ⊕1-⊕3) ----
 * "Host-pointer" = the caller hands over HOST buffers, which are copied to the device; the arithmetic always runs on the GPU. The
 * `nb_accel_host_*` entry points sketched in SURVEY.md §8(b) / §7.1 step 2 (a CPU build of the same kernel behind this ABI) are
 * deliberately NOT part of this library: a CPU path inside the product would be exactly the silent fallback the parity claims rule out.
 * The CPU build of the kernel exists as test infrastructure only — oracle/nbody_oracle.c (nbo_accel_f64_omp / nbo_accel_f32_omp: OpenMP +
 * SIMD, same argument order) — and is what bench.py's `cpu_baseline` leg times. */
int nb_set_state(nb_ctx* ctx, const void* x, const void* y, const void* z,
                 const void* vx, const void* vy, const void* vz, const void* m);
int nb_get_state(nb_ctx* ctx, void* x, void* y, void* z, void* vx, void* vy, void* vz, void* m);
/* acceleration of the context's particles due to themselves; ax/ay/az are host outputs (may be NULL
 * to only refresh the cached device-side acceleration) */
int nb_accel(nb_ctx* ctx, double eps, double G, void* ax, void* ay, void* az);
/* copy out the cached acceleration (valid after nb_accel or nb_step; NB_ERR_STATE otherwise) */
int nb_get_acc(nb_ctx* ctx, void* ax, void* ay, void* az);
/* nsteps KDK steps of size dt, entirely on the device (no host round trip inside the loop) */
int nb_step(nb_ctx* ctx, double dt, int64_t nsteps, double eps, double G);
/* kinetic and softened potential energy, accumulated in fp64. The potential is the ISOLATED pair sum -G/2 sum_{i != j} m_i m_j /
 * sqrt(r^2+eps^2) whatever solver is selected: it is not the (background-subtracted) periodic potential of the Ewald / PM / P3M modes */
int nb_energy(nb_ctx* ctx, double eps, double G, double* kinetic, double* potential);
/* nsteps KDK steps like nb_step, plus the energies after every step: kinetic[k], potential[k] (host arrays of
 * nsteps doubles, either may be NULL) at the synchronised end of step k. The potential is accumulated inside
 * the force kernel (SURVEY §8a ⊕3: one extra accumulator per sink instead of a second O(N^2) pass), so an
 * energy-monitored run costs the same pair evaluations as a plain one. Direct solver only. */
int nb_step_energy(nb_ctx* ctx, double dt, int64_t nsteps, double eps, double G, double* kinetic, double* potential);

/* ---- device-pointer surface: building blocks for the sink-decomposed multi-GPU path ---- */
/* acc[i] (+)= G * sum_{j<n_src} m_j (r_j - r_i)/(r_ij^2+eps^2)^{3/2} for i < n_snk.
 * flags: NB_ACCUMULATE | NB_UNIFORM_MASS (see above). */
int nb_accel_dev(nb_ctx* ctx, int dtype,
                 const void* xi, const void* yi, const void* zi, int64_t n_snk,
                 const void* xs, const void* ys, const void* zs, const void* ms, int64_t n_src,
                 double eps, double G, void* ax, void* ay, void* az, int flags);
/* nb_accel_dev that also returns phi[i] (+)= sum_{j<n_src} m_j / sqrt(r_ij^2+eps^2) (device double[n_snk], NOT
 * multiplied by G). No pair is skipped: when sink i is among the sources its own term m_i/eps is included
 * (eps = 0: coincident pairs contribute nothing) and the caller removes it — see nb_step_energy. */
int nb_accel_pot_dev(nb_ctx* ctx, int dtype,
                     const void* xi, const void* yi, const void* zi, int64_t n_snk,
                     const void* xs, const void* ys, const void* zs, const void* ms, int64_t n_src,
                     double eps, double G, void* ax, void* ay, void* az, double* phi, int flags);
/* Two DIFFERENT device blocks, every A-B pair evaluated ONCE (the pair-symmetric kernels in their two-block form): block A's accelerations
 * aax.. (+)= the force of B on A (NB_ACCUMULATE adds, as in nb_accel_dev), block B's arrays bax.. are OVERWRITTEN with the force of A on B —
 * the partial a rank that was handed B's positions sends back to B's owner, which adds it to its own sums (sink decomposition with half the
 * pair work: SURVEY.md §8e). phi_a / phi_b (both or neither; fp64): the fused potential, per unit G — phi_a accumulates with
 * NB_ACCUMULATE, phi_b is overwritten. NB_UNIFORM_MASS: all masses of A and B are equal. Falls back to two one-sided launches where the
 * symmetric kernel does not apply (small blocks, NB_OPT_PAIR_SYMMETRY 0): the results are the same up to summation order. */
int nb_accel_pair_dev(nb_ctx* ctx, int dtype, const void* xa, const void* ya, const void* za, const void* ma, int64_t na,
                      const void* xb, const void* yb, const void* zb, const void* mb, int64_t nb, double eps, double G,
                      void* aax, void* aay, void* aaz, void* bax, void* bay, void* baz, double* phi_a, double* phi_b, int flags);
/* w_out[0] = sum_{i<n} m_i (phi_i - self_i) (device double), phi from nb_accel_pot_dev over ALL sources; self_i is the i = j term
 * exactly as the force kernel added it (0 when eps = 0). `uniform` = the NB_UNIFORM_MASS flag the phi was computed with. The
 * caller sums over ranks and applies -G/2. */
int nb_pot_energy_dev(nb_ctx* ctx, int dtype, const double* phi, const void* m, int64_t n, double eps, int uniform, double* w_out);
/* u_out[0] (+)= sum_i m_i sum_{j != i (by global index)} m_j / sqrt(r_ij^2+eps^2)   (device double;
 * the caller applies -G/2). snk_global0 / src_global0 are the global indices of element 0 of each set. */
int nb_potential_dev(nb_ctx* ctx, int dtype,
                     const void* xi, const void* yi, const void* zi, const void* mi, int64_t n_snk,
                     int64_t snk_global0,
                     const void* xs, const void* ys, const void* zs, const void* ms, int64_t n_src,
                     int64_t src_global0, double eps, double* u_out, int accumulate);
/* k_out[0] = sum_i m_i |v_i|^2 / 2 (device double) */
int nb_kinetic_dev(nb_ctx* ctx, int dtype, const void* vx, const void* vy, const void* vz,
                   const void* m, int64_t n, double* k_out);
/* v += a*h ; x += v*dt  (each product and sum rounded separately, like the NumPy oracle) */
int nb_kick_dev(nb_ctx* ctx, int dtype, void* vx, void* vy, void* vz,
                const void* ax, const void* ay, const void* az, int64_t n, double h);
int nb_drift_dev(nb_ctx* ctx, int dtype, void* x, void* y, void* z,
                 const void* vx, const void* vy, const void* vz, int64_t n, double dt);
/* Both in one pass over the state: v += a*kick ; x += v*drift (the opening half of a KDK step; same arithmetic as nb_kick_dev + nb_drift_dev). */
int nb_kick_drift_dev(nb_ctx* ctx, int dtype, void* x, void* y, void* z, void* vx, void* vy, void* vz,
                      const void* ax, const void* ay, const void* az, int64_t n, double kick, double drift);
/* nb_accel_dev followed by the leapfrog TAIL on the sinks:  v += a*kick_close  (the closing half kick of the step the force belongs to) and,
 * when kick_open or drift is non-zero,  v += a*kick_open ; xi += v*drift  (the opening half kick and drift of the NEXT step — the sink
 * positions move; where sinks and sources are the same arrays so do the sources). Bit-identical to nb_accel_dev + nb_kick_dev
 * [+ nb_kick_drift_dev]; where the force runs on the pair-symmetric kernels the tail is applied by the kernel that completes the
 * accelerations, so a whole step is two launches (force, reduce + tail) instead of five — at N = 32 768 ... 131 072 the small launches and the
 * gaps between them were 15 ... 1.5 % of a step (profiles/r04_step_timeline_small_n.txt). The reference's integrator is the same KDK
 * sequence as separate NumPy statements. */
int nb_accel_kdk_dev(nb_ctx* ctx, int dtype, void* xi, void* yi, void* zi, int64_t n_snk,
                     const void* xs, const void* ys, const void* zs, const void* ms, int64_t n_src,
                     double eps, double G, void* ax, void* ay, void* az, int flags,
                     void* vx, void* vy, void* vz, double kick_close, double kick_open, double drift);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_AMD_H */

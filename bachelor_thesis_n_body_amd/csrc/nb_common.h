// Shared declarations for the HIP side of libnbody_amd.so (gfx950 only).
// Internal header: the public C-ABI lives in include/nbody_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>

struct nb_bh_state;   // Barnes-Hut buffers (bh.hip)
struct nb_ewald_state;   // k-vector table and structure factors (ewald.hip)
struct nb_mesh_state;    // PM / P3M grids, FFT plan and cell lists (mesh.hip)
struct nb_multi_state;   // per-device shards, receive buffers and events of a multi-device context (multi.hip)

struct nb_ctx {
  int device = 0;
  int dtype = 0;              // NB_F32 / NB_F64
  int64_t n = 0;              // particles owned by this context
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // SoA state, one allocation: [x y z m | vx vy vz | ax ay az], each of length n (element = float or double)
  void* state = nullptr;
  bool has_state = false;
  bool uniform_mass = false;  // all masses given to nb_set_state were equal (checked on the host copy)
  bool acc_valid = false;     // ax,ay,az hold a(x) for the current positions and (eps, G) below
  double acc_eps = -1.0, acc_G = 0.0;
  // scratch: split-source partial sums and block-level reductions
  void* work = nullptr;
  size_t work_bytes = 0;
  double* scal = nullptr;     // device scalars (energy reductions)
  double* phi = nullptr;      // per-particle potential sum_j m_j/sqrt(r^2+eps^2) from the fused force+potential kernel (lazy)
  double* elog = nullptr;     // (K, W) pairs of nb_step_energy, one per step (lazy, elog_cap pairs)
  int64_t elog_cap = 0;
  // one KDK step [kick_drift(h,h,dt); a(x)] captured as a hipGraph and replayed for launch-bound (small-N) runs
  hipGraphExec_t step_graph = nullptr;
  double graph_dt = 0.0, graph_eps = -1.0, graph_G = 0.0;
  // rung hierarchy (nb_rung_step): long-range acceleration kept apart from the short-range one, per-particle rungs, population histogram, tables
  void* acc_lr = nullptr;
  unsigned char* rung = nullptr;
  int* rung_hist = nullptr;          // 16 device ints
  double* rung_tab = nullptr;        // kick + drift tables of one base step (device)
  size_t rung_tab_len = 0;
  const unsigned char* sr_rung = nullptr;   // when set: the pair kernel skips chunks without a particle on rung >= sr_min_rung (nb_rung_step)
  int sr_min_rung = 0;
  bool lr_valid = false;             // acc_lr / the state's acc hold a_lr / a_sr of the current positions, computed with (lr_eps, lr_G)
  double lr_eps = -1.0, lr_G = 0.0;
  // pair-symmetric self-interaction kernels (accel_sym_*.hip): per-context options (-1 = process default) and what the last launch took
  long long sym_enable = -1, sym_min_n = -1, sym_slab_bytes = -1;
  long long sym_scratch_limit = -1;   // NB_OPT_SYM_SCRATCH_LIMIT: the most scratch the pair-symmetric kernels may ask the allocator for (-1: no limit)
  bool last_accel_sym = false;
  // Integrator tail handed to the NEXT force launch (nb_accel_kdk_dev, nb_step): the kernel that writes the accelerations of a particle also
  // applies  v += a * kick_close;  and, when `advance`,  v += a * kick_open;  x += v * drift  (the closing half kick of the step the force
  // belongs to, the opening half kick and drift of the next) — one launch instead of up to three at launch-sensitive sizes. A force path
  // that can do it sets `done`; the caller runs the plain kick / kick-drift kernels otherwise. Products and sums rounded separately, as in
  // integrate.hip.
  struct Tail {
    void *vx = nullptr, *vy = nullptr, *vz = nullptr, *x = nullptr, *y = nullptr, *z = nullptr;
    double kick_close = 0.0, kick_open = 0.0, drift = 0.0;
    bool armed = false, advance = false, done = false;
  } tail;
  bool graph_replayed = false;   // the last nb_kdk_run replayed a captured step (diagnostic: nb_last_run_used_graph)
  // force solver used by nb_accel / nb_step: direct summation (default) or Barnes-Hut tree (fp32 only)
  int solver = 0;
  double theta = 0.5;
  nb_bh_state* bh = nullptr;
  nb_ewald_state* ewald = nullptr;
  nb_mesh_state* mesh = nullptr;
  nb_multi_state* multi = nullptr;   // non-null: this context drives several devices (nb_create_multi); it owns no device state itself
  int mesh_grid = 0, mesh_interlace = 1, mesh_parts = 3;   // NB_SOLVER_PM / NB_SOLVER_P3M (nb_set_mesh)
  double mesh_scale = 1.25, mesh_range = 4.5;              // r_s in cells, cut-off in r_s
  double box = 0.0, alphaL = 6.2;   // periodic cube edge and Ewald splitting (NB_SOLVER_EWALD)
  int kmax = 7;
  std::string err;
};

// element pointer helpers into the SoA state block
template <typename T> static inline T* st_ptr(nb_ctx* c, int field) {  // 0..9 = x y z m vx vy vz ax ay az
  return reinterpret_cast<T*>(c->state) + (size_t)field * (size_t)c->n;
}

int nb_fail(nb_ctx* c, int code, const std::string& msg);
int nb_ensure_work(nb_ctx* c, size_t bytes);
// like nb_ensure_work, but a request the allocator refuses (or one above `limit` >= 0) is not an error: returns NB_WORK_UNAVAILABLE with the
// HIP error cleared and the context's scratch released (the caller retries with a smaller request or takes another kernel)
constexpr int NB_WORK_UNAVAILABLE = 1;
int nb_try_ensure_work(nb_ctx* c, size_t bytes, long long limit);

#define NB_HIP(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return nb_fail((c), NB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// launchers (defined in the kernel translation units); all asynchronous on c->stream
int nb_launch_accel_f32(nb_ctx* c, const float* xi, const float* yi, const float* zi, int64_t n_snk,
                        const float* xs, const float* ys, const float* zs, const float* ms, int64_t n_src,
                        float eps2, float G, float* ax, float* ay, float* az, int flags, double* phi = nullptr);
// pair-symmetric self-interaction (sinks == sources, each unordered pair evaluated once); *taken = 0: not applicable, use the one-sided kernel
int nb_launch_accel_sym_f32(nb_ctx* c, const float* x, const float* y, const float* z, const float* m, int64_t n,
                            float eps2, float G, float* ax, float* ay, float* az, int flags, double* phi, int* taken);
int nb_launch_accel_sym_f64(nb_ctx* c, const double* x, const double* y, const double* z, const double* m, int64_t n,
                            double eps2, double G, double* ax, double* ay, double* az, int flags, double* phi, int* taken);
// two DIFFERENT blocks, every A-B pair evaluated once: A's accelerations (+)= the force of B, B's (bax..) = the force of A; *taken = 0: not applicable
int nb_launch_accel_cross_f32(nb_ctx* c, const float* xa, const float* ya, const float* za, const float* ma, int64_t na,
                              const float* xb, const float* yb, const float* zb, const float* mb, int64_t nb,
                              float eps2, float G, float* aax, float* aay, float* aaz, float* bax, float* bay, float* baz,
                              double* phi_a, double* phi_b, int flags, int* taken);
int nb_launch_accel_cross_f64(nb_ctx* c, const double* xa, const double* ya, const double* za, const double* ma, int64_t na,
                              const double* xb, const double* yb, const double* zb, const double* mb, int64_t nb,
                              double eps2, double G, double* aax, double* aay, double* aaz, double* bax, double* bay, double* baz,
                              double* phi_a, double* phi_b, int flags, int* taken);
// cross kernel where it applies, else the one-sided kernel in both directions (capi.hip)
int nb_launch_accel_pair(nb_ctx* c, int dtype, const void* xa, const void* ya, const void* za, const void* ma, int64_t na,
                         const void* xb, const void* yb, const void* zb, const void* mb, int64_t nb, double eps2, double G,
                         void* aax, void* aay, void* aaz, void* bax, void* bay, void* baz, double* phi_a, double* phi_b, int flags);
int nb_launch_accel_f64(nb_ctx* c, const double* xi, const double* yi, const double* zi, int64_t n_snk,
                        const double* xs, const double* ys, const double* zs, const double* ms, int64_t n_src,
                        double eps2, double G, double* ax, double* ay, double* az, int flags, double* phi = nullptr);
// Barnes-Hut acceleration of n particles on themselves (fp32); falls back to the direct sum for n < 128 or theta <= 0
int nb_launch_accel_tree_f32(nb_ctx* c, const float* x, const float* y, const float* z, const float* m, int64_t n,
                             float eps2, float G, float theta, float* ax, float* ay, float* az);
// periodic Ewald summation: acceleration of all n particles on themselves, uniform background subtracted
int nb_launch_accel_ewald(nb_ctx* c, int dtype, const void* x, const void* y, const void* z, const void* m, int64_t n,
                          double L, double alphaL, int kmax, double eps2, double G, void* ax, void* ay, void* az);
int nb_launch_wrap(nb_ctx* c, int dtype, void* x, void* y, void* z, int64_t n, double L);
void nb_ewald_release(nb_ctx* c);
int nb_launch_accel_mesh(nb_ctx* c, int dtype, const void* x, const void* y, const void* z, const void* m, int64_t n,
                         double L, int ngrid, double rs_cells, double rcut_rs, int p3m, int interlace, int parts,
                         double eps2, double G, void* ax, void* ay, void* az);
int nb_launch_short_range(nb_ctx* c, int dtype, const void* x, const void* y, const void* z, const void* m, int64_t n, double L,
                          double alpha, double rcut, double eps2, double G, void* ax, void* ay, void* az);   // ADDS to acc
void nb_mesh_release(nb_ctx* c);
void nb_bh_release(nb_ctx* c);
// multi-device context (multi.hip): the public entry points of capi.hip forward to these when c->multi is set
int nb_multi_create(nb_ctx* c, int ndev, const int* dev_ids);
void nb_multi_release(nb_ctx* c);
int nb_multi_set_state(nb_ctx* c, const void* x, const void* y, const void* z, const void* vx, const void* vy, const void* vz, const void* m);
int nb_multi_get(nb_ctx* c, void* const* dst, const int* fields, int nf);   // fields: 0..9 = x y z m vx vy vz ax ay az
int nb_multi_accel(nb_ctx* c, double eps, double G);
bool nb_multi_acc_valid(const nb_ctx* c, double eps, double G);
bool nb_multi_has_acc(const nb_ctx* c);
int nb_multi_kdk(nb_ctx* c, int64_t nsteps, const double* kick1, const double* drift, const double* kick2, double eps, double G);
int nb_multi_energy(nb_ctx* c, double eps, double G, double* kinetic, double* potential);
int nb_multi_step_energy(nb_ctx* c, double dt, int64_t nsteps, double eps, double G, double* kinetic, double* potential);
int nb_multi_sync(nb_ctx* c);
int nb_multi_set_option(nb_ctx* c, int option, int64_t value);
int nb_multi_last_symmetric(const nb_ctx* c);
int nb_multi_schedule_pairwise(const nb_ctx* c);
int nb_multi_info(const nb_ctx* c, int which, int64_t* out, int cap);
int nb_bh_check(nb_ctx* c);   // call after a stream sync
int nb_launch_potential_f32(nb_ctx* c, const float* xi, const float* yi, const float* zi, const float* mi,
                            int64_t n_snk, int64_t snk_global0, const float* xs, const float* ys, const float* zs,
                            const float* ms, int64_t n_src, int64_t src_global0, float eps2, double* u_out,
                            int accumulate);
int nb_launch_potential_f64(nb_ctx* c, const double* xi, const double* yi, const double* zi, const double* mi,
                            int64_t n_snk, int64_t snk_global0, const double* xs, const double* ys, const double* zs,
                            const double* ms, int64_t n_src, int64_t src_global0, double eps2, double* u_out,
                            int accumulate);
// out[0] = (accumulate ? out[0] : 0) + scale * sum(part[0..nparts)), one block, fixed order
int nb_launch_final_sum(nb_ctx* c, const double* part, int nparts, double scale, double* out, int accumulate);
// w_out[0] = sum_i m_i (phi_i - self_i), self_i = the i==j term the fused kernel included (see integrate.hip)
int nb_launch_pot_energy(nb_ctx* c, int dtype, const double* phi, const void* m, int64_t n, double eps2, int uniform,
                         double* w_out);
int nb_launch_kinetic(nb_ctx* c, int dtype, const void* vx, const void* vy, const void* vz, const void* m,
                      int64_t n, double* k_out);
int nb_launch_kick(nb_ctx* c, int dtype, void* vx, void* vy, void* vz, const void* ax, const void* ay,
                   const void* az, int64_t n, double h);
int nb_launch_drift(nb_ctx* c, int dtype, void* x, void* y, void* z, const void* vx, const void* vy,
                    const void* vz, int64_t n, double dt);
int nb_launch_rung_assign(nb_ctx* c, int dtype, const void* sr, const void* lr, int64_t n, double dt_base, double crit, int levels, int sr_only,
                          unsigned char* rung, int* hist);
int nb_launch_rung_kick(nb_ctx* c, int dtype, void* vx, void* vy, void* vz, const void* ax, const void* ay, const void* az,
                        const unsigned char* rung, int64_t n, int phase, int s, int S, int top, const double* kc);
// phase 0: v += a*fac[3s]; x += v*fac[3s+1]   phase 1: v += a*fac[3s+2]; ++s      with s = *step read on the device (graph-replayable KDK)
int nb_launch_kdk_indexed(nb_ctx* c, int dtype, int phase, void* x, void* y, void* z, void* vx, void* vy, void* vz,
                          const void* ax, const void* ay, const void* az, int64_t n, const double* fac, int* step);
int nb_launch_kick_drift(nb_ctx* c, int dtype, void* x, void* y, void* z, void* vx, void* vy, void* vz,
                         const void* ax, const void* ay, const void* az, int64_t n, double h1, double h2, double dt);

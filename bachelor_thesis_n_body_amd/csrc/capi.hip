// C-ABI of libnbody_amd.so (declared in include/nbody_amd.h): context lifetime, host<->device state
// transfer, and the compute_forces()/step()/energy() drivers that string the HIP kernels together on one
// stream with no host round trip inside nb_step().
#include "nb_common.h"
#include "../../include/nbody_amd.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

static thread_local std::string g_create_err;

int nb_fail(nb_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_create_err = msg;
  return code;
}

int nb_ensure_work(nb_ctx* c, size_t bytes) {
  if (bytes <= c->work_bytes) return NB_OK;
  // earlier kernels on the stream may still read the old buffer
  NB_HIP(c, hipStreamSynchronize(c->stream));
  // a captured step graph holds the old scratch pointer
  if (c->step_graph) { (void)hipGraphExecDestroy(c->step_graph); c->step_graph = nullptr; }
  if (c->work) { NB_HIP(c, hipFree(c->work)); c->work = nullptr; c->work_bytes = 0; }
  size_t want = bytes + bytes / 4;
  NB_HIP(c, hipMalloc(&c->work, want));
  c->work_bytes = want;
  return NB_OK;
}

int nb_try_ensure_work(nb_ctx* c, size_t bytes, long long limit) {
  if (bytes <= c->work_bytes) return NB_OK;
  if (limit >= 0 && bytes > (size_t)limit) return NB_WORK_UNAVAILABLE;
  NB_HIP(c, hipStreamSynchronize(c->stream));
  if (c->step_graph) { (void)hipGraphExecDestroy(c->step_graph); c->step_graph = nullptr; }
  if (c->work) { NB_HIP(c, hipFree(c->work)); c->work = nullptr; c->work_bytes = 0; }
  // 25 % headroom so that a slowly growing request does not reallocate every time; the bare request if that is refused
  for (size_t want : {bytes + bytes / 4, bytes}) {
    if (limit >= 0 && want > (size_t)limit) continue;
    if (hipMalloc(&c->work, want) == hipSuccess) { c->work_bytes = want; return NB_OK; }
    c->work = nullptr;
    (void)hipGetLastError();                  // out of memory is an answer here, not a sticky error
  }
  return NB_WORK_UNAVAILABLE;
}

static inline size_t esize(int dtype) { return dtype == NB_F64 ? 8 : 4; }

static int use_device(nb_ctx* c) {
  NB_HIP(c, hipSetDevice(c->device));
  return NB_OK;
}

// Two different blocks, every A-B pair once where the pair-symmetric kernels apply, otherwise the one-sided kernel in both directions.
// A's arrays (+)= (NB_ACCUMULATE), B's are overwritten. Used by nb_accel_pair_dev and by the multi-device context.
int nb_launch_accel_pair(nb_ctx* c, int dtype, const void* xa, const void* ya, const void* za, const void* ma, int64_t na,
                         const void* xb, const void* yb, const void* zb, const void* mb, int64_t nb, double eps2, double G,
                         void* aax, void* aay, void* aaz, void* bax, void* bay, void* baz, double* phi_a, double* phi_b, int flags) {
  const int fl_b = flags & ~NB_ACCUMULATE;
  int rc, taken = 0;
  c->last_accel_sym = false;
  if (dtype == NB_F32) {
    rc = nb_launch_accel_cross_f32(c, (const float*)xa, (const float*)ya, (const float*)za, (const float*)ma, na, (const float*)xb, (const float*)yb,
                                   (const float*)zb, (const float*)mb, nb, (float)eps2, (float)G, (float*)aax, (float*)aay, (float*)aaz,
                                   (float*)bax, (float*)bay, (float*)baz, phi_a, phi_b, flags, &taken);
    if (rc || taken) return rc;
    rc = nb_launch_accel_f32(c, (const float*)xa, (const float*)ya, (const float*)za, na, (const float*)xb, (const float*)yb, (const float*)zb,
                             (const float*)mb, nb, (float)eps2, (float)G, (float*)aax, (float*)aay, (float*)aaz, flags, phi_a);
    if (rc) return rc;
    return nb_launch_accel_f32(c, (const float*)xb, (const float*)yb, (const float*)zb, nb, (const float*)xa, (const float*)ya, (const float*)za,
                               (const float*)ma, na, (float)eps2, (float)G, (float*)bax, (float*)bay, (float*)baz, fl_b, phi_b);
  }
  rc = nb_launch_accel_cross_f64(c, (const double*)xa, (const double*)ya, (const double*)za, (const double*)ma, na, (const double*)xb, (const double*)yb,
                                 (const double*)zb, (const double*)mb, nb, eps2, G, (double*)aax, (double*)aay, (double*)aaz,
                                 (double*)bax, (double*)bay, (double*)baz, phi_a, phi_b, flags, &taken);
  if (rc || taken) return rc;
  rc = nb_launch_accel_f64(c, (const double*)xa, (const double*)ya, (const double*)za, na, (const double*)xb, (const double*)yb, (const double*)zb,
                           (const double*)mb, nb, eps2, G, (double*)aax, (double*)aay, (double*)aaz, flags, phi_a);
  if (rc) return rc;
  return nb_launch_accel_f64(c, (const double*)xb, (const double*)yb, (const double*)zb, nb, (const double*)xa, (const double*)ya, (const double*)za,
                             (const double*)ma, na, eps2, G, (double*)bax, (double*)bay, (double*)baz, fl_b, phi_b);
}

extern "C" {

int nb_version(void) { return 100; }

int nb_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { g_create_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return NB_ERR_HIP; }
  return n;
}

const char* nb_last_error(const nb_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int nb_create(nb_ctx** out, int64_t n, int dtype, int device) {
  if (!out) return nb_fail(nullptr, NB_ERR_ARG, "nb_create: out is NULL");
  *out = nullptr;
  if (n <= 0) return nb_fail(nullptr, NB_ERR_ARG, "nb_create: n must be > 0");
  if (n >= (1ll << 31)) return nb_fail(nullptr, NB_ERR_ARG, "nb_create: n must be < 2^31");
  if (dtype != NB_F32 && dtype != NB_F64) return nb_fail(nullptr, NB_ERR_ARG, "nb_create: dtype must be NB_F32 or NB_F64");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return nb_fail(nullptr, NB_ERR_NODEVICE, std::string("nb_create: no HIP device visible (") + hipGetErrorString(e) + ")");
  if (device < 0 || device >= ndev) return nb_fail(nullptr, NB_ERR_NODEVICE, "nb_create: device index out of range");
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return nb_fail(nullptr, NB_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return nb_fail(nullptr, NB_ERR_NODEVICE, std::string("nb_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  nb_ctx* c = new (std::nothrow) nb_ctx();
  if (!c) return nb_fail(nullptr, NB_ERR_ARG, "nb_create: out of host memory");
  c->device = device; c->dtype = dtype; c->n = n;
  auto bail = [&](const char* what, hipError_t err) {
    std::string m = std::string(what) + ": " + hipGetErrorString(err);
    if (c->state) (void)hipFree(c->state);
    if (c->scal) (void)hipFree(c->scal);
    if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return nb_fail(nullptr, NB_ERR_HIP, m);
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  c->own_stream = true;
  if ((e = hipMalloc(&c->state, (size_t)10 * (size_t)n * esize(dtype))) != hipSuccess) return bail("hipMalloc(state)", e);
  // zero on the context's own stream: a plain hipMemset runs on the NULL stream, which a non-blocking stream does
  // not order against, so it could land AFTER the first nb_set_state copies (seen once as garbage accelerations)
  if ((e = hipMemsetAsync(c->state, 0, (size_t)10 * (size_t)n * esize(dtype), c->stream)) != hipSuccess) return bail("hipMemsetAsync(state)", e);
  if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail("hipStreamSynchronize", e);
  if ((e = hipMalloc((void**)&c->scal, 8 * sizeof(double))) != hipSuccess) return bail("hipMalloc(scalars)", e);
  *out = c;
  return NB_OK;
}

int nb_create_multi(nb_ctx** out, int64_t n, int dtype, int ndev, const int* dev_ids) {
  if (!out) return nb_fail(nullptr, NB_ERR_ARG, "nb_create_multi: out is NULL");
  *out = nullptr;
  if (ndev < 1 || ndev > 64 || !dev_ids) return nb_fail(nullptr, NB_ERR_ARG, "nb_create_multi: ndev must be in [1, 64] and dev_ids non-NULL");
  if (n < ndev) return nb_fail(nullptr, NB_ERR_ARG, "nb_create_multi: fewer particles than devices");
  if (n >= (1ll << 31)) return nb_fail(nullptr, NB_ERR_ARG, "nb_create_multi: n must be < 2^31");
  if (dtype != NB_F32 && dtype != NB_F64) return nb_fail(nullptr, NB_ERR_ARG, "nb_create_multi: dtype must be NB_F32 or NB_F64");
  int visible = 0;
  hipError_t e = hipGetDeviceCount(&visible);
  if (e != hipSuccess || visible <= 0)
    return nb_fail(nullptr, NB_ERR_NODEVICE, std::string("nb_create_multi: no HIP device visible (") + hipGetErrorString(e) + ")");
  for (int g = 0; g < ndev; ++g)
    if (dev_ids[g] < 0 || dev_ids[g] >= visible) return nb_fail(nullptr, NB_ERR_NODEVICE, "nb_create_multi: device index out of range");
  nb_ctx* c = new (std::nothrow) nb_ctx();
  if (!c) return nb_fail(nullptr, NB_ERR_ARG, "nb_create_multi: out of host memory");
  c->device = dev_ids[0]; c->dtype = dtype; c->n = n;
  int rc = nb_multi_create(c, ndev, dev_ids);
  if (rc) { g_create_err = c->err; nb_multi_release(c); delete c; return rc; }
  *out = c;
  return NB_OK;
}

int nb_multi_layout(const nb_ctx* c, int64_t* counts, int64_t* offsets, int cap) {
  if (!c) return NB_ERR_ARG;
  if (!c->multi) { if (cap > 0) { if (counts) counts[0] = c->n; if (offsets) offsets[0] = 0; } return 1; }
  if (counts) nb_multi_info(c, 0, counts, cap);
  if (offsets) nb_multi_info(c, 1, offsets, cap);
  int64_t dummy[64];
  return nb_multi_info(c, 2, dummy, 64);
}

// entry points that only make sense on one device
#define NB_SINGLE_ONLY(c, name) do { if ((c)->multi) return nb_fail((c), NB_ERR_ARG, name ": not available on a multi-device context (direct solver, host-pointer surface only)"); } while (0)

int nb_destroy(nb_ctx* c) {
  if (!c) return NB_OK;
  if (c->multi) { nb_multi_release(c); delete c; return NB_OK; }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->step_graph) (void)hipGraphExecDestroy(c->step_graph);
  nb_bh_release(c);
  nb_ewald_release(c);
  nb_mesh_release(c);
  if (c->state) (void)hipFree(c->state);
  if (c->work) (void)hipFree(c->work);
  if (c->scal) (void)hipFree(c->scal);
  if (c->phi) (void)hipFree(c->phi);
  if (c->elog) (void)hipFree(c->elog);
  if (c->acc_lr) (void)hipFree(c->acc_lr);
  if (c->rung) (void)hipFree(c->rung);
  if (c->rung_hist) (void)hipFree(c->rung_hist);
  if (c->rung_tab) (void)hipFree(c->rung_tab);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return NB_OK;
}

// The handle is BORROWED: the context enqueues on it and synchronises it (nb_synchronize, nb_destroy, the next nb_set_stream, scratch
// growth), so the caller must keep the stream alive until the next nb_set_stream or nb_destroy (include/nbody_amd.h).
int nb_set_stream(nb_ctx* c, void* hip_stream) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_set_stream");
  int rc = use_device(c); if (rc) return rc;
  NB_HIP(c, hipStreamSynchronize(c->stream));
  if (c->step_graph) { (void)hipGraphExecDestroy(c->step_graph); c->step_graph = nullptr; }
  // the handle is used as given; NULL is HIP's default (null) stream, which is what torch's default stream is
  if (c->own_stream) { NB_HIP(c, hipStreamDestroy(c->stream)); c->own_stream = false; }
  c->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return NB_OK;
}

int nb_set_solver(nb_ctx* c, int solver, double theta) {
  if (!c) return NB_ERR_ARG;
  if (c->multi && solver != NB_SOLVER_DIRECT) NB_SINGLE_ONLY(c, "nb_set_solver(tree)");
  if (c->multi) return NB_OK;
  if (solver != NB_SOLVER_DIRECT && solver != NB_SOLVER_TREE) return nb_fail(c, NB_ERR_ARG, "nb_set_solver: unknown solver (use nb_set_periodic for Ewald)");
  if (solver == NB_SOLVER_TREE && c->dtype != NB_F32) return nb_fail(c, NB_ERR_ARG, "nb_set_solver: the tree solver is fp32 only");
  if (solver == NB_SOLVER_TREE && !(theta >= 0.0 && theta < 2.0)) return nb_fail(c, NB_ERR_ARG, "nb_set_solver: theta must be in [0, 2)");
  if (c->step_graph) { (void)hipGraphExecDestroy(c->step_graph); c->step_graph = nullptr; }
  c->solver = solver; c->theta = theta; c->acc_valid = false;
  return NB_OK;
}

int nb_set_periodic(nb_ctx* c, double boxsize, double alpha_L, int kmax) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_set_periodic");
  if (c->step_graph) { (void)hipGraphExecDestroy(c->step_graph); c->step_graph = nullptr; }
  c->acc_valid = false;
  if (!(boxsize > 0.0)) { c->solver = NB_SOLVER_DIRECT; c->box = 0.0; return NB_OK; }
  if (!(alpha_L > 0.0) || kmax < 1 || kmax > 31) return nb_fail(c, NB_ERR_ARG, "nb_set_periodic: alpha_L must be > 0 and kmax in [1, 31]");
  c->solver = NB_SOLVER_EWALD; c->box = boxsize; c->alphaL = alpha_L; c->kmax = kmax;
  return NB_OK;
}

int nb_set_mesh(nb_ctx* c, int solver, double boxsize, int ngrid, double scale_cells, double range_scales, int flags) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_set_mesh");
  if (solver != NB_SOLVER_PM && solver != NB_SOLVER_P3M) return nb_fail(c, NB_ERR_ARG, "nb_set_mesh: solver must be NB_SOLVER_PM or NB_SOLVER_P3M");
  if (!(boxsize > 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_set_mesh: boxsize must be > 0");
  if (ngrid < 4 || ngrid > 2048) return nb_fail(c, NB_ERR_ARG, "nb_set_mesh: grid size must be in [4, 2048]");
  if (solver == NB_SOLVER_P3M) {
    if (!(scale_cells > 0.0) || !(range_scales > 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_set_mesh: split scale and range must be > 0");
    if (std::floor(ngrid / (scale_cells * range_scales)) < 3.0)
      return nb_fail(c, NB_ERR_ARG, "nb_set_mesh: the short-range cut-off exceeds a third of the box (grid too coarse); use nb_set_periodic");
  }
  if (c->step_graph) { (void)hipGraphExecDestroy(c->step_graph); c->step_graph = nullptr; }
  c->acc_valid = false;
  c->lr_valid = false;
  c->solver = solver; c->box = boxsize; c->mesh_grid = ngrid; c->mesh_scale = scale_cells; c->mesh_range = range_scales;
  c->mesh_interlace = (flags & NB_MESH_NO_INTERLACE) ? 0 : 1;
  c->mesh_parts = (flags & NB_MESH_LONG_ONLY) ? 1 : (flags & NB_MESH_SHORT_ONLY) ? 2 : 3;
  return NB_OK;
}

int nb_synchronize(nb_ctx* c) {
  if (!c) return NB_ERR_ARG;
  if (c->multi) return nb_multi_sync(c);
  int rc = use_device(c); if (rc) return rc;
  NB_HIP(c, hipStreamSynchronize(c->stream));
  return NB_OK;
}

int nb_set_state(nb_ctx* c, const void* x, const void* y, const void* z, const void* vx, const void* vy,
                 const void* vz, const void* m) {
  if (!c) return NB_ERR_ARG;
  if (!x || !y || !z || !vx || !vy || !vz || !m) return nb_fail(c, NB_ERR_ARG, "nb_set_state: NULL array");
  if (c->multi) return nb_multi_set_state(c, x, y, z, vx, vy, vz, m);
  int rc = use_device(c); if (rc) return rc;
  const size_t b = (size_t)c->n * esize(c->dtype);
  char* base = reinterpret_cast<char*>(c->state);
  const void* src[7] = {x, y, z, m, vx, vy, vz};  // device field order: x y z m vx vy vz ax ay az
  for (int f = 0; f < 7; ++f) NB_HIP(c, hipMemcpyAsync(base + (size_t)f * b, src[f], b, hipMemcpyHostToDevice, c->stream));
  NB_HIP(c, hipStreamSynchronize(c->stream));
  c->acc_valid = false;
  c->lr_valid = false;
  c->has_state = true;
  // a captured step graph bakes in the UNIFORM / general-mass kernel variant chosen for the previous masses
  if (c->step_graph) { (void)hipGraphExecDestroy(c->step_graph); c->step_graph = nullptr; }
  // equal masses (the standard N-body set-up) unlock the cheaper inner loop; decided on the host copy, O(N)
  bool uni = true;
  if (c->dtype == NB_F32) { const float* mm = (const float*)m; for (int64_t i = 1; i < c->n && uni; ++i) uni = mm[i] == mm[0]; }
  else { const double* mm = (const double*)m; for (int64_t i = 1; i < c->n && uni; ++i) uni = mm[i] == mm[0]; }
  c->uniform_mass = uni;
  return NB_OK;
}

int nb_get_state(nb_ctx* c, void* x, void* y, void* z, void* vx, void* vy, void* vz, void* m) {
  if (!c) return NB_ERR_ARG;
  if (!c->has_state) return nb_fail(c, NB_ERR_STATE, "nb_get_state: no state set");
  if (c->multi) { void* dst[7] = {x, y, z, m, vx, vy, vz}; const int f[7] = {0, 1, 2, 3, 4, 5, 6}; return nb_multi_get(c, dst, f, 7); }
  int rc = use_device(c); if (rc) return rc;
  const size_t b = (size_t)c->n * esize(c->dtype);
  char* base = reinterpret_cast<char*>(c->state);
  void* dst[7] = {x, y, z, m, vx, vy, vz};
  for (int f = 0; f < 7; ++f)
    if (dst[f]) NB_HIP(c, hipMemcpyAsync(dst[f], base + (size_t)f * b, b, hipMemcpyDeviceToHost, c->stream));
  NB_HIP(c, hipStreamSynchronize(c->stream));
  return NB_OK;
}

// a(x) of the context's own particles into the state's ax/ay/az fields
static int refresh_acc(nb_ctx* c, double eps, double G, double* phi = nullptr) {
  int rc;
  c->lr_valid = false;      // the acc arrays are about to hold the TOTAL force again (nb_rung_step keeps the short-range part there)
  if (phi && c->solver != NB_SOLVER_DIRECT) return nb_fail(c, NB_ERR_ARG, "the fused potential needs the direct solver");
  if (c->solver == NB_SOLVER_EWALD) {
    const size_t b = (size_t)c->n * esize(c->dtype);
    char* s = reinterpret_cast<char*>(c->state);
    rc = nb_launch_accel_ewald(c, c->dtype, s, s + b, s + 2 * b, s + 3 * b, c->n, c->box, c->alphaL, c->kmax, eps * eps, G,
                               s + 7 * b, s + 8 * b, s + 9 * b);
  } else if (c->solver == NB_SOLVER_PM || c->solver == NB_SOLVER_P3M) {
    const size_t b = (size_t)c->n * esize(c->dtype);
    char* s = reinterpret_cast<char*>(c->state);
    rc = nb_launch_accel_mesh(c, c->dtype, s, s + b, s + 2 * b, s + 3 * b, c->n, c->box, c->mesh_grid, c->mesh_scale, c->mesh_range,
                              c->solver == NB_SOLVER_P3M, c->mesh_interlace, c->mesh_parts, eps * eps, G, s + 7 * b, s + 8 * b, s + 9 * b);
  } else if (c->solver == NB_SOLVER_TREE) {
    if (c->dtype != NB_F32) return nb_fail(c, NB_ERR_ARG, "the tree solver is fp32 only");
    float* s = reinterpret_cast<float*>(c->state);
    const size_t n = (size_t)c->n;
    rc = nb_launch_accel_tree_f32(c, s, s + n, s + 2 * n, s + 3 * n, c->n, (float)(eps * eps), (float)G, (float)c->theta,
                                  s + 7 * n, s + 8 * n, s + 9 * n);
  } else if (c->dtype == NB_F32) {
    float* s = reinterpret_cast<float*>(c->state);
    const size_t n = (size_t)c->n;
    rc = nb_launch_accel_f32(c, s, s + n, s + 2 * n, c->n, s, s + n, s + 2 * n, s + 3 * n, c->n,
                             (float)(eps * eps), (float)G, s + 7 * n, s + 8 * n, s + 9 * n,
                             c->uniform_mass ? NB_UNIFORM_MASS : 0, phi);
  } else {
    double* s = reinterpret_cast<double*>(c->state);
    const size_t n = (size_t)c->n;
    rc = nb_launch_accel_f64(c, s, s + n, s + 2 * n, c->n, s, s + n, s + 2 * n, s + 3 * n, c->n, eps * eps, G,
                             s + 7 * n, s + 8 * n, s + 9 * n, c->uniform_mass ? NB_UNIFORM_MASS : 0, phi);
  }
  if (rc) return rc;
  c->acc_valid = true; c->acc_eps = eps; c->acc_G = G;
  return NB_OK;
}

// a(x) of the context's state followed by the leapfrog tail  v += a * close  [; v += a * open ; x += v * drift  when `advance`]. The force
// path applies the tail in the kernel that completes the accelerations where it can (pair-symmetric kernels: nb_ctx::Tail); otherwise the
// plain kick / kick-drift kernels run behind it — the same arithmetic either way. After an advancing tail the cached accelerations belong
// to the previous positions.
static int accel_with_tail(nb_ctx* c, double eps, double G, double close, bool advance, double open, double drift, double* phi = nullptr) {
  const size_t b = (size_t)c->n * esize(c->dtype);
  char* s = reinterpret_cast<char*>(c->state);
  void *x = s, *y = s + b, *z = s + 2 * b, *vx = s + 4 * b, *vy = s + 5 * b, *vz = s + 6 * b;
  void *ax = s + 7 * b, *ay = s + 8 * b, *az = s + 9 * b;
  c->tail = nb_ctx::Tail{};
  if (c->solver == NB_SOLVER_DIRECT) {
    c->tail.vx = vx; c->tail.vy = vy; c->tail.vz = vz; c->tail.x = x; c->tail.y = y; c->tail.z = z;
    c->tail.kick_close = close; c->tail.kick_open = open; c->tail.drift = drift; c->tail.advance = advance; c->tail.armed = true;
  }
  int rc = refresh_acc(c, eps, G, phi);
  const bool done = c->tail.done;
  c->tail = nb_ctx::Tail{};
  if (rc) return rc;
  if (!done) {
    if (advance) rc = nb_launch_kick_drift(c, c->dtype, x, y, z, vx, vy, vz, ax, ay, az, c->n, close, open, drift);
    else rc = nb_launch_kick(c, c->dtype, vx, vy, vz, ax, ay, az, c->n, close);
    if (rc) return rc;
  }
  if (advance) {
    c->acc_valid = false;
    if (c->solver >= NB_SOLVER_EWALD) { rc = nb_launch_wrap(c, c->dtype, x, y, z, c->n, c->box); if (rc) return rc; }
  }
  return NB_OK;
}

int nb_accel(nb_ctx* c, double eps, double G, void* ax, void* ay, void* az) {
  if (!c) return NB_ERR_ARG;
  if (!c->has_state) return nb_fail(c, NB_ERR_STATE, "nb_accel: call nb_set_state first");
  if (!(eps >= 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_accel: eps must be >= 0");
  if (c->multi) {
    int rcm = nb_multi_accel(c, eps, G); if (rcm) return rcm;
    void* dst[3] = {ax, ay, az}; const int f[3] = {7, 8, 9};
    return nb_multi_get(c, dst, f, 3);
  }
  int rc = use_device(c); if (rc) return rc;
  rc = refresh_acc(c, eps, G); if (rc) return rc;
  const size_t b = (size_t)c->n * esize(c->dtype);
  char* base = reinterpret_cast<char*>(c->state);
  void* dst[3] = {ax, ay, az};
  for (int f = 0; f < 3; ++f)
    if (dst[f]) NB_HIP(c, hipMemcpyAsync(dst[f], base + (size_t)(7 + f) * b, b, hipMemcpyDeviceToHost, c->stream));
  NB_HIP(c, hipStreamSynchronize(c->stream));
  return nb_bh_check(c);
}

int nb_get_acc(nb_ctx* c, void* ax, void* ay, void* az) {
  if (!c) return NB_ERR_ARG;
  if (c->multi) {
    if (!c->has_state || !nb_multi_has_acc(c)) return nb_fail(c, NB_ERR_STATE, "nb_get_acc: no cached acceleration (call nb_accel or nb_step first)");
    void* dst[3] = {ax, ay, az}; const int f[3] = {7, 8, 9};
    return nb_multi_get(c, dst, f, 3);
  }
  if (!c->has_state || !c->acc_valid) return nb_fail(c, NB_ERR_STATE, "nb_get_acc: no cached acceleration (call nb_accel or nb_step first)");
  int rc = use_device(c); if (rc) return rc;
  const size_t b = (size_t)c->n * esize(c->dtype);
  char* base = reinterpret_cast<char*>(c->state);
  void* dst[3] = {ax, ay, az};
  for (int f = 0; f < 3; ++f)
    if (dst[f]) NB_HIP(c, hipMemcpyAsync(dst[f], base + (size_t)(7 + f) * b, b, hipMemcpyDeviceToHost, c->stream));
  NB_HIP(c, hipStreamSynchronize(c->stream));
  return NB_OK;
}

int nb_step(nb_ctx* c, double dt, int64_t nsteps, double eps, double G) {
  if (!c) return NB_ERR_ARG;
  if (!c->has_state) return nb_fail(c, NB_ERR_STATE, "nb_step: call nb_set_state first");
  if (nsteps < 0) return nb_fail(c, NB_ERR_ARG, "nb_step: nsteps must be >= 0");
  if (!(eps >= 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_step: eps must be >= 0");
  if (nsteps == 0) return NB_OK;
  if (c->multi) {
    std::vector<double> kh((size_t)nsteps, 0.5 * dt), dr((size_t)nsteps, dt);
    return nb_multi_kdk(c, nsteps, kh.data(), dr.data(), kh.data(), eps, G);
  }
  int rc = use_device(c); if (rc) return rc;
  if (!c->acc_valid || c->acc_eps != eps || c->acc_G != G) { rc = refresh_acc(c, eps, G); if (rc) return rc; }
  const size_t b = (size_t)c->n * esize(c->dtype);
  char* s = reinterpret_cast<char*>(c->state);
  void *x = s, *y = s + b, *z = s + 2 * b, *vx = s + 4 * b, *vy = s + 5 * b, *vz = s + 6 * b;
  void *ax = s + 7 * b, *ay = s + 8 * b, *az = s + 9 * b;
  const double h = 0.5 * dt;
  // Step k: opening half kick + drift (a launch of its own for step 0 only), force at the new positions, and the TAIL — the closing half
  // kick of step k and, unless it is the last, the opening half kick and drift of step k + 1 — applied by the kernel that completes the
  // accelerations (accel_with_tail): two launches per step with the pair-symmetric kernels instead of five.
  // Launch-bound regime (small N, many steps): steps 1..n-2 are identical launch sequences with constant arguments, so capture one into a
  // hipGraph and replay it.
  static const bool no_graph = getenv("NB_NO_GRAPH") != nullptr;
  // (never on HIP's legacy NULL stream: it cannot be captured)
  const bool use_graph = !no_graph && c->stream != nullptr && c->solver == NB_SOLVER_DIRECT && c->n <= 65536 && nsteps >= 4;
  rc = nb_launch_kick_drift(c, c->dtype, x, y, z, vx, vy, vz, ax, ay, az, c->n, 0.0, h, dt); if (rc) return rc;
  if (c->solver >= NB_SOLVER_EWALD) { rc = nb_launch_wrap(c, c->dtype, x, y, z, c->n, c->box); if (rc) return rc; }
  for (int64_t k = 0; k < nsteps; ++k) {
    const bool last = k == nsteps - 1;
    if (use_graph && k >= 1 && !last) {
      if (!c->step_graph || c->graph_dt != dt || c->graph_eps != eps || c->graph_G != G) {
        if (c->step_graph) { (void)hipGraphExecDestroy(c->step_graph); c->step_graph = nullptr; }
        // step 0 ran eagerly above, so the scratch buffer already has its final size: no allocation in capture
        hipGraph_t g = nullptr;
        NB_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        rc = accel_with_tail(c, eps, G, h, true, h, dt);
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) return nb_fail(c, NB_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        e = hipGraphInstantiate(&c->step_graph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) { c->step_graph = nullptr; return nb_fail(c, NB_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
        c->graph_dt = dt; c->graph_eps = eps; c->graph_G = G;
      }
      NB_HIP(c, hipGraphLaunch(c->step_graph, c->stream));
      continue;
    }
    rc = accel_with_tail(c, eps, G, h, !last, h, dt); if (rc) return rc;
  }
  c->acc_valid = true; c->acc_eps = eps; c->acc_G = G;      // the last force belongs to the final positions (graph replays included)
  NB_HIP(c, hipStreamSynchronize(c->stream));
  return nb_bh_check(c);
}

static int kdk_enqueue(nb_ctx* c, double kick1, double drift, double kick2, double eps, double G) {
  const size_t b = (size_t)c->n * esize(c->dtype);
  char* s = reinterpret_cast<char*>(c->state);
  void *x = s, *y = s + b, *z = s + 2 * b, *vx = s + 4 * b, *vy = s + 5 * b, *vz = s + 6 * b;
  void *ax = s + 7 * b, *ay = s + 8 * b, *az = s + 9 * b;
  int rc = nb_launch_kick_drift(c, c->dtype, x, y, z, vx, vy, vz, ax, ay, az, c->n, 0.0, kick1, drift); if (rc) return rc;   // v += a*kick1; x += v*drift
  if (c->solver >= NB_SOLVER_EWALD) { rc = nb_launch_wrap(c, c->dtype, x, y, z, c->n, c->box); if (rc) return rc; }
  return accel_with_tail(c, eps, G, kick2, false, 0.0, 0.0);
}

int nb_kdk_step(nb_ctx* c, double kick1, double drift, double kick2, double eps, double G) {
  if (!c) return NB_ERR_ARG;
  if (!c->has_state) return nb_fail(c, NB_ERR_STATE, "nb_kdk_step: call nb_set_state first");
  if (!(eps >= 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_kdk_step: eps must be >= 0");
  if (c->multi) return nb_multi_kdk(c, 1, &kick1, &drift, &kick2, eps, G);
  int rc = use_device(c); if (rc) return rc;
  if (!c->acc_valid || c->acc_eps != eps || c->acc_G != G) { rc = refresh_acc(c, eps, G); if (rc) return rc; }
  rc = kdk_enqueue(c, kick1, drift, kick2, eps, G); if (rc) return rc;
  NB_HIP(c, hipStreamSynchronize(c->stream));
  return nb_bh_check(c);
}

int nb_kdk_run(nb_ctx* c, int64_t nsteps, const double* kick1, const double* drift, const double* kick2, double eps, double G) {
  if (!c) return NB_ERR_ARG;
  if (!c->has_state) return nb_fail(c, NB_ERR_STATE, "nb_kdk_run: call nb_set_state first");
  if (nsteps < 0 || (nsteps > 0 && (!kick1 || !drift || !kick2))) return nb_fail(c, NB_ERR_ARG, "nb_kdk_run: bad step arrays");
  if (!(eps >= 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_kdk_run: eps must be >= 0");
  if (nsteps == 0) return NB_OK;
  if (c->multi) return nb_multi_kdk(c, nsteps, kick1, drift, kick2, eps, G);
  int rc = use_device(c); if (rc) return rc;
  if (!c->acc_valid || c->acc_eps != eps || c->acc_G != G) { rc = refresh_acc(c, eps, G); if (rc) return rc; }
  // Launch-bound sizes (the 32^3 farm jobs of the thesis: ~100 small launches per P3M step, 1.6 ms per step of which the kernels are a
  // fraction): step 0 runs eagerly (it sizes every scratch buffer), then ONE step — with its kick/drift factors read from a device table
  // indexed by a device-side step counter — is captured into a hipGraph and replayed for steps 1..n-1. Any failure to capture (a library
  // call that does not support stream capture) falls back to the eager loop; results are bit-identical either way.
  static const bool no_graph = getenv("NB_NO_GRAPH") != nullptr;
  int64_t k0 = 0;
  c->graph_replayed = false;
  if (!no_graph && c->stream != nullptr && c->solver != NB_SOLVER_TREE && c->n <= 300000 && nsteps >= 8) {
    rc = kdk_enqueue(c, kick1[0], drift[0], kick2[0], eps, G); if (rc) return rc;
    k0 = 1;
    std::vector<double> fac((size_t)nsteps * 3);
    for (int64_t k = 0; k < nsteps; ++k) { fac[3 * k] = kick1[k]; fac[3 * k + 1] = drift[k]; fac[3 * k + 2] = kick2[k]; }
    double* dfac = nullptr; int* dstep = nullptr;
    NB_HIP(c, hipMalloc((void**)&dfac, fac.size() * sizeof(double) + 16));
    dstep = reinterpret_cast<int*>(dfac + fac.size());
    const int one = 1;
    hipError_t e = hipMemcpyAsync(dfac, fac.data(), fac.size() * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dstep, &one, sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    bool ok = e == hipSuccess && hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      const size_t b = (size_t)c->n * esize(c->dtype);
      char* s = reinterpret_cast<char*>(c->state);
      int rcc = nb_launch_kdk_indexed(c, c->dtype, 0, s, s + b, s + 2 * b, s + 4 * b, s + 5 * b, s + 6 * b, s + 7 * b, s + 8 * b, s + 9 * b, c->n, dfac, dstep);
      if (!rcc && c->solver >= NB_SOLVER_EWALD) rcc = nb_launch_wrap(c, c->dtype, s, s + b, s + 2 * b, c->n, c->box);
      if (!rcc) rcc = refresh_acc(c, eps, G);
      if (!rcc) rcc = nb_launch_kdk_indexed(c, c->dtype, 1, s, s + b, s + 2 * b, s + 4 * b, s + 5 * b, s + 6 * b, s + 7 * b, s + 8 * b, s + 9 * b, c->n, dfac, dstep);
      const hipError_t ec = hipStreamEndCapture(c->stream, &g);
      ok = !rcc && ec == hipSuccess && g != nullptr && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess;
    }
    (void)hipGetLastError();                          // a refused capture must not poison the eager fallback
    // Two different failures: capture / instantiate refused (nothing was enqueued: the state is the one after step 0 and the eager loop
    // takes over from step 1), or a replay failed part-way (steps 1..launched ran and advanced the device step counter: the eager loop
    // continues from step 1 + launched once the stream has drained — never from step 1 again).
    int64_t launched = 0;
    bool replay_failed = false;
    if (ok) {
      for (int64_t k = 1; k < nsteps; ++k) {
        if (hipGraphLaunch(ge, c->stream) != hipSuccess) { replay_failed = true; break; }
        ++launched;
      }
    } else {
      c->err.clear();
    }
    hipError_t es = hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    (void)hipFree(dfac);
    if (es != hipSuccess) return nb_fail(c, NB_ERR_HIP, std::string("nb_kdk_run (graph replay): ") + hipGetErrorString(es));
    k0 = 1 + launched;                                // the replays that were enqueued have completed (synchronised above)
    c->graph_replayed = ok && !replay_failed;
  }
  for (int64_t k = k0; k < nsteps; ++k) { rc = kdk_enqueue(c, kick1[k], drift[k], kick2[k], eps, G); if (rc) return rc; }
  NB_HIP(c, hipStreamSynchronize(c->stream));       // the only host round trip of the run
  return nb_bh_check(c);
}

// One base step with a rung hierarchy for the short-range force of a P3M context — the structure CONCEPT's logs show for every base step
// (…/params_id0200_*.log:56-96: long-range kick, short-range kick, "Intertwining drifts of matter with the following particle interactions: short-range",
// long-range kick): the long-range (mesh) force kicks every particle over the base step; each particle sits on rung l and receives its short-range
// kicks in steps of dt_base / 2^l; all particles drift with the finest populated step and the short-range force is re-evaluated after every drift.
int nb_rung_step(nb_ctx* c, int levels, const double* kc, const double* dc, double dt_base, double crit, double eps, double G,
                 int64_t* populations) {
  if (!c) return NB_ERR_ARG;
  if (c->multi) return nb_fail(c, NB_ERR_ARG, "nb_rung_step: not available on a multi-device context");
  if (!c->has_state) return nb_fail(c, NB_ERR_STATE, "nb_rung_step: call nb_set_state first");
  if (c->solver != NB_SOLVER_P3M) return nb_fail(c, NB_ERR_ARG, "nb_rung_step: needs a P3M context (nb_set_mesh): the rungs belong to the short-range force");
  if (levels < 1 || levels > 8 || !kc || !dc) return nb_fail(c, NB_ERR_ARG, "nb_rung_step: levels must be in [1, 8], tables non-NULL");
  if (!(eps >= 0.0) || !(dt_base > 0.0) || !(crit > 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_rung_step: eps >= 0, dt_base > 0, crit > 0");
  int rc = use_device(c); if (rc) return rc;
  const size_t b = (size_t)c->n * esize(c->dtype);
  char* s = reinterpret_cast<char*>(c->state);
  void *x = s, *y = s + b, *z = s + 2 * b, *m = s + 3 * b, *vx = s + 4 * b, *vy = s + 5 * b, *vz = s + 6 * b, *ax = s + 7 * b, *ay = s + 8 * b, *az = s + 9 * b;
  const int Smax = 1 << (levels - 1);
  const size_t tab_len = 2 * (size_t)(2 * Smax + 1);
  if (!c->acc_lr) NB_HIP(c, hipMalloc(&c->acc_lr, 3 * b));
  if (!c->rung) NB_HIP(c, hipMalloc((void**)&c->rung, (size_t)c->n));
  if (!c->rung_hist) NB_HIP(c, hipMalloc((void**)&c->rung_hist, 16 * sizeof(int)));
  if (c->rung_tab_len < tab_len) {
    NB_HIP(c, hipStreamSynchronize(c->stream));
    if (c->rung_tab) (void)hipFree(c->rung_tab);
    c->rung_tab = nullptr; c->rung_tab_len = 0;
    NB_HIP(c, hipMalloc((void**)&c->rung_tab, tab_len * sizeof(double)));
    c->rung_tab_len = tab_len;
  }
  auto force = [&](int parts) -> int {     // parts: 1 = mesh only, 2 = pairs only; result in the state's acc arrays
    return nb_launch_accel_mesh(c, c->dtype, x, y, z, m, c->n, c->box, c->mesh_grid, c->mesh_scale, c->mesh_range, 1, c->mesh_interlace, parts,
                                eps * eps, G, ax, ay, az);
  };
  c->acc_valid = false;                    // the state's acc arrays hold the short-range part only from here on
  if (!c->lr_valid || c->lr_eps != eps || c->lr_G != G) {   // the cached a_lr / a_sr belong to one (eps, G)
    c->lr_valid = false;
    rc = force(1); if (rc) return rc;
    NB_HIP(c, hipMemcpyAsync(c->acc_lr, ax, 3 * b, hipMemcpyDeviceToDevice, c->stream));
    rc = force(2); if (rc) return rc;
    c->lr_valid = true; c->lr_eps = eps; c->lr_G = G;
  }
  // rungs from the short-range acceleration at the start of the base step
  NB_HIP(c, hipMemsetAsync(c->rung_hist, 0, 16 * sizeof(int), c->stream));
  // the limiter takes the SHORT-RANGE acceleration alone — what CONCEPT's logs show ("Determining rung population ... P3M (short-range only)",
  // …/params_id0200_*.log:42-46) and what reproduces its logged populations; NB_RUNG_SR_ONLY=0 uses the total acceleration instead
  static const int sr_only = getenv("NB_RUNG_SR_ONLY") ? atoi(getenv("NB_RUNG_SR_ONLY")) : 1;
  rc = nb_launch_rung_assign(c, c->dtype, ax, c->acc_lr, c->n, dt_base, crit, levels, sr_only, c->rung, c->rung_hist); if (rc) return rc;
  int hist[16];
  NB_HIP(c, hipMemcpyAsync(hist, c->rung_hist, sizeof(hist), hipMemcpyDeviceToHost, c->stream));
  NB_HIP(c, hipStreamSynchronize(c->stream));
  int top = 0;
  for (int l = 0; l < levels; ++l) { if (populations) populations[l] = hist[l]; if (hist[l] > 0) top = l; }
  const int S = 1 << top, stride = Smax / S;      // sub-steps actually taken; the tables are given at the finest possible resolution
  // tables at the resolution in use (host side: tiny)
  std::vector<double> k2(2 * S + 1), d2(2 * S + 1);
  for (int j = 0; j <= 2 * S; ++j) { k2[j] = kc[j * stride]; d2[j] = dc[j * stride]; }
  NB_HIP(c, hipMemcpyAsync(c->rung_tab, k2.data(), k2.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  NB_HIP(c, hipStreamSynchronize(c->stream));
  const double* dkc = c->rung_tab;
  // long-range half kick for everybody
  rc = nb_launch_kick(c, c->dtype, vx, vy, vz, c->acc_lr, (char*)c->acc_lr + b, (char*)c->acc_lr + 2 * b, c->n, k2[S] - k2[0]); if (rc) return rc;
  for (int sub = 0; sub < S; ++sub) {
    rc = nb_launch_rung_kick(c, c->dtype, vx, vy, vz, ax, ay, az, c->rung, c->n, 0, sub, S, top, dkc); if (rc) return rc;
    rc = nb_launch_drift(c, c->dtype, x, y, z, vx, vy, vz, c->n, d2[2 * sub + 2] - d2[2 * sub]); if (rc) return rc;
    rc = nb_launch_wrap(c, c->dtype, x, y, z, c->n, c->box); if (rc) return rc;
    if (sub == S - 1) {                    // end of the base step: the mesh force of the new positions, kept apart
      rc = force(1); if (rc) return rc;
      NB_HIP(c, hipMemcpyAsync(c->acc_lr, ax, 3 * b, hipMemcpyDeviceToDevice, c->stream));
    }
    // after this drift only the rungs with a step boundary at sub + 1 need their pair force: l >= top - (trailing zeros of sub + 1); all at the end
    int tz = 0;
    for (int j = sub + 1; sub + 1 < S && !(j & 1); j >>= 1) ++tz;
    c->sr_rung = sub + 1 < S ? c->rung : nullptr;
    c->sr_min_rung = top - tz > 0 ? top - tz : 0;
    rc = force(2);
    c->sr_rung = nullptr; c->sr_min_rung = 0;
    if (rc) return rc;
    rc = nb_launch_rung_kick(c, c->dtype, vx, vy, vz, ax, ay, az, c->rung, c->n, 1, sub, S, top, dkc); if (rc) return rc;
  }
  rc = nb_launch_kick(c, c->dtype, vx, vy, vz, c->acc_lr, (char*)c->acc_lr + b, (char*)c->acc_lr + 2 * b, c->n, k2[2 * S] - k2[S]); if (rc) return rc;
  NB_HIP(c, hipStreamSynchronize(c->stream));
  return NB_OK;
}

int nb_last_run_used_graph(const nb_ctx* c) { return c ? (c->graph_replayed ? 1 : 0) : NB_ERR_ARG; }

int nb_set_option(nb_ctx* c, int option, int64_t value) {
  if (!c) return NB_ERR_ARG;
  const long long v = value < 0 ? -1 : (long long)value;
  switch (option) {
    case NB_OPT_PAIR_SYMMETRY: c->sym_enable = v; break;
    case NB_OPT_SYM_MIN_N: c->sym_min_n = v; break;
    case NB_OPT_SYM_SLAB_BYTES: c->sym_slab_bytes = v; break;
    case NB_OPT_SYM_SCRATCH_LIMIT: c->sym_scratch_limit = v; break;
    default: return nb_fail(c, NB_ERR_ARG, "nb_set_option: unknown option");
  }
  if (c->multi) return nb_multi_set_option(c, option, value);
  c->acc_valid = false;                               // the cached accelerations came from the other kernel's summation order
  if (c->step_graph) { (void)hipGraphExecDestroy(c->step_graph); c->step_graph = nullptr; }   // a captured step holds the old choice
  return NB_OK;
}

int nb_last_accel_symmetric(const nb_ctx* c) {
  if (!c) return NB_ERR_ARG;
  if (c->multi) return nb_multi_last_symmetric(c);          // the own-chunk launch of device 0
  return c->last_accel_sym ? 1 : 0;
}

int nb_get_option(const nb_ctx* c, int option, int64_t* value) {
  if (!c || !value) return NB_ERR_ARG;
  switch (option) {
    case NB_OPT_PAIR_SYMMETRY: *value = c->sym_enable; break;
    case NB_OPT_SYM_MIN_N: *value = c->sym_min_n; break;
    case NB_OPT_SYM_SLAB_BYTES: *value = c->sym_slab_bytes; break;
    case NB_OPT_SYM_SCRATCH_LIMIT: *value = c->sym_scratch_limit; break;
    case NB_QUERY_MULTI_SCHEDULE: *value = c->multi ? nb_multi_schedule_pairwise(c) : 0; break;
    case NB_QUERY_OWN_CHUNK_SYMMETRIC: *value = c->multi ? nb_multi_last_symmetric(c) : (c->last_accel_sym ? 1 : 0); break;
    default: return NB_ERR_ARG;
  }
  return NB_OK;
}

int nb_energy(nb_ctx* c, double eps, double G, double* kinetic, double* potential) {
  if (!c) return NB_ERR_ARG;
  if (!c->has_state) return nb_fail(c, NB_ERR_STATE, "nb_energy: call nb_set_state first");
  if (!(eps >= 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_energy: eps must be >= 0");
  if (c->multi) return nb_multi_energy(c, eps, G, kinetic, potential);
  int rc = use_device(c); if (rc) return rc;
  const size_t b = (size_t)c->n * esize(c->dtype);
  char* s = reinterpret_cast<char*>(c->state);
  rc = nb_launch_kinetic(c, c->dtype, s + 4 * b, s + 5 * b, s + 6 * b, s + 3 * b, c->n, c->scal); if (rc) return rc;
  if (c->dtype == NB_F32)
    rc = nb_launch_potential_f32(c, (float*)s, (float*)(s + b), (float*)(s + 2 * b), (float*)(s + 3 * b), c->n, 0,
                                 (float*)s, (float*)(s + b), (float*)(s + 2 * b), (float*)(s + 3 * b), c->n, 0,
                                 (float)(eps * eps), c->scal + 1, 0);
  else
    rc = nb_launch_potential_f64(c, (double*)s, (double*)(s + b), (double*)(s + 2 * b), (double*)(s + 3 * b), c->n, 0,
                                 (double*)s, (double*)(s + b), (double*)(s + 2 * b), (double*)(s + 3 * b), c->n, 0,
                                 eps * eps, c->scal + 1, 0);
  if (rc) return rc;
  double h[2];
  NB_HIP(c, hipMemcpyAsync(h, c->scal, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  NB_HIP(c, hipStreamSynchronize(c->stream));
  if (kinetic) *kinetic = h[0];
  if (potential) *potential = -0.5 * G * h[1];
  return NB_OK;
}

// KDK steps with the energies of every step: the potential comes out of the force kernel itself (one extra
// accumulator per sink, SURVEY.md §8a row ⊕3) instead of a second O(N^2) pass per diagnostic.
int nb_step_energy(nb_ctx* c, double dt, int64_t nsteps, double eps, double G, double* kinetic, double* potential) {
  if (!c) return NB_ERR_ARG;
  if (!c->has_state) return nb_fail(c, NB_ERR_STATE, "nb_step_energy: call nb_set_state first");
  if (nsteps < 0) return nb_fail(c, NB_ERR_ARG, "nb_step_energy: nsteps must be >= 0");
  if (!(eps >= 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_step_energy: eps must be >= 0");
  if (c->solver != NB_SOLVER_DIRECT) return nb_fail(c, NB_ERR_ARG, "nb_step_energy: direct solver only");
  if (nsteps == 0) return NB_OK;
  if (c->multi) return nb_multi_step_energy(c, dt, nsteps, eps, G, kinetic, potential);
  int rc = use_device(c); if (rc) return rc;
  if (!c->phi) NB_HIP(c, hipMalloc(&c->phi, (size_t)c->n * sizeof(double)));
  if (c->elog_cap < nsteps) {
    NB_HIP(c, hipStreamSynchronize(c->stream));
    if (c->elog) { (void)hipFree(c->elog); c->elog = nullptr; c->elog_cap = 0; }
    NB_HIP(c, hipMalloc(&c->elog, (size_t)nsteps * 2 * sizeof(double)));
    c->elog_cap = nsteps;
  }
  if (!c->acc_valid || c->acc_eps != eps || c->acc_G != G) { rc = refresh_acc(c, eps, G); if (rc) return rc; }
  const size_t b = (size_t)c->n * esize(c->dtype);
  char* s = reinterpret_cast<char*>(c->state);
  void *x = s, *y = s + b, *z = s + 2 * b, *m = s + 3 * b, *vx = s + 4 * b, *vy = s + 5 * b, *vz = s + 6 * b;
  void *ax = s + 7 * b, *ay = s + 8 * b, *az = s + 9 * b;
  const double h = 0.5 * dt;
  for (int64_t k = 0; k < nsteps; ++k) {
    rc = nb_launch_kick_drift(c, c->dtype, x, y, z, vx, vy, vz, ax, ay, az, c->n, 0.0, h, dt); if (rc) return rc;
    rc = accel_with_tail(c, eps, G, h, false, 0.0, 0.0, c->phi); if (rc) return rc;        // closing half kick with the force: synchronised velocities
    rc = nb_launch_kinetic(c, c->dtype, vx, vy, vz, m, c->n, c->elog + 2 * k); if (rc) return rc;
    rc = nb_launch_pot_energy(c, c->dtype, c->phi, m, c->n, eps * eps, c->dtype == NB_F32 && c->uniform_mass,
                              c->elog + 2 * k + 1); if (rc) return rc;
  }
  std::vector<double> host((size_t)nsteps * 2);
  NB_HIP(c, hipMemcpyAsync(host.data(), c->elog, host.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  NB_HIP(c, hipStreamSynchronize(c->stream));
  for (int64_t k = 0; k < nsteps; ++k) {
    if (kinetic) kinetic[k] = host[2 * k];
    if (potential) potential[k] = -0.5 * G * host[2 * k + 1];
  }
  return NB_OK;
}

/* ---------------- device-pointer surface ---------------- */

int nb_accel_dev(nb_ctx* c, int dtype, const void* xi, const void* yi, const void* zi, int64_t n_snk,
                 const void* xs, const void* ys, const void* zs, const void* ms, int64_t n_src,
                 double eps, double G, void* ax, void* ay, void* az, int flags) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_accel_dev");
  if (n_snk < 0 || n_src < 0) return nb_fail(c, NB_ERR_ARG, "nb_accel_dev: negative count");
  if (n_snk > 0 && (!xi || !yi || !zi || !ax || !ay || !az)) return nb_fail(c, NB_ERR_ARG, "nb_accel_dev: NULL sink array");
  if (n_src > 0 && (!xs || !ys || !zs || !ms)) return nb_fail(c, NB_ERR_ARG, "nb_accel_dev: NULL source array");
  if (!(eps >= 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_accel_dev: eps must be >= 0");
  int rc = use_device(c); if (rc) return rc;
  if (dtype == NB_F32)
    return nb_launch_accel_f32(c, (const float*)xi, (const float*)yi, (const float*)zi, n_snk, (const float*)xs,
                               (const float*)ys, (const float*)zs, (const float*)ms, n_src, (float)(eps * eps),
                               (float)G, (float*)ax, (float*)ay, (float*)az, flags);
  if (dtype == NB_F64)
    return nb_launch_accel_f64(c, (const double*)xi, (const double*)yi, (const double*)zi, n_snk, (const double*)xs,
                               (const double*)ys, (const double*)zs, (const double*)ms, n_src, eps * eps, G,
                               (double*)ax, (double*)ay, (double*)az, flags);
  return nb_fail(c, NB_ERR_ARG, "nb_accel_dev: unknown dtype");
}

int nb_accel_pot_dev(nb_ctx* c, int dtype, const void* xi, const void* yi, const void* zi, int64_t n_snk,
                     const void* xs, const void* ys, const void* zs, const void* ms, int64_t n_src,
                     double eps, double G, void* ax, void* ay, void* az, double* phi, int flags) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_accel_pot_dev");
  if (n_snk < 0 || n_src < 0) return nb_fail(c, NB_ERR_ARG, "nb_accel_pot_dev: negative count");
  if (n_snk > 0 && (!xi || !yi || !zi || !ax || !ay || !az || !phi)) return nb_fail(c, NB_ERR_ARG, "nb_accel_pot_dev: NULL sink array");
  if (n_src > 0 && (!xs || !ys || !zs || !ms)) return nb_fail(c, NB_ERR_ARG, "nb_accel_pot_dev: NULL source array");
  if (!(eps >= 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_accel_pot_dev: eps must be >= 0");
  int rc = use_device(c); if (rc) return rc;
  if (dtype == NB_F32)
    return nb_launch_accel_f32(c, (const float*)xi, (const float*)yi, (const float*)zi, n_snk, (const float*)xs,
                               (const float*)ys, (const float*)zs, (const float*)ms, n_src, (float)(eps * eps),
                               (float)G, (float*)ax, (float*)ay, (float*)az, flags, phi);
  if (dtype == NB_F64)
    return nb_launch_accel_f64(c, (const double*)xi, (const double*)yi, (const double*)zi, n_snk, (const double*)xs,
                               (const double*)ys, (const double*)zs, (const double*)ms, n_src, eps * eps, G,
                               (double*)ax, (double*)ay, (double*)az, flags, phi);
  return nb_fail(c, NB_ERR_ARG, "nb_accel_pot_dev: unknown dtype");
}

int nb_accel_pair_dev(nb_ctx* c, int dtype, const void* xa, const void* ya, const void* za, const void* ma, int64_t na,
                      const void* xb, const void* yb, const void* zb, const void* mb, int64_t nb, double eps, double G,
                      void* aax, void* aay, void* aaz, void* bax, void* bay, void* baz, double* phi_a, double* phi_b, int flags) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_accel_pair_dev");
  if (na < 0 || nb < 0) return nb_fail(c, NB_ERR_ARG, "nb_accel_pair_dev: negative count");
  if (na > 0 && (!xa || !ya || !za || !ma || !aax || !aay || !aaz)) return nb_fail(c, NB_ERR_ARG, "nb_accel_pair_dev: NULL array of block A");
  if (nb > 0 && (!xb || !yb || !zb || !mb || !bax || !bay || !baz)) return nb_fail(c, NB_ERR_ARG, "nb_accel_pair_dev: NULL array of block B");
  if ((phi_a == nullptr) != (phi_b == nullptr)) return nb_fail(c, NB_ERR_ARG, "nb_accel_pair_dev: phi_a and phi_b go together");
  if (xa == xb || aax == bax) return nb_fail(c, NB_ERR_ARG, "nb_accel_pair_dev: A and B must be different blocks (a block on itself: nb_accel_dev)");
  if (!(eps >= 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_accel_pair_dev: eps must be >= 0");
  if (dtype != NB_F32 && dtype != NB_F64) return nb_fail(c, NB_ERR_ARG, "nb_accel_pair_dev: unknown dtype");
  int rc = use_device(c); if (rc) return rc;
  c->last_accel_sym = false;
  if (na == 0 || nb == 0) {                                   // nothing to exchange: B's output is zero, A's untouched
    if (nb > 0) {
      const size_t bytes = (size_t)nb * esize(dtype);
      NB_HIP(c, hipMemsetAsync(bax, 0, bytes, c->stream)); NB_HIP(c, hipMemsetAsync(bay, 0, bytes, c->stream)); NB_HIP(c, hipMemsetAsync(baz, 0, bytes, c->stream));
      if (phi_b) NB_HIP(c, hipMemsetAsync(phi_b, 0, (size_t)nb * sizeof(double), c->stream));
    }
    return NB_OK;
  }
  return nb_launch_accel_pair(c, dtype, xa, ya, za, ma, na, xb, yb, zb, mb, nb, eps * eps, G, aax, aay, aaz, bax, bay, baz, phi_a, phi_b, flags);
}

int nb_pot_energy_dev(nb_ctx* c, int dtype, const double* phi, const void* m, int64_t n, double eps, int uniform, double* w_out) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_pot_energy_dev");
  if (!phi || !m || !w_out || n < 0) return nb_fail(c, NB_ERR_ARG, "nb_pot_energy_dev: bad argument");
  if (dtype != NB_F32 && dtype != NB_F64) return nb_fail(c, NB_ERR_ARG, "nb_pot_energy_dev: unknown dtype");
  if (!(eps >= 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_pot_energy_dev: eps must be >= 0");
  int rc = use_device(c); if (rc) return rc;
  return nb_launch_pot_energy(c, dtype, phi, m, n, eps * eps, dtype == NB_F32 && uniform, w_out);
}

int nb_potential_dev(nb_ctx* c, int dtype, const void* xi, const void* yi, const void* zi, const void* mi,
                     int64_t n_snk, int64_t snk_global0, const void* xs, const void* ys, const void* zs,
                     const void* ms, int64_t n_src, int64_t src_global0, double eps, double* u_out, int accumulate) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_potential_dev");
  if (!u_out) return nb_fail(c, NB_ERR_ARG, "nb_potential_dev: u_out is NULL");
  if (n_snk < 0 || n_src < 0) return nb_fail(c, NB_ERR_ARG, "nb_potential_dev: negative count");
  if (!(eps >= 0.0)) return nb_fail(c, NB_ERR_ARG, "nb_potential_dev: eps must be >= 0");
  int rc = use_device(c); if (rc) return rc;
  if (dtype == NB_F32)
    return nb_launch_potential_f32(c, (const float*)xi, (const float*)yi, (const float*)zi, (const float*)mi, n_snk,
                                   snk_global0, (const float*)xs, (const float*)ys, (const float*)zs,
                                   (const float*)ms, n_src, src_global0, (float)(eps * eps), u_out, accumulate);
  if (dtype == NB_F64)
    return nb_launch_potential_f64(c, (const double*)xi, (const double*)yi, (const double*)zi, (const double*)mi,
                                   n_snk, snk_global0, (const double*)xs, (const double*)ys, (const double*)zs,
                                   (const double*)ms, n_src, src_global0, eps * eps, u_out, accumulate);
  return nb_fail(c, NB_ERR_ARG, "nb_potential_dev: unknown dtype");
}

int nb_kinetic_dev(nb_ctx* c, int dtype, const void* vx, const void* vy, const void* vz, const void* m, int64_t n,
                   double* k_out) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_kinetic_dev");
  if (!k_out || n < 0) return nb_fail(c, NB_ERR_ARG, "nb_kinetic_dev: bad argument");
  if (dtype != NB_F32 && dtype != NB_F64) return nb_fail(c, NB_ERR_ARG, "nb_kinetic_dev: unknown dtype");
  int rc = use_device(c); if (rc) return rc;
  return nb_launch_kinetic(c, dtype, vx, vy, vz, m, n, k_out);
}

int nb_kick_dev(nb_ctx* c, int dtype, void* vx, void* vy, void* vz, const void* ax, const void* ay, const void* az,
                int64_t n, double h) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_kick_dev");
  if (dtype != NB_F32 && dtype != NB_F64) return nb_fail(c, NB_ERR_ARG, "nb_kick_dev: unknown dtype");
  int rc = use_device(c); if (rc) return rc;
  return nb_launch_kick(c, dtype, vx, vy, vz, ax, ay, az, n, h);
}

int nb_kick_drift_dev(nb_ctx* c, int dtype, void* x, void* y, void* z, void* vx, void* vy, void* vz,
                      const void* ax, const void* ay, const void* az, int64_t n, double kick, double drift) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_kick_drift_dev");
  if (dtype != NB_F32 && dtype != NB_F64) return nb_fail(c, NB_ERR_ARG, "nb_kick_drift_dev: unknown dtype");
  if (n < 0 || (n > 0 && (!x || !y || !z || !vx || !vy || !vz || !ax || !ay || !az))) return nb_fail(c, NB_ERR_ARG, "nb_kick_drift_dev: bad argument");
  int rc = use_device(c); if (rc) return rc;
  return nb_launch_kick_drift(c, dtype, x, y, z, vx, vy, vz, ax, ay, az, n, 0.0, kick, drift);
}

int nb_accel_kdk_dev(nb_ctx* c, int dtype, void* xi, void* yi, void* zi, int64_t n_snk,
                     const void* xs, const void* ys, const void* zs, const void* ms, int64_t n_src,
                     double eps, double G, void* ax, void* ay, void* az, int flags,
                     void* vx, void* vy, void* vz, double kick_close, double kick_open, double drift) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_accel_kdk_dev");
  if (dtype != NB_F32 && dtype != NB_F64) return nb_fail(c, NB_ERR_ARG, "nb_accel_kdk_dev: unknown dtype");
  if (n_snk > 0 && (!vx || !vy || !vz)) return nb_fail(c, NB_ERR_ARG, "nb_accel_kdk_dev: NULL velocity array");
  const bool advance = kick_open != 0.0 || drift != 0.0;
  c->tail = nb_ctx::Tail{};
  c->tail.vx = vx; c->tail.vy = vy; c->tail.vz = vz; c->tail.x = xi; c->tail.y = yi; c->tail.z = zi;
  c->tail.kick_close = kick_close; c->tail.kick_open = kick_open; c->tail.drift = drift; c->tail.advance = advance; c->tail.armed = true;
  int rc = nb_accel_dev(c, dtype, xi, yi, zi, n_snk, xs, ys, zs, ms, n_src, eps, G, ax, ay, az, flags);
  const bool done = c->tail.done;
  c->tail = nb_ctx::Tail{};
  if (rc || done) return rc;
  if (advance) return nb_launch_kick_drift(c, dtype, xi, yi, zi, vx, vy, vz, ax, ay, az, n_snk, kick_close, kick_open, drift);
  return nb_launch_kick(c, dtype, vx, vy, vz, ax, ay, az, n_snk, kick_close);
}

int nb_drift_dev(nb_ctx* c, int dtype, void* x, void* y, void* z, const void* vx, const void* vy, const void* vz,
                 int64_t n, double dt) {
  if (!c) return NB_ERR_ARG;
  NB_SINGLE_ONLY(c, "nb_drift_dev");
  if (dtype != NB_F32 && dtype != NB_F64) return nb_fail(c, NB_ERR_ARG, "nb_drift_dev: unknown dtype");
  int rc = use_device(c); if (rc) return rc;
  return nb_launch_drift(c, dtype, x, y, z, vx, vy, vz, n, dt);
}

}  // extern "C"

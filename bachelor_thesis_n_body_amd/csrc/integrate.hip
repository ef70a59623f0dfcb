// O(N) streaming kernels of the KDK leapfrog (SURVEY.md §8a row ⊕2) and the kinetic-energy reduction
// (row ⊕3). HBM-bound: 28-56 B per particle per step; they exist so that nb_step() never leaves the
// device. Products and sums are rounded separately (no FMA contraction) so that, given the same
// accelerations, the fp64 integrator state is bit-identical to the NumPy oracle's `v += a*h; x += v*dt`.
#include "nb_common.h"
#include "../../include/nbody_amd.h"

namespace {

#pragma clang fp contract(off)

template <typename T>
__global__ void __launch_bounds__(256)
kick_kernel(T* __restrict__ vx, T* __restrict__ vy, T* __restrict__ vz, const T* __restrict__ ax,
            const T* __restrict__ ay, const T* __restrict__ az, int64_t n, T h) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    vx[i] = vx[i] + ax[i] * h;
    vy[i] = vy[i] + ay[i] * h;
    vz[i] = vz[i] + az[i] * h;
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
drift_kernel(T* __restrict__ x, T* __restrict__ y, T* __restrict__ z, const T* __restrict__ vx,
             const T* __restrict__ vy, const T* __restrict__ vz, int64_t n, T dt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    x[i] = x[i] + vx[i] * dt;
    y[i] = y[i] + vy[i] * dt;
    z[i] = z[i] + vz[i] * dt;
  }
}

// closing half-kick of step k, opening half-kick of step k+1 and its drift in one pass over the state:
//   v += a*h1 ; v += a*h2 ; x += v*dt      (h1 = 0 for the very first step)
template <typename T>
__global__ void __launch_bounds__(256)
kick_drift_kernel(T* __restrict__ x, T* __restrict__ y, T* __restrict__ z, T* __restrict__ vx,
                  T* __restrict__ vy, T* __restrict__ vz, const T* __restrict__ ax, const T* __restrict__ ay,
                  const T* __restrict__ az, int64_t n, T h1, T h2, T dt, int two) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    T a0 = ax[i], a1 = ay[i], a2 = az[i];
    T v0 = vx[i], v1 = vy[i], v2 = vz[i];
    if (two) { v0 = v0 + a0 * h1; v1 = v1 + a1 * h1; v2 = v2 + a2 * h1; }
    v0 = v0 + a0 * h2; v1 = v1 + a1 * h2; v2 = v2 + a2 * h2;
    vx[i] = v0; vy[i] = v1; vz[i] = v2;
    x[i] = x[i] + v0 * dt; y[i] = y[i] + v1 * dt; z[i] = z[i] + v2 * dt;
  }
}

// Generalised KDK step whose factors come from DEVICE memory, indexed by a device-side step counter: the launch sequence of a step is then
// identical for every step of a run and can be captured once into a hipGraph and replayed (nb_kdk_run on launch-bound sizes).
//   pre : v += a*kick1[s] ; x += v*drift[s]      post: v += a*kick2[s]      (s = *step; same arithmetic as kick_kernel + drift_kernel)
template <typename T>
__global__ void __launch_bounds__(256)
kdk_pre_kernel(T* __restrict__ x, T* __restrict__ y, T* __restrict__ z, T* __restrict__ vx, T* __restrict__ vy, T* __restrict__ vz,
               const T* __restrict__ ax, const T* __restrict__ ay, const T* __restrict__ az, int64_t n,
               const double* __restrict__ fac, const int* __restrict__ step) {
  const int s = step[0];
  const T h = (T)fac[3 * s], dt = (T)fac[3 * s + 1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const T v0 = vx[i] + ax[i] * h, v1 = vy[i] + ay[i] * h, v2 = vz[i] + az[i] * h;
    vx[i] = v0; vy[i] = v1; vz[i] = v2;
    x[i] = x[i] + v0 * dt; y[i] = y[i] + v1 * dt; z[i] = z[i] + v2 * dt;
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
kdk_post_kernel(T* __restrict__ vx, T* __restrict__ vy, T* __restrict__ vz, const T* __restrict__ ax, const T* __restrict__ ay,
                const T* __restrict__ az, int64_t n, const double* __restrict__ fac, const int* __restrict__ step) {
  const T h = (T)fac[3 * step[0] + 2];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    vx[i] = vx[i] + ax[i] * h; vy[i] = vy[i] + ay[i] * h; vz[i] = vz[i] + az[i] * h;
  }
}

// ---- rung hierarchy (nb_rung_step): per-particle rungs for the short-range force, as in CONCEPT's time loop
// rung_i = the lowest rung l < levels with dt_base / 2^l <= sqrt(crit / |g_i|), g = a_sr + a_lr; hist[l] counts the populations.
template <typename T>
__global__ void __launch_bounds__(256)
rung_assign_kernel(const T* __restrict__ sx, const T* __restrict__ sy, const T* __restrict__ sz, const T* __restrict__ lx,
                   const T* __restrict__ ly, const T* __restrict__ lz, int64_t n, double dt_base, double crit, int levels, int sr_only,
                   unsigned char* __restrict__ rung, int* __restrict__ hist) {
  __shared__ int h[16];
  if (threadIdx.x < 16) h[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double wl = sr_only ? 0.0 : 1.0;
    const double gx = (double)sx[i] + wl * (double)lx[i], gy = (double)sy[i] + wl * (double)ly[i], gz = (double)sz[i] + wl * (double)lz[i];
    const double g = sqrt(gx * gx + gy * gy + gz * gz);
    const double dti = g > 0.0 ? sqrt(crit / g) : 1e300;
    int l = 0;
    double dt = dt_base;
    while (l < levels - 1 && dt > dti) { dt *= 0.5; ++l; }
    rung[i] = (unsigned char)l;
    atomicAdd(&h[l], 1);
  }
  __syncthreads();
  if (threadIdx.x < 16 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// short-range kicks of sub-step s of S (finest level): phase 0 = opening half-kick of every rung whose step BEGINS at s, phase 1 = closing half-kick of
// every rung whose step ENDS at s + 1. kc = cumulative kick integral at the 2 S + 1 half-sub-step boundaries of the base step (device, double).
template <typename T>
__global__ void __launch_bounds__(256)
rung_kick_kernel(T* __restrict__ vx, T* __restrict__ vy, T* __restrict__ vz, const T* __restrict__ ax, const T* __restrict__ ay,
                 const T* __restrict__ az, const unsigned char* __restrict__ rung, int64_t n, int phase, int s, int S, int top,
                 const double* __restrict__ kc) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int l = rung[i];
    l = l > top ? top : l;
    const int len = S >> l;                      // the rung's step in finest sub-steps
    T h;
    if (phase == 0) { if (s % len) continue; h = (T)(kc[2 * s + len] - kc[2 * s]); }
    else { if ((s + 1) % len) continue; h = (T)(kc[2 * (s + 1)] - kc[2 * (s + 1) - len]); }
    vx[i] = vx[i] + ax[i] * h; vy[i] = vy[i] + ay[i] * h; vz[i] = vz[i] + az[i] * h;
  }
}

__global__ void kdk_advance_kernel(int* __restrict__ step) { if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1; }

// per-block fp64 partial of sum m v^2 / 2 -> part[blockIdx.x]; fixed-order final pass below
template <typename T>
__global__ void __launch_bounds__(256)
kinetic_partial_kernel(const T* __restrict__ vx, const T* __restrict__ vy, const T* __restrict__ vz,
                       const T* __restrict__ m, int64_t n, double* __restrict__ part) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double a = (double)vx[i], b = (double)vy[i], c = (double)vz[i];
    s += (double)m[i] * (a * a + b * b + c * c);
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = 0.5 * red[0];
}

// per-block fp64 partial of sum_i m_i (phi_i - self_i). phi_i comes from the fused force+potential kernels
// (accel_f32.hip / accel_f64.hip), which do not skip j == i: the self term is what THEY added for r = 0 —
//   fp32, per-source mass:  (float)(m_i * v_rsq_f32(eps^2));   fp32, uniform mass: m_0 * (double)v_rsq_f32(eps^2);
//   fp64: m_i / sqrt(eps^2);   eps = 0: nothing (the GUARD variants drop coincident pairs).
template <typename T>
__global__ void __launch_bounds__(256)
pot_energy_partial_kernel(const double* __restrict__ phi, const T* __restrict__ m, int64_t n, double eps2, int uniform,
                          double* __restrict__ part) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const T mi = m[i];
    double self = 0.0;
    // same predicate as the force launchers' GUARD choice: fp32 drops coincident pairs for eps^2 < 1e-20f (accel_f32.hip),
    // fp64 only for eps = 0 (accel_f64.hip) — the self term is subtracted exactly when it was added
    if (sizeof(T) == 4 ? (float)eps2 >= 1e-20f : eps2 > 0.0) {
      if (sizeof(T) == 4) {
        const float ri = __builtin_amdgcn_rsqf((float)eps2);
        self = uniform ? (double)mi * (double)ri : (double)((float)mi * ri);
      } else {
        self = (double)mi / sqrt(eps2);
      }
    }
    s += (double)mi * (phi[i] - self);
  }
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// out[0] = (accumulate ? out[0] : 0) + scale * sum_b part[b]     (single block, fixed order)
__global__ void __launch_bounds__(256)
final_sum_kernel(const double* __restrict__ part, int nparts, double scale, double* __restrict__ out,
                    int accumulate) {
  __shared__ double red[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nparts; b += 256) s += part[b];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0) + scale * red[0];
}

}  // namespace

int nb_launch_final_sum(nb_ctx* c, const double* part, int nparts, double scale, double* out, int accumulate) {
  final_sum_kernel<<<1, 256, 0, c->stream>>>(part, nparts, scale, out, accumulate);
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

static inline int stream_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));  // <= 8 blocks per CU, grid-stride the rest
}

int nb_launch_kick(nb_ctx* c, int dtype, void* vx, void* vy, void* vz, const void* ax, const void* ay,
                   const void* az, int64_t n, double h) {
  if (n <= 0) return NB_OK;
  if (dtype == NB_F32)
    kick_kernel<float><<<stream_grid(n), 256, 0, c->stream>>>((float*)vx, (float*)vy, (float*)vz, (const float*)ax,
                                                              (const float*)ay, (const float*)az, n, (float)h);
  else
    kick_kernel<double><<<stream_grid(n), 256, 0, c->stream>>>((double*)vx, (double*)vy, (double*)vz, (const double*)ax,
                                                               (const double*)ay, (const double*)az, n, h);
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

int nb_launch_drift(nb_ctx* c, int dtype, void* x, void* y, void* z, const void* vx, const void* vy,
                    const void* vz, int64_t n, double dt) {
  if (n <= 0) return NB_OK;
  if (dtype == NB_F32)
    drift_kernel<float><<<stream_grid(n), 256, 0, c->stream>>>((float*)x, (float*)y, (float*)z, (const float*)vx,
                                                               (const float*)vy, (const float*)vz, n, (float)dt);
  else
    drift_kernel<double><<<stream_grid(n), 256, 0, c->stream>>>((double*)x, (double*)y, (double*)z, (const double*)vx,
                                                                (const double*)vy, (const double*)vz, n, dt);
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

int nb_launch_kick_drift(nb_ctx* c, int dtype, void* x, void* y, void* z, void* vx, void* vy, void* vz,
                         const void* ax, const void* ay, const void* az, int64_t n, double h1, double h2, double dt) {
  if (n <= 0) return NB_OK;
  const int two = h1 != 0.0;
  if (dtype == NB_F32)
    kick_drift_kernel<float><<<stream_grid(n), 256, 0, c->stream>>>(
        (float*)x, (float*)y, (float*)z, (float*)vx, (float*)vy, (float*)vz, (const float*)ax, (const float*)ay,
        (const float*)az, n, (float)h1, (float)h2, (float)dt, two);
  else
    kick_drift_kernel<double><<<stream_grid(n), 256, 0, c->stream>>>(
        (double*)x, (double*)y, (double*)z, (double*)vx, (double*)vy, (double*)vz, (const double*)ax,
        (const double*)ay, (const double*)az, n, h1, h2, dt, two);
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

int nb_launch_rung_assign(nb_ctx* c, int dtype, const void* sr, const void* lr, int64_t n, double dt_base, double crit, int levels, int sr_only,
                          unsigned char* rung, int* hist) {
  const size_t es = dtype == NB_F64 ? 8 : 4;
  const char *s = (const char*)sr, *l = (const char*)lr;
  if (dtype == NB_F32)
    rung_assign_kernel<float><<<stream_grid(n), 256, 0, c->stream>>>((const float*)s, (const float*)(s + n * es), (const float*)(s + 2 * n * es),
                                                                     (const float*)l, (const float*)(l + n * es), (const float*)(l + 2 * n * es), n, dt_base, crit, levels, sr_only, rung, hist);
  else
    rung_assign_kernel<double><<<stream_grid(n), 256, 0, c->stream>>>((const double*)s, (const double*)(s + n * es), (const double*)(s + 2 * n * es),
                                                                      (const double*)l, (const double*)(l + n * es), (const double*)(l + 2 * n * es), n, dt_base, crit, levels, sr_only, rung, hist);
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

int nb_launch_rung_kick(nb_ctx* c, int dtype, void* vx, void* vy, void* vz, const void* ax, const void* ay, const void* az,
                        const unsigned char* rung, int64_t n, int phase, int s, int S, int top, const double* kc) {
  if (dtype == NB_F32)
    rung_kick_kernel<float><<<stream_grid(n), 256, 0, c->stream>>>((float*)vx, (float*)vy, (float*)vz, (const float*)ax, (const float*)ay, (const float*)az,
                                                                   rung, n, phase, s, S, top, kc);
  else
    rung_kick_kernel<double><<<stream_grid(n), 256, 0, c->stream>>>((double*)vx, (double*)vy, (double*)vz, (const double*)ax, (const double*)ay, (const double*)az,
                                                                    rung, n, phase, s, S, top, kc);
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

int nb_launch_kdk_indexed(nb_ctx* c, int dtype, int phase, void* x, void* y, void* z, void* vx, void* vy, void* vz,
                          const void* ax, const void* ay, const void* az, int64_t n, const double* fac, int* step) {
  if (n <= 0) return NB_OK;
  if (phase == 0) {
    if (dtype == NB_F32)
      kdk_pre_kernel<float><<<stream_grid(n), 256, 0, c->stream>>>((float*)x, (float*)y, (float*)z, (float*)vx, (float*)vy, (float*)vz,
                                                                   (const float*)ax, (const float*)ay, (const float*)az, n, fac, step);
    else
      kdk_pre_kernel<double><<<stream_grid(n), 256, 0, c->stream>>>((double*)x, (double*)y, (double*)z, (double*)vx, (double*)vy, (double*)vz,
                                                                    (const double*)ax, (const double*)ay, (const double*)az, n, fac, step);
  } else {
    if (dtype == NB_F32)
      kdk_post_kernel<float><<<stream_grid(n), 256, 0, c->stream>>>((float*)vx, (float*)vy, (float*)vz, (const float*)ax, (const float*)ay,
                                                                    (const float*)az, n, fac, step);
    else
      kdk_post_kernel<double><<<stream_grid(n), 256, 0, c->stream>>>((double*)vx, (double*)vy, (double*)vz, (const double*)ax, (const double*)ay,
                                                                     (const double*)az, n, fac, step);
    kdk_advance_kernel<<<1, 64, 0, c->stream>>>(step);
  }
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

int nb_launch_kinetic(nb_ctx* c, int dtype, const void* vx, const void* vy, const void* vz, const void* m,
                      int64_t n, double* k_out) {
  const int grid = stream_grid(n);
  int rc = nb_ensure_work(c, (size_t)grid * sizeof(double));
  if (rc) return rc;
  double* part = reinterpret_cast<double*>(c->work);
  if (dtype == NB_F32)
    kinetic_partial_kernel<float><<<grid, 256, 0, c->stream>>>((const float*)vx, (const float*)vy, (const float*)vz,
                                                               (const float*)m, n, part);
  else
    kinetic_partial_kernel<double><<<grid, 256, 0, c->stream>>>((const double*)vx, (const double*)vy,
                                                                (const double*)vz, (const double*)m, n, part);
  NB_HIP(c, hipGetLastError());
  return nb_launch_final_sum(c, part, grid, 1.0, k_out, 0);
}

int nb_launch_pot_energy(nb_ctx* c, int dtype, const double* phi, const void* m, int64_t n, double eps2, int uniform,
                         double* w_out) {
  const int grid = stream_grid(n);
  int rc = nb_ensure_work(c, (size_t)grid * sizeof(double));
  if (rc) return rc;
  double* part = reinterpret_cast<double*>(c->work);
  if (dtype == NB_F32)
    pot_energy_partial_kernel<float><<<grid, 256, 0, c->stream>>>(phi, (const float*)m, n, (double)(float)eps2, uniform, part);
  else
    pot_energy_partial_kernel<double><<<grid, 256, 0, c->stream>>>(phi, (const double*)m, n, eps2, uniform, part);
  NB_HIP(c, hipGetLastError());
  return nb_launch_final_sum(c, part, grid, 1.0, w_out, 0);
}

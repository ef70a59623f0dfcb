// Periodic (Ewald) direct summation in a cubic box, fp64 and fp32 — the force law of CONCEPT's `pp` gravity, which is what the
// reference's one direct-summation run used (run evidence PPPMP3M/PP/powerspec/powerspec_a=1.00).
//
// Textbook Ewald split with a Gaussian of width 1/(2 alpha):
//   a_i = G sum_j m_j d_ij [ (r^2+eps^2)^{-3/2} - ( erf(alpha r) - (2 alpha r/sqrt(pi)) e^{-alpha^2 r^2} ) / r^3 ]     (minimum image)
//       - (4 pi G / V) sum_{k != 0} (k/k^2) e^{-k^2/(4 alpha^2)} [ S_c(k) sin(k.x_i) - S_s(k) cos(k.x_i) ]
//   S_c(k) = sum_j m_j cos(k.x_j),  S_s(k) = sum_j m_j sin(k.x_j),  k = 2 pi n / L,  |n| <= kmax
// The k = 0 term is dropped = the uniform background is subtracted, as in a cosmological box. With alpha L = 6.2 the real-space
// term is < 2e-5 of the Newtonian force at the minimum-image boundary r = L/2 and kmax = 7 leaves e^{-k^2/4 alpha^2} < 1e-5.
//   ewald_real_kernel   all-pairs, LDS-tiled like accel_f64 (sources as (x,y,z,m) in LDS, wave-uniform reads), minimum image
//                       by x - L rint(x/L); bounded by the fp64 VALU rate (+ erf/exp).
//   ewald_real_f32_kernel  the same with the long-range complement from an LDS table (fp32 contexts).
//   With alpha L >= 9.9 (and n >= 4096) the real-space terms have died out within a third of the box and the real-space part is
//   the cell-list pair sum of mesh.hip (nb_launch_short_range) instead; alpha L = 16.5, kmax = 19 is ~4x faster than 6.2 / 7.
//   ewald_sk_rows_kernel / ewald_recip_rows_kernel  reciprocal sum over ROWS of k-vectors (fixed n_x, n_y) by complex rotation:
//                       one sincos per row and particle, 4 fma per k-vector; fixed-order reductions.
//   ewald_sk_kernel / ewald_recip_kernel  the plain version, one fp64 sincos per (particle, k) (NB_EWALD_ROWS=0).
#include "nb_common.h"
#include "../../include/nbody_amd.h"
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

constexpr int BLOCK = 256;
struct d4 { double x, y, z, w; };

__global__ void __launch_bounds__(BLOCK)
ewald_real_kernel(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                  const double* __restrict__ m, int n, double L, double alpha, double eps2, double G,
                  double* __restrict__ ax, double* __restrict__ ay, double* __restrict__ az) {
  __shared__ d4 tile[BLOCK];
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  const bool live = i < n;
  const int ic = live ? i : n - 1;
  const double px = x[ic], py = y[ic], pz = z[ic];
  const double invL = 1.0 / L;
  const double two_a_sqrtpi = 2.0 * alpha * 0.5641895835477563;   // 2 alpha / sqrt(pi)
  double fx = 0.0, fy = 0.0, fz = 0.0;
  for (int j0 = 0; j0 < n; j0 += BLOCK) {
    __syncthreads();
    const int j = j0 + threadIdx.x;
    const bool ok = j < n;
    const int jc = ok ? j : n - 1;
    tile[threadIdx.x] = d4{x[jc], y[jc], z[jc], ok ? m[jc] : 0.0};
    __syncthreads();
    const int cnt = min(BLOCK, n - j0);
    for (int e = 0; e < cnt; ++e) {
      const d4 s = tile[e];
      double dx = s.x - px, dy = s.y - py, dz = s.z - pz;
      dx -= L * rint(dx * invL); dy -= L * rint(dy * invL); dz -= L * rint(dz * invL);   // minimum image
      const double r2 = dx * dx + dy * dy + dz * dz;
      if (r2 > 0.0) {
        const double r = sqrt(r2);
        const double u = alpha * r;
        const double rinv = 1.0 / r;
        const double r2s = r2 + eps2;
        const double soft = 1.0 / (r2s * sqrt(r2s));
        const double lng = (erf(u) - two_a_sqrtpi * r * exp(-u * u)) * rinv * rinv * rinv;
        const double f = s.w * (soft - lng);
        fx = fma(f, dx, fx); fy = fma(f, dy, fy); fz = fma(f, dz, fz);
      }
    }
  }
  if (live) { ax[i] = G * fx; ay[i] = G * fy; az[i] = G * fz; }
}

// fp32 version of the real-space sum. The long-range complement T(r^2) = (erf(a r) - 2 a r/sqrt(pi) e^{-a^2 r^2}) / r^3 is read
// from a table uniform in r^2 over [0, 3 L^2/4] (the largest minimum-image separation), 8192 intervals, linear interpolation
// from LDS: 3.5e-7 of the Newtonian pair force — replaces erfc/exp/rcp arithmetic (~45 full-rate VALU + 4 transcendentals per
// pair) by ~27 + 1. fp32 partial sums flushed to fp64 per source tile as in accel_f32.hip.
constexpr int ETAB = 8192;

__global__ void __launch_bounds__(BLOCK)
ewald_table_kernel(float* __restrict__ table, double alpha, double smax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > ETAB + 1) return;
  const double s = smax * (double)i / ETAB;
  double v = 4.0 * alpha * alpha * alpha / (3.0 * 1.7724538509055159);
  if (s > 1e-12 * smax) { const double r = sqrt(s), u = alpha * r; v = (erf(u) - 2.0 * u / 1.7724538509055159 * exp(-u * u)) / (s * r); }
  table[i] = (float)v;
}

__global__ void __launch_bounds__(BLOCK)
ewald_real_f32_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                      const float* __restrict__ m, int n, float L, float tscale, float eps2, float G,
                      const float* __restrict__ table, float* __restrict__ ax, float* __restrict__ ay, float* __restrict__ az) {
  __shared__ float4 tile[BLOCK];
  __shared__ float tab[ETAB + 2];
  for (int t = threadIdx.x; t < ETAB + 2; t += BLOCK) tab[t] = table[t];      // first use is behind the tile barrier
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  const bool live = i < n;
  const int ic = live ? i : n - 1;
  const float px = x[ic], py = y[ic], pz = z[ic];
  const float invL = 1.0f / L;
  double tx = 0.0, ty = 0.0, tz = 0.0;
  for (int j0 = 0; j0 < n; j0 += BLOCK) {
    __syncthreads();
    const int j = j0 + threadIdx.x;
    const bool ok = j < n;
    const int jc = ok ? j : n - 1;
    tile[threadIdx.x] = make_float4(x[jc], y[jc], z[jc], ok ? m[jc] : 0.0f);
    __syncthreads();
    const int cnt = min(BLOCK, n - j0);
    float fx = 0.f, fy = 0.f, fz = 0.f;
#pragma unroll 4
    for (int e = 0; e < cnt; ++e) {
      const float4 s = tile[e];
      float dx = s.x - px, dy = s.y - py, dz = s.z - pz;
      dx = fmaf(-L, rintf(dx * invL), dx); dy = fmaf(-L, rintf(dy * invL), dy); dz = fmaf(-L, rintf(dz * invL), dz);
      const float r2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
      const float t = r2 * tscale;
      const int it = min((int)t, ETAB);
      const float fr = t - (float)it;
      const float t0 = tab[it], t1 = tab[it + 1];
      const float sinv = __builtin_amdgcn_rsqf(r2 + eps2);
      const float soft = r2 > 0.f ? sinv * sinv * sinv : 0.f;                 // self / coincident: no contribution (d = 0)
      const float f = s.w * (soft - fmaf(fr, t1 - t0, t0));
      fx = fmaf(f, dx, fx); fy = fmaf(f, dy, fy); fz = fmaf(f, dz, fz);
    }
    tx += (double)fx; ty += (double)fy; tz += (double)fz;
  }
  if (live) { ax[i] = (float)((double)G * tx); ay[i] = (float)((double)G * ty); az[i] = (float)((double)G * tz); }
}

// kvec[4*q + (0,1,2)] = k vector, [3] = weight = 2 * (4 pi / V) e^{-k^2/4 alpha^2} / k^2   (2 = the mirrored half space)
template <typename T>
__global__ void __launch_bounds__(BLOCK)
ewald_sk_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z,
                const T* __restrict__ m, int n, const double* __restrict__ kvec, double* __restrict__ sk) {
  const int q = blockIdx.x;
  const double kx = kvec[4 * q], ky = kvec[4 * q + 1], kz = kvec[4 * q + 2];
  double sc = 0.0, ss = 0.0;
  for (int j = threadIdx.x; j < n; j += BLOCK) {
    double s, c;
    sincos(kx * (double)x[j] + ky * (double)y[j] + kz * (double)z[j], &s, &c);
    sc += (double)m[j] * c; ss += (double)m[j] * s;
  }
  __shared__ double rc[BLOCK], rs[BLOCK];
  rc[threadIdx.x] = sc; rs[threadIdx.x] = ss;
  __syncthreads();
  for (int w = BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { rc[threadIdx.x] += rc[threadIdx.x + w]; rs[threadIdx.x] += rs[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { sk[2 * q] = rc[0]; sk[2 * q + 1] = rs[0]; }
}

template <typename T>
__global__ void __launch_bounds__(BLOCK)
ewald_recip_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, int n,
                   const double* __restrict__ kvec, const double* __restrict__ sk, int nk, double G,
                   T* __restrict__ ax, T* __restrict__ ay, T* __restrict__ az) {
  extern __shared__ double lds[];          // [6 * chunk]: kx ky kz w sc ss
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  const bool live = i < n;
  const int ic = live ? i : n - 1;
  const double px = (double)x[ic], py = (double)y[ic], pz = (double)z[ic];
  double fx = 0.0, fy = 0.0, fz = 0.0;
  constexpr int CH = 256;
  for (int q0 = 0; q0 < nk; q0 += CH) {
    __syncthreads();
    const int q = q0 + threadIdx.x;
    if (threadIdx.x < CH && q < nk) {
      lds[6 * threadIdx.x + 0] = kvec[4 * q]; lds[6 * threadIdx.x + 1] = kvec[4 * q + 1]; lds[6 * threadIdx.x + 2] = kvec[4 * q + 2];
      lds[6 * threadIdx.x + 3] = kvec[4 * q + 3]; lds[6 * threadIdx.x + 4] = sk[2 * q]; lds[6 * threadIdx.x + 5] = sk[2 * q + 1];
    }
    __syncthreads();
    const int cnt = min(CH, nk - q0);
    for (int e = 0; e < cnt; ++e) {
      const double kx = lds[6 * e], ky = lds[6 * e + 1], kz = lds[6 * e + 2], w = lds[6 * e + 3];
      double s, c;
      sincos(kx * px + ky * py + kz * pz, &s, &c);
      const double t = -w * (lds[6 * e + 4] * s - lds[6 * e + 5] * c);
      fx = fma(t, kx, fx); fy = fma(t, ky, fy); fz = fma(t, kz, fz);
    }
  }
  if (live) { ax[i] = (T)((double)ax[i] + G * fx); ay[i] = (T)((double)ay[i] + G * fy); az[i] = (T)((double)az[i] + G * fz); }
}

// ---- reciprocal sum by rows of k-vectors -------------------------------------------------------------------------
// The half-space table is ordered (a, b, c) with c innermost, so the k-vectors of one (a, b) form a contiguous ROW whose members
// differ by the constant step dk = 2 pi / L along z: e^{i k.x} of the whole row follows from ONE sincos of the row's first phase
// and ONE sincos of dk z per particle by complex rotation (4 fma per k-vector instead of an fp64 sincos).
constexpr int ROWC = 36;        // k-vectors per pass of a row: a full row of kmax = 17 (fp32 contexts' default split) has 35 — one pass

// sin and cos of a phase k.x computed in fp64. fp64 contexts: sincos(). fp32 contexts (forces good to ~1e-5 by contract): the phase is
// reduced to (-1/2, 1/2] revolutions in fp64 and handed to the hardware's v_sin_f32 / v_cos_f32 (argument in revolutions, absolute error
// ~5e-7) — a dozen instructions instead of the ~150 of the fp64 sincos, which was half of the work of a (row, particle) pair.
template <typename T>
__device__ inline void phase_sincos(double phase, double* s, double* c) {
  if constexpr (sizeof(T) == 8) {
    sincos(phase, s, c);
  } else {
    const double u = phase * 0.15915494309189535;          // revolutions
    const float r = (float)(u - rint(u));
    *s = (double)__builtin_amdgcn_sinf(r); *c = (double)__builtin_amdgcn_cosf(r);
  }
}

// (cos, sin)(dk z_j): the step of the complex rotation along a row of k-vectors
template <typename T>
__global__ void __launch_bounds__(BLOCK)
ewald_zrot_kernel(const T* __restrict__ z, int n, double dk, double2* __restrict__ zrot) {
  const int j = blockIdx.x * BLOCK + threadIdx.x;
  if (j >= n) return;
  double ds, dc;
  sincos(dk * (double)z[j], &ds, &dc);
  zrot[j] = double2{dc, ds};
}

template <typename T>
__global__ void __launch_bounds__(BLOCK)
ewald_sk_rows_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, const T* __restrict__ m, int n,
                     const double* __restrict__ kvec, const int2* __restrict__ rows, double dk, const double2* __restrict__ zrot,
                     double* __restrict__ sk) {
  const int2 row = rows[blockIdx.x];
  __shared__ double red[BLOCK / 64][2 * ROWC];
  for (int c0 = 0; c0 < row.y; c0 += ROWC) {
    const int q0 = row.x + c0, cnt = min(ROWC, row.y - c0);
    const double kx = kvec[4 * q0], ky = kvec[4 * q0 + 1], kz = kvec[4 * q0 + 2];
    // fp64 contexts rotate and sum in fp64. fp32 contexts (forces good to ~1e-5 by contract) rotate and sum each thread's n / BLOCK
    // particles in fp32 — 35 rotations lose ~2e-6, a thread's sum of ~1000 terms of either sign ~2e-6 of its size — and only the
    // reduction over threads below is fp64: the inner loop runs at the fp32 rate, twice that of fp64
    typedef typename std::conditional<sizeof(T) == 8, double, float>::type R;
    // fp32 contexts: a thread's fp32 sums live for SK_FLUSH of its particles only, then go through the fp64 reduction below — the
    // rounding error of S(k), a small residual of large cancelling terms, must not grow with n / BLOCK (128^3 particles: 8 flushes)
    constexpr int SK_FLUSH = 1024;
    const int chunk = sizeof(T) == 8 ? n : SK_FLUSH * BLOCK;
    if (threadIdx.x < 2 * ROWC) for (int w = 0; w < BLOCK / 64; ++w) red[w][threadIdx.x] = 0.0;
    __syncthreads();
    for (int jb = 0; jb < n; jb += chunk) {
    const int je = min(n, jb + chunk);
    R sc[ROWC], ss[ROWC];
#pragma unroll
    for (int e = 0; e < ROWC; ++e) { sc[e] = 0; ss[e] = 0; }
    for (int j = jb + threadIdx.x; j < je; j += BLOCK) {
      const double pz = (double)z[j];
      const R mj = (R)m[j];
      double s0, c0d;
      phase_sincos<T>(kx * (double)x[j] + ky * (double)y[j] + kz * pz, &s0, &c0d);
      const double2 rot = zrot[j];                       // (cos, sin)(dk z_j): the same for every row, computed once per evaluation
      const R dc = (R)rot.x, ds = (R)rot.y;
      R s = (R)s0, c = (R)c0d;
#pragma unroll
      for (int e = 0; e < ROWC; ++e) {
        sc[e] = fma(mj, c, sc[e]); ss[e] = fma(mj, s, ss[e]);
        const R cn = c * dc - s * ds;
        s = fma(s, dc, c * ds); c = cn;
      }
    }
    // fixed-order reduction: lanes of a wave by shuffles, then the waves through LDS
#pragma unroll
    for (int e = 0; e < ROWC; ++e) {
      double a = (double)sc[e], b = (double)ss[e];
      for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
      if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][2 * e] += a; red[threadIdx.x >> 6][2 * e + 1] += b; }   // own slot of this wave: fixed order
    }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * cnt) {
      double t = 0.0;
      for (int w = 0; w < BLOCK / 64; ++w) t += red[w][threadIdx.x];
      sk[2 * q0 + threadIdx.x] = t;
    }
    __syncthreads();
  }
}

template <typename T>
__global__ void __launch_bounds__(BLOCK)
ewald_recip_rows_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, int n,
                        const double* __restrict__ kvec, const double* __restrict__ sk, const int2* __restrict__ rows, int nrows,
                        double dk, double G, T* __restrict__ ax, T* __restrict__ ay, T* __restrict__ az) {
  constexpr int MAXROW = 64;                               // >= 2 kmax + 1 for kmax <= 31
  typedef typename std::conditional<sizeof(T) == 8, double, float>::type R;   // a row's terms in fp32 for fp32 contexts (as in ewald_sk_rows_kernel);
  __shared__ R w[MAXROW], sc[MAXROW], ss[MAXROW];                               // the sum over rows stays fp64
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  const bool live = i < n;
  const int ic = live ? i : n - 1;
  const double px = (double)x[ic], py = (double)y[ic], pz = (double)z[ic];
  double dsd, dcd;
  sincos(dk * pz, &dsd, &dcd);
  const R ds = (R)dsd, dc = (R)dcd, dkr = (R)dk;
  double fx = 0.0, fy = 0.0, fz = 0.0;
  for (int r = 0; r < nrows; ++r) {
    const int2 row = rows[r];
    __syncthreads();
    if ((int)threadIdx.x < row.y) {
      const int q = row.x + threadIdx.x;
      w[threadIdx.x] = (R)kvec[4 * q + 3]; sc[threadIdx.x] = (R)sk[2 * q]; ss[threadIdx.x] = (R)sk[2 * q + 1];
    }
    __syncthreads();
    const double kx = kvec[4 * row.x], ky = kvec[4 * row.x + 1], kz0 = kvec[4 * row.x + 2];
    double s0, c0;
    phase_sincos<T>(kx * px + ky * py + kz0 * pz, &s0, &c0);
    R s = (R)s0, c = (R)c0, t0 = 0, tz = 0, kz = (R)kz0;
    for (int e = 0; e < row.y; ++e) {
      const R t = -w[e] * (sc[e] * s - ss[e] * c);
      t0 += t; tz = fma(t, kz, tz);
      const R cn = c * dc - s * ds;
      s = fma(s, dc, c * ds); c = cn;
      kz += dkr;
    }
    fx = fma((double)t0, kx, fx); fy = fma((double)t0, ky, fy); fz += (double)tz;
  }
  if (live) { ax[i] = (T)((double)ax[i] + G * fx); ay[i] = (T)((double)ay[i] + G * fy); az[i] = (T)((double)az[i] + G * fz); }
}

template <typename T>
__global__ void __launch_bounds__(BLOCK)
wrap_kernel(T* __restrict__ x, T* __restrict__ y, T* __restrict__ z, int64_t n, T L) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    T v[3] = {x[i], y[i], z[i]};
#pragma unroll
    for (int c = 0; c < 3; ++c) { v[c] -= L * floor(v[c] / L); if (v[c] >= L) v[c] -= L; if (v[c] < (T)0) v[c] = (T)0; }
    x[i] = v[0]; y[i] = v[1]; z[i] = v[2];
  }
}

}  // namespace

struct nb_ewald_state {
  double* kvec = nullptr;   // [4*nk]
  double* sk = nullptr;     // [2*nk]
  int nk = 0;
  double L = 0, alphaL = 0; int kmax = 0;
  int2* rows = nullptr;     // (first k index, count) of every (a, b) row of the table
  int nrows = 0;
  float* table = nullptr;   // T(r^2) of the fp32 real-space kernel
  double table_L = 0, table_alphaL = 0;
  double2* zrot = nullptr;  // [n] (cos, sin)(dk z_j)
  int64_t zrot_cap = 0;
};

void nb_ewald_release(nb_ctx* c) {
  if (c->ewald) {
    if (c->ewald->kvec) (void)hipFree(c->ewald->kvec);
    if (c->ewald->sk) (void)hipFree(c->ewald->sk);
    if (c->ewald->table) (void)hipFree(c->ewald->table);
    if (c->ewald->rows) (void)hipFree(c->ewald->rows);
    if (c->ewald->zrot) (void)hipFree(c->ewald->zrot);
    delete c->ewald; c->ewald = nullptr;
  }
}

int nb_launch_wrap(nb_ctx* c, int dtype, void* x, void* y, void* z, int64_t n, double L) {
  if (n <= 0) return NB_OK;
  int64_t b = (n + BLOCK - 1) / BLOCK;
  const int g = (int)(b > 2048 ? 2048 : b);
  if (dtype == NB_F32) wrap_kernel<float><<<g, BLOCK, 0, c->stream>>>((float*)x, (float*)y, (float*)z, n, (float)L);
  else wrap_kernel<double><<<g, BLOCK, 0, c->stream>>>((double*)x, (double*)y, (double*)z, n, L);
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

// structure factors + reciprocal force (ADDS to acc); rows of k-vectors by rotation unless NB_EWALD_ROWS=0
template <typename T>
static int launch_recip(nb_ctx* c, nb_ewald_state& e, const T* x, const T* y, const T* z, const T* m, int64_t n, double L, double G,
                        T* ax, T* ay, T* az) {
  static const bool by_rows = !(getenv("NB_EWALD_ROWS") && atoi(getenv("NB_EWALD_ROWS")) == 0);
  const int nb = (int)((n + BLOCK - 1) / BLOCK);
  const double dk = 2.0 * 3.14159265358979323846 / L;
  if (by_rows) {
    if (e.zrot_cap < n) {
      NB_HIP(c, hipStreamSynchronize(c->stream));
      if (e.zrot) { (void)hipFree(e.zrot); e.zrot = nullptr; e.zrot_cap = 0; }
      NB_HIP(c, hipMalloc((void**)&e.zrot, (size_t)n * sizeof(double2)));
      e.zrot_cap = n;
    }
    ewald_zrot_kernel<T><<<nb, BLOCK, 0, c->stream>>>(z, (int)n, dk, e.zrot);
    NB_HIP(c, hipGetLastError());
    ewald_sk_rows_kernel<T><<<e.nrows, BLOCK, 0, c->stream>>>(x, y, z, m, (int)n, e.kvec, e.rows, dk, e.zrot, e.sk);
    NB_HIP(c, hipGetLastError());
    ewald_recip_rows_kernel<T><<<nb, BLOCK, 0, c->stream>>>(x, y, z, (int)n, e.kvec, e.sk, e.rows, e.nrows, dk, G, ax, ay, az);
  } else {
    ewald_sk_kernel<T><<<e.nk, BLOCK, 0, c->stream>>>(x, y, z, m, (int)n, e.kvec, e.sk);
    NB_HIP(c, hipGetLastError());
    ewald_recip_kernel<T><<<nb, BLOCK, 6 * 256 * sizeof(double), c->stream>>>(x, y, z, (int)n, e.kvec, e.sk, e.nk, G, ax, ay, az);
  }
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

int nb_launch_accel_ewald(nb_ctx* c, int dtype, const void* x, const void* y, const void* z, const void* m, int64_t n,
                          double L, double alphaL, int kmax, double eps2, double G, void* ax, void* ay, void* az) {
  if (n <= 0) return NB_OK;
  if (n >= (1ll << 31)) return nb_fail(c, NB_ERR_ARG, "ewald: n must be < 2^31");
  if (!(L > 0.0) || !(alphaL > 0.0) || kmax < 1 || kmax > 31) return nb_fail(c, NB_ERR_ARG, "ewald: bad box / alpha / kmax");
  if (!c->ewald) c->ewald = new nb_ewald_state();
  nb_ewald_state& e = *c->ewald;
  const double alpha = alphaL / L;
  if (e.L != L || e.alphaL != alphaL || e.kmax != kmax) {
    // k-vector table: half space (first non-zero component positive), |n| <= kmax
    std::vector<double> kv;
    std::vector<int> rows;                                   // (first k index, count) per (a, b): members contiguous, c ascending
    const double PI = 3.14159265358979323846, V = L * L * L;
    for (int a = -kmax; a <= kmax; ++a)
      for (int b = -kmax; b <= kmax; ++b)
        for (int d = -kmax; d <= kmax; ++d) {
          const int n2 = a * a + b * b + d * d;
          if (n2 == 0 || n2 > kmax * kmax) continue;
          if (!(a > 0 || (a == 0 && (b > 0 || (b == 0 && d > 0))))) continue;
          const int q = (int)(kv.size() / 4);
          if (rows.empty() || rows[rows.size() - 2] + rows.back() != q || kv[4 * (q - 1)] != 2 * PI * a / L || kv[4 * (q - 1) + 1] != 2 * PI * b / L) {
            rows.push_back(q); rows.push_back(0);
          }
          rows.back() += 1;
          const double kx = 2 * PI * a / L, ky = 2 * PI * b / L, kz = 2 * PI * d / L, k2 = kx * kx + ky * ky + kz * kz;
          kv.push_back(kx); kv.push_back(ky); kv.push_back(kz);
          kv.push_back(2.0 * (4.0 * PI / V) * std::exp(-k2 / (4.0 * alpha * alpha)) / k2);
        }
    NB_HIP(c, hipStreamSynchronize(c->stream));
    if (e.kvec) { NB_HIP(c, hipFree(e.kvec)); e.kvec = nullptr; }
    if (e.sk) { NB_HIP(c, hipFree(e.sk)); e.sk = nullptr; }
    if (e.rows) { NB_HIP(c, hipFree(e.rows)); e.rows = nullptr; }
    e.nrows = (int)(rows.size() / 2);
    NB_HIP(c, hipMalloc((void**)&e.rows, rows.size() * sizeof(int)));
    NB_HIP(c, hipMemcpy(e.rows, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice));
    e.nk = (int)(kv.size() / 4);
    NB_HIP(c, hipMalloc((void**)&e.kvec, kv.size() * sizeof(double)));
    NB_HIP(c, hipMalloc((void**)&e.sk, 2 * (size_t)e.nk * sizeof(double)));
    NB_HIP(c, hipMemcpy(e.kvec, kv.data(), kv.size() * sizeof(double), hipMemcpyHostToDevice));
    e.L = L; e.alphaL = alphaL; e.kmax = kmax;
  }
  const int nb = (int)((n + BLOCK - 1) / BLOCK);
  // A splitting parameter large enough that the real-space terms have died out (alpha r = 3.3: 7e-5 of the pair force) within a
  // third of the box turns the all-pairs minimum-image sum into the cell-list pair sum of mesh.hip: 27 of nc^3 cells per sink.
  const double rcut = 3.3 / alpha;
  if (rcut <= L / 3.0 && n >= 4096) {
    const size_t eb = dtype == NB_F32 ? sizeof(float) : sizeof(double);
    NB_HIP(c, hipMemsetAsync(ax, 0, (size_t)n * eb, c->stream));
    NB_HIP(c, hipMemsetAsync(ay, 0, (size_t)n * eb, c->stream));
    NB_HIP(c, hipMemsetAsync(az, 0, (size_t)n * eb, c->stream));
    int rc = nb_launch_short_range(c, dtype, x, y, z, m, n, L, alpha, rcut, eps2, G, ax, ay, az);
    if (rc) return rc;
    if (dtype == NB_F32)
      return launch_recip<float>(c, e, (const float*)x, (const float*)y, (const float*)z, (const float*)m, n, L, G, (float*)ax, (float*)ay, (float*)az);
    return launch_recip<double>(c, e, (const double*)x, (const double*)y, (const double*)z, (const double*)m, n, L, G, (double*)ax, (double*)ay,
                                (double*)az);
  }
  if (dtype == NB_F32) {
    const float *fx = (const float*)x, *fy = (const float*)y, *fz = (const float*)z, *fm = (const float*)m;
    const double smax = 0.75 * L * L * (1.0 + 1e-6);           // largest minimum-image r^2
    if (!e.table || e.table_L != L || e.table_alphaL != alphaL) {
      if (!e.table) NB_HIP(c, hipMalloc((void**)&e.table, (ETAB + 2) * sizeof(float)));
      ewald_table_kernel<<<(ETAB + 2 + BLOCK - 1) / BLOCK, BLOCK, 0, c->stream>>>(e.table, alpha, smax);
      NB_HIP(c, hipGetLastError());
      e.table_L = L; e.table_alphaL = alphaL;
    }
    ewald_real_f32_kernel<<<nb, BLOCK, 0, c->stream>>>(fx, fy, fz, fm, (int)n, (float)L, (float)(ETAB / smax), (float)eps2, (float)G,
                                                       e.table, (float*)ax, (float*)ay, (float*)az);
    NB_HIP(c, hipGetLastError());
    return launch_recip<float>(c, e, fx, fy, fz, fm, n, L, G, (float*)ax, (float*)ay, (float*)az);
  } else {
    const double *dx = (const double*)x, *dy = (const double*)y, *dz = (const double*)z, *dm = (const double*)m;
    ewald_real_kernel<<<nb, BLOCK, 0, c->stream>>>(dx, dy, dz, dm, (int)n, L, alpha, eps2, G, (double*)ax, (double*)ay, (double*)az);
    NB_HIP(c, hipGetLastError());
    return launch_recip<double>(c, e, dx, dy, dz, dm, n, L, G, (double*)ax, (double*)ay, (double*)az);
  }
}

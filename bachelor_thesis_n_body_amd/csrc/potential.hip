// Softened pair potential (SURVEY.md §8a row ⊕3):  W = sum_i m_i sum_{j != i} m_j / sqrt(r_ij^2 + eps^2)
// (the caller applies -G/2). Same LDS-tiled all-pairs sweep as the acceleration kernel, 6 VALU + 1 rsq +
// 2 (index exclusion) per interaction; per-sink sums are flushed to fp64 at every tile boundary and the
// block results are reduced in a fixed order (bit-reproducible).
// The j == i exclusion is by GLOBAL particle index so that the same kernel serves the sink-decomposed
// multi-GPU path, where sources arrive as chunks with an index offset. A set on itself takes the upper triangle of tiles only.
#include "nb_common.h"
#include "../../include/nbody_amd.h"
#include <type_traits>

namespace {

constexpr int BLOCK = 256;
constexpr int KS = 2;

template <typename T> struct vec4 { T x, y, z, w; };

__device__ inline float nb_rsqrt(float v) { return __builtin_amdgcn_rsqf(v); }
__device__ inline double nb_rsqrt(double v) {
  // v_rsq_f64 seed (~2^-26 relative) + one Halley step (cubic): full fp64 accuracy
  double y = __builtin_amdgcn_rsq(v);
  const double h = fma(-v * y, y, 1.0);
  return y * fma(h, fma(0.375, h, 0.5), 1.0);
}

template <typename T, int TILE>
__global__ void __launch_bounds__(BLOCK)
potential_kernel(const T* __restrict__ xi, const T* __restrict__ yi, const T* __restrict__ zi,
                 const T* __restrict__ mi, int n_snk, long long snk_g0,
                 const T* __restrict__ xs, const T* __restrict__ ys, const T* __restrict__ zs,
                 const T* __restrict__ ms, int n_src, long long src_g0, int src_per_split, T eps2,
                 double* __restrict__ part, int tri) {
  // tri: a set on itself (same arrays, same global offset). m_i m_j / r is symmetric, so only the source tiles on and above the sink
  // block's own tile are walked (a sink block is exactly one tile wide): the tiles above count twice, the own tile — which holds
  // both (i, j) and (j, i) — once. Half the pair work; the sums differ from the full sweep in rounding order only.
  constexpr int LPT = TILE / BLOCK;
  __shared__ vec4<T> tile[TILE];
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * (BLOCK * KS) + tid;
  T px[KS], py[KS], pz[KS];
  long long gi[KS];
  double phi[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    int i = i0 + k * BLOCK;
    i = i < n_snk ? i : n_snk - 1;
    px[k] = xi[i]; py[k] = yi[i]; pz[k] = zi[i];
    gi[k] = snk_g0 + i;
    phi[k] = 0.0;
  }
  const int jbeg = blockIdx.y * src_per_split;
  const int jend = min(n_src, jbeg + src_per_split);
  static_assert(TILE == BLOCK * KS, "the triangular sweep relies on sink block == one source tile");
  for (int jt = jbeg; jt < jend; jt += TILE) {
    const int own = tri ? jt / TILE - (int)blockIdx.x : 1;      // < 0: below the diagonal, 0: the own tile, > 0: above
    if (tri && own < 0) continue;
    const double wgt = (tri && own > 0) ? 2.0 : 1.0;
    __syncthreads();
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int j = jt + l * BLOCK + tid;
      const bool ok = j < jend;
      const int jc = ok ? j : jend - 1;
      tile[l * BLOCK + tid] = vec4<T>{xs[jc], ys[jc], zs[jc], ok ? ms[jc] : (T)0};
    }
    __syncthreads();
    T acc[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) acc[k] = (T)0;
    // local index of each sink inside this tile (or -1): wave-uniform j compared per lane
    int self[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const long long d = gi[k] - (src_g0 + jt);
      self[k] = (d >= 0 && d < TILE) ? (int)d : -1;
    }
#pragma unroll 4
    for (int j = 0; j < TILE; ++j) {
      const vec4<T> s = tile[j];
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const T dx = s.x - px[k], dy = s.y - py[k], dz = s.z - pz[k];
        const T r2 = dx * dx + (dy * dy + (dz * dz + eps2));
        T rinv = nb_rsqrt(r2);
        rinv = (j == self[k] || !(r2 > (T)0)) ? (T)0 : rinv;
        acc[k] += s.w * rinv;
      }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) phi[k] += wgt * (double)acc[k];
  }
  // block reduction of sum_k m_i * phi_i
  double w = 0.0;
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int i = i0 + k * BLOCK;
    if (i < n_snk) w += (double)mi[i] * phi[k];
  }
  __shared__ double red[BLOCK];
  __syncthreads();
  red[tid] = w;
  __syncthreads();
  for (int s = BLOCK / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

// ---- a set on itself, large n: the same upper-triangle sweep with KS sinks per lane (fp32: packed pairs, as accel_f32.hip) -------------------
// Sink block b = 256 * KS consecutive particles = SET / TILE consecutive source tiles (its "own" tiles). Tiles below them are skipped, the own tiles
// count once with the i == j term excluded by index, the tiles above count twice and need no exclusion at all: 7 packed ops + 2 v_rsq_f32 per two
// pairs in the fp32 loop (the general kernel above: 9 plain ops + rsq per pair at 2 sinks per lane).
typedef float v2f __attribute__((ext_vector_type(2)));

template <typename T, int KS, int TILE, bool GUARD>
__global__ void __launch_bounds__(BLOCK)
potential_self_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, const T* __restrict__ m, int n,
                      int tiles_per_split, T eps2, double* __restrict__ part) {
  static_assert(KS % 2 == 0 && (BLOCK * KS) % TILE == 0 && TILE % BLOCK == 0, "shape");
  constexpr int SET = BLOCK * KS, OWN = SET / TILE, LPT = TILE / BLOCK, KP = KS / 2;
  __shared__ vec4<T> tile[TILE];
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * SET + tid;
  T px[KS], py[KS], pz[KS];
  double phi[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    int i = i0 + k * BLOCK;
    i = i < n ? i : n - 1;
    px[k] = x[i]; py[k] = y[i]; pz[k] = z[i];
    phi[k] = 0.0;
  }
  const int own_lo = blockIdx.x * OWN, own_hi = own_lo + OWN;
  const int ntiles_total = (n + TILE - 1) / TILE;
  const int t_beg = max(own_lo, (int)blockIdx.y * tiles_per_split), t_end = min(ntiles_total, ((int)blockIdx.y + 1) * tiles_per_split);
  for (int t = t_beg; t < t_end; ++t) {
    const bool own = t < own_hi;                 // t >= own_lo by construction
    __syncthreads();
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int j = t * TILE + l * BLOCK + tid;
      const bool ok = j < n;
      const int jc = ok ? j : n - 1;
      tile[l * BLOCK + tid] = vec4<T>{x[jc], y[jc], z[jc], ok ? m[jc] : (T)0};
    }
    __syncthreads();
    int self[KS];                                 // local index of each sink inside this tile, or -1
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int d = i0 + k * BLOCK - t * TILE;
      self[k] = (own && d >= 0 && d < TILE) ? d : -1;
    }
    if constexpr (sizeof(T) == 4) {
      v2f acc[KP];
#pragma unroll
      for (int p = 0; p < KP; ++p) acc[p] = v2f{0.f, 0.f};
      const v2f e2 = v2f{(float)eps2, (float)eps2};
      auto sweep = [&](auto owntag) {
        constexpr bool OWNT = decltype(owntag)::value;
#pragma unroll 2
        for (int j = 0; j < TILE; ++j) {
          const vec4<T> s = tile[j];
#pragma unroll
          for (int p = 0; p < KP; ++p) {
            const v2f dx = v2f{(float)s.x, (float)s.x} - v2f{(float)px[2 * p], (float)px[2 * p + 1]};
            const v2f dy = v2f{(float)s.y, (float)s.y} - v2f{(float)py[2 * p], (float)py[2 * p + 1]};
            const v2f dz = v2f{(float)s.z, (float)s.z} - v2f{(float)pz[2 * p], (float)pz[2 * p + 1]};
            const v2f r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, e2)));
            v2f rinv = v2f{__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
            if (GUARD) { rinv.x = r2.x > 0.f ? rinv.x : 0.f; rinv.y = r2.y > 0.f ? rinv.y : 0.f; }     // eps = 0: coincident pairs contribute nothing
            if (OWNT) { rinv.x = j == self[2 * p] ? 0.f : rinv.x; rinv.y = j == self[2 * p + 1] ? 0.f : rinv.y; }
            acc[p] = __builtin_elementwise_fma(v2f{(float)s.w, (float)s.w}, rinv, acc[p]);
          }
        }
      };
      if (own) sweep(std::true_type{}); else sweep(std::false_type{});
      const double wgt = own ? 1.0 : 2.0;
#pragma unroll
      for (int p = 0; p < KP; ++p) { phi[2 * p] += wgt * (double)acc[p].x; phi[2 * p + 1] += wgt * (double)acc[p].y; }
    } else {
      T acc[KS];
#pragma unroll
      for (int k = 0; k < KS; ++k) acc[k] = (T)0;
#pragma unroll 2
      for (int j = 0; j < TILE; ++j) {
        const vec4<T> s = tile[j];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const T dx = s.x - px[k], dy = s.y - py[k], dz = s.z - pz[k];
          const T r2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, eps2)));
          T rinv = nb_rsqrt(r2);
          rinv = (j == self[k] || (GUARD && !(r2 > (T)0))) ? (T)0 : rinv;
          acc[k] = fma(s.w, rinv, acc[k]);
        }
      }
      const double wgt = own ? 1.0 : 2.0;
#pragma unroll
      for (int k = 0; k < KS; ++k) phi[k] += wgt * (double)acc[k];
    }
  }
  double w = 0.0;
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int i = i0 + k * BLOCK;
    if (i < n) w += (double)m[i] * phi[k];
  }
  __shared__ double red[BLOCK];
  __syncthreads();
  red[tid] = w;
  __syncthreads();
  for (int s2 = BLOCK / 2; s2 > 0; s2 >>= 1) {
    if (tid < s2) red[tid] += red[tid + s2];
    __syncthreads();
  }
  if (tid == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

template <typename T>
int launch_potential_self(nb_ctx* c, const T* x, const T* y, const T* z, const T* m, int64_t n, T eps2, double* u_out, int accumulate) {
  constexpr int TILE = 512, KS = sizeof(T) == 4 ? 8 : 4, SET = BLOCK * KS;
  const int nblk = (int)((n + SET - 1) / SET);
  const int ntiles_total = (int)((n + TILE - 1) / TILE);
  int nsplit = nblk >= 2048 ? 1 : (2048 + nblk - 1) / nblk;          // ~2048 blocks; the triangle leaves about half of them with work
  if (nsplit > ntiles_total) nsplit = ntiles_total;
  const int tps = (ntiles_total + nsplit - 1) / nsplit;
  nsplit = (ntiles_total + tps - 1) / tps;
  const size_t nparts = (size_t)nblk * nsplit;
  int rc = nb_ensure_work(c, nparts * sizeof(double));
  if (rc) return rc;
  double* part = reinterpret_cast<double*>(c->work);
  if (eps2 > (T)0) potential_self_kernel<T, KS, TILE, false><<<dim3(nblk, nsplit), BLOCK, 0, c->stream>>>(x, y, z, m, (int)n, tps, eps2, part);
  else potential_self_kernel<T, KS, TILE, true><<<dim3(nblk, nsplit), BLOCK, 0, c->stream>>>(x, y, z, m, (int)n, tps, eps2, part);
  NB_HIP(c, hipGetLastError());
  return nb_launch_final_sum(c, part, (int)nparts, 1.0, u_out, accumulate);
}

template <typename T>
int launch_potential(nb_ctx* c, const T* xi, const T* yi, const T* zi, const T* mi, int64_t n_snk, int64_t snk_g0,
                     const T* xs, const T* ys, const T* zs, const T* ms, int64_t n_src, int64_t src_g0, T eps2,
                     double* u_out, int accumulate) {
  if (n_snk >= (1ll << 31) || n_src >= (1ll << 31)) return nb_fail(c, NB_ERR_ARG, "n must be < 2^31");
  constexpr int TILE = 512;
  if (n_snk <= 0 || n_src <= 0) {
    if (!accumulate) NB_HIP(c, hipMemsetAsync(u_out, 0, sizeof(double), c->stream));
    return NB_OK;
  }
  if (xi == xs && yi == ys && zi == zs && mi == ms && n_snk == n_src && snk_g0 == src_g0 && n_snk >= 8192)
    return launch_potential_self<T>(c, xi, yi, zi, mi, n_snk, eps2, u_out, accumulate);
  const int spb = BLOCK * KS;
  const int nblk = (int)((n_snk + spb - 1) / spb);
  const int ntiles_total = (int)((n_src + TILE - 1) / TILE);
  int nsplit = nblk >= 2048 ? 1 : (2048 + nblk - 1) / nblk;
  if (nsplit > ntiles_total) nsplit = ntiles_total;
  int tps = (ntiles_total + nsplit - 1) / nsplit;
  nsplit = (ntiles_total + tps - 1) / tps;
  const long long sps_ll = (long long)tps * TILE;
  const int sps = sps_ll > 0x7fffffff ? 0x7fffffff : (int)sps_ll;
  const size_t nparts = (size_t)nblk * nsplit;
  int rc = nb_ensure_work(c, nparts * sizeof(double));
  if (rc) return rc;
  double* part = reinterpret_cast<double*>(c->work);
  potential_kernel<T, TILE><<<dim3(nblk, nsplit), BLOCK, 0, c->stream>>>(
      xi, yi, zi, mi, (int)n_snk, (long long)snk_g0, xs, ys, zs, ms, (int)n_src, (long long)src_g0, sps, eps2, part,
      (xi == xs && yi == ys && zi == zs && mi == ms && n_snk == n_src && snk_g0 == src_g0) ? 1 : 0);
  NB_HIP(c, hipGetLastError());
  return nb_launch_final_sum(c, part, (int)nparts, 1.0, u_out, accumulate);
}

}  // namespace

int nb_launch_potential_f32(nb_ctx* c, const float* xi, const float* yi, const float* zi, const float* mi,
                            int64_t n_snk, int64_t snk_g0, const float* xs, const float* ys, const float* zs,
                            const float* ms, int64_t n_src, int64_t src_g0, float eps2, double* u_out,
                            int accumulate) {
  return launch_potential<float>(c, xi, yi, zi, mi, n_snk, snk_g0, xs, ys, zs, ms, n_src, src_g0, eps2, u_out, accumulate);
}

int nb_launch_potential_f64(nb_ctx* c, const double* xi, const double* yi, const double* zi, const double* mi,
                            int64_t n_snk, int64_t snk_g0, const double* xs, const double* ys, const double* zs,
                            const double* ms, int64_t n_src, int64_t src_g0, double eps2, double* u_out,
                            int accumulate) {
  return launch_potential<double>(c, xi, yi, zi, mi, n_snk, snk_g0, xs, ys, zs, ms, n_src, src_g0, eps2, u_out, accumulate);
}

// Particle-mesh (PM) and particle-particle-particle-mesh (P3M) gravity in a periodic cube — SURVEY.md §8f row 4
// ("then periodic/comoving PP (Ewald) and PM/P3M long-range via hipFFT"). These are the solvers the reference's farms
// actually select: `select_forces = {'matter': {'gravity': 'p3m'}}` with `potential_options` gridsize = cbrt(N),
// 'Fourier' differentiation, CIC interpolation, deconvolve (True, True), interlace (True, True)
// (TheFeedingPipeline/NN.param:53-82, SnapVsPK/Uncertain.param:66-95); `'gravity': 'pm'` in
// Different_Sizes/Giant.param:28-68. CONCEPT's source is not in the reference, so this follows the published method
// (Hockney & Eastwood; CONCEPT's documented defaults: split scale r_s = 1.25 cells, short-range cut-off 4.5 r_s).
//
// Force split (alpha = 1 / (2 r_s)), identical in form to the Ewald split of ewald.hip, so P3M -> Ewald as h -> 0:
//   long range :  phi_k = -4 pi G rho_k / k^2 * exp(-k^2 r_s^2)        (PM: no Gaussian factor, no short range)
//   short range:  a_i += G m_j d [ (r^2+eps^2)^{-3/2} - (erf(alpha r) - 2 alpha r/sqrt(pi) e^{-alpha^2 r^2}) / r^3 ],  r < r_cut
//
// Long-range pipeline, all grids fp64, ONE complex-to-complex forward FFT and THREE inverse FFTs per evaluation:
//   deposit    CIC mass assignment; lattice 1 -> real part, the lattice shifted by half a cell (interlacing) -> imaginary
//              part of the same complex grid. n >= 1e5: particles sorted by 8^3-cell brick, both lattices accumulated in
//              a 10^3-cell LDS tile (ds_add_f64), one global atomic per touched cell; small n: global_atomic_add_f64
//              per particle and corner                                                              [HBM/atomic bound]
//   FFT        hipFFT Z2Z forward; the two real fields are separated with the Hermitian symmetry in the k-space kernel
//   k-space    rho_k = (A_k + B_k e^{-i theta})/2, times Green's function, CIC deconvolution (twice: deposit and
//              interpolation), i k differentiation; Nyquist planes are zeroed. The six real output fields
//              (g_x, g_y, g_z on both lattices) are packed pairwise into three complex grids      [HBM bound, 16 B in, 48 B out per mode]
//   inverse    3 x hipFFT Z2Z backward
//   gather     CIC interpolation of both lattices to the particles, averaged; n >= 1e5: the six force fields of a brick
//              staged in a 48 KB LDS tile                                                           [HBM gather bound]
// Short range: particles are sorted (rocPRIM radix sort) by cell, then by Morton sub-cell; cells are contiguous in memory along z,
// so the neighbourhood of a group of sinks is a handful of contiguous segments, one per neighbouring ROW of cells, and the
// periodic image is a per-segment shift, not a per-pair minimum image.                              [VALU bound]
//   fp32, >= 16 384 particles, >= 4 per cell: cells of side >= r_cut/3, one wave per 16 (8) consecutive sinks replicated over 4 (8)
//              source slots; segments chosen from the blob's bounding box, every loaded source tested against the box and the
//              survivors compacted in LDS; 64-pair steps with a wave vote (short_range_tile_kernel)
//   otherwise (fp64, sparse cells, small systems): cells of side >= r_cut, one wave per chunk of 128 consecutive particles
//              of a row (two 64-sink sets per lane), source tiles read back from LDS as wave-uniform broadcasts (the
//              accel_f32.hip pattern), a wave vote per source and sink set (short_range_kernel)
#include "nb_common.h"
#include "../../include/nbody_amd.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <hipfft/hipfft.h>

struct nb_mesh_state {
  int ngrid = 0;
  bool have_fft = false;
  hipfftHandle fft = 0;
  hipStream_t fft_stream = nullptr;
  hipfftDoubleComplex* grid[4] = {nullptr, nullptr, nullptr, nullptr};   // density (two lattices) + 3 packed force grids
  // short range
  int64_t n_alloc = 0;
  int esz = 0;
  uint32_t *keys = nullptr, *keys_s = nullptr, *idx = nullptr, *idx_s = nullptr;
  void* sorted = nullptr;            // {x, y, z, m} per particle in cell order, positions wrapped into [0, L)
  int ncell3_alloc = 0;
  int *cell_start = nullptr, *chunk_cnt = nullptr, *chunk_off = nullptr;
  int4* chunk_info = nullptr;        // tile kernel: (row, first, last) per chunk
  int64_t chunk_info_cap = 0;
  double* stage = nullptr;           // brick deposit: one 2 x 10^3 tile per chunk, summed into the grid by brick_assemble_kernel
  int64_t stage_cap = 0;             // in chunks
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  float* table = nullptr;            // long-range complement of the pair force tabulated in r^2 (fp32 short-range kernel)
  double table_alpha = 0.0, table_rcut = 0.0;
  float2* ltab = nullptr;            // tile kernel: the whole pair factor on a grid logarithmic in x = r^2 + eps^2 (LT_N entries)
  double ltab_alpha = 0.0, ltab_rcut = 0.0, ltab_eps2 = -1.0;
};

namespace {

constexpr int SUBBITS = 6; // sub-cell bits below the cell index in the sort key

struct d4 { double x, y, z, w; };

// periodic wrap of a cell index. A 64-bit remainder by a run-time divisor is ~100 instructions on this hardware and the particle kernels
// need six to nine of them per particle: particles inside the box (the usual case) take none
__device__ inline int wrap_index(long long i, int n) {
  if ((unsigned long long)i < (unsigned long long)n) return (int)i;
  int r = (int)(i % n);
  return r < 0 ? r + n : r;
}
// the same for an index known to lie in [-n, 2n) (tile cells around a brick)
__device__ inline int wrap_near(int i, int n) {
  if (i < 0) return i + n;
  if (i >= n) { i -= n; if (i >= n) i %= n; }          // (a mesh smaller than a brick tile wraps more than once)
  return i;
}

// CIC deposit of particle masses onto one lattice; `shift` = 0 (lattice 1, real part) or 0.5 (lattice displaced by half
// a cell along every axis, imaginary part). grid is the complex grid viewed as doubles: element 2*cell + comp.
template <typename T>
__global__ void __launch_bounds__(256)
mesh_deposit_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, const T* __restrict__ m,
                    int64_t n, double invh, int N, double shift, int comp, double* __restrict__ grid) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const double ux = (double)x[p] * invh - shift, uy = (double)y[p] * invh - shift, uz = (double)z[p] * invh - shift;
    const double fx = floor(ux), fy = floor(uy), fz = floor(uz);
    const double wx1 = ux - fx, wy1 = uy - fy, wz1 = uz - fz;
    const int i0 = wrap_index((long long)fx, N), j0 = wrap_index((long long)fy, N), k0 = wrap_index((long long)fz, N);
    const int i1 = i0 + 1 == N ? 0 : i0 + 1, j1 = j0 + 1 == N ? 0 : j0 + 1, k1 = k0 + 1 == N ? 0 : k0 + 1;
    const double mass = (double)m[p];
    const double wx[2] = {(1.0 - wx1) * mass, wx1 * mass}, wy[2] = {1.0 - wy1, wy1}, wz[2] = {1.0 - wz1, wz1};
    const int ii[2] = {i0, i1}, jj[2] = {j0, j1}, kk[2] = {k0, k1};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          atomicAdd(&grid[2 * (((size_t)ii[a] * N + jj[b]) * N + kk[c]) + comp], (wx[a] * wy[b]) * wz[c]);
  }
}

__device__ inline double inv_sinc2(double t) {   // 1 / sinc(t)^2, sinc(t) = sin(t)/t
  if (fabs(t) < 1e-8) return 1.0;
  const double s = t / sin(t);
  return s * s;
}

// One thread per mode. in = FFT of (lattice-1 density) + i (lattice-2 density); out: three complex grids holding
// (g_x + i g_y), (g_z + i g'_x), (g'_y + i g'_z) in k-space, primes = sampled on the shifted lattice.
// fac = 4 pi G / L^3 (deposit is mass per cell, the unnormalised inverse FFT supplies the sum; 1/N^3 and 1/h^3 combine).
__global__ void __launch_bounds__(256)
mesh_kspace_kernel(const hipfftDoubleComplex* __restrict__ in, hipfftDoubleComplex* __restrict__ o1,
                   hipfftDoubleComplex* __restrict__ o2, hipfftDoubleComplex* __restrict__ o3, int N, double L,
                   double rs, double fac, int interlace, int gaussian) {
  const size_t cells = (size_t)N * N * N;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= cells) return;
  int ix, iy, iz;
  if (cells <= 0xffffffffull) {           // two 32-bit divisions instead of three 64-bit ones (~100 instructions each on this hardware)
    const unsigned id = (unsigned)idx, row = id / (unsigned)N;
    iz = (int)(id - row * (unsigned)N); ix = (int)(row / (unsigned)N); iy = (int)(row - (unsigned)ix * (unsigned)N);
  } else {
    iz = (int)(idx % N); iy = (int)((idx / N) % N); ix = (int)(idx / ((size_t)N * N));
  }
  const int half = N / 2;
  const bool nyq = (N % 2 == 0) && (ix == half || iy == half || iz == half);
  hipfftDoubleComplex z0 = {0.0, 0.0};
  if ((ix == 0 && iy == 0 && iz == 0) || nyq) { o1[idx] = z0; o2[idx] = z0; o3[idx] = z0; return; }
  const int mx = ix ? N - ix : 0, my = iy ? N - iy : 0, mz = iz ? N - iz : 0;
  const hipfftDoubleComplex C = in[idx], Cm = in[((size_t)mx * N + my) * N + mz];
  // A = (C + conj(Cm)) / 2,  B = (C - conj(Cm)) / (2 i)
  const double Ar = 0.5 * (C.x + Cm.x), Ai = 0.5 * (C.y - Cm.y);
  const double Br = 0.5 * (C.y + Cm.y), Bi = -0.5 * (C.x - Cm.x);
  const double dk = 2.0 * 3.14159265358979323846 / L, h = L / N;
  const double kx = dk * (ix <= half ? ix : ix - N), ky = dk * (iy <= half ? iy : iy - N), kz = dk * (iz <= half ? iz : iz - N);
  const double theta = 0.5 * h * (kx + ky + kz);
  double st, ct;
  sincos(theta, &st, &ct);
  double rr = Ar, ri = Ai;
  if (interlace) {   // (A + B e^{-i theta}) / 2
    rr = 0.5 * (Ar + Br * ct + Bi * st);
    ri = 0.5 * (Ai + Bi * ct - Br * st);
  }
  const double k2 = kx * kx + ky * ky + kz * kz;
  double g = fac / k2;
  if (gaussian) g *= exp(-k2 * rs * rs);
  const double dec = inv_sinc2(0.5 * kx * h) * inv_sinc2(0.5 * ky * h) * inv_sinc2(0.5 * kz * h);   // 1 / W_CIC
  g *= dec * dec;                                                                                      // deposit and interpolation
  // g_c = i k_c g rho:  i (rr + i ri) = -ri + i rr
  const double pr = -ri * g, pi = rr * g;
  const double gxr = kx * pr, gxi = kx * pi, gyr = ky * pr, gyi = ky * pi, gzr = kz * pr, gzi = kz * pi;
  // primes: times e^{+i theta}
  const double qr = pr * ct - pi * st, qi = pr * st + pi * ct;
  const double hxr = kx * qr, hxi = kx * qi, hyr = ky * qr, hyi = ky * qi, hzr = kz * qr, hzi = kz * qi;
  // pack P + i Q = (Pr - Qi) + i (Pi + Qr)
  o1[idx] = hipfftDoubleComplex{gxr - gyi, gxi + gyr};
  o2[idx] = hipfftDoubleComplex{gzr - hxi, gzi + hxr};
  o3[idx] = hipfftDoubleComplex{hyr - hzi, hyi + hzr};
}

template <typename T>
__global__ void __launch_bounds__(256)
mesh_gather_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, int64_t n, double invh,
                   int N, const double* __restrict__ f1, const double* __restrict__ f2, const double* __restrict__ f3,
                   int interlace, T* __restrict__ ax, T* __restrict__ ay, T* __restrict__ az) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int lat = 0; lat < (interlace ? 2 : 1); ++lat) {
      const double shift = lat ? 0.5 : 0.0;
      const double ux = (double)x[p] * invh - shift, uy = (double)y[p] * invh - shift, uz = (double)z[p] * invh - shift;
      const double fx = floor(ux), fy = floor(uy), fz = floor(uz);
      const double wx1 = ux - fx, wy1 = uy - fy, wz1 = uz - fz;
      const int i0 = wrap_index((long long)fx, N), j0 = wrap_index((long long)fy, N), k0 = wrap_index((long long)fz, N);
      const int i1 = i0 + 1 == N ? 0 : i0 + 1, j1 = j0 + 1 == N ? 0 : j0 + 1, k1 = k0 + 1 == N ? 0 : k0 + 1;
      const double wx[2] = {1.0 - wx1, wx1}, wy[2] = {1.0 - wy1, wy1}, wz[2] = {1.0 - wz1, wz1};
      const int ii[2] = {i0, i1}, jj[2] = {j0, j1}, kk[2] = {k0, k1};
      // lattice 1: g_x = Re f1, g_y = Im f1, g_z = Re f2;   lattice 2: g_x = Im f2, g_y = Re f3, g_z = Im f3
      const double* gx = lat ? f2 + 1 : f1;
      const double* gy = lat ? f3 : f1 + 1;
      const double* gz = lat ? f3 + 1 : f2;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const size_t cell = 2 * (((size_t)ii[a] * N + jj[b]) * N + kk[c]);
            const double w = (wx[a] * wy[b]) * wz[c];
            sx += w * gx[cell]; sy += w * gy[cell]; sz += w * gz[cell];
          }
    }
    const double s = interlace ? 0.5 : 1.0;
    ax[p] = (T)(s * sx); ay[p] = (T)(s * sy); az[p] = (T)(s * sz);
  }
}

/* ------------------------------------------------------------------ LDS-privatised bricks ----
 * Large particle counts: particles are radix-sorted by the 8^3-cell brick of the mesh they deposit into; one block per
 * (brick, chunk of <= 1024 particles) accumulates both lattices in a 10^3-cell LDS tile (ds_add_f64) and flushes the tile with
 * one global atomic per touched cell — (10/8)^3 = 2 global atomics per mesh cell and lattice instead of 8 per particle, all
 * of them z-contiguous — and the interpolation reads the six force fields of a brick from a 48 KB LDS tile instead of 48
 * scattered 8-byte loads per particle from L2. */
constexpr int BR = 8;              // brick edge in mesh cells
constexpr int BT = BR + 2;         // tile edge: local cells -1 .. BR (lattice 2 reaches one cell below, both one above)
constexpr int BT3 = BT * BT * BT;
constexpr int BCH = 1024;          // particles per block

template <typename T>
__global__ void __launch_bounds__(256)
brick_key_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, int64_t n, double invh, int N, int nbr,
                 uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int i0 = wrap_index((long long)floor((double)x[p] * invh), N), j0 = wrap_index((long long)floor((double)y[p] * invh), N),
            k0 = wrap_index((long long)floor((double)z[p] * invh), N);
  keys[p] = (uint32_t)(((i0 / BR) * nbr + j0 / BR) * nbr + k0 / BR);
  idx[p] = (uint32_t)p;
}

// block -> (brick, first particle, one-past-last particle) from the chunk table brick_prepare builds (chunk_info_kernel: one binary search per
// chunk in a streaming kernel — 15 dependent loads at the start of every short-lived block cost the gather 0.2 ms of 0.75 at 256^3); false
// when the block is beyond the last chunk
__device__ inline bool brick_of_block(const int4* __restrict__ chunk_info, int w, int& brick, int& p0, int& p1) {
  const int4 ci = chunk_info[w];
  brick = ci.x; p0 = ci.y; p1 = ci.z + 1;
  return ci.x >= 0;
}
// CIC weights and brick-local base cells of one particle on both lattices (0: as is, 1: shifted by half a cell). The brick kernels are bound
// by instruction issue, not by memory (ablation in DESIGN.md §6.7), so this is written for few instructions: the cell index through a 32-bit
// conversion when the coordinate is inside the box (else the general 64-bit wrap), and lattice 2 from lattice 1's fraction f instead of a second
// floor: (u - 1/2) - floor(u - 1/2) is f - 1/2 in the upper half of a cell and f + 1/2, one cell lower, in the lower half — the same bits
// (all of these subtractions are exact in fp64).
struct cic_axis { int l1, l2; double w1, w2; };
__device__ inline cic_axis cic_axis_in_brick(double u, int N, int origin) {
  const double f1 = floor(u);
  cic_axis r;
  const int cell = (f1 >= 0.0 && f1 < (double)N) ? (int)f1 : wrap_index((long long)f1, N);
  r.l1 = cell - origin;
  r.w1 = u - f1;
  const bool upper = r.w1 >= 0.5;
  r.w2 = upper ? r.w1 - 0.5 : r.w1 + 0.5;
  r.l2 = upper ? r.l1 : r.l1 - 1;
  return r;
}
struct cic_local { int lx, ly, lz; double wx, wy, wz; };
struct cic_both { cic_axis x, y, z; };
template <typename T>
__device__ inline cic_both cic_in_brick(T px, T py, T pz, double invh, int N, int ox, int oy, int oz) {
  cic_both r;
  r.x = cic_axis_in_brick((double)px * invh, N, ox); r.y = cic_axis_in_brick((double)py * invh, N, oy); r.z = cic_axis_in_brick((double)pz * invh, N, oz);
  return r;
}
__device__ inline cic_local cic_lattice(const cic_both& c, int lat) {
  cic_local r;
  if (lat == 0) { r.lx = c.x.l1; r.ly = c.y.l1; r.lz = c.z.l1; r.wx = c.x.w1; r.wy = c.y.w1; r.wz = c.z.w1; }
  else { r.lx = c.x.l2; r.ly = c.y.l2; r.lz = c.z.l2; r.wx = c.x.w2; r.wy = c.y.w2; r.wz = c.z.w2; }
  return r;
}

constexpr int BPT = BCH / 256;      // particles per thread of a chunk
constexpr int BHEAVY = 8;           // a brick with more chunks than this is "crowded": its chunks add their tiles to the grid with atomics, in parallel

// Both brick kernels are made of short blocks (<= 1024 particles, two or three dependent trips to memory each): what they cost is latency
// unless every load a block needs is in flight at once — so a thread first fetches the indices and then the coordinates of ALL its
// particles, and only then touches the tile (zeroing it / filling it with 12 loads per thread in flight); 256^3 particles on a 256^3
// mesh: gather 0.71 -> 0.61 ms (profiles/r04_mesh_brick_kernels_ablation.txt).
template <typename T>
__global__ void __launch_bounds__(256)
brick_deposit_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, const T* __restrict__ m,
                     const uint32_t* __restrict__ idx_s, const int4* __restrict__ chunk_info,
                     int nbr, double invh, int N, int interlace, double* __restrict__ grid, double* __restrict__ stage,
                     const int* __restrict__ chunk_off, int which) {
  __shared__ double tile[2][BT3];
  int brick, p0, p1;
  if (!brick_of_block(chunk_info, (int)blockIdx.x, brick, p0, p1)) return;
  // which = 1: only the chunks of bricks with at most BHEAVY chunks (their tiles go to `stage`); 2: only those of crowded bricks (atomic
  // flush onto the grid brick_assemble_kernel has written); 0: every chunk, atomic flush
  if (which) { const bool heavy = chunk_off[brick + 1] - chunk_off[brick] > BHEAVY; if (heavy != (which == 2)) return; }
  const int ox = (brick / (nbr * nbr)) * BR, oy = ((brick / nbr) % nbr) * BR, oz = (brick % nbr) * BR;
  uint32_t q[BPT];
  T px[BPT], py[BPT], pz[BPT], pm[BPT];
#pragma unroll
  for (int u = 0; u < BPT; ++u) { const int p = p0 + u * 256 + (int)threadIdx.x; q[u] = p < p1 ? idx_s[p] : 0u; }
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    const bool live = p0 + u * 256 + (int)threadIdx.x < p1;
    px[u] = live ? x[q[u]] : T(0); py[u] = live ? y[q[u]] : T(0); pz[u] = live ? z[q[u]] : T(0); pm[u] = live ? m[q[u]] : T(0);
  }
  for (int t = threadIdx.x; t < BT3; t += 256) { tile[0][t] = 0.0; tile[1][t] = 0.0; }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    if (p0 + u * 256 + (int)threadIdx.x >= p1) break;
    const double mass = (double)pm[u];
    const cic_both cb = cic_in_brick(px[u], py[u], pz[u], invh, N, ox, oy, oz);
    for (int lat = 0; lat < (interlace ? 2 : 1); ++lat) {
      const cic_local c = cic_lattice(cb, lat);
      const double wx[2] = {(1.0 - c.wx) * mass, c.wx * mass}, wy[2] = {1.0 - c.wy, c.wy}, wz[2] = {1.0 - c.wz, c.wz};
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int d = 0; d < 2; ++d)
            atomicAdd(&tile[lat][((c.lx + 1 + a) * BT + (c.ly + 1 + b)) * BT + (c.lz + 1 + d)], (wx[a] * wy[b]) * wz[d]);
    }
  }
  __syncthreads();
  if (stage) {                                             // the whole tile, as it is, to this chunk's slot: brick_assemble_kernel sums the slots
    double* out = stage + (size_t)blockIdx.x * (2 * BT3);
    for (int t = threadIdx.x; t < BT3; t += 256) { out[t] = tile[0][t]; if (interlace) out[BT3 + t] = tile[1][t]; }
    return;
  }
  for (int t = threadIdx.x; t < BT3; t += 256) {
    const double v0 = tile[0][t], v1 = tile[1][t];
    if (v0 == 0.0 && v1 == 0.0) continue;
    const int gx = wrap_near(ox - 1 + t / (BT * BT), N), gy = wrap_near(oy - 1 + (t / BT) % BT, N), gz = wrap_near(oz - 1 + t % BT, N);
    double* cell = grid + 2 * (((size_t)gx * N + gy) * N + gz);
    if (v0 != 0.0) atomicAdd(cell, v0);
    if (v1 != 0.0) atomicAdd(cell + 1, v1);
  }
}

// The density grid from the staged tiles, without atomics and without clearing it first: one block per brick writes the brick's own 8^3 cells
// (both lattices: re, im), each the sum — in a fixed order — of that cell in the tiles of every chunk of every brick whose tile covers it:
// the brick itself and, for the cells on its low / high faces, the neighbour below (whose tile reaches one cell above its brick: local 8)
// / above (local -1). Needs a mesh of whole bricks (N a multiple of 8); other meshes keep the atomic flush of brick_deposit_kernel.
// At 256^3 particles on a 256^3 mesh the atomic flush was 0.6 of the deposit's 0.84 ms (65 M global fp64 atomics) plus 0.075 ms of clearing.
__global__ void __launch_bounds__(256)
brick_assemble_kernel(const double* __restrict__ stage, const int* __restrict__ chunk_off, int nbr, int N, int interlace,
                      double2* __restrict__ grid) {
  __shared__ int coff[27][2];                                  // chunk ranges of the 27 bricks around this one
  const int brick = blockIdx.x;
  const int bx = brick / (nbr * nbr), by = (brick / nbr) % nbr, bz = brick % nbr;
  if (threadIdx.x < 27) {
    const int da = (int)threadIdx.x / 9 - 1, db = ((int)threadIdx.x / 3) % 3 - 1, dd = (int)threadIdx.x % 3 - 1;
    int sx = bx + da, sy = by + db, sz = bz + dd;
    sx = sx < 0 ? sx + nbr : (sx >= nbr ? sx - nbr : sx); sy = sy < 0 ? sy + nbr : (sy >= nbr ? sy - nbr : sy); sz = sz < 0 ? sz + nbr : (sz >= nbr ? sz - nbr : sz);
    const int src = (sx * nbr + sy) * nbr + sz;
    const int c0 = chunk_off[src], c1 = chunk_off[src + 1];
    const bool heavy = c1 - c0 > BHEAVY;                       // a crowded brick's chunks add their tiles themselves, afterwards (atomics)
    coff[threadIdx.x][0] = heavy ? 0 : c0; coff[threadIdx.x][1] = heavy ? 0 : c1;
  }
  __syncthreads();
  constexpr int CPT = BR * BR * BR / 256;                      // cells per thread
  double v0[CPT], v1[CPT];
  int k0[CPT][8], k1[CPT][8], tt[CPT][8];
  // first the first chunk of each of the (up to eight) covering bricks, all loads in flight together: the usual case is one chunk per brick
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int cell = (int)threadIdx.x + 256 * u;
    const int a = cell / (BR * BR), b = (cell / BR) % BR, d = cell % BR;
    const int na = a == 0 ? -1 : (a == BR - 1 ? 1 : 0), nb = b == 0 ? -1 : (b == BR - 1 ? 1 : 0), nd = d == 0 ? -1 : (d == BR - 1 ? 1 : 0);
    double f0[8], f1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ia = j >> 2, ib = (j >> 1) & 1, id = j & 1;
      const bool covered = (!ia || na) && (!ib || nb) && (!id || nd);
      const int da = ia ? na : 0, db = ib ? nb : 0, dd = id ? nd : 0;
      const int w = (da + 1) * 9 + (db + 1) * 3 + (dd + 1);
      k0[u][j] = covered ? coff[w][0] : 0; k1[u][j] = covered ? coff[w][1] : 0;
      tt[u][j] = ((a - BR * da + 1) * BT + (b - BR * db + 1)) * BT + (d - BR * dd + 1);
      const bool has = k0[u][j] < k1[u][j];
      const double* in = stage + (size_t)k0[u][j] * (2 * BT3) + tt[u][j];
      f0[j] = has ? in[0] : 0.0;
      f1[j] = has && interlace ? in[BT3] : 0.0;
    }
    v0[u] = 0.0; v1[u] = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v0[u] += f0[j]; v1[u] += f1[j]; }
  }
  // then whatever further chunks a crowded brick has (fixed order)
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      for (int k = k0[u][j] + 1; k < k1[u][j]; ++k) {
        const double* in = stage + (size_t)k * (2 * BT3) + tt[u][j];
        v0[u] += in[0];
        if (interlace) v1[u] += in[BT3];
      }
    const int cell = (int)threadIdx.x + 256 * u;
    const int a = cell / (BR * BR), b = (cell / BR) % BR, d = cell % BR;
    grid[((size_t)(bx * BR + a) * N + (by * BR + b)) * N + (bz * BR + d)] = make_double2(v0[u], v1[u]);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
brick_gather_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z,
                    const uint32_t* __restrict__ idx_s, const int4* __restrict__ chunk_info,
                    int nbr, double invh, int N, const double2* __restrict__ f1, const double2* __restrict__ f2,
                    const double2* __restrict__ f3, int interlace, T* __restrict__ ax, T* __restrict__ ay, T* __restrict__ az) {
  // g_x g_y g_z on lattice 1, then on lattice 2 — in the OUTPUT precision: an fp32 context interpolates in fp32 (the cell fractions still come
  // from fp64 arithmetic on the coordinates): 24 KB of LDS instead of 48 (six blocks per CU instead of three) and a third of the fp64
  // instructions, which run at half the fp32 rate — the kernel was bound by them, not by memory (ablation: DESIGN.md §6.7)
  __shared__ T tile[6][BT3];
  int brick, p0, p1;
  if (!brick_of_block(chunk_info, (int)blockIdx.x, brick, p0, p1)) return;
  const int ox = (brick / (nbr * nbr)) * BR, oy = ((brick / nbr) % nbr) * BR, oz = (brick % nbr) * BR;
  uint32_t q[BPT];
  T px[BPT], py[BPT], pz[BPT];
#pragma unroll
  for (int u = 0; u < BPT; ++u) { const int p = p0 + u * 256 + (int)threadIdx.x; q[u] = p < p1 ? idx_s[p] : 0u; }
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    const bool live = p0 + u * 256 + (int)threadIdx.x < p1;
    px[u] = live ? x[q[u]] : T(0); py[u] = live ? y[q[u]] : T(0); pz[u] = live ? z[q[u]] : T(0);
  }
  {
    constexpr int TPT = (BT3 + 255) / 256;
    double2 fa[TPT], fb[TPT], fc[TPT];
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
      const int t = (int)threadIdx.x + 256 * u;
      if (t < BT3) {
        const int gx = wrap_near(ox - 1 + t / (BT * BT), N), gy = wrap_near(oy - 1 + (t / BT) % BT, N), gz = wrap_near(oz - 1 + t % BT, N);
        const size_t cell = ((size_t)gx * N + gy) * N + gz;
        fa[u] = f1[cell]; fb[u] = f2[cell];
        if (interlace) fc[u] = f3[cell];
      }
    }
#pragma unroll
    for (int u = 0; u < TPT; ++u) {
      const int t = (int)threadIdx.x + 256 * u;
      if (t < BT3) {
        tile[0][t] = (T)fa[u].x; tile[1][t] = (T)fa[u].y; tile[2][t] = (T)fb[u].x;
        if (interlace) { tile[3][t] = (T)fb[u].y; tile[4][t] = (T)fc[u].x; tile[5][t] = (T)fc[u].y; }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    if (p0 + u * 256 + (int)threadIdx.x >= p1) break;
    T sx = 0, sy = 0, sz = 0;
    const cic_both cb = cic_in_brick(px[u], py[u], pz[u], invh, N, ox, oy, oz);
    for (int lat = 0; lat < (interlace ? 2 : 1); ++lat) {
      const cic_local c = cic_lattice(cb, lat);
      const T wx[2] = {(T)(1.0 - c.wx), (T)c.wx}, wy[2] = {(T)(1.0 - c.wy), (T)c.wy}, wz0 = (T)(1.0 - c.wz), wz1 = (T)c.wz;
      const int tb = ((c.lx + 1) * BT + (c.ly + 1)) * BT + (c.lz + 1);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int t = tb + (a * BT + b) * BT;               // the two cells along z are neighbours in the tile: one two-word read each
          const T w = wx[a] * wy[b];
          sx += w * (wz0 * tile[3 * lat][t] + wz1 * tile[3 * lat][t + 1]);
          sy += w * (wz0 * tile[3 * lat + 1][t] + wz1 * tile[3 * lat + 1][t + 1]);
          sz += w * (wz0 * tile[3 * lat + 2][t] + wz1 * tile[3 * lat + 2][t + 1]);
        }
    }
    const T sc = interlace ? (T)0.5 : (T)1;
    ax[q[u]] = sc * sx; ay[q[u]] = sc * sy; az[q[u]] = sc * sz;
  }
}

/* ------------------------------------------------------------------ short range ---- */

template <typename T>
__global__ void __launch_bounds__(256)
cell_key_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, int64_t n, double L,
                int nc, uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const double invL = 1.0 / L;
  double px = (double)x[p], py = (double)y[p], pz = (double)z[p];
  px -= L * floor(px * invL); py -= L * floor(py * invL); pz -= L * floor(pz * invL);
  const double s = nc * invL;
  const int cx = min(nc - 1, max(0, (int)(px * s))), cy = min(nc - 1, max(0, (int)(py * s))), cz = min(nc - 1, max(0, (int)(pz * s)));
  // 6 low bits: Morton index of the particle's 4x4x4 sub-cell, so that 64 consecutive sorted particles (= the sinks of one
  // wave) form a compact blob and the wave-level cut-off vote of the pair loop can skip most sources
  const int sx = min(3, max(0, (int)((px * s - cx) * 4.0))), sy = min(3, max(0, (int)((py * s - cy) * 4.0))), sz = min(3, max(0, (int)((pz * s - cz) * 4.0)));
  const uint32_t sub = (uint32_t)(((sx >> 1) << 5) | ((sy >> 1) << 4) | ((sz >> 1) << 3) | ((sx & 1) << 2) | ((sy & 1) << 1) | (sz & 1));
  keys[p] = ((uint32_t)((cx * nc + cy) * nc + cz) << SUBBITS) | sub;
  idx[p] = (uint32_t)p;
}

template <typename T, typename V4>
__global__ void __launch_bounds__(256)
sort_gather_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z, const T* __restrict__ m,
                   const uint32_t* __restrict__ idx_s, int64_t n, double L, V4* __restrict__ sorted) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t q = idx_s[p];
  const double invL = 1.0 / L;
  double px = (double)x[q], py = (double)y[q], pz = (double)z[q];
  px -= L * floor(px * invL); py -= L * floor(py * invL); pz -= L * floor(pz * invL);
  V4 v; v.x = (T)px; v.y = (T)py; v.z = (T)pz; v.w = m[q];
  sorted[p] = v;
}

// cell_start[c] = first sorted position whose key >= c, for c = 0 .. ncell3 (cell_start[ncell3] = n)
__global__ void __launch_bounds__(256)
cell_start_kernel(const uint32_t* __restrict__ keys_s, int64_t n, int ncell3, int shift, int* __restrict__ cell_start) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n) return;
  const int lo = p == 0 ? 0 : (int)(keys_s[p - 1] >> shift) + 1;
  const int hi = p == n ? ncell3 : (int)(keys_s[p] >> shift);
  for (int c = lo; c <= hi; ++c) cell_start[c] = (int)p;
}

__global__ void __launch_bounds__(256)
chunk_count_kernel(const int* __restrict__ cell_start, int nrows, int nc, int chunk, int* __restrict__ cnt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > nrows) return;
  cnt[r] = r < nrows ? (cell_start[(r + 1) * nc] - cell_start[r * nc] + chunk - 1) / chunk : 0;
}

// fp32 contexts read the long-range complement  T(r^2) = (erf(a r) - 2 a r/sqrt(pi) e^{-a^2 r^2}) / r^3  from a table uniform in
// r^2 with linear interpolation — as CONCEPT tabulates its short-range force (`shortrange_params` 'tablesize': 4096). T is smooth
// (T(0) = 4 a^3 / (3 sqrt(pi))): 2048 intervals give 3.5e-7 relative to T, 2e-7 of the Newtonian pair force.
constexpr int TABN = 2048;

__global__ void __launch_bounds__(256)
sr_table_kernel(float* __restrict__ table, double alpha, double rcut2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > TABN + 1) return;
  const double s = rcut2 * (double)i / TABN;
  double v = 4.0 * alpha * alpha * alpha / (3.0 * 1.7724538509055159);
  if (s > 1e-12 * rcut2) { const double r = sqrt(s), u = alpha * r; v = (erf(u) - 2.0 * u / 1.7724538509055159 * exp(-u * u)) / (s * r); }
  table[i] = (float)v;
}

// erfc-split pair factor (fp64 contexts, exact):  (r^2+eps^2)^{-3/2} - (erf(a r) - 2 a r/sqrt(pi) e^{-a^2 r^2}) / r^3
__device__ inline double sr_factor(double r2, double eps2, double alpha, double c2, double /*unused*/) {
  if (!(r2 > 0.0)) return 0.0;
  const double r = sqrt(r2), u = alpha * r, rinv = 1.0 / r;
  const double r2s = r2 + eps2;
  const double soft = 1.0 / (r2s * sqrt(r2s));
  const double lng = (erf(u) - c2 * r * exp(-u * u)) * rinv * rinv * rinv;
  return soft - lng;
}

// One block (one wave) per chunk of <= 64*KS consecutive sorted particles of one ROW of cells (fixed cx, cy; cells consecutive in z are
// consecutive in memory). Sparse cells are packed together into full blocks; a dense cell is split into several chunks.
// The sinks span cells lo..hi of the row; sources are the unwrapped cells lo-1..hi+1 of the 9 neighbouring rows, walked as
// contiguous memory segments (split where the periodic image changes). Images out of reach contribute nothing through the
// cut-off test, so every sink sees each of its 27 neighbour-cell images exactly once.
constexpr int SRW = 4;      // waves (= independent chunks) per block; they share only the look-up table
constexpr int SRT = 8;      // the same for the tile kernel (its table is twice the size)

// SOFT: eps > 0, so the softened term is finite for a coincident pair (the particle itself): no r = 0 guard in the loop (its dx = 0 kills the
// term). UNI: all masses equal — the per-pair mass multiply is dropped and the common mass applied once with G at the end.
template <typename T, typename V4, int KS, bool SOFT, bool UNI>
__global__ void __launch_bounds__(64 * SRW)
short_range_kernel(const V4* __restrict__ sorted, const uint32_t* __restrict__ keys_s, const int* __restrict__ cell_start,
                   const int* __restrict__ chunk_off, int nrows, int nc, T L, T alpha, T rcut2, T eps2, double G,
                   const uint32_t* __restrict__ idx_s, const float* __restrict__ table, T* __restrict__ ax, T* __restrict__ ay,
                   T* __restrict__ az, double* __restrict__ part, int64_t n, const unsigned char* __restrict__ rung, int min_rung) {
  // gridDim.y = 1: the wave walks all 9 neighbouring rows and adds its result to ax/ay/az. gridDim.y = 9 * zparts (few chunks: the launch
  // could not fill the chip, and a lone wave per SIMD crawls through ~7000 sources at one instruction every few cycles — the walk is a chain
  // of dependent LDS round trips): wave (chunk, blockIdx.y) walks ONE neighbouring row and, of its cells, every zparts-th one, and stores its
  // partial sums in part[blockIdx.y][3][n] (sorted order); sr_reduce_kernel adds them in a fixed order. (An unrolled branch-free walk was
  // tried instead and measured slower: without the cut-off vote every candidate pays the full pair arithmetic.)
  // One WAVE per chunk of 64*KS consecutive sorted particles; lane l owns particles l, l+64, ... of the chunk, so each of the KS
  // sink sets is a compact 64-particle blob with its own cut-off vote, while one LDS broadcast of a source serves all KS sets.
  // SRW independent waves share a block only for the look-up table. (PMC of the one-sink-per-lane version: LDS 74 % busy — tile
  // broadcasts and random-address table reads — VALU 27 %; two sinks per lane: 256^3 P3M 43 -> 37 ms. A degree-10 polynomial
  // instead of the table was tried and measured slower: its serial Horner chain costs more than the look-up.)
  constexpr bool TABLE = sizeof(T) == 4;        // fp32: long-range complement from the LDS table; fp64: exact erf/exp
  constexpr int CH = 64 * KS;
  __shared__ V4 tiles[SRW][64];
  __shared__ float tab[TABLE ? TABN + 2 : 1];
  if (TABLE) {                                  // the only block-wide step: every wave takes part before any may leave
    for (int t = threadIdx.x; t < TABN + 2; t += 64 * SRW) tab[t] = table[t];
    __syncthreads();
  }
  V4* tile = tiles[threadIdx.x >> 6];           // private to the wave: wave-scope fences order its writes and reads
  const int total = chunk_off[nrows];
  const int w = blockIdx.x * SRW + (int)(threadIdx.x >> 6);
  if (w >= total) return;                       // the grid is an upper bound on the number of chunks
  // row = last r with chunk_off[r] <= w  (wave-uniform binary search)
  int lo_r = 0, hi_r = nrows;
  while (hi_r - lo_r > 1) { const int mid = (lo_r + hi_r) >> 1; if (chunk_off[mid] <= w) lo_r = mid; else hi_r = mid; }
  const int row = lo_r, chunk = w - chunk_off[row];
  const int cx = row / nc, cy = row % nc;
  const int r1 = cell_start[(row + 1) * nc];
  const int i_first = cell_start[row * nc] + chunk * CH, i_last = min(r1, i_first + CH) - 1;
  const int zlo = (int)((keys_s[i_first] >> SUBBITS) % (uint32_t)nc), zhi = (int)((keys_s[i_last] >> SUBBITS) % (uint32_t)nc);
  const int lane = threadIdx.x & 63;
  V4 me[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) me[k] = sorted[min(i_first + k * 64 + lane, i_last)];
  if (rung) {     // rung hierarchy: only particles on rung >= min_rung need their force now; a chunk without any is skipped (its sinks keep 0)
    bool act = false;
#pragma unroll
    for (int k = 0; k < KS; ++k) { const int i = i_first + k * 64 + lane; if (i <= i_last) act |= rung[idx_s[i]] >= min_rung; }
    if (__builtin_amdgcn_ballot_w64(act) == 0) return;
  }
  const T c2 = (T)2 * alpha * (T)0.5641895835477563, a2l = 0;
  const float tscale = TABLE ? (float)TABN / (float)rcut2 : 0.f;
  double tx[KS], ty[KS], tz[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) { tx[k] = 0.0; ty[k] = 0.0; tz[k] = 0.0; }
  const int zparts = gridDim.y == 1 ? 1 : (int)gridDim.y / 9, zpart = gridDim.y == 1 ? 0 : (int)blockIdx.y / 9;
  const int nb_lo = gridDim.y == 1 ? 0 : (int)blockIdx.y % 9, nb_hi = gridDim.y == 1 ? 9 : nb_lo + 1;
  for (int nb = nb_lo; nb < nb_hi; ++nb) {
    int qx = cx + nb / 3 - 1, qy = cy + nb % 3 - 1;
    T shx = 0, shy = 0;                         // periodic image of the whole neighbour row
    if (qx < 0) { qx += nc; shx = -L; } else if (qx >= nc) { qx -= nc; shx = L; }
    if (qy < 0) { qy += nc; shy = -L; } else if (qy >= nc) { qy -= nc; shy = L; }
    const int qbase = (qx * nc + qy) * nc;
    for (int a = zlo - 1; a <= zhi + 1;) {
      const int img = a < 0 ? -1 : a / nc;                      // which periodic image along z
      int b = min(zhi + 1, (img + 1) * nc - 1);                 // last cell of this segment
      if (zparts > 1) {                                         // one cell at a time, every zparts-th cell is this wave's
        b = a;
        if ((a - (zlo - 1)) % zparts != zpart) { a = b + 1; continue; }
      }
      const T shz = (T)img * L;
      // The periodic image: L is SUBTRACTED from whichever of the two lies next to the far face — from the source for a shift of -L, from
      // the sinks (once per segment) for +L. x - L is exact for L/2 <= x <= 2L, so a close pair across a box face keeps its fp32 precision
      // (adding +L to a source near 0 rounds it to ulp(L) = 8e-6 at L = 80: 4e-5 of the force of a pair 0.08 apart).
      const T sox = shx < 0 ? shx : (T)0, soy = shy < 0 ? shy : (T)0, soz = shz < 0 ? shz : (T)0;
      V4 ms[KS];
#pragma unroll
      for (int k = 0; k < KS; ++k) { ms[k].x = me[k].x - (shx - sox); ms[k].y = me[k].y - (shy - soy); ms[k].z = me[k].z - (shz - soz); }
      const int b0 = cell_start[qbase + (a - img * nc)], b1 = cell_start[qbase + (b - img * nc) + 1];
      a = b + 1;
      for (int j0 = b0; j0 < b1; j0 += 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous tile has been consumed
        const int j = j0 + lane;
        if (j < b1) {
          V4 s = sorted[j];
          s.x += sox; s.y += soy; s.z += soz;
          tile[lane] = s;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // LDS serves a wave's accesses in order: the tile is visible
        const int cnt = min(64, b1 - j0);
        T fx[KS], fy[KS], fz[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) { fx[k] = 0; fy[k] = 0; fz[k] = 0; }
          for (int e = 0; e < cnt; ++e) {
            const V4 s = tile[e];                   // wave-uniform address: one LDS broadcast for all KS sink sets
#pragma unroll
            for (int k = 0; k < KS; ++k) {
              const T dx = s.x - ms[k].x, dy = s.y - ms[k].y, dz = s.z - ms[k].z;
              const T r2 = dx * dx + dy * dy + dz * dz;
              // wave-uniform branch: the 64 sinks of a set are a compact blob (sub-cell sort), most sources of the walked cells
              // are out of reach of all of them and cost only the distance test
              if (__builtin_amdgcn_ballot_w64(r2 < rcut2) != 0) {
                T pair;
                if constexpr (TABLE) {
                  const float t = fminf((float)r2 * tscale, (float)TABN);   // lanes beyond the cut-off are masked below
                  const int it = (int)t;
                  const float fr = __builtin_amdgcn_fractf(t);             // t - floor(t): one instruction, no int -> float round trip
                  const float t0 = tab[it], t1 = tab[it + 1];
                  const float sinv = __builtin_amdgcn_rsqf((float)r2 + (float)eps2);
                  const float soft = (SOFT || r2 > (T)0) ? sinv * sinv * sinv : 0.f;
                  pair = soft - fmaf(fr, t1 - t0, t0);
                } else {
                  pair = sr_factor(r2, eps2, alpha, c2, a2l);
                }
                T f = UNI ? pair : s.w * pair;
                f = r2 < rcut2 ? f : (T)0;
                fx[k] += f * dx; fy[k] += f * dy; fz[k] += f * dz;
              }
            }
          }
#pragma unroll
        for (int k = 0; k < KS; ++k) { tx[k] += (double)fx[k]; ty[k] += (double)fy[k]; tz[k] += (double)fz[k]; }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int i = i_first + k * 64 + lane;
    if (i <= i_last) {
      const double g = UNI ? G * (double)sorted[0].w : G;          // equal masses: the common mass comes in here
      if (gridDim.y == 1) {
        const uint32_t o = idx_s[i];
        ax[o] += (T)(g * tx[k]); ay[o] += (T)(g * ty[k]); az[o] += (T)(g * tz[k]);
      } else {
        double* p = part + (size_t)blockIdx.y * 3 * (size_t)n;
        const double m0 = UNI ? (double)sorted[0].w : 1.0;        // sr_reduce_kernel applies G
        p[i] = m0 * tx[k]; p[(size_t)n + i] = m0 * ty[k]; p[2 * (size_t)n + i] = m0 * tz[k];
      }
    }
  }
}

// a[idx_s[i]] += G * sum_{y<ny} part[y][c][i]   (fixed order: bit-reproducible)
template <typename T>
__global__ void __launch_bounds__(256)
sr_reduce_kernel(const double* __restrict__ part, int ny, const uint32_t* __restrict__ idx_s, int64_t n, double G, T* __restrict__ ax,
                 T* __restrict__ ay, T* __restrict__ az) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for (int y = 0; y < ny; ++y) {
    const double* p = part + (size_t)y * 3 * (size_t)n;
    sx += p[i]; sy += p[(size_t)n + i]; sz += p[2 * (size_t)n + i];
  }
  const uint32_t o = idx_s[i];
  ax[o] += (T)(G * sx); ay[o] += (T)(G * sy); az[o] += (T)(G * sz);
}

// ---- fp32 pair kernel on W-sink tiles (round 2): its table, its chunk list, the kernel ----
// The tile kernel reads the WHOLE pair factor  g(x) = x^{-3/2} - T(x - eps^2),  x = r^2 + eps^2,  from a table logarithmic in x: interval i
// is the set of floats whose bits >> LT_SHIFT equal base + i (2^LT_M intervals per octave, LT_OCT octaves below x_cut = r_cut^2 + eps^2), and
// stores (a, b) with g ~ a + b x on it — the index is two integer instructions on the bits of x, the value one fma: no v_rsq_f32, no
// fraction. Chord through the end points, lowered by 2/3 of its mid-point error (zero mean error over the interval): 2^8 intervals per
// octave give |error| <= 4.8e-6 of the Newtonian term, +-1e-6 of a summed short-range force. x below the table (closer than r_cut / 64,
// the particle itself) takes a branch with v_rsq_f32 and the series of T.
constexpr int LT_M = 8, LT_SHIFT = 23 - LT_M, LT_OCT = 12, LT_N = LT_OCT << LT_M;

__device__ inline double sr_pair_factor(double x, double eps2, double alpha) {
  const double s = x > eps2 ? x - eps2 : 0.0;
  double t = 4.0 * alpha * alpha * alpha / (3.0 * 1.7724538509055159);
  if (s > 1e-14 / (alpha * alpha)) { const double r = sqrt(s), u = alpha * r; t = (erf(u) - 2.0 * u / 1.7724538509055159 * exp(-u * u)) / (s * r); }
  return 1.0 / (x * sqrt(x)) - t;
}

__global__ void __launch_bounds__(256)
sr_logtable_kernel(float2* __restrict__ tab, int base, double alpha, double eps2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= LT_N) return;
  const double x0 = (double)__uint_as_float((uint32_t)(base + i) << LT_SHIFT), x1 = (double)__uint_as_float((uint32_t)(base + i + 1) << LT_SHIFT);
  const double g0 = sr_pair_factor(x0, eps2, alpha), g1 = sr_pair_factor(x1, eps2, alpha);
  const double b = (g1 - g0) / (x1 - x0), xm = 0.5 * (x0 + x1);
  const double a = g0 - b * x0 + (2.0 / 3.0) * (sr_pair_factor(xm, eps2, alpha) - (g0 + b * (xm - x0)));
  tab[i] = make_float2((float)a, (float)b);
}

// v of the lane E places further on in this lane's row of 16 (rotation within the row): a DPP modifier on the consuming instruction
template <int E>
__device__ inline float dpp_row_ror(float v) {
  if constexpr (E == 0) return v;
  else return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + E, 0xf, 0xf, false));
}

// chunk k of the tile kernel -> (row, first sorted particle, last sorted particle): one binary search per chunk here, in a streaming
// kernel, instead of 13 dependent loads at the start of every (short-lived) wave
__global__ void __launch_bounds__(256)
chunk_info_kernel(const int* __restrict__ cell_start, const int* __restrict__ chunk_off, int nrows, int nf, int W, int cap,
                  int4* __restrict__ info) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cap) return;
  if (k >= chunk_off[nrows]) { info[k] = make_int4(-1, 0, -1, 0); return; }   // the grid is an upper bound on the number of chunks
  int lo = 0, hi = nrows;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (chunk_off[mid] <= k) lo = mid; else hi = mid; }
  const int i0 = cell_start[lo * nf] + (k - chunk_off[lo]) * W;
  info[k] = make_int4(lo, i0, min(cell_start[(lo + 1) * nf], i0 + W) - 1, 0);
}

// The 64-sink wave above tests every particle of 27 cells of edge >= r_cut against its blob: at one particle per mesh cell and r_cut = 5.6
// cells that is ~5 000 candidates for the ~740 partners a sink really has, and only 37 % of the lanes of a passing candidate do useful
// work — a 64-particle blob plus its reach fills the 3^3 block whatever the sub-cell order. Here a wave owns only W (8 or 16) consecutive
// sorted sinks of a row — a blob a fraction of r_cut wide, every lane l holding sink l % W.
//   candidates  cells of edge >= r_cut / 3 (or / 2); per neighbouring ROW of cells (fixed cx, cy: contiguous in memory along z) one
//               segment, chosen from the blob's bounding box: a row whose column is out of reach is dropped, of the others only the
//               z-interval within sqrt(r_cut^2 - d_xy^2) of the box is loaded (<= 49 rows x 3 periodic images; the lanes build the list
//               in LDS) — ~2 600 candidates per 16-sink blob instead of 5 000 per 64-sink one
//   culling     every loaded source is tested against the box (a quarter of an instruction per candidate and lane); those in reach are
//               appended, compacted, to an LDS buffer; four 64-source loads are in flight
//   pair steps  64 buffered sources at a time: lane l takes source l into registers and in step e meets the source of the lane e places
//               further on in its row of 16 (DPP row rotation) — W steps of 64 pairs, no LDS access per pair but the table read; a wave
//               vote ends the steps without a pair in reach; the pair factor is one fma on a log-spaced table entry
template <int W, bool SOFT, bool UNI>
__global__ void __launch_bounds__(64 * SRT)
short_range_tile_kernel(const float4* __restrict__ sorted, const int* __restrict__ cell_start, const int4* __restrict__ chunk_info,
                        int total, int nf, int reach, float L, float rcut, float eps2, double G,
                        const uint32_t* __restrict__ idx_s, const float2* __restrict__ ltab, int lbase, float t0c, float t1c,
                        float* __restrict__ ax, float* __restrict__ ay, float* __restrict__ az, const unsigned char* __restrict__ rung,
                        int min_rung) {
  constexpr int MAXSEG = 49 * 3 + 1;           // (2*3+1)^2 neighbouring rows x up to three periodic images along z
  __shared__ float4 tiles[SRT][128];            // sources in reach of the blob: 64 being consumed + up to 63 waiting
  __shared__ float2 ltb[LT_N];                  // pair factor ~ ltb[i].x + ltb[i].y * x on the floats x with bits >> LT_SHIFT == lbase + i
  // segment: first particle; count (26 bits) | periodic image << 26 (2 bits per axis: 0: -L, 1: 0, 2: +L)
  __shared__ int seg_b0[SRT][MAXSEG], seg_ci[SRT][MAXSEG];
  for (int t = threadIdx.x; t < LT_N; t += 64 * SRT) ltb[t] = ltab[t];
  __syncthreads();                              // the only block-wide step
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4* tile = tiles[wv];
  const int w = blockIdx.x * SRT + wv;
  if (w >= total) return;
  const int4 ci = chunk_info[w];
  const int row = ci.x, i_first = ci.y, i_last = ci.z;
  if (row < 0) return;
  const int cx = row / nf, cy = row % nf;
  const int sk = lane & (W - 1), slot = lane / W;      // sink of this lane; slot 0 stores the result
  const float4 me = sorted[min(i_first + sk, i_last)];   // padding sinks repeat the last one: inside the box, never stored
  if (rung) {
    const int i = i_first + sk;
    if (__builtin_amdgcn_ballot_w64(i <= i_last && rung[idx_s[i]] >= min_rung) == 0) return;
  }
  // bounding box of the blob (every group of W lanes holds all W sinks)
  float lox = me.x, loy = me.y, loz = me.z, hix = me.x, hiy = me.y, hiz = me.z;
#pragma unroll
  for (int o = 1; o < W; o <<= 1) {
    lox = fminf(lox, __shfl_xor(lox, o)); loy = fminf(loy, __shfl_xor(loy, o)); loz = fminf(loz, __shfl_xor(loz, o));
    hix = fmaxf(hix, __shfl_xor(hix, o)); hiy = fmaxf(hiy, __shfl_xor(hiy, o)); hiz = fmaxf(hiz, __shfl_xor(hiz, o));
  }
  // ---- candidate segments: lane (dx, dy) looks at one neighbouring row (still in box coordinates)
  const int R = 2 * reach + 1;
  const float c = L / (float)nf, invc = (float)nf / L, margin = 1e-4f * c, rc = rcut * 1.0001f;
  int nmine = 0, sb0[3], scn[3], simg[3];
  if (lane < R * R) {
    int qx = cx + lane / R - reach, qy = cy + lane % R - reach;
    const float colx = (float)qx * c, coly = (float)qy * c;          // the row's column, unwrapped
    const float gx = fmaxf(0.f, fmaxf(colx - hix, lox - (colx + c)) - margin), gy = fmaxf(0.f, fmaxf(coly - hiy, loy - (coly + c)) - margin);
    const float d2 = gx * gx + gy * gy;
    if (d2 < rc * rc) {
      const float rz = sqrtf(rc * rc - d2) + margin;
      const int za = (int)floorf((loz - rz) * invc), zb = (int)floorf((hiz + rz) * invc);
      int code = 1 | (1 << 2);
      if (qx < 0) { qx += nf; code = (code & ~3) | 0; } else if (qx >= nf) { qx -= nf; code = (code & ~3) | 2; }
      if (qy < 0) { qy += nf; code = (code & ~12) | (0 << 2); } else if (qy >= nf) { qy -= nf; code = (code & ~12) | (2 << 2); }
      const int qbase = (qx * nf + qy) * nf;
      for (int img = -1; img <= 1; ++img) {
        const int a = max(za, img * nf), b = min(zb, img * nf + nf - 1);
        if (a > b) continue;
        const int b0 = cell_start[qbase + a - img * nf], b1 = cell_start[qbase + b - img * nf + 1];
        if (b1 > b0) { sb0[nmine] = b0; scn[nmine] = b1 - b0; simg[nmine] = code | ((img + 1) << 4); ++nmine; }
      }
    }
  }
  int nloads = 0;                               // 64-source loads of the whole walk
  for (int k = 0; k < nmine; ++k) nloads += (scn[k] + 63) >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) nloads += __shfl_xor(nloads, o);
  nloads = __builtin_amdgcn_readfirstlane(nloads);
  int nseg = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(nmine > k);
    if (nmine > k) {
      const int pos = nseg + __popcll(mk & ((1ull << lane) - 1ull));
      seg_b0[wv][pos] = sb0[k]; seg_ci[wv][pos] = scn[k] | (simg[k] << 26);
    }
    nseg += __popcll(mk);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const float xcut = rcut * rcut + eps2;        // the cut-off on x = r^2 + eps^2
  const float xlo = __uint_as_float((uint32_t)lbase << LT_SHIFT);
  const float far = 1e18f;
  // From here on coordinates are relative to the low corner of the blob's box: a source that reaches the sinks through a periodic image
  // is shifted by (image - corner), a small number, instead of by +-L — in fp32 a close pair across a box face keeps the precision it
  // has anywhere else (with the plain +-L shift it lost ulp(L) = 8e-6 of its separation).
  // (Only for a blob narrower than r_cut: a straggling one — 16 particles of a void — would trade the exact differences of unshifted
  // coordinates for rounding at the scale of its own extent; it keeps box coordinates.)
  const bool compact = fmaxf(fmaxf(hix - lox, hiy - loy), hiz - loz) < rcut;
  const float ox = compact ? lox : 0.f, oy = compact ? loy : 0.f, oz = compact ? loz : 0.f;
  const float4 mel = make_float4(me.x - ox, me.y - oy, me.z - oz, me.w);
  const float ax0 = lox - ox, ay0 = loy - oy, az0 = loz - oz, bx = hix - ox, by = hiy - oy, bz = hiz - oz;   // the box in those coordinates
  double tx = 0.0, ty = 0.0, tz = 0.0;
  // 64 buffered sources against the W sinks, no LDS traffic per pair: lane l takes source l into registers (one conflict-free
  // ds_read_b128), so each row of 16 lanes holds 16 sources, and in step e a lane meets the source of the lane e places further in its
  // row (DPP row rotation folded into the subtractions). W steps: lanes l and l + W of a row hold the same sink and between them meet all
  // 16 sources of the row once. (With one ds_read_b96 per pair step the four SIMDs of a CU kept its LDS 70 % busy — the tile reads and
  // the random table reads together — and the kernel was LDS bound, not VALU bound.)
  auto consume64 = [&]() {
    const float4 src = tile[lane];
    float fx = 0.f, fy = 0.f, fz = 0.f;
#define NB_TILE_STEP(E)                                                                                                         \
    if constexpr ((E) < W) {                                                                                                    \
      const float qx = dpp_row_ror<(E)>(src.x), qy = dpp_row_ror<(E)>(src.y), qz = dpp_row_ror<(E)>(src.z);                    \
      const float dx = qx - mel.x, dy = qy - mel.y, dz = qz - mel.z;                                                            \
      const float x = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));                                                           \
      if (__builtin_amdgcn_ballot_w64(x < xcut) != 0) {          /* wave-uniform: some pair of this step is in reach */        \
        /* lanes beyond the cut-off read anything (or nothing: LDS reads out of range return 0) and are masked below */         \
        const float2 ab = ltb[(int)(__float_as_uint(x) >> LT_SHIFT) - lbase];                                                   \
        float f = fmaf(ab.y, x, ab.x);                                                                                          \
        if (__builtin_amdgcn_ballot_w64(x < xlo) != 0) {         /* closer than r_cut/64 (the particle itself, eps -> 0): rare */ \
          asm volatile("" ::: "memory");                         /* a real branch: not to be speculated into every step */      \
          const float y = __builtin_amdgcn_rsqf(x);                                                                             \
          const float g = ((SOFT || x > eps2) ? y * y * y : 0.f) - t0c * fmaf(t1c, x - eps2, 1.f);   /* T(s) = T(0) (1 - 3/5 alpha^2 s + ...) */ \
          f = x < xlo ? g : f;                                                                                                  \
        }                                                                                                                       \
        if (!UNI) f *= dpp_row_ror<(E)>(src.w);                                                                                 \
        f = x < xcut ? f : 0.f;                                                                                                 \
        fx = fmaf(f, dx, fx); fy = fmaf(f, dy, fy); fz = fmaf(f, dz, fz);                                                       \
      }                                                                                                                         \
    }
    NB_TILE_STEP(0) NB_TILE_STEP(1) NB_TILE_STEP(2) NB_TILE_STEP(3) NB_TILE_STEP(4) NB_TILE_STEP(5) NB_TILE_STEP(6) NB_TILE_STEP(7)
    NB_TILE_STEP(8) NB_TILE_STEP(9) NB_TILE_STEP(10) NB_TILE_STEP(11) NB_TILE_STEP(12) NB_TILE_STEP(13) NB_TILE_STEP(14) NB_TILE_STEP(15)
#undef NB_TILE_STEP
    tx += (double)fx; ty += (double)fy; tz += (double)fz;
  };
  // The (segment, 64-source load) walk: a scalar iterator runs PF loads ahead of the consumer, so PF global loads are in flight per wave
  // (a wave makes ~50 loads, one per short segment: one at a time they are a chain of ~50 memory latencies, which a 32^3 system — a few
  // waves per SIMD — cannot hide). Each loaded source is tested against the blob's bounding box (a quarter of an instruction per
  // candidate and lane) and only those in reach of it are appended to the LDS buffer; 64 buffered sources are consumed at a time.
  constexpr int PF = 4;
  int sg = 0, j0 = 0, b0 = 0, cnt = 0, fill = 0;
  float sx = 0.f, sy = 0.f, sz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;
  auto open_segment = [&]() {
    b0 = __builtin_amdgcn_readfirstlane(seg_b0[wv][sg]);
    const int ci = __builtin_amdgcn_readfirstlane(seg_ci[wv][sg]), code = (int)((unsigned)ci >> 26);
    cnt = ci & ((1 << 26) - 1);
    // v + shift - origin in two exact steps: L is subtracted from whichever lies next to the far face (the source for a shift of -L: first
    // term; the blob's origin for +L: second term), cf. the 64-sink kernel above
    const float hx = (float)((code & 3) - 1) * L, hy = (float)(((code >> 2) & 3) - 1) * L, hz = (float)(((code >> 4) & 3) - 1) * L;
    nx = fminf(hx, 0.f); ny = fminf(hy, 0.f); nz = fminf(hz, 0.f);
    sx = fmaxf(hx, 0.f) - ox; sy = fmaxf(hy, 0.f) - oy; sz = fmaxf(hz, 0.f) - oz;
    j0 = 0;
  };
  auto fetch_next = [&]() {                                    // the iterator's next 64 sources (all padding once it is exhausted)
    float4 v = make_float4(far, far, far, 0.f);              // lanes past the end of the segment: beyond every cut-off
    if (sg < nseg) {
      if (j0 + lane < cnt) { v = sorted[b0 + j0 + lane]; v.x = (v.x + nx) + sx; v.y = (v.y + ny) + sy; v.z = (v.z + nz) + sz; }
      j0 += 64;
      if (j0 >= cnt) { ++sg; if (sg < nseg) open_segment(); }
    }
    return v;
  };
  const float reach2 = rc * rc;
  auto handle = [&](const float4& v) {
    const float ex = fmaxf(fmaxf(ax0 - v.x, v.x - bx), 0.f), ey = fmaxf(fmaxf(ay0 - v.y, v.y - by), 0.f), ez = fmaxf(fmaxf(az0 - v.z, v.z - bz), 0.f);
    const bool in = ex * ex + ey * ey + ez * ez < reach2;
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(in);
    if (in) tile[fill + __popcll(mk & ((1ull << lane) - 1ull))] = v;
    fill += __popcll(mk);
    if (fill >= 64) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      consume64();
      const float4 r = tile[64 + lane];                      // the overflow moves to the front
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      fill -= 64;
      if (lane < fill) tile[lane] = r;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  };
  if (nseg > 0) open_segment();
  float4 vq[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) vq[u] = fetch_next();
  for (int k = 0; k < nloads; k += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const float4 v = vq[u];
      vq[u] = fetch_next();
      if (k + u < nloads) handle(v);
    }
  }
  if (fill > 0) {
    if (lane >= fill) tile[lane] = make_float4(far, far, far, 0.f);     // the last, partial batch: padding beyond every cut-off
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    consume64();
  }
  // the 64 / W lanes that hold the same sink, added in a fixed order
#pragma unroll
  for (int o = W; o < 64; o <<= 1) { tx += __shfl_xor(tx, o); ty += __shfl_xor(ty, o); tz += __shfl_xor(tz, o); }
  const int i = i_first + sk;
  if (slot == 0 && i <= i_last) {
    const double g = UNI ? G * (double)sorted[0].w : G;
    const uint32_t o = idx_s[i];
    ax[o] += (float)(g * tx); ay[o] += (float)(g * ty); az[o] += (float)(g * tz);
  }
}

inline int stream_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

#define M_FFT(c, call) do { hipfftResult r_ = (call); if (r_ != HIPFFT_SUCCESS) \
  return nb_fail((c), NB_ERR_HIP, std::string(#call) + ": hipfft error " + std::to_string((int)r_)); } while (0)

int ensure_grids(nb_ctx* c, nb_mesh_state& s, int ngrid) {
  if (s.ngrid == ngrid && s.have_fft && s.fft_stream == c->stream) return NB_OK;
  NB_HIP(c, hipStreamSynchronize(c->stream));
  if (s.ngrid != ngrid) {
    for (auto& g : s.grid) if (g) { (void)hipFree(g); g = nullptr; }
    if (s.have_fft) { (void)hipfftDestroy(s.fft); s.have_fft = false; }
    const size_t bytes = (size_t)ngrid * ngrid * ngrid * sizeof(hipfftDoubleComplex);
    for (auto& g : s.grid) NB_HIP(c, hipMalloc((void**)&g, bytes));
    M_FFT(c, hipfftPlan3d(&s.fft, ngrid, ngrid, ngrid, HIPFFT_Z2Z));
    s.have_fft = true;
    s.ngrid = ngrid;
  }
  M_FFT(c, hipfftSetStream(s.fft, c->stream));
  s.fft_stream = c->stream;
  return NB_OK;
}

int ensure_cells(nb_ctx* c, nb_mesh_state& s, int64_t n, int esz, int ncell3) {
  if (s.n_alloc < n || s.esz != esz) {
    NB_HIP(c, hipStreamSynchronize(c->stream));
    for (void* p : {(void*)s.keys, (void*)s.keys_s, (void*)s.idx, (void*)s.idx_s, s.sorted}) if (p) (void)hipFree(p);
    s.keys = s.keys_s = s.idx = s.idx_s = nullptr; s.sorted = nullptr;
    NB_HIP(c, hipMalloc((void**)&s.keys, (size_t)n * 4)); NB_HIP(c, hipMalloc((void**)&s.keys_s, (size_t)n * 4));
    NB_HIP(c, hipMalloc((void**)&s.idx, (size_t)n * 4)); NB_HIP(c, hipMalloc((void**)&s.idx_s, (size_t)n * 4));
    NB_HIP(c, hipMalloc(&s.sorted, (size_t)n * 4 * (size_t)esz));
    s.n_alloc = n; s.esz = esz;
    if (s.tmp) { (void)hipFree(s.tmp); s.tmp = nullptr; s.tmp_bytes = 0; }
  }
  if (s.ncell3_alloc < ncell3) {
    NB_HIP(c, hipStreamSynchronize(c->stream));
    for (int* p : {s.cell_start, s.chunk_cnt, s.chunk_off}) if (p) (void)hipFree(p);
    NB_HIP(c, hipMalloc((void**)&s.cell_start, ((size_t)ncell3 + 1) * 4));
    NB_HIP(c, hipMalloc((void**)&s.chunk_cnt, ((size_t)ncell3 + 1) * 4));
    NB_HIP(c, hipMalloc((void**)&s.chunk_off, ((size_t)ncell3 + 1) * 4));
    s.ncell3_alloc = ncell3;
    if (s.tmp) { (void)hipFree(s.tmp); s.tmp = nullptr; s.tmp_bytes = 0; }
  }
  if (!s.tmp) {
    size_t t1 = 0, t2 = 0;
    NB_HIP(c, rocprim::radix_sort_pairs((void*)nullptr, t1, s.keys, s.keys_s, s.idx, s.idx_s, (size_t)s.n_alloc, 0, 32, c->stream));
    NB_HIP(c, rocprim::exclusive_scan((void*)nullptr, t2, s.chunk_cnt, s.chunk_off, 0, (size_t)s.ncell3_alloc + 1, rocprim::plus<int>(), c->stream));
    s.tmp_bytes = t1 > t2 ? t1 : t2;
    if (s.tmp_bytes < 256) s.tmp_bytes = 256;
    NB_HIP(c, hipMalloc(&s.tmp, s.tmp_bytes));
  }
  return NB_OK;
}

template <typename T, typename V4>
int short_range(nb_ctx* c, nb_mesh_state& s, const T* x, const T* y, const T* z, const T* m, int64_t n, double L, int nc,
                double alpha, double rcut, double eps2, double G, T* ax, T* ay, T* az) {
  // fp32 and >= 16 384 particles: the W-sink tile kernel on cells of edge >= r_cut / div, div = 3 or 2 — the finest with >= 4 particles per
  // cell on average (emptier cells mean many short segments per blob: measured slower than the 64-sink kernel below). NB_SR_TILES=0/1
  // forces it off/on, NB_SR_W (8 | 16) and NB_SR_DIV (1..3) choose the blob width and the cell size.
  static const int tiles_env = getenv("NB_SR_TILES") ? atoi(getenv("NB_SR_TILES")) : -1;
  static const int w_env = getenv("NB_SR_W") ? atoi(getenv("NB_SR_W")) : 0, div_env = getenv("NB_SR_DIV") ? atoi(getenv("NB_SR_DIV")) : 0;
  const int tile_w = (w_env == 8 || w_env == 16) ? w_env : (n < 100000 ? 8 : 16);   // small systems: more, shorter waves (32^3 farm run 0.132 -> 0.116 s)
  int reach = 0, nf = 0;
  bool use_tiles = false;
  if (sizeof(T) == 4 && tiles_env != 0 && (tiles_env > 0 || n >= 16384) && n < (1ll << 26)) {   // (a segment's count is stored in 26 bits)
    const int d_hi = (div_env >= 1 && div_env <= 3) ? div_env : 3, d_lo = (div_env >= 1 && div_env <= 3) ? div_env : 2;
    for (int d = d_hi; d >= d_lo && !use_tiles; --d) {
      const int f = (int)floor((double)d * L / rcut);
      const double f3 = (double)f * f * f;
      if (f >= 2 * d + 2 && f3 <= 6.0e7 && (tiles_env > 0 || (double)n >= 4.0 * f3)) { reach = d; nf = f; use_tiles = true; }
    }
  }
  if (use_tiles) nc = nf;
  const int ncell3 = nc * nc * nc;
  int rc = ensure_cells(c, s, n, (int)sizeof(T), ncell3);
  if (rc) return rc;
  hipStream_t st = c->stream;
  const int nb = (int)((n + 255) / 256);
  cell_key_kernel<T><<<nb, 256, 0, st>>>(x, y, z, n, L, nc, s.keys, s.idx);
  NB_HIP(c, hipGetLastError());
  int bits = 1; while ((1u << bits) < (unsigned)ncell3) ++bits;
  bits += SUBBITS;
  size_t tb = s.tmp_bytes;
  NB_HIP(c, rocprim::radix_sort_pairs(s.tmp, tb, s.keys, s.keys_s, s.idx, s.idx_s, (size_t)n, 0, (unsigned)bits, st));
  sort_gather_kernel<T, V4><<<nb, 256, 0, st>>>(x, y, z, m, s.idx_s, n, L, (V4*)s.sorted);
  NB_HIP(c, hipGetLastError());
  cell_start_kernel<<<(int)((n + 1 + 255) / 256), 256, 0, st>>>(s.keys_s, n, ncell3, SUBBITS, s.cell_start);
  NB_HIP(c, hipGetLastError());
  const int nrows = nc * nc;
  static const int ks_env = getenv("NB_SR_KS") ? atoi(getenv("NB_SR_KS")) : 0;
  // sinks per lane (chunk = 64 * ks particles): 2 halves the LDS broadcasts per pair; with very full cells (the Ewald cut-off:
  // thousands of particles per cell, hardly any source out of reach of a 64-sink blob) 1 measured 6 % faster
  const int ks = (ks_env == 1 || ks_env == 2 || ks_env == 4) ? ks_env : (n / ncell3 >= 1024 ? 1 : 2);
  const int chunk = use_tiles ? tile_w : 64 * ks;
  chunk_count_kernel<<<(nrows + 1 + 255) / 256, 256, 0, st>>>(s.cell_start, nrows, nc, chunk, s.chunk_cnt);
  NB_HIP(c, hipGetLastError());
  tb = s.tmp_bytes;
  NB_HIP(c, rocprim::exclusive_scan(s.tmp, tb, s.chunk_cnt, s.chunk_off, 0, (size_t)nrows + 1, rocprim::plus<int>(), st));
  const int64_t max_chunks = (int64_t)nrows + n / chunk + 1;    // sum_r ceil(n_r / chunk) <= nrows + n / chunk
  if (sizeof(T) == 4 && (!s.table || s.table_alpha != alpha || s.table_rcut != rcut)) {
    if (!s.table) NB_HIP(c, hipMalloc((void**)&s.table, (TABN + 2) * sizeof(float)));
    sr_table_kernel<<<(TABN + 2 + 255) / 256, 256, 0, st>>>(s.table, alpha, rcut * rcut);
    NB_HIP(c, hipGetLastError());
    s.table_alpha = alpha; s.table_rcut = rcut;
  }
  if constexpr (sizeof(T) == 4) {
    if (use_tiles) {
      if (s.chunk_info_cap < max_chunks) {
        NB_HIP(c, hipStreamSynchronize(st));
        if (s.chunk_info) { (void)hipFree(s.chunk_info); s.chunk_info = nullptr; s.chunk_info_cap = 0; }
        NB_HIP(c, hipMalloc((void**)&s.chunk_info, (size_t)(max_chunks + max_chunks / 8) * sizeof(int4)));
        s.chunk_info_cap = max_chunks + max_chunks / 8;
      }
      chunk_info_kernel<<<(unsigned)((max_chunks + 255) / 256), 256, 0, st>>>(s.cell_start, s.chunk_off, nrows, nc, tile_w, (int)max_chunks, s.chunk_info);
      NB_HIP(c, hipGetLastError());
      const float eps2f = (float)eps2, rcf = (float)rcut, xcut = rcf * rcf + eps2f;     // as the kernel computes it
      uint32_t xb; memcpy(&xb, &xcut, 4);
      const int lbase = (int)(xb >> LT_SHIFT) + 1 - LT_N;
      if (!s.ltab || s.ltab_alpha != alpha || s.ltab_rcut != rcut || s.ltab_eps2 != eps2) {
        if (!s.ltab) NB_HIP(c, hipMalloc((void**)&s.ltab, LT_N * sizeof(float2)));
        sr_logtable_kernel<<<(LT_N + 255) / 256, 256, 0, st>>>(s.ltab, lbase, alpha, (double)eps2f);
        NB_HIP(c, hipGetLastError());
        s.ltab_alpha = alpha; s.ltab_rcut = rcut; s.ltab_eps2 = eps2;
      }
      const float t0c = (float)(4.0 * alpha * alpha * alpha / (3.0 * 1.7724538509055159)), t1c = (float)(-0.6 * alpha * alpha);
      const unsigned gridx = (unsigned)((max_chunks + SRT - 1) / SRT);
      const bool soft_t = eps2 > 0.0 && (float)eps2 > 1e-30f, uni_t = c->uniform_mass;
#define NB_ST2(WW, SF, UN) short_range_tile_kernel<WW, SF, UN><<<gridx, 64 * SRT, 0, st>>>((const float4*)s.sorted, s.cell_start, s.chunk_info, (int)max_chunks, \
    nc, reach, (float)L, rcf, eps2f, G, s.idx_s, s.ltab, lbase, t0c, t1c, ax, ay, az, c->sr_rung, c->sr_min_rung)
#define NB_ST(WW) do { if (soft_t && uni_t) NB_ST2(WW, true, true); else if (soft_t) NB_ST2(WW, true, false); else if (uni_t) NB_ST2(WW, false, true); \
                       else NB_ST2(WW, false, false); } while (0)
      if (tile_w == 16) NB_ST(16); else NB_ST(8);
#undef NB_ST
#undef NB_ST2
      NB_HIP(c, hipGetLastError());
      return NB_OK;
    }
  }
  // fewer chunks than ~4 waves per SIMD: one wave per (chunk, neighbouring row, z-part) — 9 x zparts the waves, each that fraction of the
  // serial walk (the 32^3 farm jobs of the thesis: 256 chunks; short-range kernel 1.41 ms -> see DESIGN.md §6.7); partial sums go
  // through an fp64 slab and are added in a fixed order
  static const int zp_env = getenv("NB_SR_ZPARTS") ? atoi(getenv("NB_SR_ZPARTS")) : 0;
  int ny = 1;
  if (max_chunks < 4096) {
    int zparts = zp_env > 0 ? zp_env : (max_chunks < 1024 ? 3 : (max_chunks < 2048 ? 2 : 1));
    if (zparts > 4) zparts = 4;
    ny = 9 * zparts;
  }
  double* part = nullptr;
  if (ny > 1) {
    rc = nb_ensure_work(c, (size_t)ny * 3 * (size_t)n * sizeof(double));
    if (rc) return rc;
    part = reinterpret_cast<double*>(c->work);
    NB_HIP(c, hipMemsetAsync(part, 0, (size_t)ny * 3 * (size_t)n * sizeof(double), st));   // waves without work write nothing
  }
  const dim3 sr_grid((unsigned)((max_chunks + SRW - 1) / SRW), (unsigned)ny);
  const bool soft = eps2 > 0.0 && (sizeof(T) == 8 || (float)eps2 > 1e-30f), uni = c->uniform_mass;
#define NB_SR2(K, SF, UN) short_range_kernel<T, V4, K, SF, UN><<<sr_grid, 64 * SRW, 0, st>>>((const V4*)s.sorted, s.keys_s, \
    s.cell_start, s.chunk_off, nrows, nc, (T)L, (T)alpha, (T)(rcut * rcut), (T)eps2, G, s.idx_s, s.table, ax, ay, az, part, n, c->sr_rung, c->sr_min_rung)
#define NB_SR(K) do { if (soft && uni) NB_SR2(K, true, true); else if (soft) NB_SR2(K, true, false); else if (uni) NB_SR2(K, false, true); \
                      else NB_SR2(K, false, false); } while (0)
  if (ks == 1) NB_SR(1); else if (ks == 2) NB_SR(2); else NB_SR(4);
#undef NB_SR
#undef NB_SR2
  NB_HIP(c, hipGetLastError());
  if (ny > 1) {
    sr_reduce_kernel<T><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(part, ny, s.idx_s, n, G, ax, ay, az);
    NB_HIP(c, hipGetLastError());
  }
  return NB_OK;
}

// sort the particles by mesh brick and build the (brick, chunk) work list; returns the launch grid in max_chunks
template <typename T>
int brick_prepare(nb_ctx* c, nb_mesh_state& s, const T* x, const T* y, const T* z, int64_t n, double invh, int N, int& nbr, int& nb3,
                  int64_t& max_chunks) {
  nbr = (N + BR - 1) / BR;
  nb3 = nbr * nbr * nbr;
  int rc = ensure_cells(c, s, n, (int)sizeof(T), nb3);
  if (rc) return rc;
  hipStream_t st = c->stream;
  brick_key_kernel<T><<<(int)((n + 255) / 256), 256, 0, st>>>(x, y, z, n, invh, N, nbr, s.keys, s.idx);
  NB_HIP(c, hipGetLastError());
  int bits = 1; while ((1u << bits) < (unsigned)nb3) ++bits;
  size_t tb = s.tmp_bytes;
  NB_HIP(c, rocprim::radix_sort_pairs(s.tmp, tb, s.keys, s.keys_s, s.idx, s.idx_s, (size_t)n, 0, (unsigned)bits, st));
  cell_start_kernel<<<(int)((n + 1 + 255) / 256), 256, 0, st>>>(s.keys_s, n, nb3, 0, s.cell_start);
  NB_HIP(c, hipGetLastError());
  chunk_count_kernel<<<(nb3 + 1 + 255) / 256, 256, 0, st>>>(s.cell_start, nb3, 1, BCH, s.chunk_cnt);
  NB_HIP(c, hipGetLastError());
  tb = s.tmp_bytes;
  NB_HIP(c, rocprim::exclusive_scan(s.tmp, tb, s.chunk_cnt, s.chunk_off, 0, (size_t)nb3 + 1, rocprim::plus<int>(), st));
  max_chunks = (int64_t)nb3 + n / BCH + 1;
  if (s.chunk_info_cap < max_chunks) {
    NB_HIP(c, hipStreamSynchronize(st));
    if (s.chunk_info) { (void)hipFree(s.chunk_info); s.chunk_info = nullptr; s.chunk_info_cap = 0; }
    NB_HIP(c, hipMalloc((void**)&s.chunk_info, (size_t)(max_chunks + max_chunks / 8) * sizeof(int4)));
    s.chunk_info_cap = max_chunks + max_chunks / 8;
  }
  chunk_info_kernel<<<(unsigned)((max_chunks + 255) / 256), 256, 0, st>>>(s.cell_start, s.chunk_off, nb3, 1, BCH, (int)max_chunks, s.chunk_info);
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

template <typename T>
int long_range(nb_ctx* c, nb_mesh_state& s, const T* x, const T* y, const T* z, const T* m, int64_t n, double L, int ngrid, double rs,
               int p3m, int interlace, double G, T* ax, T* ay, T* az) {
  hipStream_t st = c->stream;
  const double h = L / ngrid;
  const size_t cells = (size_t)ngrid * ngrid * ngrid;
  // LDS-privatised bricks pay for their sort from ~10^5 particles on (NB_MESH_BRICKS=0/1 forces the choice)
  static const char* env = getenv("NB_MESH_BRICKS");
  const bool bricks = env && *env ? atoi(env) != 0 : n >= 100000;
  int nbr = 0, nb3 = 0;
  int64_t max_chunks = 0;
  if (bricks) { int rc = brick_prepare<T>(c, s, x, y, z, n, 1.0 / h, ngrid, nbr, nb3, max_chunks); if (rc) return rc; }
  // a mesh of whole bricks: the tiles go to one slot per chunk and a second kernel sums them into the grid (no atomics, no clearing);
  // NB_MESH_STAGE=0 keeps the atomic flush
  static const bool stage_env = !(getenv("NB_MESH_STAGE") && atoi(getenv("NB_MESH_STAGE")) == 0);
  // (and enough particles per brick: the staging buffer holds a 16 KB tile per chunk, every brick has at least one)
  const bool staged = bricks && stage_env && ngrid % BR == 0 && (int64_t)nb3 * 32 <= n;
  if (staged && s.stage_cap < max_chunks) {
    NB_HIP(c, hipStreamSynchronize(st));
    if (s.stage) { (void)hipFree(s.stage); s.stage = nullptr; s.stage_cap = 0; }
    NB_HIP(c, hipMalloc((void**)&s.stage, (size_t)(max_chunks + max_chunks / 8) * 2 * BT3 * sizeof(double)));
    s.stage_cap = max_chunks + max_chunks / 8;
  }
  if (!staged) NB_HIP(c, hipMemsetAsync(s.grid[0], 0, cells * sizeof(hipfftDoubleComplex), st));
  if (bricks) {
    // staged: the chunks of ordinary bricks leave their tiles in `stage`, brick_assemble_kernel writes every grid cell from them, and only then
    // the chunks of crowded bricks (strong clustering: summing thousands of tiles one after the other in the one block that owns the cells
    // would serialise the heaviest part) add theirs with atomics
    brick_deposit_kernel<T><<<(unsigned)max_chunks, 256, 0, st>>>(x, y, z, m, s.idx_s, s.chunk_info, nbr, 1.0 / h, ngrid, interlace,
                                                                 (double*)s.grid[0], staged ? s.stage : nullptr, s.chunk_off, staged ? 1 : 0);
    NB_HIP(c, hipGetLastError());
    if (staged) {
      brick_assemble_kernel<<<nb3, 256, 0, st>>>(s.stage, s.chunk_off, nbr, ngrid, interlace, (double2*)s.grid[0]);
      NB_HIP(c, hipGetLastError());
      if (n > (int64_t)BHEAVY * BCH) {                       // (no brick can be crowded with fewer particles than that)
        brick_deposit_kernel<T><<<(unsigned)max_chunks, 256, 0, st>>>(x, y, z, m, s.idx_s, s.chunk_info, nbr, 1.0 / h, ngrid, interlace,
                                                                     (double*)s.grid[0], nullptr, s.chunk_off, 2);
        NB_HIP(c, hipGetLastError());
      }
    }
  } else {
    for (int lat = 0; lat < (interlace ? 2 : 1); ++lat) {
      mesh_deposit_kernel<T><<<stream_grid(n), 256, 0, st>>>(x, y, z, m, n, 1.0 / h, ngrid, lat ? 0.5 : 0.0, lat, (double*)s.grid[0]);
      NB_HIP(c, hipGetLastError());
    }
  }
  M_FFT(c, hipfftExecZ2Z(s.fft, s.grid[0], s.grid[0], HIPFFT_FORWARD));
  const double fac = 4.0 * 3.14159265358979323846 * G / (L * L * L);
  mesh_kspace_kernel<<<(unsigned)((cells + 255) / 256), 256, 0, st>>>(s.grid[0], s.grid[1], s.grid[2], s.grid[3], ngrid, L, rs, fac,
                                                                      interlace, p3m);
  NB_HIP(c, hipGetLastError());
  for (int g = 1; g <= (interlace ? 3 : 2); ++g) M_FFT(c, hipfftExecZ2Z(s.fft, s.grid[g], s.grid[g], HIPFFT_BACKWARD));
  if (bricks)
    brick_gather_kernel<T><<<(unsigned)max_chunks, 256, 0, st>>>(x, y, z, s.idx_s, s.chunk_info, nbr, 1.0 / h, ngrid,
                                                                (const double2*)s.grid[1], (const double2*)s.grid[2],
                                                                (const double2*)s.grid[3], interlace, ax, ay, az);
  else
    mesh_gather_kernel<T><<<stream_grid(n), 256, 0, st>>>(x, y, z, n, 1.0 / h, ngrid, (const double*)s.grid[1], (const double*)s.grid[2],
                                                          (const double*)s.grid[3], interlace, ax, ay, az);
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

}  // namespace

// The cell-list pair sum on its own — the real-space part of an Ewald sum with a cut-off (ewald.hip): ADDS
// G m_j d [ (r^2+eps^2)^{-3/2} - (erf(alpha r) - 2 alpha r/sqrt(pi) e^{-alpha^2 r^2}) / r^3 ] over r < rcut to ax, ay, az.
int nb_launch_short_range(nb_ctx* c, int dtype, const void* x, const void* y, const void* z, const void* m, int64_t n, double L,
                          double alpha, double rcut, double eps2, double G, void* ax, void* ay, void* az) {
  if (n <= 0) return NB_OK;
  const int nc = (int)floor(L / rcut);
  if (nc < 3) return nb_fail(c, NB_ERR_ARG, "short-range sum: the cut-off exceeds a third of the box");
  if (!c->mesh) c->mesh = new nb_mesh_state();
  if (dtype == NB_F32)
    return short_range<float, float4>(c, *c->mesh, (const float*)x, (const float*)y, (const float*)z, (const float*)m, n, L, nc, alpha, rcut,
                                      eps2, G, (float*)ax, (float*)ay, (float*)az);
  return short_range<double, d4>(c, *c->mesh, (const double*)x, (const double*)y, (const double*)z, (const double*)m, n, L, nc, alpha, rcut,
                                 eps2, G, (double*)ax, (double*)ay, (double*)az);
}

void nb_mesh_release(nb_ctx* c) {
  if (!c->mesh) return;
  nb_mesh_state& s = *c->mesh;
  for (auto& g : s.grid) if (g) (void)hipFree(g);
  if (s.have_fft) (void)hipfftDestroy(s.fft);
  for (void* p : {(void*)s.keys, (void*)s.keys_s, (void*)s.idx, (void*)s.idx_s, s.sorted, (void*)s.cell_start,
                  (void*)s.chunk_cnt, (void*)s.chunk_off, s.tmp, (void*)s.table, (void*)s.chunk_info, (void*)s.ltab, (void*)s.stage})
    if (p) (void)hipFree(p);
  delete c->mesh;
  c->mesh = nullptr;
}

// parts: bit 0 = long range (mesh), bit 1 = short range (cells). acc is overwritten.
int nb_launch_accel_mesh(nb_ctx* c, int dtype, const void* x, const void* y, const void* z, const void* m, int64_t n,
                         double L, int ngrid, double rs_cells, double rcut_rs, int p3m, int interlace, int parts,
                         double eps2, double G, void* ax, void* ay, void* az) {
  if (n <= 0) return NB_OK;
  if (n >= (1ll << 31)) return nb_fail(c, NB_ERR_ARG, "mesh solver: n must be < 2^31");
  if (ngrid < 4 || ngrid > 2048) return nb_fail(c, NB_ERR_ARG, "mesh solver: grid size must be in [4, 2048]");
  if (!c->mesh) c->mesh = new nb_mesh_state();
  nb_mesh_state& s = *c->mesh;
  hipStream_t st = c->stream;
  const double h = L / ngrid, rs = rs_cells * h, rcut = rcut_rs * rs;
  const size_t esz = dtype == NB_F32 ? 4 : 8;
  if (parts & 1) {
    int rc = ensure_grids(c, s, ngrid);
    if (rc) return rc;
    if (dtype == NB_F32)
      rc = long_range<float>(c, s, (const float*)x, (const float*)y, (const float*)z, (const float*)m, n, L, ngrid, rs, p3m, interlace, G,
                             (float*)ax, (float*)ay, (float*)az);
    else
      rc = long_range<double>(c, s, (const double*)x, (const double*)y, (const double*)z, (const double*)m, n, L, ngrid, rs, p3m, interlace,
                              G, (double*)ax, (double*)ay, (double*)az);
    if (rc) return rc;
  } else {
    NB_HIP(c, hipMemsetAsync(ax, 0, (size_t)n * esz, st));
    NB_HIP(c, hipMemsetAsync(ay, 0, (size_t)n * esz, st));
    NB_HIP(c, hipMemsetAsync(az, 0, (size_t)n * esz, st));
  }
  if (p3m && (parts & 2)) {
    const int nc = (int)floor(L / rcut);
    if (nc < 3) return nb_fail(c, NB_ERR_ARG, "p3m: the short-range cut-off exceeds a third of the box (grid too coarse); use the Ewald solver");
    if ((double)nc * nc * nc > 6.0e7) return nb_fail(c, NB_ERR_ARG, "p3m: too many short-range cells");
    const double alpha = 1.0 / (2.0 * rs);
    if (dtype == NB_F32)
      return short_range<float, float4>(c, s, (const float*)x, (const float*)y, (const float*)z, (const float*)m, n, L, nc, alpha,
                                        rcut, eps2, G, (float*)ax, (float*)ay, (float*)az);
    return short_range<double, d4>(c, s, (const double*)x, (const double*)y, (const double*)z, (const double*)m, n, L, nc, alpha,
                                   rcut, eps2, G, (double*)ax, (double*)ay, (double*)az);
  }
  return NB_OK;
}

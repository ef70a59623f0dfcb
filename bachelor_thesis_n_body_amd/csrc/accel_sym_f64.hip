// fp64 softened all-pairs acceleration of a particle set ON ITSELF, each unordered pair evaluated once — the fp64 twin of
// accel_sym_f32.hip (same skewed walk, same slabs; SURVEY.md §8(a) row ⊕1, configs[3]).
//
// accel_f64.hip spends 13 fp64 VALU + the v_rsq_f64 seed per ORDERED pair. Here a pair costs 3 sub, 3 fma (r^2), the seed, 4 for
// (2/3) r^-3 (accel_f64.hip's first-order refinement: u = y*y; q = fma(-r2, u, 5/3); f = (u*y)*q), 3 fma for the sink and 3 fma for the
// source: 16 + seed per UNORDERED pair = 8 + 1/2 per ordered pair (per-particle masses: two more multiplies per unordered pair).
//  * sets of 256*KS particles; block (b, y) holds set b as sinks (KS per lane) and walks the sets b + d for the offsets d of its group
//    (d = 1 .. (nsets-1)/2 two-sided; d = 0 and, for an even set count, d = nsets/2 one-sided);
//  * in step t lane l meets source (t + l) mod TJ of the staged tile — 64 different sources per instruction (per-lane ds_read, the tile
//    carries 64 wrap-around copies) — adds the pair term to its sinks in registers and the opposite term to the source's running sum in
//    LDS by a plain read / add / write: 64 different addresses per instruction, a private accumulator copy per wave, and the entry lane l
//    updates in step t is the one lane l + 1 updated in step t - 1 (LDS operations of a wave complete in order): exact, bit-reproducible;
//  * per tile the waves' copies are added in a fixed order and stored to an fp64 slab [offset][3][n]; the sinks' sums go to a slab
//    [group][3][n]; a second kernel adds everything in a fixed order. No atomics anywhere.
#include "nb_common.h"
#include <cmath>
#include "../../include/nbody_amd.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr double P_SCALE = 0.5;     // and the 1/2 of the potential's (r^-1 = y (3 - r2 y^2) / 2)
constexpr double F_SCALE = 1.5;     // the 3/2 of the first-order rsqrt refinement, applied with G in the epilogue (see accel_f64.hip)
struct alignas(16) d4 { double x, y, z, w; };
struct alignas(16) d3 { double x, y, z; };

template <bool UNIFORM> __device__ inline d4 ld_src(const d4* p) {
  if (UNIFORM) { const d3 v = *reinterpret_cast<const d3*>(p); return d4{v.x, v.y, v.z, 0.0}; }
  return *p;
}

// ---- how the work of a launch is dealt to the blocks: equal shares of STEPS, exactly as in accel_sym_f32.hip (see the comment there) -------
__host__ __device__ inline long long part_begin(long long k, long long G, int B) { return k >= B ? G : ((k * G / B) & ~1LL); }
__host__ __device__ inline int part_block_of(long long s, long long G, int B) {       // the block whose range holds step s
  int k = (int)(s * B / G);
  if (k > B - 1) k = B - 1;
  if (k + 1 < B && part_begin(k + 1, G, B) <= s) ++k;
  return k;
}
__device__ inline int sym_set_of(int b, int o, int cross, int nsets, int nsets_src, int H) {
  if (cross) return (b + o) % nsets_src;
  const int d = o <= H ? o : nsets / 2; const int c = b + d; return c >= nsets ? c - nsets : c;
}

template <int KS, int TJ, bool GUARD, bool UNIFORM, bool POT>
__global__ void __launch_bounds__(BLOCK, 2)      // two blocks per CU (two waves per SIMD): at most 256 registers, AGPRs included
accel_sym_f64_kernel(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z, const double* __restrict__ m,
                     int n, int nsets, int H, int o_begin, int tps, long long G, double eps2,
                     double* __restrict__ part, double* __restrict__ slab, double* __restrict__ bslab, size_t n_pad,
                     const double* __restrict__ xs, const double* __restrict__ ys, const double* __restrict__ zs, const double* __restrict__ ms,
                     int n_src, int nsets_src, size_t n_pad_src, int cross, int b0, int turn_shift) {
  // cross = 0: the set on itself (xs.. are the same arrays); cross = 1: block A = (x..) against a different block B = (xs..), every A-B pair
  // two-sided, B's sums to slab slot b - b0 (see accel_sym_f32.hip). The launch's steps g = ((bb * tps + tau) * TJ + t) are dealt to the
  // blocks in equal shares; a tile cut by a block boundary leaves the second block's share in the bslab.
  static_assert(TJ % BLOCK == 0 && TJ % 2 == 0, "shape");
  constexpr int SET = BLOCK * KS;
  constexpr int AS = TJ + 64;
  constexpr int LPT = TJ / BLOCK;
  constexpr int QT = SET / TJ;
  constexpr int HS = TJ;             // steps per tile (one source per step)
  static_assert(SET % TJ == 0, "a set is a whole number of tiles");
  constexpr int NC = POT ? 4 : 3;    // POT: the pair potential rides along as a fourth component on both sides
  __shared__ d4 tile[AS];
  __shared__ double acc[WAVES][NC][AS];
  __shared__ unsigned turn_slots[WAVES];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double far = 1e150;          // UNIFORM padding: y = r2^-1/2 ~ 1e-150, y^3 underflows to exactly 0
  const int nblk = gridDim.x, kb = blockIdx.x;
  const long long g_begin = part_begin(kb, G, nblk), g_end = part_begin(kb + 1, G, nblk);
  const long long S = (long long)tps * HS;

  double px[KS], py[KS], pz[KS], pm[KS], fx[KS], fy[KS], fz[KS], fp[KS];
  auto load_sinks = [&](int b) {
    const int i0 = b * SET + tid;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int i = i0 + k * BLOCK;
      const bool ok = i < n;
      const int ic = ok ? i : n - 1;
      if (UNIFORM) { px[k] = ok ? x[ic] : far; py[k] = ok ? y[ic] : far; pz[k] = ok ? z[ic] : far; pm[k] = 0.0; }
      else { px[k] = x[ic]; py[k] = y[ic]; pz[k] = z[ic]; pm[k] = ok ? m[ic] : 0.0; }   // padded sinks: zero mass, they push nothing
      fx[k] = 0.0; fy[k] = 0.0; fz[k] = 0.0; fp[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) asm volatile("" : "+v"(px[k]), "+v"(py[k]), "+v"(pz[k]), "+v"(pm[k]));
  };
  auto store_sinks = [&](int b, int seg) {
    double* pp = part + (size_t)seg * NC * n_pad;
    const int i0 = b * SET + tid;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int i = i0 + k * BLOCK;
      if (i < n) { pp[i] = fx[k]; pp[n_pad + i] = fy[k]; pp[2 * n_pad + i] = fz[k]; if (POT) pp[3 * n_pad + i] = fp[k]; }
    }
  };
  double e2 = eps2;
  asm volatile("" : "+v"(e2));

  for (int e = tid; e < WAVES * NC * AS; e += BLOCK) (&acc[0][0][0])[e] = 0.0;

  d4 stage[LPT];
  auto fetch = [&](long long tile_id) {
    const int bb = (int)(tile_id / tps), tau = (int)(tile_id - (long long)bb * tps);
    const int c = sym_set_of(b0 + bb, o_begin + tau / QT, cross, nsets, nsets_src, H), q = tau % QT;
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int j = c * SET + q * TJ + l * BLOCK + tid;
      const bool ok = j < n_src;
      const int jc = ok ? j : n_src - 1;
      stage[l] = UNIFORM ? d4{ok ? xs[jc] : far, ok ? ys[jc] : far, ok ? zs[jc] : far, 0.0}
                         : d4{xs[jc], ys[jc], zs[jc], ok ? ms[jc] : 0.0};
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int l = 0; l < LPT; ++l) tile[l * BLOCK + tid] = stage[l];
    if (tid < 64) tile[TJ + tid] = stage[0];
  };

  unsigned wave_slot;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(wave_slot));      // wave slot within its SIMD: the block's turn id below
  if (lane == 0) turn_slots[wv] = wave_slot;
  if (g_begin < g_end) { fetch(g_begin / HS); commit(); }
  __syncthreads();

  const d4* tl = tile + lane;
  double* ac = &acc[wv][0][lane];
  // the two co-resident waves of a SIMD take turns at the raised issue priority in time slices of the constant-rate counter (accel_sym_f32.hip)
  // The turn is the BLOCK's, not the wave's: all four waves of a block must be favoured in the same slices (they meet at every tile's barrier,
  // so a block runs at the pace of its slowest wave). The block's id for the turns is the smallest wave-slot id among its waves: where the
  // CU holds nothing else that is 0 for the first block and 1 for the second on every SIMD; where a third wave sits on one SIMD (bench.py's
  // clock probe: one sleeping wave per XCD) the per-wave ids there are 1 and 2 against 0 and 1 elsewhere, and per-WAVE turns put one wave of
  // each block out of phase with its three partners — that CU then straggled by 6 % and, under a deal with few rounds, so did the launch
  // (bench.py with the probe: +4-8 %; without: -3 % ... 0; profiles/r04_priority_turns_between_coresident_waves.txt).
  unsigned slot_id = min(min(turn_slots[0], turn_slots[1]), min(turn_slots[2], turn_slots[3]));
  slot_id = __builtin_amdgcn_readfirstlane(slot_id);

  auto walk = [&](auto symtag, int t_lo, int t_hi) {
    constexpr bool SYM = decltype(symtag)::value;
    auto step = [&](const d4& s, d4& nx, int t) {
      double gx = 0.0, gy = 0.0, gz = 0.0, gp = 0.0;
      asm volatile("" ::: "memory");     // no LDS access may cross a step boundary (lane l reads what lane l + 1 wrote one step earlier)
      nx = ld_src<UNIFORM>(tl + t + 1);
      if (SYM) { gx = ac[0 * AS + t]; gy = ac[1 * AS + t]; gz = ac[2 * AS + t]; if (POT) gp = ac[3 * AS + t]; }
      double dx[KS], dy[KS], dz[KS], f[KS], pr[KS];
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        dx[k] = s.x - px[k]; dy[k] = s.y - py[k]; dz[k] = s.z - pz[k];
        const double r2 = fma(dx[k], dx[k], fma(dy[k], dy[k], fma(dz[k], dz[k], e2)));
        const double yv = __builtin_amdgcn_rsq(r2);
        const double u = yv * yv;
        const double q = fma(-r2, u, 5.0 / 3.0);
        f[k] = (u * yv) * q;                         // (2/3) r^-3
        if (POT) pr[k] = yv * fma(-r2, u, 3.0);      // 2 r^-1
        if (GUARD) { const bool apart = r2 > e2; f[k] = apart ? f[k] : 0.0; if (POT) pr[k] = apart ? pr[k] : 0.0; }   // coincident pair at eps = 0: nothing
        const double fi = UNIFORM ? f[k] : f[k] * s.w;
        fx[k] = fma(fi, dx[k], fx[k]); fy[k] = fma(fi, dy[k], fy[k]); fz[k] = fma(fi, dz[k], fz[k]);
        if (POT) fp[k] = UNIFORM ? fp[k] + pr[k] : fma(pr[k], s.w, fp[k]);
      }
      if (SYM) {
        __builtin_amdgcn_sched_barrier(0);           // the source-side chains (which wait for the accumulator reads) stay behind the sink-side work
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const double fj = UNIFORM ? f[k] : f[k] * pm[k];
          gx = fma(-fj, dx[k], gx); gy = fma(-fj, dy[k], gy); gz = fma(-fj, dz[k], gz);
          if (POT) gp = UNIFORM ? gp + pr[k] : fma(pr[k], pm[k], gp);
        }
        ac[0 * AS + t] = gx; ac[1 * AS + t] = gy; ac[2 * AS + t] = gz; if (POT) ac[3 * AS + t] = gp;
      }
    };
    d4 sa = ld_src<UNIFORM>(tl + t_lo), sb;
    for (int tc = t_lo; tc < t_hi; tc += 8) {
      const int te = tc + 8 < t_hi ? tc + 8 : t_hi;
      unsigned long long now = 0;
      if (turn_shift > 0) now = __builtin_amdgcn_s_memrealtime();
      for (int t = tc; t < te; t += 2) { step(sa, sb, t); step(sb, sa, t + 1); }
      if (turn_shift > 0) { if (((now >> turn_shift) ^ slot_id) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    }
  };

  int cur = -1, seg = 0;
  for (long long g = g_begin; g < g_end;) {
    const long long tile_id = g / HS;
    const int t_lo = (int)(g - tile_id * HS);
    const int t_hi = (int)((long long)HS < t_lo + (g_end - g) ? (long long)HS : t_lo + (g_end - g));
    const int bb = (int)(tile_id / tps), tau = (int)(tile_id - (long long)bb * tps);
    const int o = o_begin + tau / QT, q = tau % QT;
    const bool sym = cross || (o >= 1 && o <= H);
    if (bb != cur) {
      if (cur >= 0) store_sinks(b0 + cur, seg);
      seg = cur < 0 ? kb - part_block_of((long long)bb * S, G, nblk) : 0;
      cur = bb;
      load_sinks(b0 + bb);
    }
    const long long g_next = g + (t_hi - t_lo);
    if (g_next < g_end) fetch(g_next / HS);
    if (sym) walk(std::true_type{}, t_lo, t_hi); else walk(std::false_type{}, t_lo, t_hi);
    __syncthreads();
    if (sym) {
      const int c = sym_set_of(b0 + bb, o, cross, nsets, nsets_src, H);
      double* sl = t_lo == 0 ? slab + (size_t)(cross ? bb : o - max(o_begin, 1)) * NC * n_pad_src + (size_t)c * SET + (size_t)q * TJ
                             : bslab + (size_t)kb * NC * TJ;
      const size_t cstride = t_lo == 0 ? n_pad_src : (size_t)TJ;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
        for (int e = tid; e < TJ; e += BLOCK) {
          double v = 0.0;
#pragma unroll
          for (int w2 = 0; w2 < WAVES; ++w2) { v += acc[w2][cc][e]; acc[w2][cc][e] = 0.0; }
          if (e < 64) {
#pragma unroll
            for (int w2 = 0; w2 < WAVES; ++w2) { v += acc[w2][cc][TJ + e]; acc[w2][cc][TJ + e] = 0.0; }
          }
          sl[(size_t)cc * cstride + e] = v;
        }
    }
    if (g_next < g_end) commit();
    __syncthreads();
    g = g_next;
  }
  if (cur >= 0) store_sinks(b0 + cur, seg);
}

// two-block launches: slab entry of a tile cut by a block boundary += the share of the block that started inside it (a set on itself: reduce kernel)
template <int KS, int TJ, bool POT>
__global__ void __launch_bounds__(BLOCK)
accel_sym_boundary_f64_kernel(double* __restrict__ slab, const double* __restrict__ bslab, int nsets, int H, int o_begin, int tps, long long G,
                              int nsets_src, size_t n_pad_src, int cross, int b0) {
  constexpr int NC = POT ? 4 : 3, SET = BLOCK * KS, QT = SET / TJ, HS = TJ;
  const int kb = blockIdx.x, nblk = gridDim.x;
  const long long g = part_begin(kb, G, nblk);
  if (g >= part_begin(kb + 1, G, nblk)) return;
  const long long tile_id = g / HS;
  if (g == tile_id * HS) return;
  const int bb = (int)(tile_id / tps), tau = (int)(tile_id - (long long)bb * tps);
  const int o = o_begin + tau / QT, q = tau % QT;
  if (!(cross || (o >= 1 && o <= H))) return;
  const int c = sym_set_of(b0 + bb, o, cross, nsets, nsets_src, H);
  double* sl = slab + (size_t)(cross ? bb : o - max(o_begin, 1)) * NC * n_pad_src + (size_t)c * SET + (size_t)q * TJ;
  const double* bs = bslab + (size_t)kb * NC * TJ;
  for (int cc = 0; cc < NC; ++cc)
    for (int e = threadIdx.x; e < TJ; e += BLOCK) sl[(size_t)cc * n_pad_src + e] += bs[cc * TJ + e];
}

// cut tiles finished by the reduce kernel and the integrator tail it may apply: see accel_sym_f32.hip
struct SymCut { const double* bslab; int nsets, o_begin, s_begin, tps, tj, hs; };
struct SymTail { double *vx, *vy, *vz, *x, *y, *z; double kick_close, kick_open, drift; int advance; };

// a[c][i] (+)= g * ( sum_seg part[seg][c][i] + sum_s slab[s][c][i] ), fixed order; several passes collect in `tot` first. The part slots of
// a particle are those of the blocks that shared its sink set (nblk = 0: no part slab); cut.bslab non-null: the share of a block that started
// inside the particle's tile is added to the slab entry here; tail.vx non-null: the leapfrog tail (nb_ctx::Tail) — see accel_sym_f32.hip
__global__ void __launch_bounds__(256)
accel_sym_reduce_f64_kernel(const double* __restrict__ part, long long Gs, int nblk, long long S, int set_size, int b0,
                            const double* __restrict__ slab, int nslots, int n, size_t n_pad,
                            double G, const double* __restrict__ m0, double* __restrict__ tot, int first, int last,
                            double* __restrict__ ax, double* __restrict__ ay, double* __restrict__ az, int accumulate, double* __restrict__ phi,
                            int i_begin, SymCut cut, SymTail tail) {
  __shared__ int cut_block[256];
  const int nc = phi ? 4 : 3;
  const int i0 = i_begin + blockIdx.x * blockDim.x;                    // particles [i_begin, n)
  const int i = i0 + threadIdx.x;
  const bool live = i < n;
  double sx = 0.0, sy = 0.0, sz = 0.0, sp = 0.0;
  if (live && nblk > 0) {
    const long long bb = i / set_size - b0;
    const int nseg = part_block_of((bb + 1) * S - 1, Gs, nblk) - part_block_of(bb * S, Gs, nblk) + 1;
    for (int g = 0; g < nseg; ++g) {
      const double* p = part + (size_t)g * nc * n_pad;
      sx += p[i]; sy += p[n_pad + i]; sz += p[2 * n_pad + i];
      if (phi) sp += p[3 * n_pad + i];
    }
  }
  if (!cut.bslab) {
    if (live) {
#pragma unroll 8
      for (int s = 0; s < nslots; ++s) {
        const double* p = slab + (size_t)s * nc * n_pad;
        sx += p[i]; sy += p[n_pad + i]; sz += p[2 * n_pad + i];
        if (phi) sp += p[3 * n_pad + i];
      }
    }
  } else {
    const int c = i0 / set_size, r0 = i0 - c * set_size, q = r0 / cut.tj, e = r0 - q * cut.tj + (int)threadIdx.x;
    for (int s0 = 0; s0 < nslots; s0 += 256) {
      const int ns = nslots - s0 < 256 ? nslots - s0 : 256;
      if (s0 > 0) __syncthreads();
      if ((int)threadIdx.x < ns) {
        const int o = cut.s_begin + s0 + (int)threadIdx.x;
        int bb = c - o; if (bb < 0) bb += cut.nsets;
        const long long g0 = ((long long)bb * cut.tps + (long long)(o - cut.o_begin) * (set_size / cut.tj) + q) * cut.hs;
        const int k0 = part_block_of(g0, Gs, nblk);
        cut_block[threadIdx.x] = part_begin(k0 + 1, Gs, nblk) < g0 + cut.hs ? k0 + 1 : -1;
      }
      __syncthreads();
      if (live) {
#pragma unroll 4
        for (int s = 0; s < ns; ++s) {
          const double* p = slab + (size_t)(s0 + s) * nc * n_pad;
          double vx = p[i], vy = p[n_pad + i], vz = p[2 * n_pad + i], vp = phi ? p[3 * n_pad + i] : 0.0;
          const int kb = cut_block[s];
          if (kb >= 0) {
            const double* bs = cut.bslab + (size_t)kb * nc * cut.tj + e;
            vx += bs[0]; vy += bs[cut.tj]; vz += bs[2 * cut.tj];
            if (phi) vp += bs[3 * cut.tj];
          }
          sx += vx; sy += vy; sz += vz;
          if (phi) sp += vp;
        }
      }
    }
  }
  if (!live) return;
  if (tot) {
    if (!first) { sx += tot[i]; sy += tot[(size_t)n + i]; sz += tot[2 * (size_t)n + i]; if (phi) sp += tot[3 * (size_t)n + i]; }
    if (!last) { tot[i] = sx; tot[(size_t)n + i] = sy; tot[2 * (size_t)n + i] = sz; if (phi) tot[3 * (size_t)n + i] = sp; return; }
  }
  if (phi) { const double ph = P_SCALE * (m0 ? m0[0] * sp : sp); phi[i] = accumulate ? phi[i] + ph : ph; }   // per unit G, self term included
  const double g = F_SCALE * (m0 ? G * m0[0] : G);
  double ox = g * sx, oy = g * sy, oz = g * sz;
  if (accumulate) { ox += ax[i]; oy += ay[i]; oz += az[i]; }
  ax[i] = ox; ay[i] = oy; az[i] = oz;
  if (tail.vx) {                                            // as kick_kernel / kick_drift_kernel (integrate.hip): products and sums rounded one by one
#pragma clang fp contract(off)
    double v0 = tail.vx[i] + ox * tail.kick_close;
    double v1 = tail.vy[i] + oy * tail.kick_close;
    double v2 = tail.vz[i] + oz * tail.kick_close;
    if (tail.advance) {
      v0 = v0 + ox * tail.kick_open; v1 = v1 + oy * tail.kick_open; v2 = v2 + oz * tail.kick_open;
      tail.x[i] = tail.x[i] + v0 * tail.drift;
      tail.y[i] = tail.y[i] + v1 * tail.drift;
      tail.z[i] = tail.z[i] + v2 * tail.drift;
    }
    tail.vx[i] = v0; tail.vy[i] = v1; tail.vz[i] = v2;
  }
}

// Slice length of the priority turns (the kernel's turn_shift: slices of 2^shift ticks of the 100 MHz counter) for blocks expected to run for
// `block_ticks`: a third of that, to the nearest power of two. Measured (bench.py --n N on one box, profiles/r04_turn_shift_sweep.txt): the best
// slice grows with the launch — 2^11..12 at N = 24 576 (0.15 ms), 2^12 at 32 768, 2^14 at 49 152 ... 65 536, 2^15 at 98 304 ... 131 072: a
// short launch needs several turns to be fair at all, a long one loses a little with every change of hands. Launches of many rounds are
// indifferent (a CU always holds two blocks there).
int auto_turn_shift(double block_ticks) {
  const int s = (int)std::lround(std::log2(std::max(block_ticks / 3.0, 1.0)));
  return std::min(22, std::max(10, s));
}

long long env_ll(const char* name, long long dflt) {
  const char* v = getenv(name);
  return v && *v ? atoll(v) : dflt;
}

// blocks of a launch: the device's resident block slots for this kernel x rounds, at most one per tile (accel_sym_f32.hip: sym_blocks)
template <typename K>
int sym_blocks(nb_ctx* c, K kernel, long long tiles, long long env_blocks, long long env_rounds) {
  // resident block slots of this kernel on this device: asked of the runtime once per (kernel, device), not on every launch
  struct Slot { const void* k; int dev; long long slots; };
  static thread_local std::vector<Slot> cache;
  long long slots = 0;
  for (const Slot& e : cache) if (e.k == (const void*)kernel && e.dev == c->device) { slots = e.slots; break; }
  if (!slots) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLOCK, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus < 1) { (void)hipGetLastError(); cus = 256; }
    slots = (long long)per_cu * cus;
    cache.push_back(Slot{(const void*)kernel, c->device, slots});
  }
  // rounds: 1 is the static optimum when every slot runs at the same speed. They do not: the XCDs of one board hold clocks 3.5 - 6 % apart, and
  // under a static deal the slowest XCD sets the launch time. Before the co-resident waves took turns at the issue priority (see the kernel) finer
  // shares also evened out the unfair sharing of a CU (N = 2^20: 163.4 ms with 1 round, 160.0 / 158.8 / 157.6 with 2 / 4 / 8). With the turns: on a
  // box whose XCDs differ by 2 % the round count hardly matters at N = 2^20 (157.9 - 158.7 ms for 1 ... 8) and one round is 1 - 2.5 % faster up
  // to N ~ 500 000 (profiles/r04_sym_rounds_sweep_with_priority_turns.txt); on a box whose XCDs differ by 6 % five rounds at N = 2^20 cost 5 %
  // against the eight of the record before (profiles/r04_bench_n1_driver_form_5_rounds_6pct_xcd_spread.json). So: as many rounds as leave 16
  // tiles per block, at most 8 — large launches tolerate the clock spread, small ones keep their few tiles whole.
  const long long rounds = env_rounds > 0 ? env_rounds : std::min<long long>(8, std::max<long long>(1, tiles / (16 * slots)));
  long long nblk = env_blocks > 0 ? env_blocks : slots * rounds;
  if (nblk > tiles) nblk = tiles;
  if (nblk > 65535 * 16) nblk = 65535 * 16;
  return (int)std::max<long long>(1, nblk);
}

template <int KS, int TJ, bool POT>
auto sym_kernel(bool guard, bool uniform) {
  if (guard) return uniform ? accel_sym_f64_kernel<KS, TJ, true, true, POT> : accel_sym_f64_kernel<KS, TJ, true, false, POT>;
  return uniform ? accel_sym_f64_kernel<KS, TJ, false, true, POT> : accel_sym_f64_kernel<KS, TJ, false, false, POT>;
}

template <int KS, bool POT>
int run_sym(nb_ctx* c, const double* x, const double* y, const double* z, const double* m, int64_t n,
            double eps2, double G, double* ax, double* ay, double* az, int flags, long long budget, long long env_blocks, long long env_rounds, int turn_shift, double* phi) {
  constexpr int TJ = 256, SET = BLOCK * KS, NC = POT ? 4 : 3;
  constexpr int QT = SET / TJ, HS = TJ;
  const int nsets = (int)((n + SET - 1) / SET);
  const size_t n_pad = (size_t)nsets * SET;
  const int H = (nsets - 1) / 2;
  const int n_o = H + 1 + (nsets % 2 == 0 ? 1 : 0);
  const size_t slot_bytes = NC * n_pad * sizeof(double);
  // clamped in 64 bits BEFORE the cast: a budget meant as "unlimited" (INT64_MAX) must not wrap
  int slots_per_pass = (int)std::min<long long>(std::max<long long>(1, budget / (long long)slot_bytes), (long long)std::max(H, 1));
  const bool uniform = (flags & NB_UNIFORM_MASS) != 0;
  const bool guard = !(eps2 > 0.0);
  const int accumulate = flags & NB_ACCUMULATE;
  const auto kernel = sym_kernel<KS, TJ, POT>(guard, uniform);
  int npass, nblk_max, nseg_max;
  size_t part_bytes, bslab_bytes, slab_bytes, tot_bytes;
  for (;;) {                                                 // the allocator may refuse the scratch: fewer slots per pass, then give up (one-sided kernel)
    npass = H > 0 ? (H + slots_per_pass - 1) / slots_per_pass : 1;
    nblk_max = sym_blocks(c, kernel, (long long)nsets * std::min(n_o, slots_per_pass + 2) * QT, env_blocks, env_rounds);
    nseg_max = nblk_max / nsets + 2;
    part_bytes = (size_t)nseg_max * NC * n_pad * sizeof(double);
    bslab_bytes = (size_t)nblk_max * NC * TJ * sizeof(double);
    slab_bytes = (size_t)slots_per_pass * slot_bytes;
    tot_bytes = npass > 1 ? NC * (size_t)n * sizeof(double) : 0;
    const int rc = nb_try_ensure_work(c, part_bytes + bslab_bytes + slab_bytes + tot_bytes, c->sym_scratch_limit);
    if (rc == NB_OK) break;
    if (rc != NB_WORK_UNAVAILABLE || slots_per_pass <= 1) return rc;
    slots_per_pass = (slots_per_pass + 1) / 2;
  }
  char* w = reinterpret_cast<char*>(c->work);
  double* part = reinterpret_cast<double*>(w);
  double* bslab = reinterpret_cast<double*>(w + part_bytes);
  double* slab = reinterpret_cast<double*>(w + part_bytes + bslab_bytes);
  double* tot = npass > 1 ? reinterpret_cast<double*>(w + part_bytes + bslab_bytes + slab_bytes) : nullptr;
  for (int ps = 0; ps < npass; ++ps) {
    const int s_begin = 1 + ps * slots_per_pass, s_end = 1 + std::min(H, (ps + 1) * slots_per_pass);
    const int o_begin = ps == 0 ? 0 : s_begin;
    const int o_end = (ps == npass - 1) ? n_o : s_end;
    const int tps = (o_end - o_begin) * QT;                   // tiles per sink set in this pass
    const long long Gs = (long long)nsets * tps * HS;        // steps of this pass
    const int nblk = std::min(nblk_max, sym_blocks(c, kernel, (long long)nsets * tps, env_blocks, env_rounds));
    const int nslots = std::max(0, s_end - s_begin);
    if (nblk / nsets + 2 > nseg_max) return nb_fail(c, NB_ERR_ARG, "accel_sym_f64: internal slot count");
    kernel<<<nblk, BLOCK, 0, c->stream>>>(x, y, z, m, (int)n, nsets, H, o_begin, tps, Gs, eps2, part, slab, bslab, n_pad,
                                          x, y, z, m, (int)n, nsets, n_pad, 0, 0,
                                          turn_shift >= 0 ? turn_shift : auto_turn_shift((double)Gs / nblk * KS * 8.1));
    NB_HIP(c, hipGetLastError());
    // the tiles a block boundary cut are finished inside the reduce kernel; the last pass also applies the integrator tail, if one is armed
    const SymCut cut{nslots > 0 ? bslab : nullptr, nsets, o_begin, s_begin, tps, TJ, HS};
    SymTail tail{};
    if (c->tail.armed && ps == npass - 1) {
      tail = SymTail{(double*)c->tail.vx, (double*)c->tail.vy, (double*)c->tail.vz, (double*)c->tail.x, (double*)c->tail.y, (double*)c->tail.z,
                     c->tail.kick_close, c->tail.kick_open, c->tail.drift, c->tail.advance ? 1 : 0};
      c->tail.done = true;
    }
    accel_sym_reduce_f64_kernel<<<(int)((n + 255) / 256), 256, 0, c->stream>>>(part, Gs, nblk, (long long)tps * HS, SET, 0, slab, nslots, (int)n, n_pad, G,
                                                                                 uniform ? m : nullptr, tot, ps == 0, ps == npass - 1,
                                                                                 ax, ay, az, accumulate, phi, 0, cut, tail);
    NB_HIP(c, hipGetLastError());
  }
  return NB_OK;
}

// Block A against a different block B, every A-B pair once (see accel_sym_f32.hip: run_cross)
template <int KS, bool POT>
int run_cross(nb_ctx* c, const double* xa, const double* ya, const double* za, const double* ma, int64_t na,
              const double* xb, const double* yb, const double* zb, const double* mb, int64_t nb,
              double eps2, double G, double* aax, double* aay, double* aaz, double* bax, double* bay, double* baz,
              double* phi_a, double* phi_b, int flags, long long budget, long long env_blocks, long long env_rounds, int turn_shift) {
  constexpr int TJ = 256, SET = BLOCK * KS, NC = POT ? 4 : 3;
  constexpr int QT = SET / TJ, HS = TJ;
  const int nsa = (int)((na + SET - 1) / SET), nsb = (int)((nb + SET - 1) / SET);
  const size_t npa = (size_t)nsa * SET, npb = (size_t)nsb * SET;
  const size_t slot_bytes = NC * npb * sizeof(double);
  int sets_per_pass = (int)std::min<long long>(std::max<long long>(1, budget / (long long)slot_bytes), (long long)std::max(nsa, 1));
  const bool uniform = (flags & NB_UNIFORM_MASS) != 0;
  const bool guard = !(eps2 > 0.0);
  const auto kernel = sym_kernel<KS, TJ, POT>(guard, uniform);
  const int tps = nsb * QT;                                   // every set of B, tile by tile
  int nblk_max, nseg_max;
  size_t part_bytes, bslab_bytes;
  for (;;) {                                                 // as run_sym: fewer sets of A per pass if the allocator refuses, then give up
    nblk_max = sym_blocks(c, kernel, (long long)sets_per_pass * tps, env_blocks, env_rounds);
    nseg_max = nblk_max / sets_per_pass + 2;                 // blocks that can share one sink set (+ the two cut ends)
    const int rest = nsa % sets_per_pass;                    // a last pass with fewer sets shares its blocks among fewer sets
    if (rest > 0) nseg_max = std::max(nseg_max, sym_blocks(c, kernel, (long long)rest * tps, env_blocks, env_rounds) / rest + 2);
    part_bytes = (size_t)nseg_max * NC * npa * sizeof(double);
    bslab_bytes = (size_t)nblk_max * NC * TJ * sizeof(double);
    const int rc = nb_try_ensure_work(c, part_bytes + bslab_bytes + (size_t)sets_per_pass * slot_bytes, c->sym_scratch_limit);
    if (rc == NB_OK) break;
    if (rc != NB_WORK_UNAVAILABLE || sets_per_pass <= 1) return rc;
    sets_per_pass = (sets_per_pass + 1) / 2;
  }
  char* w = reinterpret_cast<char*>(c->work);
  double* part = reinterpret_cast<double*>(w);
  double* bslab = reinterpret_cast<double*>(w + part_bytes);
  double* slab = reinterpret_cast<double*>(w + part_bytes + bslab_bytes);
  for (int b0 = 0; b0 < nsa; b0 += sets_per_pass) {
    const int nb_sets = std::min(sets_per_pass, nsa - b0);
    const long long Gs = (long long)nb_sets * tps * HS;
    const int nb_launch = std::min(nblk_max, sym_blocks(c, kernel, (long long)nb_sets * tps, env_blocks, env_rounds));
    if (nb_launch / nb_sets + 2 > nseg_max) return nb_fail(c, NB_ERR_ARG, "accel_sym_f64: internal slot count");
    kernel<<<nb_launch, BLOCK, 0, c->stream>>>(xa, ya, za, ma, (int)na, nsa, 0, 0, tps, Gs, eps2, part, slab, bslab, npa,
                                               xb, yb, zb, mb, (int)nb, nsb, npb, 1, b0,
                                               turn_shift >= 0 ? turn_shift : auto_turn_shift((double)Gs / nb_launch * KS * 8.1));
    NB_HIP(c, hipGetLastError());
    accel_sym_boundary_f64_kernel<KS, TJ, POT><<<nb_launch, BLOCK, 0, c->stream>>>(slab, bslab, nsa, 0, 0, tps, Gs, nsb, npb, 1, b0);
    NB_HIP(c, hipGetLastError());
    const int i_begin = b0 * SET, i_end = (int)std::min<int64_t>(na, (int64_t)(b0 + nb_sets) * SET);
    accel_sym_reduce_f64_kernel<<<(i_end - i_begin + 255) / 256, 256, 0, c->stream>>>(part, Gs, nb_launch, (long long)tps * HS, SET, b0, nullptr, 0, i_end, npa, G,
                                                                                       uniform ? ma : nullptr, nullptr, 1, 1, aax, aay, aaz,
                                                                                       flags & NB_ACCUMULATE, phi_a, i_begin, SymCut{}, SymTail{});
    NB_HIP(c, hipGetLastError());
    accel_sym_reduce_f64_kernel<<<(int)((nb + 255) / 256), 256, 0, c->stream>>>(nullptr, 0, 0, 0, SET, 0, slab, nb_sets, (int)nb, npb, G, uniform ? ma : nullptr,
                                                                                 nullptr, 1, 1, bax, bay, baz, b0 > 0, phi_b, 0, SymCut{}, SymTail{});
    NB_HIP(c, hipGetLastError());
  }
  return NB_OK;
}

// options of the pair-symmetric kernels: the context's (nb_set_option) or, where unset, the process-wide defaults from the environment
struct SymOpt { long long mode, min_n, budget, blocks, rounds, ks, turn; };
SymOpt sym_options(const nb_ctx* c) {
  static const long long env_mode = env_ll("NB_F64_SYM", 1), env_min_n = env_ll("NB_F64_SYM_MIN_N", 10240),
                         env_budget = env_ll("NB_SYM_SLAB_BYTES", 4ll << 30), env_blocks = env_ll("NB_F64_SYM_BLOCKS", 0),
                         env_rounds = env_ll("NB_F64_SYM_ROUNDS", 0), env_ks = env_ll("NB_F64_SYM_KS", 0),
                         env_turn = env_ll("NB_F64_SYM_TURN_SHIFT", -1);    // -1: slice length from the launch (auto_turn_shift), 0: no turns
  return SymOpt{c->sym_enable >= 0 ? c->sym_enable : env_mode, c->sym_min_n >= 0 ? c->sym_min_n : env_min_n,
                c->sym_slab_bytes >= 0 ? c->sym_slab_bytes : env_budget, env_blocks, env_rounds, env_ks, env_turn};
}

}  // namespace

// Self-interaction launch (sinks == sources); *taken = 0: not applicable (small n, switched off with NB_F64_SYM=0) — use the one-sided kernel
int nb_launch_accel_sym_f64(nb_ctx* c, const double* x, const double* y, const double* z, const double* m, int64_t n,
                            double eps2, double G, double* ax, double* ay, double* az, int flags, double* phi, int* taken) {
  *taken = 0;
  c->last_accel_sym = false;
  const SymOpt opt = sym_options(c);
  const long long mode = opt.mode, min_n = opt.min_n, budget = opt.budget, env_ks = opt.ks;
  if (!mode || n < min_n || n >= (1ll << 31) - 65536 || n <= BLOCK * 2) return NB_OK;
  // sinks per lane: 8 for large n (N = 2^19, equal masses: 80.4 ms against 85.1 with 4 and 96.9 with 2; the one-sided kernel: 126.9), smaller
  // sets for smaller n so that the launch still has ~1000 (set, offset, tile) work items (sweep: profiles/r03_sym_kernel_development_log.txt)
  auto items = [&](int ks) { const long long ns = (n + BLOCK * ks - 1) / (BLOCK * ks); return ns * ((ns - 1) / 2 + 1) * ks; };   // tile = 256: ks tiles per set
  int ks = 8;
  if (items(8) < 1000) ks = 4;
  if (ks == 4 && items(4) < 1000) ks = 2;
  if (env_ks == 2 || env_ks == 4 || env_ks == 8) ks = (int)env_ks;
#define NB_RUN(K, PT) run_sym<K, PT>(c, x, y, z, m, n, eps2, G, ax, ay, az, flags, budget, opt.blocks, opt.rounds, (int)opt.turn, phi)
  const int rc = phi ? (ks == 8 ? NB_RUN(8, true) : ks == 4 ? NB_RUN(4, true) : NB_RUN(2, true))
                     : (ks == 8 ? NB_RUN(8, false) : ks == 4 ? NB_RUN(4, false) : NB_RUN(2, false));
#undef NB_RUN
  if (rc == NB_WORK_UNAVAILABLE) return NB_OK;          // no scratch for even one slab slot: *taken stays 0, the caller runs the one-sided kernel
  if (rc) return rc;
  *taken = 1;
  c->last_accel_sym = true;
  return NB_OK;
}

// Two different blocks, every A-B pair once; *taken = 0: not applicable — the caller runs the one-sided kernel in both directions
int nb_launch_accel_cross_f64(nb_ctx* c, const double* xa, const double* ya, const double* za, const double* ma, int64_t na,
                              const double* xb, const double* yb, const double* zb, const double* mb, int64_t nb,
                              double eps2, double G, double* aax, double* aay, double* aaz, double* bax, double* bay, double* baz,
                              double* phi_a, double* phi_b, int flags, int* taken) {
  *taken = 0;
  const SymOpt opt = sym_options(c);
  const long long mode = opt.mode, min_n = opt.min_n, budget = opt.budget;
  const int64_t nmin = std::min(na, nb);
  if (!mode || nmin * 2 < min_n || na >= (1ll << 31) - 65536 || nb >= (1ll << 31) - 65536 || nmin < 1) return NB_OK;
  auto items = [&](int ks) { const long long sa = (na + BLOCK * ks - 1) / (BLOCK * ks), sb = (nb + BLOCK * ks - 1) / (BLOCK * ks); return sa * sb * ks; };
  const bool pot = phi_a != nullptr;
  int ks = 8;
  if (items(8) < 1000) ks = 4;
  if (ks == 4 && items(4) < 1000) ks = 2;
#define NB_RUN(K, PT) run_cross<K, PT>(c, xa, ya, za, ma, na, xb, yb, zb, mb, nb, eps2, G, aax, aay, aaz, bax, bay, baz, phi_a, phi_b, flags, budget, opt.blocks, opt.rounds, (int)opt.turn)
  const int rc = pot ? (ks == 8 ? NB_RUN(8, true) : ks == 4 ? NB_RUN(4, true) : NB_RUN(2, true))
                     : (ks == 8 ? NB_RUN(8, false) : ks == 4 ? NB_RUN(4, false) : NB_RUN(2, false));
#undef NB_RUN
  if (rc == NB_WORK_UNAVAILABLE) return NB_OK;          // no scratch for even one slab slot: *taken stays 0, the caller runs the one-sided kernel
  if (rc) return rc;
  *taken = 1;
  c->last_accel_sym = true;
  return NB_OK;
}

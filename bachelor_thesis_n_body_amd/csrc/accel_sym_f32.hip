// fp32 softened all-pairs acceleration of a particle set ON ITSELF, each unordered pair evaluated ONCE (Newton's third law) —
// SURVEY.md §8(a) row ⊕1 for the self-interaction case (sinks == sources: nb_accel / nb_step, the own-chunk launch of a sharded run).
//
//   a_i = G * sum_j m_j (r_j - r_i) / (|r_j - r_i|^2 + eps^2)^{3/2}
//
// accel_f32.hip sits at its instruction-issue floor (11 full-rate VALU + 1 v_rsq_f32 per ORDERED pair). The pair term
// s_ij (r_j - r_i) serves both particles: evaluated once, the r^2 / rsqrt / cube work (8 + 1 of those 12 instructions) is shared and
// only the second accumulation (3 fma) is added — 14 full-rate + 1 transcendental per UNORDERED pair = 7 + 1/2 per ordered pair.
// What that needs is a place for the second particle's sum. On a GPU the classic sink-per-lane loop reads every source as a wave-wide
// BROADCAST, so all 64 lanes would owe a contribution to the same source at the same time (a 64-lane reduction per source: dearer than
// the work saved). Here the walk is SKEWED instead:
//
//  * particles are cut into SETS of 256*KS; block (b, y) keeps set b in VGPRs as sinks (KS per lane, packed pairs, as accel_f32) and
//    walks the sets c = b + d (mod nsets) for the offsets d of its group y; d = 1 .. (nsets-1)/2 covers every unordered pair of
//    distinct sets exactly once. d = 0 (a set with itself) and, for an even set count, d = nsets/2 (met from both sides) are walked
//    one-sidedly (no second accumulation): 2 of nsets/2 + 1 tiles.
//  * a set streams through LDS in tiles of TJ sources. In step t lane l of a wave meets source (t + l) mod TJ — every lane a DIFFERENT
//    source, read with one conflict-free ds_read_b96 / b128 per lane (the tile carries 64 wrap-around copies, so the address is base + t).
//    The lane adds the pair's term to its sinks in registers and the opposite term to the source's accumulator in LDS — a plain
//    read / add / write of 64 different addresses per instruction: no conflicts, no reduction, no atomics (measured: the same walk
//    with ds_add_f32 runs 6x slower — LDS float atomics retire a few lanes per clock). Each wave owns a private copy of the tile's
//    accumulators; the entry lane l updates in step t is the one lane l + 1 updated in step t - 1, and within a wave LDS operations
//    complete in program order, so the sums are exact and bit-reproducible.
//  * after a tile the block adds its waves' copies in a fixed order and stores the tile's partial accelerations to a slab
//    [offset][3][n] (fp32, written once, no atomics); the sinks' own sums go to an fp64 slab [group][3][n]; a second kernel adds
//    both in a fixed order. Slab traffic at N = 2^20 (12 sinks per lane): 2.4 GB written + 2.2 GB read per launch (PMC) — the reduce kernel
//    takes 0.45 ms of a ~160 ms launch, the writes hide behind the math.
//  * accuracy as accel_f32: fp32 partial sums live for one tile (sinks) / one set of 256*KS sinks (sources) and are then widened.
#include "nb_common.h"
#include <cmath>
#include "../../include/nbody_amd.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
typedef float v2f __attribute__((ext_vector_type(2)));

// one staged source (16 bytes, one ds_read_b128 per lane). Equal masses: (x, y, z, z) — the packed subtractions take (x, x) and (y, y) from the
// low / high half of the first register pair by operand selection, and (z, z) IS the second pair (a 12-byte read leaves z alone in an
// odd-partnered register and costs a v_mov per source to pair it up: 2 of 198 instructions per two steps — measured neutral, A/B on one box:
// 156.9 / 156.9 against 156.5 / 157.2 ms at N = 2^20; kept because it is the simpler code)
template <bool UNIFORM> __device__ inline float4 ld_src(const float4* p) { return *p; }

// ---- the work of a launch and how it is dealt to the blocks -------------------------------------------------------------------------
// Offsets of a launch are numbered o = 0 .. : o = 0 -> d = 0 (one-sided), o = 1 .. H -> d = o (two-sided), o = H + 1 -> d = nsets / 2
// (one-sided; only when nsets is even); a launch covers the offsets [o_begin, o_end) for the sink sets b0 .. b0 + nb_sets - 1. Its work is
// the sequence of STEPS  g = ((bb * tps + tau) * HS + t):  sink set bb of the launch, tile tau = (o - o_begin) * QT + q of its tps tiles,
// step t of the HS = TJ / U steps of a tile — G steps in all, every step the same amount of work (KS * U pair terms per lane).
// Block k of B takes the steps [part_begin(k), part_begin(k + 1)): equal shares to within one pair of steps, whatever the set and tile
// counts are. (Round 3 dealt whole runs of tiles to a (set, group) grid: 342 x 12 = 4104 equal blocks on 512 resident slots at N = 2^20 is
// 8.02 rounds — the ninth, with 8 blocks, cost 6 % of the launch; at N = 65 536, 2176 one-tile blocks = 4.25 rounds cost 15 %.)
// A block walks its tiles in order; when the sink set changes it stores the set's sink sums to slot (ordinal of this block among the blocks
// that share the set) of the fp64 `part` slab and loads the next set. A tile cut by a block boundary is walked by two blocks: the one that
// holds its step 0 writes the tile's source sums to the regular slab slot, the other to its own row of the `bslab`, which is added afterwards
// (B <= G / HS, so a tile is cut at most once): a set on itself — by the reduce kernel; two different blocks — by a small kernel of its own.
// All of it is a fixed function of (n, B): bit-reproducible.
__host__ __device__ inline long long part_begin(long long k, long long G, int B) { return k >= B ? G : ((k * G / B) & ~1LL); }
__host__ __device__ inline int part_block_of(long long s, long long G, int B) {       // the block whose range holds step s
  int k = (int)(s * B / G);
  if (k > B - 1) k = B - 1;
  if (k + 1 < B && part_begin(k + 1, G, B) <= s) ++k;
  return k;
}
// set met by sink set b at offset o
__device__ inline int sym_set_of(int b, int o, int cross, int nsets, int nsets_src, int H) {
  if (cross) return (b + o) % nsets_src;
  const int d = o <= H ? o : nsets / 2; const int c = b + d; return c >= nsets ? c - nsets : c;
}

template <int KS, int U, int TJ, bool GUARD, bool UNIFORM, bool POT>
__global__ void __launch_bounds__(BLOCK, 2)      // two blocks per CU (two waves per SIMD): at most 256 registers, AGPRs included
accel_sym_f32_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const float* __restrict__ m,
                     int n, int nsets, int H, int o_begin, int tps, long long G, float eps2,
                     double* __restrict__ part, float* __restrict__ slab, float* __restrict__ bslab, size_t n_pad,
                     const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs, const float* __restrict__ ms,
                     int n_src, int nsets_src, size_t n_pad_src, int cross, int b0, int turn_shift, long long* __restrict__ trace) {
  // trace (debugging aid, normally null; NB_SYM_TRACE=<file>): [block][start tick, end tick, HW_ID | XCC_ID << 32] of the constant-rate counter
  // cross = 0: the set (x, y, z, m) on itself (xs.. are the same arrays). cross = 1: block A = (x..) against a DIFFERENT block B = (xs..):
  // sink set b of A walks every set of B two-sidedly (o = 0 .. nsets_src - 1 -> set (b + o) mod nsets_src); B's sums go to slab slot
  // b - b0 (one slot per sink set of this launch).
  static_assert(KS % 2 == 0 && TJ % BLOCK == 0 && TJ % U == 0, "shape");
  constexpr int NC = POT ? 4 : 3;    // POT: the pair potential m / sqrt(r^2 + eps^2) rides along as a fourth component on both sides
  constexpr int SET = BLOCK * KS;
  constexpr int KP = KS / 2;
  constexpr int AS = TJ + 64;        // tile entries incl. the 64 wrap-around copies
  constexpr int LPT = TJ / BLOCK;
  constexpr int QT = SET / TJ;       // tiles per set
  constexpr int HS = TJ / U;         // steps per tile
  static_assert(HS % 2 == 0, "steps are taken in pairs");
  __shared__ float4 tile[AS];
  constexpr int AW = 2;              // the two halves of a packed partial sum keep separate accumulators (no add per step)
  __shared__ float acc[WAVES][NC][AS * AW];
  __shared__ unsigned turn_slots[WAVES];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float far = 1e20f;           // UNIFORM padding: r^2 overflows to +inf, v_rsq_f32(+inf) = +0, every term exactly 0
  const int nblk = gridDim.x, kb = blockIdx.x;
  const long long g_begin = part_begin(kb, G, nblk), g_end = part_begin(kb + 1, G, nblk);
  const long long S = (long long)tps * HS;                 // steps per sink set
  if (trace && tid == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    trace[3 * kb] = __builtin_amdgcn_s_memrealtime();
    trace[3 * kb + 2] = (long long)hw | ((long long)xcc << 32);
  }

  v2f px[KP], py[KP], pz[KP], pm[KP];
  double tx[KS], ty[KS], tz[KS], tp[KS];
  auto load_sinks = [&](int b) {
    const int i0 = b * SET + tid;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
      const int ia = i0 + (2 * p) * BLOCK, ib = ia + BLOCK;
      const bool oa = ia < n, ob = ib < n;
      const int ca = oa ? ia : n - 1, cb = ob ? ib : n - 1;
      if (UNIFORM) {
        px[p] = v2f{oa ? x[ca] : far, ob ? x[cb] : far}; py[p] = v2f{oa ? y[ca] : far, ob ? y[cb] : far}; pz[p] = v2f{oa ? z[ca] : far, ob ? z[cb] : far};
        pm[p] = v2f{0.f, 0.f};
      } else {                       // padded sinks: a real position with zero mass (they push nothing onto the sources)
        px[p] = v2f{x[ca], x[cb]}; py[p] = v2f{y[ca], y[cb]}; pz[p] = v2f{z[ca], z[cb]};
        pm[p] = v2f{oa ? m[ca] : 0.f, ob ? m[cb] : 0.f};
      }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) { tx[k] = 0.0; ty[k] = 0.0; tz[k] = 0.0; tp[k] = 0.0; }
#pragma unroll
    for (int p = 0; p < KP; ++p) asm volatile("" : "+v"(px[p]), "+v"(py[p]), "+v"(pz[p]), "+v"(pm[p]));
  };
  auto store_sinks = [&](int b, int seg) {                 // the set's sink sums of this block: slot `seg` of the part slab
    double* pp = part + (size_t)seg * NC * n_pad;
    const int i0 = b * SET + tid;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int i = i0 + k * BLOCK;
      if (i < n) { pp[i] = tx[k]; pp[n_pad + i] = ty[k]; pp[2 * n_pad + i] = tz[k]; if (POT) pp[3 * n_pad + i] = tp[k]; }
    }
  };
  v2f e2;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2" : "=v"(e2.x), "=v"(e2.y) : "s"(eps2));

  for (int e = tid; e < WAVES * NC * AS * AW; e += BLOCK) (&acc[0][0][0])[e] = 0.f;

  float4 stage[LPT];
  auto fetch = [&](long long tile_id) {                    // sources of tile `tile_id` of the launch into registers
    const int bb = (int)(tile_id / tps), tau = (int)(tile_id - (long long)bb * tps);
    const int c = sym_set_of(b0 + bb, o_begin + tau / QT, cross, nsets, nsets_src, H), q = tau % QT;
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int j = c * SET + q * TJ + l * BLOCK + tid;
      const bool ok = j < n_src;
      const int jc = ok ? j : n_src - 1;
      const float zv = UNIFORM ? (ok ? zs[jc] : far) : zs[jc];
      stage[l] = UNIFORM ? make_float4(ok ? xs[jc] : far, ok ? ys[jc] : far, zv, zv)
                         : make_float4(xs[jc], ys[jc], zv, ok ? ms[jc] : 0.0f);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int l = 0; l < LPT; ++l) tile[l * BLOCK + tid] = stage[l];
    if (tid < 64) tile[TJ + tid] = stage[0];       // wrap-around copies of entries 0..63
  };

  unsigned wave_slot;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(wave_slot));      // wave slot within its SIMD: the block's turn id below
  if (lane == 0) turn_slots[wv] = wave_slot;
  if (g_begin < g_end) { fetch(g_begin / HS); commit(); }
  __syncthreads();

  const float4* tl = tile + lane;
  float* ac = &acc[wv][0][lane * AW];
  // Two blocks share a CU (one wave of each per SIMD) and the issue arbiter prefers the OLDER wave whenever both are ready (MI355X_MICROARCH.md:
  // "priority, then age"): of two co-resident blocks with equal work the first-dispatched runs at 0.75 of the SIMD's issue rate, the other
  // at 0.22, and once the first has ended the second continues alone at the 0.8-0.9 a lone wave reaches (measured with NB_SYM_TRACE at
  // N = 65 536: 462 against 764 us per CU; a lone wave per SIMD over a whole launch: 0.81 / 0.90 of the two-wave rate at N = 65 536 / 2^20) —
  // the pair finishes 10 % later than two waves sharing fairly would. So the two blocks take TURNS at the raised priority: time slices of
  // 2^turn_shift ticks of the constant-rate counter (100 MHz, the same for every wave of the device); the block whose turn id has the slice's
  // parity runs at s_setprio 1, the other at 0. The counter is read when a chunk of 8 steps begins and acted on when it ends (scalar work
  // outside the step loop). The slice must be long against that look-up period and short against the launch (the host chooses it:
  // auto_turn_shift); N = 65 536 with 2^14 ticks = 164 us: per CU 667 against 727 us, launch 819 -> 776 us. First attempts failed for
  // exactly that reason — turns counted in each wave's OWN steps let the leader reach its next favoured segment first (63 : 37), slices of
  // 10 us looked at every 32 steps (~20-40 us) left the two priorities uncorrelated (profiles/r04_priority_turns_between_coresident_waves.txt).
  // Pinning the step's LDS reads above its arithmetic does not make a lone wave faster.
  // The turn id is the BLOCK's, not the wave's: all four waves of a block must be favoured in the same slices (they meet at every tile's
  // barrier, so a block runs at the pace of its slowest wave). It is the smallest wave-slot id among the block's waves: where the CU holds
  // nothing else that is 0 for the first block and 1 for the second on every SIMD; where a third wave sits on one SIMD (bench.py's clock
  // probe: one sleeping wave per XCD) the per-wave ids there are 1 and 2 against 0 and 1 elsewhere, and per-WAVE turns put one wave of each
  // block out of phase with its three partners — that CU then straggled by 6 % and, under a deal with few rounds, so did the launch
  // (bench.py with the probe: +4-8 %; without: -3 % ... 0; same file).
  unsigned slot_id = min(min(turn_slots[0], turn_slots[1]), min(turn_slots[2], turn_slots[3]));
  slot_id = __builtin_amdgcn_readfirstlane(slot_id);

  auto walk = [&](auto symtag, int t_lo, int t_hi) {
    constexpr bool SYM = decltype(symtag)::value;
    v2f fx[KP], fy[KP], fz[KP], fp[KP];
#pragma unroll
    for (int p = 0; p < KP; ++p) { fx[p] = v2f{0.f, 0.f}; fy[p] = v2f{0.f, 0.f}; fz[p] = v2f{0.f, 0.f}; fp[p] = v2f{0.f, 0.f}; }
    // step t: lane l meets the sources t + u * TJ/U + l (u < U) — 64 different sources per instruction, and the U sets of a step are disjoint.
    // The sources of step t + 1 are read during step t into the other of two register sets (the tile has 64 entries beyond the last
    // step's); the accumulators of a step are read at its top and consumed at its end (the scheduler may not pull the source-side
    // chains above the sink-side work), so no LDS round trip is exposed.
    auto step = [&](float4 (&s)[U], float4 (&nx)[U], int t) {
      v2f gx[U], gy[U], gz[U], gp[U];
      // lane l reads the accumulator lane l + 1 wrote one step earlier (same wave: LDS operations complete in program order); the
      // compiler sees only this lane's addresses, so no LDS access may be moved across a step boundary
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < U; ++u) {
        nx[u] = ld_src<UNIFORM>(tl + t + 1 + u * HS);
      }
      if (SYM) {   // the source's running sums start the partial-sum chains of this step
#pragma unroll
        for (int u = 0; u < U; ++u) {
          gx[u] = *reinterpret_cast<const v2f*>(ac + (0 * AS + t + u * HS) * 2);
          gy[u] = *reinterpret_cast<const v2f*>(ac + (1 * AS + t + u * HS) * 2);
          gz[u] = *reinterpret_cast<const v2f*>(ac + (2 * AS + t + u * HS) * 2);
          if (POT) gp[u] = *reinterpret_cast<const v2f*>(ac + (3 * AS + t + u * HS) * 2);
        }
      }
      v2f dx[U][KP], dy[U][KP], dz[U][KP], r2[U][KP], rinv[U][KP], fj[U][KP], pj[U][KP];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < KP; ++p) {
          dx[u][p] = v2f{s[u].x, s[u].x} - px[p];
          dy[u][p] = v2f{s[u].y, s[u].y} - py[p];
          dz[u][p] = (UNIFORM ? v2f{s[u].z, s[u].w} : v2f{s[u].z, s[u].z}) - pz[p];
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < KP; ++p)
          r2[u][p] = __builtin_elementwise_fma(dx[u][p], dx[u][p], __builtin_elementwise_fma(dy[u][p], dy[u][p], __builtin_elementwise_fma(dz[u][p], dz[u][p], e2)));
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < KP; ++p) {
          rinv[u][p] = v2f{__builtin_amdgcn_rsqf(r2[u][p].x), __builtin_amdgcn_rsqf(r2[u][p].y)};
          if (GUARD) {
            rinv[u][p].x = r2[u][p].x > e2.x ? rinv[u][p].x : 0.0f;
            rinv[u][p].y = r2[u][p].y > e2.y ? rinv[u][p].y : 0.0f;
          }
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < KP; ++p) {
          // factors of the sinks' sums (m_j / r^3, m_j / r) and of the source's sums (m_i / r^3, m_i / r); equal masses: applied at the end
          const v2f rinv2 = rinv[u][p] * rinv[u][p];
          v2f fi, pi;
          if (UNIFORM) { fi = rinv2 * rinv[u][p]; fj[u][p] = fi; pi = rinv[u][p]; pj[u][p] = pi; }
          else if (POT) { pi = rinv[u][p] * v2f{s[u].w, s[u].w}; pj[u][p] = rinv[u][p] * pm[p]; fi = pi * rinv2; fj[u][p] = pj[u][p] * rinv2; }
          else { const v2f r3 = rinv2 * rinv[u][p]; fi = r3 * v2f{s[u].w, s[u].w}; fj[u][p] = r3 * pm[p]; }
          fx[p] = __builtin_elementwise_fma(fi, dx[u][p], fx[p]);
          fy[p] = __builtin_elementwise_fma(fi, dy[u][p], fy[p]);
          fz[p] = __builtin_elementwise_fma(fi, dz[u][p], fz[p]);
          if (POT) fp[p] = fp[p] + pi;
        }
      if (SYM) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int p = 0; p < KP; ++p) {                                   // a_j -= fj (r_j - r_i)
            gx[u] = __builtin_elementwise_fma(-fj[u][p], dx[u][p], gx[u]);
            gy[u] = __builtin_elementwise_fma(-fj[u][p], dy[u][p], gy[u]);
            gz[u] = __builtin_elementwise_fma(-fj[u][p], dz[u][p], gz[u]);
            if (POT) gp[u] = gp[u] + pj[u][p];
          }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          *reinterpret_cast<v2f*>(ac + (0 * AS + t + u * HS) * 2) = gx[u];
          *reinterpret_cast<v2f*>(ac + (1 * AS + t + u * HS) * 2) = gy[u];
          *reinterpret_cast<v2f*>(ac + (2 * AS + t + u * HS) * 2) = gz[u];
          if (POT) *reinterpret_cast<v2f*>(ac + (3 * AS + t + u * HS) * 2) = gp[u];
        }
      }
    };
    float4 sa[U], sb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) sa[u] = ld_src<UNIFORM>(tl + t_lo + u * HS);
    for (int tc = t_lo; tc < t_hi; tc += 8) {
      const int te = tc + 8 < t_hi ? tc + 8 : t_hi;
      unsigned long long now = 0;
      if (turn_shift > 0) now = __builtin_amdgcn_s_memrealtime();
      for (int t = tc; t < te; t += 2) { step(sa, sb, t); step(sb, sa, t + 1); }
      if (turn_shift > 0) { if (((now >> turn_shift) ^ slot_id) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    }
#pragma unroll
    for (int p = 0; p < KP; ++p) {
      tx[2 * p] += (double)fx[p].x; ty[2 * p] += (double)fy[p].x; tz[2 * p] += (double)fz[p].x;
      tx[2 * p + 1] += (double)fx[p].y; ty[2 * p + 1] += (double)fy[p].y; tz[2 * p + 1] += (double)fz[p].y;
      if (POT) { tp[2 * p] += (double)fp[p].x; tp[2 * p + 1] += (double)fp[p].y; }
    }
  };

  int cur = -1, seg = 0;                                   // sink set (of the launch) in registers, and this block's slot among the set's blocks
  for (long long g = g_begin; g < g_end;) {
    const long long tile_id = g / HS;
    const int t_lo = (int)(g - tile_id * HS);
    const int t_hi = (int)((long long)HS < t_lo + (g_end - g) ? (long long)HS : t_lo + (g_end - g));
    const int bb = (int)(tile_id / tps), tau = (int)(tile_id - (long long)bb * tps);
    const int o = o_begin + tau / QT, q = tau % QT;
    const bool sym = cross || (o >= 1 && o <= H);
    if (bb != cur) {
      if (cur >= 0) store_sinks(b0 + cur, seg);
      seg = cur < 0 ? kb - part_block_of((long long)bb * S, G, nblk) : 0;     // a set that begins inside this block's range: its first block
      cur = bb;
      load_sinks(b0 + bb);
    }
    const long long g_next = g + (t_hi - t_lo);
    if (g_next < g_end) fetch(g_next / HS);
    if (sym) walk(std::true_type{}, t_lo, t_hi); else walk(std::false_type{}, t_lo, t_hi);
    __syncthreads();                                   // every wave has finished the tile: accumulators complete, tile free
    if (sym) {
      const int c = sym_set_of(b0 + bb, o, cross, nsets, nsets_src, H);
      // the block that holds step 0 of a tile owns its slab entry; a block that starts inside the tile leaves its share in its bslab row
      float* sl = t_lo == 0 ? slab + (size_t)(cross ? bb : o - max(o_begin, 1)) * NC * n_pad_src + (size_t)c * SET + (size_t)q * TJ
                            : bslab + (size_t)kb * NC * TJ;
      const size_t cstride = t_lo == 0 ? n_pad_src : (size_t)TJ;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
        for (int e = tid; e < TJ; e += BLOCK) {
          float v = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < WAVES; ++w2)
#pragma unroll
            for (int h = 0; h < AW; ++h) { v += acc[w2][cc][e * AW + h]; acc[w2][cc][e * AW + h] = 0.f; }
          if (e < 64) {
#pragma unroll
            for (int w2 = 0; w2 < WAVES; ++w2)
#pragma unroll
              for (int h = 0; h < AW; ++h) { v += acc[w2][cc][(TJ + e) * AW + h]; acc[w2][cc][(TJ + e) * AW + h] = 0.f; }
          }
          sl[(size_t)cc * cstride + e] = v;
        }
    }
    if (g_next < g_end) commit();
    __syncthreads();
    g = g_next;
  }
  if (cur >= 0) store_sinks(b0 + cur, seg);
  if (trace && tid == 0) trace[3 * kb + 1] = __builtin_amdgcn_s_memrealtime();
}

// two-block launches: slab entry of a tile cut by a block boundary += the share of the block that started inside it (one such block per tile:
// B <= G / HS). A set on itself has this done by its reduce kernel.
template <int KS, int U, int TJ, bool POT>
__global__ void __launch_bounds__(BLOCK)
accel_sym_boundary_f32_kernel(float* __restrict__ slab, const float* __restrict__ bslab, int nsets, int H, int o_begin, int tps, long long G,
                              int nsets_src, size_t n_pad_src, int cross, int b0) {
  constexpr int NC = POT ? 4 : 3, SET = BLOCK * KS, QT = SET / TJ, HS = TJ / U;
  const int kb = blockIdx.x, nblk = gridDim.x;
  const long long g = part_begin(kb, G, nblk);
  if (g >= part_begin(kb + 1, G, nblk)) return;
  const long long tile_id = g / HS;
  if (g == tile_id * HS) return;                           // starts on a tile boundary: owns its first tile
  const int bb = (int)(tile_id / tps), tau = (int)(tile_id - (long long)bb * tps);
  const int o = o_begin + tau / QT, q = tau % QT;
  if (!(cross || (o >= 1 && o <= H))) return;              // one-sided tiles have no source sums
  const int c = sym_set_of(b0 + bb, o, cross, nsets, nsets_src, H);
  float* sl = slab + (size_t)(cross ? bb : o - max(o_begin, 1)) * NC * n_pad_src + (size_t)c * SET + (size_t)q * TJ;
  const float* bs = bslab + (size_t)kb * NC * TJ;
  for (int cc = 0; cc < NC; ++cc)
    for (int e = threadIdx.x; e < TJ; e += BLOCK) sl[(size_t)cc * n_pad_src + e] += bs[cc * TJ + e];
}

// What the reduce kernel needs to finish the tiles that a block boundary cut (self launches; bslab = nullptr: nothing to finish, or the
// separate boundary kernel above has run), and the integrator tail it may apply to the particles it completes (vx = nullptr: none).
struct SymCut { const float* bslab; int nsets, o_begin, s_begin, tps, tj, hs; };
struct SymTail { float *vx, *vy, *vz, *x, *y, *z; float kick_close, kick_open, drift; int advance; };

// a[c][i] (+)= g * ( sum_seg part[seg][c][i] + sum_s slab[s][c][i] ), fixed order; `tot` non-null: the sums of several passes are collected in
// fp64 first (first: tot = sum, otherwise tot += sum) and turned into accelerations by the last pass. The part slots of a particle are those of
// the blocks that shared its sink set (part_block_of: the same function of (G, nblk) the force kernel dealt its work by); nblk = 0: no part slab.
// cut.bslab non-null: slab entry s of a particle lacks the share of the block that STARTED inside its tile, if one did — block
// part_block_of(first step of the tile) + 1 when that block begins before the tile ends; its row of the bslab is added here, in fp32 and before
// the conversion, exactly as the boundary kernel would have (same bits). The 256 particles of a block of this kernel lie in one tile (sets and
// tiles are multiples of 256), so the blocks are looked up once per slot by the first `nslots` threads and passed through LDS.
// tail.vx non-null (last pass only): the leapfrog tail of nb_ctx::Tail on the particles this thread has just completed.
__global__ void __launch_bounds__(256)
accel_sym_reduce_f32_kernel(const double* __restrict__ part, long long G, int nblk, long long S, int set_size, int b0,
                            const float* __restrict__ slab, int nslots, int n, size_t n_pad,
                            float G_newton, const float* __restrict__ m0, double* __restrict__ tot, int first, int last,
                            float* __restrict__ ax, float* __restrict__ ay, float* __restrict__ az, int accumulate, double* __restrict__ phi,
                            int i_begin, SymCut cut, SymTail tail) {
  __shared__ int cut_block[256];
  const int nc = phi ? 4 : 3;
  const int i0 = i_begin + blockIdx.x * blockDim.x;                    // particles [i_begin, n)
  const int i = i0 + threadIdx.x;
  const bool live = i < n;
  double sx = 0.0, sy = 0.0, sz = 0.0, sp = 0.0;
  if (live && nblk > 0) {
    const long long bb = i / set_size - b0;
    const int nseg = part_block_of((bb + 1) * S - 1, G, nblk) - part_block_of(bb * S, G, nblk) + 1;
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
        const float* p = slab + (size_t)s * nc * n_pad;
        sx += (double)p[i]; sy += (double)p[n_pad + i]; sz += (double)p[2 * n_pad + i];
        if (phi) sp += (double)p[3 * n_pad + i];
      }
    }
  } else {
    const int c = i0 / set_size, r0 = i0 - c * set_size, q = r0 / cut.tj, e = r0 - q * cut.tj + (int)threadIdx.x;
    for (int s0 = 0; s0 < nslots; s0 += 256) {
      const int ns = nslots - s0 < 256 ? nslots - s0 : 256;
      if (s0 > 0) __syncthreads();
      if ((int)threadIdx.x < ns) {
        const int o = cut.s_begin + s0 + (int)threadIdx.x;               // the offset of slot s; its sink set bb met this particle's set c
        int bb = c - o; if (bb < 0) bb += cut.nsets;
        const long long g0 = ((long long)bb * cut.tps + (long long)(o - cut.o_begin) * (set_size / cut.tj) + q) * cut.hs;
        const int k0 = part_block_of(g0, G, nblk);
        cut_block[threadIdx.x] = part_begin(k0 + 1, G, nblk) < g0 + cut.hs ? k0 + 1 : -1;
      }
      __syncthreads();
      if (live) {
#pragma unroll 4
        for (int s = 0; s < ns; ++s) {
          const float* p = slab + (size_t)(s0 + s) * nc * n_pad;
          float vx = p[i], vy = p[n_pad + i], vz = p[2 * n_pad + i], vp = phi ? p[3 * n_pad + i] : 0.f;
          const int kb = cut_block[s];
          if (kb >= 0) {
            const float* bs = cut.bslab + (size_t)kb * nc * cut.tj + e;
            vx += bs[0]; vy += bs[cut.tj]; vz += bs[2 * cut.tj];
            if (phi) vp += bs[3 * cut.tj];
          }
          sx += (double)vx; sy += (double)vy; sz += (double)vz;
          if (phi) sp += (double)vp;
        }
      }
    }
  }
  if (!live) return;
  if (tot) {
    if (!first) { sx += tot[i]; sy += tot[(size_t)n + i]; sz += tot[2 * (size_t)n + i]; if (phi) sp += tot[3 * (size_t)n + i]; }
    if (!last) { tot[i] = sx; tot[(size_t)n + i] = sy; tot[2 * (size_t)n + i] = sz; if (phi) tot[3 * (size_t)n + i] = sp; return; }
  }
  if (phi) { const double ph = (m0 ? (double)m0[0] : 1.0) * sp; phi[i] = accumulate ? phi[i] + ph : ph; }   // per unit G, self term included
  const double g = m0 ? (double)G_newton * (double)m0[0] : (double)G_newton;
  float ox = (float)(g * sx), oy = (float)(g * sy), oz = (float)(g * sz);
  if (accumulate) { ox += ax[i]; oy += ay[i]; oz += az[i]; }
  ax[i] = ox; ay[i] = oy; az[i] = oz;
  if (tail.vx) {                                            // as kick_kernel / kick_drift_kernel (integrate.hip): products and sums rounded one by one
#pragma clang fp contract(off)
    float v0 = tail.vx[i] + ox * tail.kick_close;
    float v1 = tail.vy[i] + oy * tail.kick_close;
    float v2 = tail.vz[i] + oz * tail.kick_close;
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

// Blocks of a launch: the resident block slots of the device for this kernel (occupancy x CUs) times `rounds` (NB_F32_SYM_ROUNDS; default: up
// to 8, see below — any multiple of the slots stays balanced because the shares are equal to a step), never more than one block per tile
// (a tile may be cut once only); NB_F32_SYM_BLOCKS overrides the product.
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

// the instantiation for (coincident-pair guard, equal masses). 12 sinks per lane exist for equal masses only (per-particle masses would need
// 256 VGPRs): there the per-particle-mass slot names the equal-mass kernel again, so that it is never instantiated (callers pass uniform = true)
template <int KS, int U, int TJ, bool POT>
auto sym_kernel(bool guard, bool uniform) {
  constexpr bool ONLY_UNIFORM = KS == 12;
  if (guard) return uniform ? accel_sym_f32_kernel<KS, U, TJ, true, true, POT> : accel_sym_f32_kernel<KS, U, TJ, true, ONLY_UNIFORM, POT>;
  return uniform ? accel_sym_f32_kernel<KS, U, TJ, false, true, POT> : accel_sym_f32_kernel<KS, U, TJ, false, ONLY_UNIFORM, POT>;
}

template <int KS, int U, bool POT>
int run_sym(nb_ctx* c, const float* x, const float* y, const float* z, const float* m, int64_t n,
            float eps2, float G, float* ax, float* ay, float* az, int flags, long long budget, long long env_blocks, long long env_rounds, int turn_shift, double* phi) {
  constexpr int TJ = POT ? 256 : 512, SET = BLOCK * KS, NC = POT ? 4 : 3;   // POT: a fourth accumulator per source — half the tile keeps the LDS footprint
  constexpr int QT = SET / TJ, HS = TJ / U;                  // tiles per set, steps per tile
  const int nsets = (int)((n + SET - 1) / SET);
  const size_t n_pad = (size_t)nsets * SET;
  const int H = (nsets - 1) / 2;
  const int n_o = H + 1 + (nsets % 2 == 0 ? 1 : 0);          // offsets incl. the one-sided ones
  const size_t slot_bytes = NC * n_pad * sizeof(float);
  // clamped in 64 bits BEFORE the cast: a budget meant as "unlimited" (INT64_MAX) must not wrap
  int slots_per_pass = (int)std::min<long long>(std::max<long long>(1, budget / (long long)slot_bytes), (long long)std::max(H, 1));
  const bool uniform = (flags & NB_UNIFORM_MASS) != 0;
  const bool guard = !(eps2 >= 1e-20f);
  const int accumulate = flags & NB_ACCUMULATE;
  // the kernel instantiation of this call (for the occupancy query and the launches)
  const auto kernel = sym_kernel<KS, U, TJ, POT>(guard, uniform);
  // pass 0 takes o = 0 and the first two-sided offsets, the last pass also the one-sided half offset. The blocks of the LARGEST pass size the
  // scratch: part slab [blocks sharing a set + 1][NC][n_pad] fp64, boundary slab [blocks][NC][TJ].
  int npass, nblk_max, nseg_max;
  size_t part_bytes, bslab_bytes, slab_bytes, tot_bytes;
  for (;;) {                                                 // the allocator may refuse the scratch: fewer slots per pass, then give up (one-sided kernel)
    npass = H > 0 ? (H + slots_per_pass - 1) / slots_per_pass : 1;
    const long long tiles_max = (long long)nsets * std::min(n_o, slots_per_pass + 2) * QT;
    nblk_max = sym_blocks(c, kernel, tiles_max, env_blocks, env_rounds);
    nseg_max = nblk_max / nsets + 2;
    part_bytes = (size_t)nseg_max * NC * n_pad * sizeof(double);
    bslab_bytes = (size_t)nblk_max * NC * TJ * sizeof(float);
    slab_bytes = (size_t)slots_per_pass * slot_bytes;
    tot_bytes = npass > 1 ? NC * (size_t)n * sizeof(double) : 0;
    const int rc = nb_try_ensure_work(c, part_bytes + bslab_bytes + slab_bytes + tot_bytes, c->sym_scratch_limit);
    if (rc == NB_OK) break;
    if (rc != NB_WORK_UNAVAILABLE || slots_per_pass <= 1) return rc;
    slots_per_pass = (slots_per_pass + 1) / 2;
  }
  char* w = reinterpret_cast<char*>(c->work);
  double* part = reinterpret_cast<double*>(w);
  float* bslab = reinterpret_cast<float*>(w + part_bytes);
  float* slab = reinterpret_cast<float*>(w + part_bytes + bslab_bytes);
  double* tot = npass > 1 ? reinterpret_cast<double*>(w + part_bytes + bslab_bytes + slab_bytes) : nullptr;
  for (int ps = 0; ps < npass; ++ps) {
    // two-sided offsets of this pass: o in [1 + ps * spp, 1 + min(H, (ps + 1) * spp)); o = 0 rides with pass 0, o = H + 1 with the last
    const int s_begin = 1 + ps * slots_per_pass, s_end = 1 + std::min(H, (ps + 1) * slots_per_pass);
    const int o_begin = ps == 0 ? 0 : s_begin;
    const int o_end = (ps == npass - 1) ? n_o : s_end;
    const int tps = (o_end - o_begin) * QT;                   // tiles per sink set in this pass
    const long long Gs = (long long)nsets * tps * HS;        // steps of this pass
    const int nblk = std::min(nblk_max, sym_blocks(c, kernel, (long long)nsets * tps, env_blocks, env_rounds));
    const int nslots = std::max(0, s_end - s_begin);
    if (nblk / nsets + 2 > nseg_max) return nb_fail(c, NB_ERR_ARG, "accel_sym_f32: internal slot count");
    static const char* trace_path = getenv("NB_SYM_TRACE");      // debugging aid: per-block start / end ticks and placement of every launch
    long long* trace = nullptr;
    if (trace_path && *trace_path) NB_HIP(c, hipMalloc(&trace, (size_t)nblk * 3 * sizeof(long long)));
    kernel<<<nblk, BLOCK, 0, c->stream>>>(x, y, z, m, (int)n, nsets, H, o_begin, tps, Gs, eps2, part, slab, bslab, n_pad,
                                          x, y, z, m, (int)n, nsets, n_pad, 0, 0,
                                          turn_shift >= 0 ? turn_shift : auto_turn_shift((double)Gs / nblk * (KS / 2) * U * 7.3), trace);
    NB_HIP(c, hipGetLastError());
    if (trace) {
      std::vector<long long> h((size_t)nblk * 3);
      NB_HIP(c, hipStreamSynchronize(c->stream));
      NB_HIP(c, hipMemcpy(h.data(), trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
      (void)hipFree(trace);
      if (FILE* f = fopen(trace_path, "a")) {
        fprintf(f, "# launch n=%lld KS=%d blocks=%d steps=%lld\n", (long long)n, KS, nblk, Gs);
        for (int k = 0; k < nblk; ++k) fprintf(f, "%d %lld %lld %lld %lld\n", k, h[3 * k], h[3 * k + 1], h[3 * k + 2] & 0xffffffffll, h[3 * k + 2] >> 32);
        fclose(f);
      }
    }
    // the tiles a block boundary cut are finished inside the reduce kernel; the last pass also applies the integrator tail, if one is armed
    const SymCut cut{nslots > 0 ? bslab : nullptr, nsets, o_begin, s_begin, tps, TJ, HS};
    SymTail tail{};
    if (c->tail.armed && ps == npass - 1) {
      tail = SymTail{(float*)c->tail.vx, (float*)c->tail.vy, (float*)c->tail.vz, (float*)c->tail.x, (float*)c->tail.y, (float*)c->tail.z,
                     (float)c->tail.kick_close, (float)c->tail.kick_open, (float)c->tail.drift, c->tail.advance ? 1 : 0};
      c->tail.done = true;
    }
    accel_sym_reduce_f32_kernel<<<(int)((n + 255) / 256), 256, 0, c->stream>>>(part, Gs, nblk, (long long)tps * HS, SET, 0, slab, nslots, (int)n, n_pad, G,
                                                                                 uniform ? m : nullptr, tot, ps == 0, ps == npass - 1,
                                                                                 ax, ay, az, accumulate, phi, 0, cut, tail);
    NB_HIP(c, hipGetLastError());
  }
  return NB_OK;
}

// Block A (sinks that stay: accelerations += / = into aax..) against a different block B (accelerations due to A written to bax..): every
// A-B pair evaluated once. Several passes over ranges of A's sets when the slab [sets of A][NC][n_b] exceeds the scratch budget.
template <int KS, int U, bool POT>
int run_cross(nb_ctx* c, const float* xa, const float* ya, const float* za, const float* ma, int64_t na,
              const float* xb, const float* yb, const float* zb, const float* mb, int64_t nb,
              float eps2, float G, float* aax, float* aay, float* aaz, float* bax, float* bay, float* baz,
              double* phi_a, double* phi_b, int flags, long long budget, long long env_blocks, long long env_rounds, int turn_shift) {
  constexpr int TJ = POT ? 256 : 512, SET = BLOCK * KS, NC = POT ? 4 : 3;
  constexpr int QT = SET / TJ, HS = TJ / U;
  const int nsa = (int)((na + SET - 1) / SET), nsb = (int)((nb + SET - 1) / SET);
  const size_t npa = (size_t)nsa * SET, npb = (size_t)nsb * SET;
  const size_t slot_bytes = NC * npb * sizeof(float);
  int sets_per_pass = (int)std::min<long long>(std::max<long long>(1, budget / (long long)slot_bytes), (long long)std::max(nsa, 1));
  const bool uniform = (flags & NB_UNIFORM_MASS) != 0;
  const bool guard = !(eps2 >= 1e-20f);
  const auto kernel = sym_kernel<KS, U, TJ, POT>(guard, uniform);
  const int tps = nsb * QT;                                   // every set of B, tile by tile
  int nblk_max, nseg_max;
  size_t part_bytes, bslab_bytes;
  for (;;) {                                                 // as run_sym: fewer sets of A per pass if the allocator refuses, then give up
    nblk_max = sym_blocks(c, kernel, (long long)sets_per_pass * tps, env_blocks, env_rounds);
    nseg_max = nblk_max / sets_per_pass + 2;                 // blocks that can share one sink set (+ the two cut ends)
    const int rest = nsa % sets_per_pass;                    // a last pass with fewer sets shares its blocks among fewer sets
    if (rest > 0) nseg_max = std::max(nseg_max, sym_blocks(c, kernel, (long long)rest * tps, env_blocks, env_rounds) / rest + 2);
    part_bytes = (size_t)nseg_max * NC * npa * sizeof(double);
    bslab_bytes = (size_t)nblk_max * NC * TJ * sizeof(float);
    const int rc = nb_try_ensure_work(c, part_bytes + bslab_bytes + (size_t)sets_per_pass * slot_bytes, c->sym_scratch_limit);
    if (rc == NB_OK) break;
    if (rc != NB_WORK_UNAVAILABLE || sets_per_pass <= 1) return rc;
    sets_per_pass = (sets_per_pass + 1) / 2;
  }
  char* w = reinterpret_cast<char*>(c->work);
  double* part = reinterpret_cast<double*>(w);
  float* bslab = reinterpret_cast<float*>(w + part_bytes);
  float* slab = reinterpret_cast<float*>(w + part_bytes + bslab_bytes);
  for (int b0 = 0; b0 < nsa; b0 += sets_per_pass) {
    const int nb_sets = std::min(sets_per_pass, nsa - b0);
    const long long Gs = (long long)nb_sets * tps * HS;
    const int nb_launch = std::min(nblk_max, sym_blocks(c, kernel, (long long)nb_sets * tps, env_blocks, env_rounds));
    if (nb_launch / nb_sets + 2 > nseg_max) return nb_fail(c, NB_ERR_ARG, "accel_sym_f32: internal slot count");
    kernel<<<nb_launch, BLOCK, 0, c->stream>>>(xa, ya, za, ma, (int)na, nsa, 0, 0, tps, Gs, eps2, part, slab, bslab, npa,
                                               xb, yb, zb, mb, (int)nb, nsb, npb, 1, b0,
                                               turn_shift >= 0 ? turn_shift : auto_turn_shift((double)Gs / nb_launch * (KS / 2) * U * 7.3), nullptr);
    NB_HIP(c, hipGetLastError());
    accel_sym_boundary_f32_kernel<KS, U, TJ, POT><<<nb_launch, BLOCK, 0, c->stream>>>(slab, bslab, nsa, 0, 0, tps, Gs, nsb, npb, 1, b0);
    NB_HIP(c, hipGetLastError());
    // A's sinks of this pass: complete after it
    const int i_begin = b0 * SET, i_end = (int)std::min<int64_t>(na, (int64_t)(b0 + nb_sets) * SET);
    accel_sym_reduce_f32_kernel<<<(i_end - i_begin + 255) / 256, 256, 0, c->stream>>>(part, Gs, nb_launch, (long long)tps * HS, SET, b0, nullptr, 0, i_end, npa, G,
                                                                                       uniform ? ma : nullptr, nullptr, 1, 1, aax, aay, aaz,
                                                                                       flags & NB_ACCUMULATE, phi_a, i_begin, SymCut{}, SymTail{});
    NB_HIP(c, hipGetLastError());
    // B's sums: this pass' slots, added to what earlier passes left
    accel_sym_reduce_f32_kernel<<<(int)((nb + 255) / 256), 256, 0, c->stream>>>(nullptr, 0, 0, 0, SET, 0, slab, nb_sets, (int)nb, npb, G, uniform ? ma : nullptr,
                                                                                 nullptr, 1, 1, bax, bay, baz, b0 > 0, phi_b, 0, SymCut{}, SymTail{});
    NB_HIP(c, hipGetLastError());
  }
  return NB_OK;
}

// options of the pair-symmetric kernels: the context's (nb_set_option) or, where unset, the process-wide defaults from the environment
struct SymOpt { long long mode, min_n, budget, blocks, rounds, ks, turn; };
SymOpt sym_options(const nb_ctx* c) {
  static const long long env_mode = env_ll("NB_F32_SYM", 1), env_min_n = env_ll("NB_F32_SYM_MIN_N", 18432),
                         env_budget = env_ll("NB_SYM_SLAB_BYTES", 4ll << 30), env_blocks = env_ll("NB_F32_SYM_BLOCKS", 0),
                         env_rounds = env_ll("NB_F32_SYM_ROUNDS", 0), env_ks = env_ll("NB_F32_SYM_KS", 0),
                         // priority turns: time slices of 2^shift ticks of the 100 MHz counter; -1 = from the launch (auto_turn_shift), 0 = off
                         env_turn = env_ll("NB_F32_SYM_TURN_SHIFT", -1);
  return SymOpt{c->sym_enable >= 0 ? c->sym_enable : env_mode, c->sym_min_n >= 0 ? c->sym_min_n : env_min_n,
                c->sym_slab_bytes >= 0 ? c->sym_slab_bytes : env_budget, env_blocks, env_rounds, env_ks, env_turn};
}

}  // namespace

// Self-interaction launch (sinks == sources). Returns NB_OK and sets *taken = 1 when the pair-symmetric kernel ran, *taken = 0 when the
// caller should use the one-sided kernel (small n, switched off with NB_F32_SYM=0). phi non-null: the fused potential (per unit G, self term included).
int nb_launch_accel_sym_f32(nb_ctx* c, const float* x, const float* y, const float* z, const float* m, int64_t n,
                            float eps2, float G, float* ax, float* ay, float* az, int flags, double* phi, int* taken) {
  *taken = 0;
  c->last_accel_sym = false;
  const SymOpt opt = sym_options(c);
  const long long mode = opt.mode, min_n = opt.min_n, budget = opt.budget, env_ks = opt.ks;
  if (!mode || n < min_n || n >= (1ll << 31) - 65536 || n <= BLOCK * 4) return NB_OK;
  // Sinks per lane (set = 256 * KS particles). Equal masses, large n: 12 sinks x 1 source per step (244 VGPRs; fewest accumulator round trips
  // per pair: 159.6 ms at N = 2^20 against 163.7 with 8 sinks x 2 sources, one box); per-particle masses carry two more registers per sink
  // pair: 8 x 2. Smaller n: smaller sets, so that the launch still has enough (set, offset, tile) work items to spread over 256 CUs — measured
  // (profiles/r03_sym_kernel_development_log.txt, sinks-per-lane sweep): 12 from ~3300 items (n ~ 100 000), 8 from ~500 (n ~ 30 000), else 4.
  auto items = [&](int ks, int tj) { const long long ns = (n + BLOCK * ks - 1) / (BLOCK * ks); return ns * ((ns - 1) / 2 + 1) * (BLOCK * ks / tj); };
  const int tj = phi ? 256 : 512;
  int ks = (flags & NB_UNIFORM_MASS) && !phi ? 12 : 8;
  if (ks == 12 && items(12, tj) < 3300) ks = 8;
  if (ks == 8 && items(8, tj) < 500) ks = 4;      // (700 until the blocks of a CU took turns: N = 32 768, 576 items, now runs 6 % faster on 8-sink sets — profiles/r04_sym_ks_sweep_final_kernels.txt)
  if (env_ks == 4 || env_ks == 8 || (env_ks == 12 && (flags & NB_UNIFORM_MASS) && !phi)) ks = (int)env_ks;
#define NB_RUN(K, UU, PT) run_sym<K, UU, PT>(c, x, y, z, m, n, eps2, G, ax, ay, az, flags, budget, opt.blocks, opt.rounds, (int)opt.turn, phi)
  const int rc = phi ? (ks == 8 ? NB_RUN(8, 1, true) : NB_RUN(4, 2, true))
                     : (ks == 12 ? NB_RUN(12, 1, false) : ks == 8 ? NB_RUN(8, 2, false) : NB_RUN(4, 2, false));
#undef NB_RUN
  if (rc == NB_WORK_UNAVAILABLE) return NB_OK;          // no scratch for even one slab slot: *taken stays 0, the caller runs the one-sided kernel
  if (rc) return rc;
  *taken = 1;
  c->last_accel_sym = true;
  return NB_OK;
}

// Two different blocks, every A-B pair once: A's accelerations (+)= the force of B, B's accelerations (bax..) = the force of A (overwritten).
// *taken = 0: not applicable (small blocks, switched off) — the caller runs the one-sided kernel in both directions.
int nb_launch_accel_cross_f32(nb_ctx* c, const float* xa, const float* ya, const float* za, const float* ma, int64_t na,
                              const float* xb, const float* yb, const float* zb, const float* mb, int64_t nb,
                              float eps2, float G, float* aax, float* aay, float* aaz, float* bax, float* bay, float* baz,
                              double* phi_a, double* phi_b, int flags, int* taken) {
  *taken = 0;
  const SymOpt opt = sym_options(c);
  const long long mode = opt.mode, min_n = opt.min_n, budget = opt.budget;
  const int64_t nmin = std::min(na, nb);
  if (!mode || nmin * 2 < min_n || na >= (1ll << 31) - 65536 || nb >= (1ll << 31) - 65536 || nmin < 1) return NB_OK;   // (na x nb pairs ~ a self launch of sqrt(2 na nb))
  const bool pot = phi_a != nullptr;
  const int tj = pot ? 256 : 512;
  auto items = [&](int ks) { const long long sa = (na + BLOCK * ks - 1) / (BLOCK * ks), sb = (nb + BLOCK * ks - 1) / (BLOCK * ks); return sa * sb * (BLOCK * ks / tj); };
  int ks = (flags & NB_UNIFORM_MASS) && !pot ? 12 : 8;      // as for a set on itself (nb_launch_accel_sym_f32)
  if (ks == 12 && items(12) < 3300) ks = 8;
  if (ks == 8 && items(8) < 700) ks = 4;
#define NB_RUN(K, UU, PT) run_cross<K, UU, PT>(c, xa, ya, za, ma, na, xb, yb, zb, mb, nb, eps2, G, aax, aay, aaz, bax, bay, baz, phi_a, phi_b, flags, budget, opt.blocks, opt.rounds, (int)opt.turn)
  const int rc = pot ? (ks == 8 ? NB_RUN(8, 1, true) : NB_RUN(4, 2, true))
                     : (ks == 12 ? NB_RUN(12, 1, false) : ks == 8 ? NB_RUN(8, 2, false) : NB_RUN(4, 2, false));
#undef NB_RUN
  if (rc == NB_WORK_UNAVAILABLE) return NB_OK;          // no scratch for even one slab slot: *taken stays 0, the caller runs the one-sided kernel
  if (rc) return rc;
  *taken = 1;
  c->last_accel_sym = true;
  return NB_OK;
}

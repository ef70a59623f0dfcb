// fp32 softened all-pairs acceleration for gfx950 (MI355X) — SURVEY.md §8(a) row ⊕1.
//
//   a_i = G * sum_j m_j (r_j - r_i) / (|r_j - r_i|^2 + eps^2)^{3/2}
//
// Design (vector-FP32 bound; not HBM, not MFMA — SURVEY.md §8(d)):
//  * sinks: SoA x/y/z, KS sinks per lane held in VGPRs for the whole kernel (coalesced dword loads,
//    sink i = block*256*KS + k*256 + lane), accumulators in VGPRs.
//  * sources: streamed from the SoA x/y/z/m arrays with coalesced dword loads, re-packed in flight to
//    (x,y,z,m) float4 and staged in a double-buffered LDS tile; the inner loop reads one source per
//    ds_read_b128 with a wave-uniform address (LDS broadcast), so one LDS instruction feeds 13*KS VALU
//    instructions. Global loads of tile t+1 are issued before the math on tile t (register staging),
//    written to the other LDS buffer after it: one barrier per tile.
//  * per interaction: 3 sub, 3 fma (r^2+eps^2), 1 v_rsq_f32, 3 mul (m*rinv^3), 3 fma = 12 full-rate
//    VALU + 1 transcendental. The 12 are issued as 6 packed v_pk_{add,fma,mul}_f32 over a PAIR of
//    sinks per lane: on gfx950 a packed op costs two plain ops (no flop gain) but leaves issue slots in
//    which the quarter-rate v_rsq_f32 overlaps — measured 14 ns vs 16 ns per 64 interactions per SIMD
//    (tools/ubench_valu.hip, profiles/r01_ubench_valu.txt).
//  * accuracy: fp32 partial sums live only for one source tile (TILE sources); at every tile boundary
//    they are flushed into fp64 running totals (3*KS v_add_f64 per TILE*KS*13 VALU ops — free), which
//    removes the O(sqrt(N)) growth of the fp32 accumulation error at N = 2^20.
//  * uniform masses (flag NB_UNIFORM_MASS, the standard equal-mass N-body set-up — every CONCEPT run of the thesis):
//    the per-source `m*rinv` multiply is dropped (11 instead of 12 full-rate ops) and the common mass is applied once
//    with G in the fp64 epilogue; padding sources then sit at 1e20 so that r^2 = +inf and rinv = 0 exactly.
//  * fused potential (POT, requested by passing `phi`): phi_i += m_j rinv costs ONE more packed add per sink pair — the pair
//    potential of the same positions for 1/12 of a second O(N^2) pass (SURVEY.md §8a row ⊕3: "fuse with ⊕1").
//  * small N: the source range is split over gridDim.y so the grid still fills 256 CUs; partial sums go
//    to an fp64 slab [split][3][n_snk] and a second tiny kernel reduces them in a fixed order
//    (bit-reproducible — no float atomics).
#include "nb_common.h"
#include "../../include/nbody_amd.h"
#include <cstdlib>

namespace {

constexpr int BLOCK = 256;
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KS, int U, int TILE, bool GUARD, bool UNIFORM, bool POT>
__global__ void __launch_bounds__(BLOCK)
accel_f32_kernel(const float* __restrict__ xi, const float* __restrict__ yi, const float* __restrict__ zi,
                 int n_snk,
                 const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
                 const float* __restrict__ ms, int n_src, int src_per_split,
                 float eps2, float G,
                 float* __restrict__ ax, float* __restrict__ ay, float* __restrict__ az,
                 double* __restrict__ part, int accumulate, double* __restrict__ phi) {
  static_assert(TILE % BLOCK == 0, "tile must be a multiple of the block size");
  constexpr int LPT = TILE / BLOCK;  // staged sources per thread
  __shared__ float4 tile[2][TILE];

  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * (BLOCK * KS) + tid;

  static_assert(KS % 2 == 0, "sinks are processed as packed pairs");
  static_assert(TILE % U == 0, "tile must be a multiple of the source group");
  constexpr int KP = KS / 2;
  // sink pair p = sinks (2p, 2p+1) of this lane: one 64-bit VGPR pair per coordinate -> v_pk_* math
  v2f px[KP], py[KP], pz[KP];
#pragma unroll
  for (int p = 0; p < KP; ++p) {
    int ia = i0 + (2 * p) * BLOCK, ib = ia + BLOCK;
    ia = ia < n_snk ? ia : n_snk - 1;  // clamp: padded lanes recompute the last sink, never stored
    ib = ib < n_snk ? ib : n_snk - 1;
    px[p] = v2f{xi[ia], xi[ib]}; py[p] = v2f{yi[ia], yi[ib]}; pz[p] = v2f{zi[ia], zi[ib]};
  }
  double tx[KS], ty[KS], tz[KS], tp[KS];   // tp: potential sum_j m_j / sqrt(r^2+eps^2) (POT only; dead code otherwise)
#pragma unroll
  for (int k = 0; k < KS; ++k) { tx[k] = 0.0; ty[k] = 0.0; tz[k] = 0.0; tp[k] = 0.0; }
  // eps^2 lives in a VGPR pair: an SGPR operand halves the VALU issue rate on gfx950 (tools/ubench_valu)
  v2f e2;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2" : "=v"(e2.x), "=v"(e2.y) : "s"(eps2));

  const int jbeg = blockIdx.y * src_per_split;
  const int jend = min(n_src, jbeg + src_per_split);
  const int ntiles = (jend - jbeg + TILE - 1) / TILE;

  float4 stage[LPT];
  auto fetch = [&](int t) {
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      int j = jbeg + t * TILE + l * BLOCK + tid;
      bool ok = j < jend;
      int jc = ok ? j : jend - 1;
      // padding sources carry zero mass: they add exactly 0 (their r^2 + eps^2 > 0, or GUARD zeroes them).
      // In UNIFORM mode the mass is not read per source, so padding sits at 1e20 instead: r^2 overflows to +inf,
      // v_rsq_f32(+inf) = +0, and both the force and the potential term are exactly 0.
      const float far = 1e20f;
      stage[l] = UNIFORM ? make_float4(ok ? xs[jc] : far, ok ? ys[jc] : far, ok ? zs[jc] : far, 0.0f)
                         : make_float4(xs[jc], ys[jc], zs[jc], ok ? ms[jc] : 0.0f);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int l = 0; l < LPT; ++l) tile[buf][l * BLOCK + tid] = stage[l];
  };

  if (ntiles > 0) { fetch(0); commit(0); }
  // make the sink loads complete here, so the compiler's vmcnt bookkeeping does not wait for the
  // prefetched source tile inside the hot loop
#pragma unroll
  for (int p = 0; p < KP; ++p) asm volatile("" : "+v"(px[p]), "+v"(py[p]), "+v"(pz[p]));
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) fetch(t + 1);

    v2f fx[KP], fy[KP], fz[KP], fp[KP];
#pragma unroll
    for (int p = 0; p < KP; ++p) { fx[p] = v2f{0.f, 0.f}; fy[p] = v2f{0.f, 0.f}; fz[p] = v2f{0.f, 0.f}; fp[p] = v2f{0.f, 0.f}; }

    // U sources x KP sink pairs = U*KP independent dependency chains per group, written breadth-first (all
    // chains do stage 1, then all do stage 2, ...) so that a wave never issues an instruction that needs the
    // result of the one just before it: the pk ops of other chains fill the v_rsq_f32 latency.
    for (int j = 0; j < TILE; j += U) {
      float4 s[U];
#pragma unroll
      for (int u = 0; u < U; ++u) s[u] = tile[buf][j + u];  // wave-uniform address: LDS broadcast
      v2f dx[U][KP], dy[U][KP], dz[U][KP], r2[U][KP], rinv[U][KP];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < KP; ++p) {
          dx[u][p] = v2f{s[u].x, s[u].x} - px[p];
          dy[u][p] = v2f{s[u].y, s[u].y} - py[p];
          dz[u][p] = v2f{s[u].z, s[u].z} - pz[p];
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
          if (GUARD) {  // coincident pair (dx=dy=dz=0): contributes nothing
            rinv[u][p].x = r2[u][p].x > e2.x ? rinv[u][p].x : 0.0f;
            rinv[u][p].y = r2[u][p].y > e2.y ? rinv[u][p].y : 0.0f;
          }
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < KP; ++p) {
          const v2f rinv2 = rinv[u][p] * rinv[u][p];
          v2f f;
          if (UNIFORM) {                                                        // common mass applied once, at the end
            f = rinv2 * rinv[u][p];
            if (POT) fp[p] = fp[p] + rinv[u][p];
          } else {
            const v2f mr = v2f{s[u].w, s[u].w} * rinv[u][p];
            f = mr * rinv2;
            if (POT) fp[p] = fp[p] + mr;
          }
          fx[p] = __builtin_elementwise_fma(f, dx[u][p], fx[p]);
          fy[p] = __builtin_elementwise_fma(f, dy[u][p], fy[p]);
          fz[p] = __builtin_elementwise_fma(f, dz[u][p], fz[p]);
        }
    }
#pragma unroll
    for (int p = 0; p < KP; ++p) {
      tx[2 * p] += (double)fx[p].x; ty[2 * p] += (double)fy[p].x; tz[2 * p] += (double)fz[p].x;
      tx[2 * p + 1] += (double)fx[p].y; ty[2 * p + 1] += (double)fy[p].y; tz[2 * p + 1] += (double)fz[p].y;
      if (POT) { tp[2 * p] += (double)fp[p].x; tp[2 * p + 1] += (double)fp[p].y; }
    }

    if (t + 1 < ntiles) commit(buf ^ 1);
    __syncthreads();
  }

  if (gridDim.y == 1) {
    const double g = UNIFORM ? (double)G * (double)ms[0] : (double)G;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int i = i0 + k * BLOCK;
      if (i < n_snk) {
        float ox = (float)(g * tx[k]), oy = (float)(g * ty[k]), oz = (float)(g * tz[k]);
        if (accumulate) { ox += ax[i]; oy += ay[i]; oz += az[i]; }
        ax[i] = ox; ay[i] = oy; az[i] = oz;
        if (POT) {   // potential per unit G: sum_j m_j / sqrt(r^2+eps^2), self term included (removed by the caller)
          const double ph = (UNIFORM ? (double)ms[0] : 1.0) * tp[k];
          phi[i] = accumulate ? phi[i] + ph : ph;
        }
      }
    }
  } else {
    constexpr int NC = POT ? 4 : 3;
    double* p = part + (size_t)blockIdx.y * NC * (size_t)n_snk;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const int i = i0 + k * BLOCK;
      if (i < n_snk) {
        p[i] = tx[k]; p[(size_t)n_snk + i] = ty[k]; p[2 * (size_t)n_snk + i] = tz[k];
        if (POT) p[3 * (size_t)n_snk + i] = tp[k];
      }
    }
  }
}

// out[c][i] (+)= G * sum_s part[s][c][i], fixed summation order
__global__ void __launch_bounds__(256)
accel_reduce_f32_kernel(const double* __restrict__ part, int nsplit, int n_snk, float G, const float* __restrict__ m0,
                        float* __restrict__ ax, float* __restrict__ ay, float* __restrict__ az, int accumulate,
                        double* __restrict__ phi) {
  const int nc = phi ? 4 : 3;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_snk) return;
  double sx = 0.0, sy = 0.0, sz = 0.0, sp = 0.0;
  for (int s = 0; s < nsplit; ++s) {
    const double* p = part + (size_t)s * nc * (size_t)n_snk;
    sx += p[i]; sy += p[(size_t)n_snk + i]; sz += p[2 * (size_t)n_snk + i];
    if (phi) sp += p[3 * (size_t)n_snk + i];
  }
  if (phi) { const double ph = (m0 ? (double)m0[0] : 1.0) * sp; phi[i] = accumulate ? phi[i] + ph : ph; }
  const double g = m0 ? (double)G * (double)m0[0] : (double)G;   // uniform-mass mode: common mass applied here
  float ox = (float)(g * sx), oy = (float)(g * sy), oz = (float)(g * sz);
  if (accumulate) { ox += ax[i]; oy += ay[i]; oz += az[i]; }
  ax[i] = ox; ay[i] = oy; az[i] = oz;
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

template <int KS, int U, int TILE>
void launch_variant(bool guard, bool uniform, dim3 grid, hipStream_t st, const float* xi, const float* yi, const float* zi,
                    int n_snk, const float* xs, const float* ys, const float* zs, const float* ms, int n_src,
                    int sps, float eps2, float G, float* ax, float* ay, float* az, double* part, int accumulate, double* phi) {
#define NB_LAUNCH(GD, UF, PT) accel_f32_kernel<KS, U, TILE, GD, UF, PT><<<grid, BLOCK, 0, st>>>( \
    xi, yi, zi, n_snk, xs, ys, zs, ms, n_src, sps, eps2, G, ax, ay, az, part, accumulate, phi)
  const int v = (guard ? 4 : 0) | (uniform ? 2 : 0) | (phi ? 1 : 0);
  switch (v) {
    case 0: NB_LAUNCH(false, false, false); break;
    case 1: NB_LAUNCH(false, false, true); break;
    case 2: NB_LAUNCH(false, true, false); break;
    case 3: NB_LAUNCH(false, true, true); break;
    case 4: NB_LAUNCH(true, false, false); break;
    case 5: NB_LAUNCH(true, false, true); break;
    case 6: NB_LAUNCH(true, true, false); break;
    default: NB_LAUNCH(true, true, true); break;
  }
#undef NB_LAUNCH
}

}  // namespace

int nb_launch_accel_f32(nb_ctx* c, const float* xi, const float* yi, const float* zi, int64_t n_snk,
                        const float* xs, const float* ys, const float* zs, const float* ms, int64_t n_src,
                        float eps2, float G, float* ax, float* ay, float* az, int flags, double* phi) {
  if (n_snk <= 0) return NB_OK;
  const int accumulate = flags & NB_ACCUMULATE;
  const bool uniform = (flags & NB_UNIFORM_MASS) && n_src > 0;
  if (n_snk >= (1ll << 31) || n_src >= (1ll << 31)) return nb_fail(c, NB_ERR_ARG, "n must be < 2^31");
  c->last_accel_sym = false;
  if (xi == xs && yi == ys && zi == zs && n_snk == n_src) {   // a set on itself: every unordered pair once (accel_sym_f32.hip)
    int taken = 0;
    const int rc = nb_launch_accel_sym_f32(c, xs, ys, zs, ms, n_src, eps2, G, ax, ay, az, flags, phi, &taken);
    if (rc || taken) return rc;
  }
  constexpr int TILE = 512;
  // sinks per lane: 8 (four packed pairs) as soon as there are enough sinks to fill lanes; measured on MI355X
  // (profiles/r01_tune_f32.txt) KS=8/U=2 with many source splits is fastest from N = 2^14 to 2^20.
  static const int env_ks = env_int("NB_F32_KS", 0), env_split = env_int("NB_F32_SPLIT", 0);   // tuning knobs, read once per process
  int ks = env_ks;
  if (ks != 2 && ks != 4 && ks != 8) ks = n_snk >= 16384 ? 8 : (n_snk >= 4096 ? 4 : 2);
  const int spb = BLOCK * ks;
  const int nblk = (int)((n_snk + spb - 1) / spb);
  // split the source range so that the grid has ~16 blocks per CU (256 CUs) when sinks alone cannot
  int nsplit = env_split;
  const int ntiles_total = (int)((n_src + TILE - 1) / TILE);
  if (nsplit <= 0) {
    // ~4096 blocks (8 rounds of the 512 resident KS = 8 blocks): measured A/B at N = 2^20 (profiles/r02_f32_split_sweep.txt) — no split
    // (512 blocks = exactly one round) 238.8 ms, 2-way 232.1, 4-way 228.7, 8-way 227.2: blocks of one round do not finish together, later
    // rounds fill the gaps. The price is an fp64 slab [split][3][N] written and re-read once (0.4 GB at N = 2^20 = 0.07 ms).
    const int target_blocks = 4096;
    nsplit = nblk >= target_blocks ? 1 : (target_blocks + nblk - 1) / nblk;
  }
  if (nsplit > ntiles_total) nsplit = ntiles_total > 0 ? ntiles_total : 1;
  if (nsplit > 65535) nsplit = 65535;
  int tiles_per_split = nsplit > 0 ? (ntiles_total + nsplit - 1) / nsplit : 1;
  if (tiles_per_split < 1) tiles_per_split = 1;
  nsplit = ntiles_total > 0 ? (ntiles_total + tiles_per_split - 1) / tiles_per_split : 1;
  const long long sps_ll = (long long)tiles_per_split * TILE;
  const int sps = sps_ll > 0x7fffffff ? 0x7fffffff : (int)sps_ll;

  double* part = nullptr;
  if (nsplit > 1) {
    int rc = nb_ensure_work(c, (size_t)nsplit * (phi ? 4 : 3) * (size_t)n_snk * sizeof(double));
    if (rc) return rc;
    part = reinterpret_cast<double*>(c->work);
  }
  const bool guard = !(eps2 >= 1e-20f);
  dim3 grid(nblk, nsplit);
#define NB_GO(K, UU) launch_variant<K, UU, TILE>(guard, uniform, grid, c->stream, xi, yi, zi, (int)n_snk, xs, ys, zs, ms, (int)n_src, sps, eps2, G, ax, ay, az, part, accumulate, phi)
  switch (ks) {
    case 2: NB_GO(2, 2); break;
    case 4: NB_GO(4, 2); break;
    default: NB_GO(8, 2); break;
  }
#undef NB_GO
  NB_HIP(c, hipGetLastError());
  if (nsplit > 1) {
    accel_reduce_f32_kernel<<<(int)((n_snk + 255) / 256), 256, 0, c->stream>>>(part, nsplit, (int)n_snk, G, uniform ? ms : nullptr, ax, ay, az, accumulate, phi);
    NB_HIP(c, hipGetLastError());
  }
  return NB_OK;
}

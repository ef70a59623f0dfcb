// Standalone harness for the MFMA-assisted fp32 force kernel: r^2 tiles from v_mfma_f32_32x32x2_f32 on
// tile-centred coordinates, rsq / cube / accumulation on the VALU in the MFMA's shadow. Host side generates a Plummer sphere,
// Morton-sorts it, uploads it and checks sampled sinks against an fp64 direct sum.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_accel mfma_accel.hip && ./mfma_accel [N] [eps] [kfar] [reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BLOCK = 256, WAVES = 4;
constexpr int CHUNK = 32;            // sources per MFMA tile (columns)
constexpr int SUPER = 512;           // sources per LDS stage
constexpr int NCH = SUPER / CHUNK;   // chunks per stage

__device__ inline int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }   // C/D layout of 32x32 MFMA

// one butterfly level of the reduce-scatter over the 32 lanes of a half: W values kept
template <int W>
__device__ inline void bfly(float (&v)[64], bool bit, int mask) {
#pragma unroll
  for (int r = 0; r < W; ++r) {
    const float keep = bit ? v[r + W] : v[r];
    const float send = bit ? v[r] : v[r + W];
    v[r] = keep + __shfl_xor(send, mask);
  }
}

#ifndef PIPE
#define PIPE 1
#endif
#ifdef WPE
#define WPE_ATTR __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
#else
#define WPE_ATTR
#endif
template <bool UNIFORM>
__global__ void __launch_bounds__(BLOCK) WPE_ATTR
accel_mfma_kernel(const float4* __restrict__ snk, int n_tiles, const float4* __restrict__ tile_meta,   // (cx,cy,cz,rho^2)
                  const float4* __restrict__ src, int n_super,
                  const float4* __restrict__ sup_lo, const float4* __restrict__ sup_hi,              // per super-chunk bbox
                  const float4* __restrict__ ch_lo, const float4* __restrict__ ch_hi,                // per chunk bbox
                  float eps2, float kfar2, int flush_every, double* __restrict__ out /* [n_tiles*32][4]: sum s x', y', z', s -> a */) {
  __shared__ float4 stage[2][SUPER];
  __shared__ float4 sinkS[WAVES][32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, h = lane >> 5, col = lane & 31;
  int tile = blockIdx.x * WAVES + wave;
  const bool active = tile < n_tiles;
  if (!active) tile = n_tiles - 1;

  const float4 meta = tile_meta[tile];
  const float cx = meta.x, cy = meta.y, cz = meta.z, near2 = kfar2 * meta.w;
  const float4 me = snk[(size_t)tile * 32 + col];
  const float xi = me.x - cx, yi = me.y - cy, zi = me.z - cz;
  const float si2 = fmaf(xi, xi, fmaf(yi, yi, fmaf(zi, zi, eps2)));
  if (h == 0) sinkS[wave][col] = make_float4(xi, yi, zi, si2);
  // r^2 = si2*1 + xi'(-2 xj') + yi'(-2 yj') + zi'(-2 zj') + 1*sj2 as a K = 6 fmaf chain over three 32x32x2 MFMAs (C = 0 inline):
  // k = 0,1 (lanes < 32 / >= 32 supply the even / odd k of each instruction): A = (si2, 0)    B = (1, 0)
  // k = 2,3:                                                                   A = (-2xi', -2yi')  B = (xj', yj')
  // k = 4,5:                                                                   A = (-2zi', 1)      B = (zj', sj2)
  const float A0 = h ? 0.0f : si2, B0 = h ? 0.0f : 1.0f;
  const float A1 = h ? -2.0f * yi : -2.0f * xi;
  const float A2 = h ? 1.0f : -2.0f * zi;
  __syncthreads();

  float fx[16], fy[16], fz[16], fs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { fx[r] = 0.f; fy[r] = 0.f; fz[r] = 0.f; fs[r] = 0.f; }
  double T0 = 0.0, T1 = 0.0;

  // LDS staging of the sources, double-buffered: thread t loads elements t and t+256 of a super-chunk
  float4 pre0, pre1;
  auto fetch = [&](int s) { pre0 = src[(size_t)s * SUPER + tid]; pre1 = src[(size_t)s * SUPER + BLOCK + tid]; };
  auto commit = [&](int buf) { stage[buf][tid] = pre0; stage[buf][BLOCK + tid] = pre1; };
  fetch(0); commit(0);
  __syncthreads();

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // matrix-pipe half of a far chunk: tile-relative source coordinates and the 32 x 32 block of r^2 + eps^2
  auto far_r2 = [&](const float4 q, float& xj, float& yj, float& zj) -> f32x16 {
    xj = q.x - cx; yj = q.y - cy; zj = q.z - cz;
    const float sj2 = fmaf(xj, xj, fmaf(yj, yj, zj * zj));
    const float B1 = h ? yj : xj;
    const float B2 = h ? sj2 : zj;
#ifdef NO_MFMA      // timing experiment only: a fake r^2 without the matrix pipe
    f32x16 r2;
#pragma unroll
    for (int r = 0; r < 16; ++r) r2[r] = B1 + B2 + A0;
    asm volatile("" : "+v"(r2));
    return r2;
#else
    f32x16 r2 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, zero16, 0, 0, 0);
    r2 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, r2, 0, 0, 0);
    r2 = __builtin_amdgcn_mfma_f32_32x32x2f32(A2, B2, r2, 0, 0, 0);
    return r2;
#endif
  };
  // vector half: s = (r^2+eps^2)^{-3/2}, sum s xj', sum s  (16 sinks per lane, this lane's source column)
  auto far_acc = [&](const f32x16 r2, float xj, float yj, float zj, float mj) {
    float wx = xj, wy = yj, wz = zj;
    if (!UNIFORM) { wx *= mj; wy *= mj; wz *= mj; }
#ifdef NO_VALU      // timing experiment only: consume the MFMA result with one op
    fx[0] += r2[0] + r2[15];
    return;
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ri = __builtin_amdgcn_rsqf(r2[r]);
      const float s = ri * ri * ri;
      fx[r] = fmaf(s, wx, fx[r]); fy[r] = fmaf(s, wy, fy[r]); fz[r] = fmaf(s, wz, fz[r]);
      if (UNIFORM) fs[r] += s; else fs[r] = fmaf(s, mj, fs[r]);
    }
  };
  auto far_chunk = [&](const float4 q) {
    float xj, yj, zj;
    const f32x16 r2 = far_r2(q, xj, yj, zj);
    far_acc(r2, xj, yj, zj, q.w);
  };
  auto near_chunk = [&](const float4 q) {
    const float xj = q.x - cx, yj = q.y - cy, zj = q.z - cz;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 p = sinkS[wave][rowmap(r, h)];
      const float dx = xj - p.x, dy = yj - p.y, dz = zj - p.z;
      const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));
      const float ri = __builtin_amdgcn_rsqf(r2);
      float s = ri * ri * ri;
      if (!UNIFORM) s *= q.w;
      // the two halves of the wave hold the SAME 32 source columns: each half sums all 32 columns for its own 16 sink rows
      fx[r] = fmaf(s, dx, fx[r]); fy[r] = fmaf(s, dy, fy[r]); fz[r] = fmaf(s, dz, fz[r]);
      __builtin_amdgcn_sched_barrier(0);     // rare path: keep it compact, do not hoist the 16 LDS reads (register pressure)
    }
  };
  auto flush = [&]() {
    float v[64];
#pragma unroll
    for (int r = 0; r < 16; ++r) { v[4 * r] = fx[r]; v[4 * r + 1] = fy[r]; v[4 * r + 2] = fz[r]; v[4 * r + 3] = fs[r]; fx[r] = 0.f; fy[r] = 0.f; fz[r] = 0.f; fs[r] = 0.f; }
    bfly<32>(v, lane & 1, 1); bfly<16>(v, lane & 2, 2); bfly<8>(v, lane & 4, 4); bfly<4>(v, lane & 8, 8); bfly<2>(v, lane & 16, 16);
    T0 += (double)v[0]; T1 += (double)v[1];
  };

  for (int s = 0; s < n_super; ++s) {
    const int buf = s & 1;
    if (s + 1 < n_super) fetch(s + 1);
    if (active) {
      // distance of the tile centre from the bounding box of the 512 sources of this stage (wave-uniform)
      const float4 lo = sup_lo[s], hi = sup_hi[s];
      const float gx = fmaxf(fmaxf(lo.x - cx, cx - hi.x), 0.f), gy = fmaxf(fmaxf(lo.y - cy, cy - hi.y), 0.f), gz = fmaxf(fmaxf(lo.z - cz, cz - hi.z), 0.f);
      const bool near_super = fmaf(gx, gx, fmaf(gy, gy, gz * gz)) < near2;
      if (!__builtin_amdgcn_readfirstlane(near_super)) {
#if !PIPE
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) far_chunk(stage[buf][c * CHUNK + col]);
#else
        // software pipeline: the MFMAs of chunk c+1 are issued before the vector work on chunk c, which runs in their shadow
        float xa, ya, za, xb, yb, zb;
        float4 qa = stage[buf][col], qb;
        f32x16 ra = far_r2(qa, xa, ya, za), rb;
#pragma unroll 1
        for (int c = 0; c < NCH; c += 2) {
          qb = stage[buf][(c + 1) * CHUNK + col];
          rb = far_r2(qb, xb, yb, zb);
          far_acc(ra, xa, ya, za, qa.w);
          if (c + 2 < NCH) { qa = stage[buf][(c + 2) * CHUNK + col]; ra = far_r2(qa, xa, ya, za); }
          far_acc(rb, xb, yb, zb, qb.w);
        }
#endif
      } else {
        for (int c = 0; c < NCH; ++c) {
          const float4 l2 = ch_lo[s * NCH + c], h2 = ch_hi[s * NCH + c];
          const float ex = fmaxf(fmaxf(l2.x - cx, cx - h2.x), 0.f), ey = fmaxf(fmaxf(l2.y - cy, cy - h2.y), 0.f), ez = fmaxf(fmaxf(l2.z - cz, cz - h2.z), 0.f);
          const bool near = fmaf(ex, ex, fmaf(ey, ey, ez * ez)) < near2;
          const float4 q = stage[buf][c * CHUNK + col];
          if (__builtin_amdgcn_readfirstlane(near)) near_chunk(q); else far_chunk(q);
        }
      }
      if ((s + 1) % flush_every == 0) flush();
    }
    if (s + 1 < n_super) commit(buf ^ 1);
    __syncthreads();
  }
  if (!active) return;
  flush();
  // lane bits (within its half): q = 8 b0 + 4 b1 + 2 b2 + b3 is the sink register index, b4 selects (x,y) or (z,s)
  const int q = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
  const int row = rowmap(q, h);
  const float4 p = sinkS[wave][row];
  // partner (xor 16) holds the other two components of the same sink
  const double P1 = __shfl_xor(T1, 16);
  double* o = out + ((size_t)tile * 32 + row) * 4;
  if (!(lane & 16)) { o[0] = T0 - (double)p.x * P1; o[1] = T1 - (double)p.y * P1; }      // near-path sums carry no xi' * sum(s) term: see note
  else { o[2] = T0 - (double)p.z * T1; o[3] = T1; }
}

// ------------------------------------------------------------------ host side
static uint64_t rng_s = 88172645463325252ull;
static double urand() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (rng_s >> 11) * (1.0 / 9007199254740992.0); }
static uint64_t spread21(uint32_t v) {
  uint64_t x = v & 0x1fffff;
  x = (x | x << 32) & 0x1f00000000ffffull; x = (x | x << 16) & 0x1f0000ff0000ffull; x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull; x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1 << 20;
  const float eps = argc > 2 ? (float)atof(argv[2]) : 0.01f;
  const float kfar = argc > 3 ? (float)atof(argv[3]) : 2.0f;
  const int reps = argc > 4 ? atoi(argv[4]) : 3;
  const int flush_every = argc > 5 ? atoi(argv[5]) : 8;
  const int uniform = argc > 6 ? atoi(argv[6]) : 1;
  if (n % SUPER) { printf("n must be a multiple of %d in this harness\n", SUPER); return 1; }
  std::vector<float4> p(n);
  const double a = 3.0 * M_PI / 16.0;
  for (int i = 0; i < n; ++i) {
    double r;
    do { double u = urand(); r = a / sqrt(pow(u, -2.0 / 3.0) - 1.0); } while (r > 10 * a);
    double ct = 2 * urand() - 1, ph = 2 * M_PI * urand(), st = sqrt(1 - ct * ct);
    p[i] = make_float4((float)(r * st * cos(ph)), (float)(r * st * sin(ph)), (float)(r * ct), uniform ? 1.0f / n : (float)((0.5 + urand()) / n));
  }
  float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
  for (auto& q : p) { float v[3] = {q.x, q.y, q.z}; for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], v[d]); hi[d] = std::max(hi[d], v[d]); } }
  const float ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2])) * 1.0001f;
  std::vector<std::pair<uint64_t, int>> key(n);
  for (int i = 0; i < n; ++i) {
    uint32_t qx = (uint32_t)((p[i].x - lo[0]) / ext * 2097152.0f), qy = (uint32_t)((p[i].y - lo[1]) / ext * 2097152.0f), qz = (uint32_t)((p[i].z - lo[2]) / ext * 2097152.0f);
    key[i] = {spread21(qx) << 2 | spread21(qy) << 1 | spread21(qz), i};
  }
  std::sort(key.begin(), key.end());
  std::vector<float4> sp(n);
  for (int i = 0; i < n; ++i) sp[i] = p[key[i].second];
  const int n_tiles = n / 32, n_super = n / SUPER, n_chunks = n / CHUNK;
  std::vector<float4> meta(n_tiles), slo(n_super), shi(n_super), clo(n_chunks), chi(n_chunks);
  auto box = [&](int b, int cnt, float4& l, float4& hh) {
    l = make_float4(1e30f, 1e30f, 1e30f, 0); hh = make_float4(-1e30f, -1e30f, -1e30f, 0);
    for (int i = b; i < b + cnt; ++i) { l.x = std::min(l.x, sp[i].x); l.y = std::min(l.y, sp[i].y); l.z = std::min(l.z, sp[i].z); hh.x = std::max(hh.x, sp[i].x); hh.y = std::max(hh.y, sp[i].y); hh.z = std::max(hh.z, sp[i].z); }
  };
  for (int t = 0; t < n_tiles; ++t) {
    float4 l, hh; box(t * 32, 32, l, hh);
    float cx = 0.5f * (l.x + hh.x), cy = 0.5f * (l.y + hh.y), cz = 0.5f * (l.z + hh.z), r2 = 0;
    for (int i = t * 32; i < t * 32 + 32; ++i) { float dx = sp[i].x - cx, dy = sp[i].y - cy, dz = sp[i].z - cz; r2 = std::max(r2, dx * dx + dy * dy + dz * dz); }
    meta[t] = make_float4(cx, cy, cz, r2);
  }
  for (int s = 0; s < n_super; ++s) box(s * SUPER, SUPER, slo[s], shi[s]);
  for (int c = 0; c < n_chunks; ++c) box(c * CHUNK, CHUNK, clo[c], chi[c]);

  float4 *d_sp, *d_meta, *d_slo, *d_shi, *d_clo, *d_chi; double* d_out;
  CK(hipMalloc(&d_sp, n * sizeof(float4))); CK(hipMalloc(&d_meta, n_tiles * sizeof(float4)));
  CK(hipMalloc(&d_slo, n_super * sizeof(float4))); CK(hipMalloc(&d_shi, n_super * sizeof(float4)));
  CK(hipMalloc(&d_clo, n_chunks * sizeof(float4))); CK(hipMalloc(&d_chi, n_chunks * sizeof(float4)));
  CK(hipMalloc(&d_out, (size_t)n * 4 * sizeof(double)));
  CK(hipMemcpy(d_sp, sp.data(), n * sizeof(float4), hipMemcpyHostToDevice)); CK(hipMemcpy(d_meta, meta.data(), n_tiles * sizeof(float4), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_slo, slo.data(), n_super * sizeof(float4), hipMemcpyHostToDevice)); CK(hipMemcpy(d_shi, shi.data(), n_super * sizeof(float4), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_clo, clo.data(), n_chunks * sizeof(float4), hipMemcpyHostToDevice)); CK(hipMemcpy(d_chi, chi.data(), n_chunks * sizeof(float4), hipMemcpyHostToDevice));
  CK(hipMemset(d_out, 0, (size_t)n * 4 * sizeof(double)));
  const int grid = (n_tiles + WAVES - 1) / WAVES;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int it = 0; it < reps; ++it) {
    CK(hipEventRecord(e0));
    if (uniform) accel_mfma_kernel<true><<<grid, BLOCK>>>(d_sp, n_tiles, d_meta, d_sp, n_super, d_slo, d_shi, d_clo, d_chi, eps * eps, kfar * kfar, flush_every, d_out);
    else accel_mfma_kernel<false><<<grid, BLOCK>>>(d_sp, n_tiles, d_meta, d_sp, n_super, d_slo, d_shi, d_clo, d_chi, eps * eps, kfar * kfar, flush_every, d_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("launch %d: %.3f ms = %.3e interactions/s\n", it, ms, (double)n * n / (ms * 1e-3));
    best = std::min(best, ms);
  }
  std::vector<double> out((size_t)n * 4);
  CK(hipMemcpy(out.data(), d_out, out.size() * sizeof(double), hipMemcpyDeviceToHost));
  // sampled sinks vs fp64 direct sum on the same fp32 inputs
  double worst = 0, sum = 0; int cnt = 0;
  for (int k = 0; k < 256; ++k) {
    const int i = (int)((long long)k * (n - 1) / 255);
    double rx = 0, ry = 0, rz = 0;
    for (int j = 0; j < n; ++j) {
      double dx = (double)sp[j].x - sp[i].x, dy = (double)sp[j].y - sp[i].y, dz = (double)sp[j].z - sp[i].z;
      double d2 = dx * dx + dy * dy + dz * dz + (double)eps * eps; double di = 1.0 / sqrt(d2); double s = (uniform ? 1.0 : (double)sp[j].w) * di * di * di;
      rx += s * dx; ry += s * dy; rz += s * dz;
    }
    const double* o = &out[(size_t)i * 4];
    double e = sqrt((o[0] - rx) * (o[0] - rx) + (o[1] - ry) * (o[1] - ry) + (o[2] - rz) * (o[2] - rz)) / sqrt(rx * rx + ry * ry + rz * rz);
    worst = std::max(worst, e); sum += e; ++cnt;
  }
  printf("N=%d eps=%g kfar=%g flush=%d uniform=%d: best %.3f ms = %.4e interactions/s (%.3f of the 20-flop fp32 roofline); max rel err over %d sampled sinks %.3e (mean %.3e)\n",
         n, eps, kfar, flush_every, uniform, best, (double)n * n / (best * 1e-3), (double)n * n / (best * 1e-3) * 20 / 157.3e12, cnt, worst, sum / cnt);
  return 0;
}

// Barnes-Hut tree gravity for gfx950 (BASELINE.json configs[4]: theta = 0.5, N = 16M, one MI355X, fp32).
//
// The reference contains no tree code; the algorithm is fixed here and restated sequentially in
// oracle/bh_oracle.c so that the GPU result can be checked node for node:
//   1. bounding cube of all particles; 63-bit Morton key per particle (21 bits per axis)        [morton_kernel]
//   2. radix sort of (key, index) pairs (rocPRIM), gather to sorted (x,y,z,m) float4               [gather_kernel]
//   3. binary radix tree over the sorted keys (Karras 2012): internal node i in [0, n-1), leaf j = node n-1+j; every
//      node covers a CONTIGUOUS range of the sorted particles                                      [build_kernel]
//   4. node mass / centre of mass from fp64 prefix sums of m, m*x, m*y, m*z over the sorted order (no bottom-up
//      pass, no inter-workgroup hand-off); node size l = edge of the Morton cell of the node's common prefix
//                                                                                                  [node_kernel]
//   5. traversal, one 64-lane wave per group of 64 consecutive sorted particles. Two walks implement it:
//      (list walk, default) [walk_list_kernel]: every pass pops up to 64 nodes from a wave-private stack, ONE NODE PER
//      LANE, so 64 node records are fetched and tested at once instead of one dependent fetch per node. A node is
//      accepted when l^2 < theta^2 d^2 for every particle of the group: d = the distance from the node's centre of mass to the nearest of
//      the group's acceptance volumes (the bounding boxes of its 8 runs of 8 consecutive particles, in LDS; a run that straddles a jump
//      of the Morton curve is entered particle by particle) — a lower bound of every particle's distance;
//      wave ballots compact the accepted nodes into an LDS monopole list, rejected nodes with <= NLEAF particles into
//      a leaf list, and push the children of the rest. Then all lanes sweep the lists for their own particle: monopoles
//      as LDS broadcasts, leaf particles staged 64 at a time into an LDS tile (the direct kernel's inner loop).
//      (ballot walk, NB_BH_WALK=0) [walk_kernel]: one node at a time, accepted when l^2 < theta^2 d^2 holds for EVERY
//      lane's own particle (__all); 15x slower (latency bound: one dependent node fetch per step), kept as a cross-check.
//      Accepted nodes interact as monopoles: a += G m_node d / (d^2 + eps^2)^{3/2}.
// Irregular, latency/issue bound (not a roofline kernel): measured numbers are in DESIGN.md.
#include "nb_common.h"
#include "../../include/nbody_amd.h"
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int NLEAF = 16;          // ballot walk: a rejected node with at most this many particles is summed directly
constexpr int NLEAF_LIST = 4;      // list walk: with the 8-box acceptance test a node test is cheap, opening small nodes beats summing them
                                   // (N = 16M sweep, profiles/r02_bh_nleaf_sweep.txt: 2: 42.5 ms, 4: 41.2, 8: 42.3, 16: 46.0, 24: 50.2)
constexpr int STACK = 192;         // per-wave stack entries (binary tree depth <= 63 key bits + 32 index bits)

struct BHBuf {
  float* bbox = nullptr;                     // 6 floats: min xyz, max xyz ; then [6] = cube edge
  unsigned long long *keys = nullptr, *keys_sorted = nullptr;
  int *idx = nullptr, *idx_sorted = nullptr;
  float4* sp = nullptr;                      // sorted (x,y,z,m)
  double* pre = nullptr;                     // 4 prefix arrays [4][n] (m, mx, my, mz), inclusive
  int2* child = nullptr;                     // internal nodes: (left, right) node ids
  int2* range = nullptr;                     // all nodes: (first, count)
  float4* com = nullptr;                     // all nodes: (cx, cy, cz, mass)
  float* l2 = nullptr;                       // all nodes: cell edge squared
  float4* pk = nullptr;                      // (x,y,z,m) in the caller's order: one 16-byte read per particle for the gather
  int4* rec = nullptr;                       // all nodes: (l2 bits, first, count, left child) for the list walk
  int* gstack = nullptr;                     // wave-private stacks of the persistent list-walk waves
  int stack_blocks = 0;                      // blocks gstack was sized for: the list walk never launches more
  int* flags = nullptr;                      // [0] = stack overflow seen
  void* tmp = nullptr; size_t tmp_bytes = 0; // rocPRIM scratch
  int64_t cap = 0;
};

__device__ inline unsigned long long spread21(unsigned int v) {
  unsigned long long x = v & 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__global__ void __launch_bounds__(256)
bbox_partial_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n,
                    float* __restrict__ part) {
  float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float v[3] = {x[i], y[i], z[i]};
#pragma unroll
    for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], v[c]); hi[c] = fmaxf(hi[c], v[c]); }
  }
  __shared__ float red[6][256];
#pragma unroll
  for (int c = 0; c < 3; ++c) { red[c][threadIdx.x] = lo[c]; red[3 + c][threadIdx.x] = hi[c]; }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        red[c][threadIdx.x] = fminf(red[c][threadIdx.x], red[c][threadIdx.x + w]);
        red[3 + c][threadIdx.x] = fmaxf(red[3 + c][threadIdx.x], red[3 + c][threadIdx.x + w]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void bbox_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ bbox) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int b = 0; b < nparts; ++b)
      for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], part[b * 6 + c]); hi[c] = fmaxf(hi[c], part[b * 6 + 3 + c]); }
    float edge = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
    edge = edge > 0.f ? edge * 1.0001f : 1.0f;       // strictly inside the cube; a single point gets a unit cube
    for (int c = 0; c < 3; ++c) { bbox[c] = lo[c]; bbox[3 + c] = hi[c]; }
    bbox[6] = edge;
  }
}

__global__ void __launch_bounds__(256)
morton_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, const float* __restrict__ m, int n,
              const float* __restrict__ bbox, unsigned long long* __restrict__ keys, int* __restrict__ idx, float4* __restrict__ pk) {
  const float edge = bbox[6];
  const float scale = 2097152.0f / edge;   // 2^21 cells per axis
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    unsigned int q[3];
    const float v[3] = {x[i], y[i], z[i]};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float t = (v[c] - bbox[c]) * scale;
      t = fminf(fmaxf(t, 0.0f), 2097151.0f);
      q[c] = (unsigned int)t;
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    idx[i] = i;
    pk[i] = make_float4(v[0], v[1], v[2], m[i]);
  }
}

__global__ void __launch_bounds__(256)
gather_kernel(const float4* __restrict__ pk, const int* __restrict__ idx, int n, float4* __restrict__ sp, double* __restrict__ pre) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = pk[idx[i]];
    sp[i] = p;
    const double mm = (double)p.w;
    pre[i] = mm; pre[(size_t)n + i] = mm * (double)p.x; pre[2 * (size_t)n + i] = mm * (double)p.y; pre[3 * (size_t)n + i] = mm * (double)p.z;
  }
}

// length of the common prefix of the (key, position) strings of sorted elements i and j; -1 outside [0, n)
__device__ inline int delta(const unsigned long long* __restrict__ k, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = k[i], b = k[j];
  if (a != b) return __clzll(a ^ b);
  return 64 + __clz((unsigned int)i ^ (unsigned int)j);
}

// Karras (2012) binary radix tree: internal node i covers the sorted range [first, last]; children are internal
// nodes or leaves (leaf j has node id n-1+j).
__global__ void __launch_bounds__(256)
build_kernel(const unsigned long long* __restrict__ k, int n, int2* __restrict__ child, int2* __restrict__ range,
             float* __restrict__ l2, const float* __restrict__ bbox) {
  const float edge = bbox[6];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
    const int d = delta(k, n, i, i + 1) - delta(k, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(k, n, i, i - d);
    int lmax = 2;
    while (delta(k, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
      if (delta(k, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(k, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
      if (delta(k, n, i, i + (s + t) * d) > dnode) s += t;
      if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    const int left = (first == gamma) ? (n - 1 + gamma) : gamma;
    const int right = (last == gamma + 1) ? (n - 1 + gamma + 1) : (gamma + 1);
    child[i] = make_int2(left, right);
    range[i] = make_int2(first, last - first + 1);
    // Morton cell of the common key prefix: level = prefix bits / 3 (the key's top bit is unused: 63 = 3*21 bits)
    const int pbits = dnode >= 64 ? 63 : max(dnode - 1, 0);
    const int level = pbits / 3;
    const float cell = ldexpf(edge, -level);
    l2[i] = cell * cell;
  }
}

__global__ void __launch_bounds__(256)
node_kernel(const double* __restrict__ pre, const float4* __restrict__ sp, int n, int2* __restrict__ range,
            float4* __restrict__ com, float* __restrict__ l2, const int2* __restrict__ child, int4* __restrict__ rec) {
  const int total = 2 * n - 1;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gridDim.x * blockDim.x) {
    if (v >= n - 1) {                       // leaf
      const int j = v - (n - 1);
      range[v] = make_int2(j, 1);
      com[v] = sp[j];
      l2[v] = 0.0f;
      rec[v] = make_int4(0, j, 1, -1);
      continue;
    }
    const int2 r = range[v];
    const int a = r.x, b = r.x + r.y - 1;
    double s[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) s[c] = pre[(size_t)c * n + b] - (a > 0 ? pre[(size_t)c * n + a - 1] : 0.0);
    float4 o;
    if (s[0] > 0.0) o = make_float4((float)(s[1] / s[0]), (float)(s[2] / s[0]), (float)(s[3] / s[0]), (float)s[0]);
    else {                                  // massless range: geometric anchor = first particle, mass 0
      const float4 p = sp[a];
      o = make_float4(p.x, p.y, p.z, 0.0f);
    }
    com[v] = o;
    rec[v] = make_int4(__float_as_int(l2[v]), r.x, r.y, child[v].x);
  }
}

template <bool GUARD>
__global__ void __launch_bounds__(256)
walk_kernel(const float4* __restrict__ sp, int n, const int2* __restrict__ child, const int2* __restrict__ range,
            const float4* __restrict__ com, const float* __restrict__ l2, float theta2, float eps2,
            const int* __restrict__ idx, float G, float* __restrict__ out_x, float* __restrict__ out_y, float* __restrict__ out_z,
            int* __restrict__ flags) {
  __shared__ int stack_all[4][STACK];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  volatile int* stack = stack_all[wave];
  const int group = blockIdx.x * 4 + wave;
  const int i = group * 64 + lane;
  const bool live = i < n;
  const float4 me = sp[live ? i : n - 1];
  float ax = 0.f, ay = 0.f, az = 0.f;
  if (group * 64 >= n) return;              // whole wave out of range (wave-uniform)

  int sp_top = 0;
  if (lane == 0) stack[0] = 0;              // root = internal node 0 (n >= 2 guaranteed by the launcher)
  sp_top = 1;
  while (sp_top > 0) {
    const int v = __builtin_amdgcn_readfirstlane(stack[sp_top - 1]);
    --sp_top;
    const float4 c = com[v];
    const float dx = c.x - me.x, dy = c.y - me.y, dz = c.z - me.z;
    const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));   // same expression as oracle/bh_oracle.c: identical MAC decisions
    const int2 r = range[v];
    const bool is_leaf = v >= n - 1;
    // accept (all lanes): cell small enough seen from every particle of the group; padding lanes vote yes
    const bool ok = is_leaf ? true : (!live || l2[v] < theta2 * d2);
    if (!is_leaf && __all(ok)) {
      const float r2 = d2 + eps2;
      float rinv = __builtin_amdgcn_rsqf(r2);
      if (GUARD) rinv = r2 > 0.f ? rinv : 0.f;
      const float f = c.w * rinv * rinv * rinv;
      ax = fmaf(f, dx, ax); ay = fmaf(f, dy, ay); az = fmaf(f, dz, az);
    } else if (is_leaf || r.y <= NLEAF) {
      for (int j = r.x; j < r.x + r.y; ++j) {
        const float4 q = sp[j];                            // wave-uniform address
        const float ex = q.x - me.x, ey = q.y - me.y, ez = q.z - me.z;
        const float r2 = fmaf(ex, ex, fmaf(ey, ey, ez * ez)) + eps2;
        float rinv = __builtin_amdgcn_rsqf(r2);
        if (GUARD) rinv = r2 > 0.f ? rinv : 0.f;           // the particle itself / coincident pairs at eps = 0
        const float f = q.w * rinv * rinv * rinv;
        ax = fmaf(f, ex, ax); ay = fmaf(f, ey, ay); az = fmaf(f, ez, az);
      }
    } else {
      const int2 ch = child[v];
      if (sp_top + 2 <= STACK) {
        if (lane == 0) { stack[sp_top] = ch.y; stack[sp_top + 1] = ch.x; }   // left child is popped first
        sp_top += 2;
      } else if (lane == 0) flags[0] = 1;                                    // overflow: results invalid, reported by the host
    }
  }
  if (live) { const int j = idx[i]; out_x[j] = G * ax; out_y[j] = G * ay; out_z[j] = G * az; }   // straight to the caller's particle order
}

constexpr int LSTACK = 12288;      // entries per wave-private stack (<= 128 nodes left behind per tree level, depth <= 96)

// The list walk (see the header). Stack entries live in global memory (L2-resident, wave-private); entries written by
// one lane are read by another lane of the same wave in a later pass, so every store is drained (s_waitcnt vmcnt(0))
// before the next pass loads, and the loads bypass the vector L1.
#ifndef NB_BH_UMAC
#define NB_BH_UMAC 4
#endif
#ifndef NB_BH_USWEEP
#define NB_BH_USWEEP 8
#endif
#ifndef NB_BH_WPE
#define NB_BH_WPE 6                // waves per SIMD the list walk is compiled for (79 VGPRs; N = 16M: 4: 43.5 ms, 5: 42.2, 6: 41.7, 7: 41.9)
#endif
template <bool GUARD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB_BH_WPE, NB_BH_WPE)))
walk_list_kernel(const float4* __restrict__ sp, int n, const float4* __restrict__ com, const int4* __restrict__ rec,
                 float theta2, float eps2, const int* __restrict__ idx, float G, float* __restrict__ out_x, float* __restrict__ out_y,
                 float* __restrict__ out_z, int* __restrict__ gstack, int ngroups, int* __restrict__ flags, int nleaf) {
  __shared__ float4 mono_all[4][64];
  __shared__ float4 tile_all[4][64];
  __shared__ float4 vlo_all[4][64];
  __shared__ float4 vhi_all[4][64];
  __shared__ int lf_first_all[4][64];
  __shared__ int lf_pref_all[4][65];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // wave-private LDS lists: written by some lanes, then read by all lanes of the SAME wave. LDS operations of one wave
  // execute in order, so only the compiler has to be told (wavefront-scope fences around a wave barrier) — volatile
  // accesses would serialise the sweeps below into dependent dword reads (measured 15x slower).
  float4* mono = mono_all[wave];
  float4* tile = tile_all[wave];
  float4* vlo = vlo_all[wave];
  float4* vhi = vhi_all[wave];
  int* lf_first = lf_first_all[wave];
  int* lf_pref = lf_pref_all[wave];
#define NB_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
  const int wid = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
  int* stack = gstack + (size_t)wid * LSTACK;

  // groups are handed out dynamically (flags[1] is a work counter zeroed by the host): their cost varies by >10x
  // between the halo and the core, static striding left a tail of late waves
  (void)nwaves;
  for (;;) {
    int group = 0;
    if (lane == 0) group = atomicAdd(&flags[1], 1);
    group = __builtin_amdgcn_readfirstlane(group);
    if (group >= ngroups) break;
    const int i = group * 64 + lane;
    const bool live = i < n;
    const float4 me = sp[live ? i : n - 1];
    // Acceptance volumes of the group: the bounding boxes of its 8 runs of 8 consecutive (Morton-ordered) particles. A node is accepted when
    // l^2 < theta^2 d^2 with d the distance from its centre of mass to the NEAREST of these boxes — a lower bound of the distance to every
    // particle of the group, so the criterion holds for each of them (conservative: never larger errors than the per-particle test, a few per
    // cent more interactions), at 8 box tests per node instead of 64 particle tests (the exact scan was ~40 % of the walk's VALU work).
    // 64 CONSECUTIVE Morton particles can straddle a jump of the curve; one box spanning the jump would force everything between its two
    // ends to be opened to the leaves (measured with a single group box: a few groups ran 100x longer). A run whose box diagonal exceeds 0.6
    // of the group's is therefore entered particle by particle (8 degenerate boxes): the list has 8 .. 64 entries, 64 being the exact test.
    float lox = me.x, loy = me.y, loz = me.z, hix = me.x, hiy = me.y, hiz = me.z;
#pragma unroll
    for (int w = 1; w < 8; w <<= 1) {
      lox = fminf(lox, __shfl_xor(lox, w)); loy = fminf(loy, __shfl_xor(loy, w)); loz = fminf(loz, __shfl_xor(loz, w));
      hix = fmaxf(hix, __shfl_xor(hix, w)); hiy = fmaxf(hiy, __shfl_xor(hiy, w)); hiz = fmaxf(hiz, __shfl_xor(hiz, w));
    }
    float glx = lox, gly = loy, glz = loz, ghx = hix, ghy = hiy, ghz = hiz;
#pragma unroll
    for (int w = 8; w < 64; w <<= 1) {
      glx = fminf(glx, __shfl_xor(glx, w)); gly = fminf(gly, __shfl_xor(gly, w)); glz = fminf(glz, __shfl_xor(glz, w));
      ghx = fmaxf(ghx, __shfl_xor(ghx, w)); ghy = fmaxf(ghy, __shfl_xor(ghy, w)); ghz = fmaxf(ghz, __shfl_xor(ghz, w));
    }
    const float bx = hix - lox, by = hiy - loy, bz = hiz - loz, gx_ = ghx - glx, gy_ = ghy - gly, gz_ = ghz - glz;
    const float diag2 = fmaf(bx, bx, fmaf(by, by, bz * bz)), gdiag2 = fmaf(gx_, gx_, fmaf(gy_, gy_, gz_ * gz_));
    const bool big = diag2 > 0.36f * gdiag2;                               // same for the 8 lanes of a run
    const bool lead = (lane & 7) == 0 && !big;
    const unsigned long long m_lead = __ballot(lead), m_big = __ballot(big), below_l = (1ull << lane) - 1ull;
    const int n_lead = __popcll(m_lead), nvol = n_lead + __popcll(m_big);
    if (lead) { const int e = __popcll(m_lead & below_l); vlo[e] = make_float4(lox, loy, loz, 0.f); vhi[e] = make_float4(hix, hiy, hiz, 0.f); }
    if (big) { const int e = n_lead + __popcll(m_big & below_l); vlo[e] = me; vhi[e] = me; }
    NB_WAVE_SYNC();
    float ax = 0.f, ay = 0.f, az = 0.f;
    int top = 1;
    if (lane == 0) stack[0] = 0;                 // root
    while (top > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stack stores of the previous pass have reached L2
      const int k = top < 64 ? top : 64;
      top -= k;
      int v = -1;
      if (lane < k) v = __hip_atomic_load(&stack[top + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
      int4 r = make_int4(0, 0, 0, -1);
      if (v >= 0) { c = com[v]; r = rec[v]; }
      // squared distance from this lane's node to the nearest acceptance volume of the group (wave-uniform trip count)
      float d2 = 3.4e38f;
#pragma unroll NB_BH_UMAC
      for (int e = 0; e < nvol; ++e) {
        const float4 lo = vlo[e], hi = vhi[e];                        // LDS broadcasts
        const float ex = fmaxf(fmaxf(lo.x - c.x, c.x - hi.x), 0.f), ey = fmaxf(fmaxf(lo.y - c.y, c.y - hi.y), 0.f),
                    ez = fmaxf(fmaxf(lo.z - c.z, c.z - hi.z), 0.f);
        d2 = fminf(d2, fmaf(ex, ex, fmaf(ey, ey, ez * ez)));
      }
      const bool is_leaf = v >= n - 1;
      const bool accept = v >= 0 && !is_leaf && __int_as_float(r.x) < theta2 * d2;
      const bool leafy = v >= 0 && !accept && (is_leaf || r.z <= nleaf);
      const bool open = v >= 0 && !accept && !leafy;
      const unsigned long long m_acc = __ballot(accept), m_leaf = __ballot(leafy), m_open = __ballot(open);
      const unsigned long long below = (1ull << lane) - 1ull;
      const int n_acc = __popcll(m_acc), n_leaf = __popcll(m_leaf), n_open = __popcll(m_open);
      if (accept) mono[__popcll(m_acc & below)] = c;
      // inclusive prefix of the leaf counts, in lane order
      int cnt = leafy ? r.z : 0, pre = cnt;
#pragma unroll
      for (int w = 1; w < 64; w <<= 1) { const int t = __shfl_up(pre, w); if (lane >= w) pre += t; }
      // compact: entry e (e-th leafy lane) -> first particle and exclusive prefix
      if (leafy) { const int p = __popcll(m_leaf & below); lf_first[p] = r.y; lf_pref[p] = pre - cnt; }
      const int total_leaf = __shfl(pre, 63);
      if (lane == 0) lf_pref[n_leaf] = total_leaf;
      if (open) {
        const int p = __popcll(m_open & below);
        const int left = r.w;
        const int gamma = left >= n - 1 ? left - (n - 1) : left;
        const int right = (r.y + r.z - 1 == gamma + 1) ? (n - 1 + gamma + 1) : (gamma + 1);
        if (top + 2 * n_open <= LSTACK) { stack[top + 2 * p] = right; stack[top + 2 * p + 1] = left; }
      }
      if (n_open > 0) {
        if (top + 2 * n_open <= LSTACK) top += 2 * n_open;
        else if (lane == 0) flags[0] = 1;                       // overflow: results invalid, reported by the host
      }
      NB_WAVE_SYNC();
      // ---- monopoles of this pass
#pragma unroll NB_BH_USWEEP
      for (int e = 0; e < n_acc; ++e) {
        const float4 q = mono[e];                                                    // LDS broadcast (ds_read_b128)
        const float qm = q.w;
        const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
        const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));
        float rinv = __builtin_amdgcn_rsqf(r2);
        if (GUARD) rinv = r2 > 0.f ? rinv : 0.f;
        const float f = qm * rinv * rinv * rinv;
        ax = fmaf(f, dx, ax); ay = fmaf(f, dy, ay); az = fmaf(f, dz, az);
      }
      // ---- particles of the leaf ranges, 64 at a time through an LDS tile
      for (int base = 0; base < total_leaf; base += 64) {
        const int t = base + lane;
        if (t < total_leaf) {
          int a = 0, b = n_leaf;                                  // last entry with lf_pref[e] <= t
          while (b - a > 1) { const int mid = (a + b) >> 1; if (lf_pref[mid] <= t) a = mid; else b = mid; }
          tile[lane] = sp[lf_first[a] + (t - lf_pref[a])];
        }
        NB_WAVE_SYNC();
        const int m = total_leaf - base < 64 ? total_leaf - base : 64;
#pragma unroll NB_BH_USWEEP
        for (int e = 0; e < m; ++e) {
          const float4 q = tile[e];
          const float qm = q.w;
          const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
          const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));
          float rinv = __builtin_amdgcn_rsqf(r2);
          if (GUARD) rinv = r2 > 0.f ? rinv : 0.f;
          const float f = qm * rinv * rinv * rinv;
          ax = fmaf(f, dx, ax); ay = fmaf(f, dy, ay); az = fmaf(f, dz, az);
        }
        NB_WAVE_SYNC();                        // the tile is rewritten by the next chunk
      }
      NB_WAVE_SYNC();                          // the lists are rewritten by the next pass
    }
    if (live) { const int j = idx[i]; out_x[j] = G * ax; out_y[j] = G * ay; out_z[j] = G * az; }   // straight to the caller's particle order
    NB_WAVE_SYNC();                            // the acceptance volumes are rewritten for the next group
  }
#undef NB_WAVE_SYNC
}

inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

struct nb_bh_state { BHBuf b; };

static void bh_free(BHBuf& b) {
  void* p[] = {b.bbox, b.keys, b.keys_sorted, b.idx, b.idx_sorted, b.sp, b.pre, b.child, b.range, b.com, b.l2, b.pk, b.tmp,
               b.rec, b.gstack, b.flags};
  for (void* q : p) if (q) (void)hipFree(q);
  b = BHBuf();
}

// after a stream sync: did a list-walk stack overflow (results invalid)?
int nb_bh_check(nb_ctx* c) {
  if (!c->bh || !c->bh->b.flags) return NB_OK;
  int f = 0;
  NB_HIP(c, hipMemcpy(&f, c->bh->b.flags, sizeof(int), hipMemcpyDeviceToHost));
  if (f) { (void)hipMemset(c->bh->b.flags, 0, sizeof(int)); return nb_fail(c, NB_ERR_STATE, "tree walk: stack overflow (results invalid); set NB_BH_WALK=0"); }
  return NB_OK;
}

void nb_bh_release(nb_ctx* c) {
  if (c->bh) { bh_free(c->bh->b); delete c->bh; c->bh = nullptr; }
}

int nb_launch_accel_tree_f32(nb_ctx* c, const float* x, const float* y, const float* z, const float* m, int64_t n,
                             float eps2, float G, float theta, float* ax, float* ay, float* az) {
  if (n <= 0) return NB_OK;
  if (n >= (1ll << 29)) return nb_fail(c, NB_ERR_ARG, "tree solver: n must be < 2^29");
  if (n < 2 * 64 || !(theta > 0.f)) {
    // tiny systems (or theta = 0: open everything) are exactly the direct sum
    return nb_launch_accel_f32(c, x, y, z, n, x, y, z, m, n, eps2, G, ax, ay, az, c->uniform_mass ? NB_UNIFORM_MASS : 0);
  }
  if (!c->bh) c->bh = new nb_bh_state();
  BHBuf& b = c->bh->b;
  hipStream_t st = c->stream;
  const int nn = (int)n;
  if (b.cap < n) {
    NB_HIP(c, hipStreamSynchronize(st));
    bh_free(b);
    const size_t N = (size_t)n, T = 2 * N;
    NB_HIP(c, hipMalloc((void**)&b.bbox, (8 + 6 * 1024) * sizeof(float)));
    NB_HIP(c, hipMalloc((void**)&b.keys, N * 8)); NB_HIP(c, hipMalloc((void**)&b.keys_sorted, N * 8));
    NB_HIP(c, hipMalloc((void**)&b.idx, N * 4)); NB_HIP(c, hipMalloc((void**)&b.idx_sorted, N * 4));
    NB_HIP(c, hipMalloc((void**)&b.sp, N * sizeof(float4)));
    NB_HIP(c, hipMalloc((void**)&b.pre, 4 * N * sizeof(double)));
    NB_HIP(c, hipMalloc((void**)&b.child, N * sizeof(int2)));
    NB_HIP(c, hipMalloc((void**)&b.range, T * sizeof(int2)));
    NB_HIP(c, hipMalloc((void**)&b.com, T * sizeof(float4)));
    NB_HIP(c, hipMalloc((void**)&b.l2, T * sizeof(float)));
    NB_HIP(c, hipMalloc((void**)&b.pk, N * sizeof(float4)));
    NB_HIP(c, hipMalloc((void**)&b.rec, T * sizeof(int4)));
    {
      const int groups = (int)((N + 63) / 64), blocks = (groups + 3) / 4 < 2048 ? (groups + 3) / 4 : 2048;
      NB_HIP(c, hipMalloc((void**)&b.gstack, (size_t)blocks * 4 * LSTACK * sizeof(int)));
      b.stack_blocks = blocks;
    }
    NB_HIP(c, hipMalloc((void**)&b.flags, 4 * sizeof(int)));
    NB_HIP(c, hipMemsetAsync(b.flags, 0, 4 * sizeof(int), st));
    size_t t1 = 0, t2 = 0;
    NB_HIP(c, rocprim::radix_sort_pairs((void*)nullptr, t1, b.keys, b.keys_sorted, b.idx, b.idx_sorted, N, 0, 63, st));
    NB_HIP(c, rocprim::inclusive_scan((void*)nullptr, t2, b.pre, b.pre, N, rocprim::plus<double>(), st));
    b.tmp_bytes = (t1 > t2 ? t1 : t2) + 256;
    NB_HIP(c, hipMalloc(&b.tmp, b.tmp_bytes));
    b.cap = n;
  }
  float* part = b.bbox + 8;
  const int gb = grid_for(n) > 1024 ? 1024 : grid_for(n);
  bbox_partial_kernel<<<gb, 256, 0, st>>>(x, y, z, nn, part);
  bbox_final_kernel<<<1, 64, 0, st>>>(part, gb, b.bbox);
  morton_kernel<<<grid_for(n), 256, 0, st>>>(x, y, z, m, nn, b.bbox, b.keys, b.idx, b.pk);
  NB_HIP(c, hipGetLastError());
  size_t tb = b.tmp_bytes;
  NB_HIP(c, rocprim::radix_sort_pairs(b.tmp, tb, b.keys, b.keys_sorted, b.idx, b.idx_sorted, (size_t)n, 0, 63, st));
  gather_kernel<<<grid_for(n), 256, 0, st>>>(b.pk, b.idx_sorted, nn, b.sp, b.pre);
  NB_HIP(c, hipGetLastError());
  for (int k = 0; k < 4; ++k) {
    tb = b.tmp_bytes;
    NB_HIP(c, rocprim::inclusive_scan(b.tmp, tb, b.pre + (size_t)k * n, b.pre + (size_t)k * n, (size_t)n, rocprim::plus<double>(), st));
  }
  build_kernel<<<grid_for(n), 256, 0, st>>>(b.keys_sorted, nn, b.child, b.range, b.l2, b.bbox);
  node_kernel<<<grid_for(2 * n), 256, 0, st>>>(b.pre, b.sp, nn, b.range, b.com, b.l2, b.child, b.rec);
  NB_HIP(c, hipGetLastError());
  const int groups = (nn + 63) / 64;
  const int blocks = (groups + 3) / 4;
  static const int walk_mode = getenv("NB_BH_WALK") ? atoi(getenv("NB_BH_WALK")) : 1;
  if (walk_mode == 0) {
    if (eps2 > 0.f) walk_kernel<false><<<blocks, 256, 0, st>>>(b.sp, nn, b.child, b.range, b.com, b.l2, theta * theta, eps2, b.idx_sorted, G, ax, ay, az, b.flags);
    else walk_kernel<true><<<blocks, 256, 0, st>>>(b.sp, nn, b.child, b.range, b.com, b.l2, theta * theta, eps2, b.idx_sorted, G, ax, ay, az, b.flags);
  } else {
    static const int maxb = getenv("NB_BH_BLOCKS") ? atoi(getenv("NB_BH_BLOCKS")) : 1792;   // 7 blocks per CU (VGPR-limited residency); <= 2048 (stack allocation)
    int pblocks = blocks < maxb ? blocks : maxb;            // persistent waves; groups are handed out dynamically
    if (pblocks > b.stack_blocks) pblocks = b.stack_blocks; // never more waves than wave-private stacks were allocated
    if (pblocks < 1) pblocks = 1;
    static const int nleaf = getenv("NB_BH_NLEAF") ? atoi(getenv("NB_BH_NLEAF")) : NLEAF_LIST;
    NB_HIP(c, hipMemsetAsync(b.flags + 1, 0, sizeof(int), st));                 // dynamic group counter
    if (eps2 > 0.f) walk_list_kernel<false><<<pblocks, 256, 0, st>>>(b.sp, nn, b.com, b.rec, theta * theta, eps2, b.idx_sorted, G, ax, ay, az, b.gstack, groups, b.flags, nleaf);
    else walk_list_kernel<true><<<pblocks, 256, 0, st>>>(b.sp, nn, b.com, b.rec, theta * theta, eps2, b.idx_sorted, G, ax, ay, az, b.gstack, groups, b.flags, nleaf);
  }
  NB_HIP(c, hipGetLastError());
  return NB_OK;
}

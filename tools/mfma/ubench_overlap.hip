// Which pipes of a gfx950 SIMD run concurrently? Two waves per SIMD (512-thread blocks, one block per CU), the first four
// waves of a block run role A, the last four role B; time A alone, B alone and both together. together ~ max(A, B): separate
// pipes; together ~ A + B: one shared resource.
//   roles: 0 idle, 1 v_fma_f32, 2 v_pk_fma_f32, 3 v_rsq_f32, 4 mfma_f32_32x32x2_f32, 5 mfma_f32_32x32x16_bf16, 6 v_fma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ROLE>
__device__ __forceinline__ void work(int iters, float* sink) {
  const float seed = (float)threadIdx.x * 1e-3f + 1.0f;
  if (ROLE == 1) {
    float a[8]; for (int k = 0; k < 8; ++k) a[k] = seed + k;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(seed), "v"(a[(k + 1) & 7]));
    float s = 0; for (int k = 0; k < 8; ++k) s += a[k]; if (s == 123.f) *sink = s;
  } else if (ROLE == 2) {
    v2f a[8]; for (int k = 0; k < 8; ++k) a[k] = v2f{seed + k, seed - k};
    v2f b = v2f{seed, seed};
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(a[(k + 1) & 7]));
    float s = 0; for (int k = 0; k < 8; ++k) s += a[k].x + a[k].y; if (s == 123.f) *sink = s;
  } else if (ROLE == 3) {
    float a[8]; for (int k = 0; k < 8; ++k) a[k] = seed + k;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
    float s = 0; for (int k = 0; k < 8; ++k) s += a[k]; if (s == 123.f) *sink = s;
  } else if (ROLE == 4) {
    f32x16 c[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed, c[k], 0, 0, 0);
    float s = 0; for (int k = 0; k < 4; ++k) s += c[k][0]; if (s == 123.f) *sink = s;
  } else if (ROLE == 5) {
    f32x16 c[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
    bf16x8 a; for (int k = 0; k < 8; ++k) a[k] = (__bf16)seed;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c[k], 0, 0, 0);
    float s = 0; for (int k = 0; k < 4; ++k) s += c[k][0]; if (s == 123.f) *sink = s;
  } else if (ROLE == 6) {
    double a[8]; for (int k = 0; k < 8; ++k) a[k] = seed + k;
    const double b = seed;
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(a[(k + 1) & 7]));
    double s = 0; for (int k = 0; k < 8; ++k) s += a[k]; if (s == 123.0) *sink = (float)s;
  }
}

// roles are dealt out per SIMD in arrival order (HW_ID.simd_id), so that every SIMD really hosts one A and one B wave
template <int RA, int RB>
__global__ void __launch_bounds__(512) bench(int ia, int ib, float* sink, int* census) {
  __shared__ int cnt[4];
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
  __syncthreads();
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const int simd = (hwid >> 4) & 3;
  int slot = 0;
  if ((threadIdx.x & 63) == 0) slot = atomicAdd(&cnt[simd], 1);
  slot = __builtin_amdgcn_readfirstlane(slot);
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) census[threadIdx.x >> 6] = simd * 16 + slot;
  __syncthreads();
  if ((slot & 1) == 0) work<RA>(ia, sink); else work<RB>(ib, sink);
}

template <int RA, int RB>
float run(int ia, int ib, float* d) {
  static int* census = nullptr; if (!census) CK(hipMalloc(&census, 64));
#define bench_args ia, ib, d, census
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  bench<RA, RB><<<256, 512>>>(bench_args); CK(hipDeviceSynchronize());
  { static bool shown = false; if (!shown) { int h[8]; CK(hipMemcpy(h, census, 32, hipMemcpyDeviceToHost)); printf("block 0 waves -> (simd,slot):"); for (int w = 0; w < 8; ++w) printf(" %d:(%d,%d)", w, h[w] / 16, h[w] % 16); printf("\n"); shown = true; } }
  CK(hipEventRecord(e0)); bench<RA, RB><<<256, 512>>>(bench_args); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

#define TRIPLE(A, B, NA, NB, IA, IB) do { float a = run<A, 0>(IA, 0, d), b = run<0, B>(0, IB, d), ab = run<A, B>(IA, IB, d); \
  printf("%-22s alone %.3f ms | %-22s alone %.3f ms | together %.3f ms  (max %.3f, sum %.3f)\n", NA, a, NB, b, ab, a > b ? a : b, a + b); } while (0)

int main() {
  float* d; CK(hipMalloc(&d, 4));
  const int I = 20000;
  // per iteration: VALU roles issue 64 instructions; MFMA roles 16. Iteration counts chosen so that each side alone takes similar time.
  TRIPLE(1, 1, "v_fma_f32", "v_fma_f32", I, I);
  TRIPLE(4, 4, "mfma_f32_32x32x2", "mfma_f32_32x32x2", I / 4, I / 4);
  TRIPLE(4, 1, "mfma_f32_32x32x2", "v_fma_f32", I / 4, 2 * I);
  TRIPLE(4, 2, "mfma_f32_32x32x2", "v_pk_fma_f32", I / 4, I);
  TRIPLE(4, 3, "mfma_f32_32x32x2", "v_rsq_f32", I / 4, I / 2);
  TRIPLE(5, 1, "mfma_f32_32x32x16_bf16", "v_fma_f32", I / 2, 2 * I);
  TRIPLE(5, 3, "mfma_f32_32x32x16_bf16", "v_rsq_f32", I / 2, I / 2);
  TRIPLE(1, 3, "v_fma_f32", "v_rsq_f32", 2 * I, I / 2);
  TRIPLE(6, 1, "v_fma_f64", "v_fma_f32", I, 2 * I);
  TRIPLE(4, 6, "mfma_f32_32x32x2", "v_fma_f64", I / 4, I);
  TRIPLE(2, 2, "v_pk_fma_f32", "v_pk_fma_f32", I, I);
  TRIPLE(3, 3, "v_rsq_f32", "v_rsq_f32", I / 2, I / 2);
  TRIPLE(2, 3, "v_pk_fma_f32", "v_rsq_f32", I, I / 2);
  TRIPLE(5, 2, "mfma_f32_32x32x16_bf16", "v_pk_fma_f32", I / 2, I);
  return 0;
}

// DPP cross-lane microbenchmark for gfx950 (round 3): what a wave-wide rotation by one lane costs next to the packed math of the
// pair-symmetric kernel — the question behind "source coordinates and source-side sums rotated through DPP instead of LDS".
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_dpp.hip -o tools/ubench_dpp ; run: tools/ubench_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int TEST>
__global__ void __launch_bounds__(256) ub(float* out, int iters, long long* cyc) {
  float a[16];
  v2f p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = 1.0f + 1e-3f * (threadIdx.x + i); p[i] = v2f{a[i], a[i] + 0.5f}; }
  v2f pb = {0.999f, 0.998f}, pc = {1e-6f, 2e-6f};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (TEST == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
    } else if constexpr (TEST == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(a[i]));
    } else if constexpr (TEST == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32_dpp %0, %1, %0 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
    } else if constexpr (TEST == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
    } else if constexpr (TEST == 4 || TEST == 5 || TEST == 6) {
      // one source step of the 12-sink kernel: 84 packed ops + 12 rsq; TEST 5 adds 9 wave rotations, TEST 6 adds 9 row rotations
#pragma unroll
      for (int g = 0; g < 6; ++g) {
#pragma unroll
        for (int i = 0; i < 14; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(g + i) & 15]) : "v"(pb), "v"(pc));
        asm volatile("v_rsq_f32 %0, %0" : "+v"(a[g]));
        asm volatile("v_rsq_f32 %0, %0" : "+v"(a[g + 6]));
      }
      if constexpr (TEST == 5) {
#pragma unroll
        for (int i = 0; i < 9; ++i) asm volatile("v_add_f32_dpp %0, %1, %0 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[12 + (i & 3)]));
      }
      if constexpr (TEST == 6) {
#pragma unroll
        for (int i = 0; i < 9; ++i) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[12 + (i & 3)]));
      }
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void semantics(int* out) {
  int v = threadIdx.x, w = -1, r = -1;
  asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(w) : "v"(v));
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
  out[threadIdx.x] = w; out[64 + threadIdx.x] = r;
}

template <int TEST> void run(const char* name, int instr, float* out, long long* cyc) {
  const int blocks = 256 * 2, iters = 20000;          // 2 blocks of 4 waves per CU = 2 waves per SIMD
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  ub<TEST><<<blocks, 256>>>(out, 10, cyc); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0)); ub<TEST><<<blocks, 256>>>(out, iters, cyc); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CHECK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
  // per SIMD: 2 waves x iters x instr wave-instructions in c shader cycles (clock64 counts at a fixed 100 MHz on gfx9: use the event time and 2.3 GHz as a guide)
  const double ns_per_instr = ms * 1e6 / (2.0 * iters * instr);
  printf("%-52s %8.3f ms  %7.3f ns per wave-instruction and SIMD (= %5.2f cycles at 2.3 GHz)\n", name, ms, ns_per_instr, ns_per_instr * 2.3);
}

int main() {
  float* out; long long* cyc; int* sem;
  CHECK(hipMalloc(&out, 512 * 256 * sizeof(float))); CHECK(hipMalloc(&cyc, 8)); CHECK(hipMalloc(&sem, 128 * sizeof(int)));
  semantics<<<1, 64>>>(sem); int h[128]; CHECK(hipMemcpy(h, sem, sizeof(h), hipMemcpyDeviceToHost));
  printf("wave_ror:1 lane 0 <- %d, lane 1 <- %d, lane 16 <- %d, lane 63 <- %d ; row_ror:1 lane 0 <- %d, lane 16 <- %d, lane 15 <- %d\n", h[0], h[1], h[16], h[63], h[64], h[80], h[79]);
  run<0>("v_pk_fma_f32 x16", 16, out, cyc);
  run<1>("v_mov_b32_dpp wave_ror:1 x16 (+ s_nop 1)", 16, out, cyc);
  run<2>("v_add_f32_dpp wave_ror:1 x16", 16, out, cyc);
  run<3>("v_add_f32_dpp row_ror:1 x16", 16, out, cyc);
  run<4>("84 pk_fma + 12 rsq (one 12-sink source step)", 96, out, cyc);
  run<5>("84 pk_fma + 12 rsq + 9 v_add_f32_dpp wave_ror:1", 105, out, cyc);
  run<6>("84 pk_fma + 12 rsq + 9 v_add_f32_dpp row_ror:1", 105, out, cyc);
  return 0;
}

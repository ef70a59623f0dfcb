// VALU issue-rate microbenchmark for gfx950 (MI355X).
//
// Purpose: pin the instruction-level ceiling of the pairwise-gravity inner loop
// (SURVEY.md §7.3 items 2-3, §8d "Instruction-level ceiling"): how many cycles a
// wave64 v_fma_f32 / v_pk_fma_f32 / v_rsq_f32 / v_fma_f64 costs per SIMD at 1..8
// waves per SIMD, and whether the transcendental unit overlaps with plain VALU.
//
// Build:  hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
// Run  :  tools/ubench_valu            (prints one line per (test, waves/SIMD))
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

enum Test { T_FMA32 = 0, T_PKFMA32, T_RSQ32, T_MIX12_1, T_FMA64, T_RSQ64, T_FMA32_SGPR,
            T_PKMUL32, T_PKADD32, T_MIX_PK, T_MIX12_1_INTERLEAVED, T_RCP32, T_SQRT32, T_MIX6PK_1, T_NTESTS };

static const char* kNames[] = {"v_fma_f32 x16", "v_pk_fma_f32 x16", "v_rsq_f32 x16", "12 fma + 1 rsq (blocked x4)",
  "v_fma_f64 x16", "v_rsq_f64 x16", "v_fma_f32 (sgpr src) x16", "v_pk_mul_f32 x16", "v_pk_add_f32 x16",
  "6 pk_fma + 1 rsq x4", "12 fma + 1 rsq (interleaved x4)", "v_rcp_f32 x16", "v_sqrt_f32 x16", "12pk(=24 fma) + 2 rsq x2"};
// number of wave-instructions per loop iteration for each test
static const int kInstr[] = {16, 16, 16, 52, 16, 16, 16, 16, 16, 28, 52, 16, 16, 28};

template <int TEST>
__global__ void __launch_bounds__(1024) ubench(float* out, int iters, float sarg, long long* cyc, long long* rt) {
  extern __shared__ float lds_pin[];  // sized by the host so that exactly 1 (or 2) blocks fit per CU
  if (iters < 0) lds_pin[threadIdx.x] = sarg;
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 1.0f + 1e-3f * (threadIdx.x + i);
  float b = 0.999f, c = 1e-6f;
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { p[i].x = a[i]; p[i].y = a[i] + 0.5f; }
  float2v pb = {0.999f, 0.998f}, pc = {1e-6f, 2e-6f};
  double d[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) d[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  double db = 0.999, dc = 1e-6;

  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (TEST == T_FMA32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    } else if constexpr (TEST == T_FMA32_SGPR) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sarg), "v"(c));
    } else if constexpr (TEST == T_PKFMA32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
    } else if constexpr (TEST == T_PKMUL32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
    } else if constexpr (TEST == T_PKADD32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
    } else if constexpr (TEST == T_RSQ32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
    } else if constexpr (TEST == T_RCP32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
    } else if constexpr (TEST == T_SQRT32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
    } else if constexpr (TEST == T_MIX12_1) {
      // 4 groups of (12 independent fma on a[0..11], then 1 rsq on a[12+g])
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        asm volatile("v_rsq_f32 %0, %0" : "+v"(a[12 + g]));
      }
    } else if constexpr (TEST == T_MIX12_1_INTERLEAVED) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        asm volatile("v_rsq_f32 %0, %0" : "+v"(a[12 + g]));
#pragma unroll
        for (int i = 6; i < 12; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      }
    } else if constexpr (TEST == T_MIX_PK) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
        asm volatile("v_rsq_f32 %0, %0" : "+v"(a[12 + g]));
      }
    } else if constexpr (TEST == T_MIX6PK_1) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
        asm volatile("v_rsq_f32 %0, %0" : "+v"(a[12 + 2 * g]));
        asm volatile("v_rsq_f32 %0, %0" : "+v"(a[13 + 2 * g]));
      }
    } else if constexpr (TEST == T_FMA64) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));
    } else if constexpr (TEST == T_RSQ64) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
    }
  }
  long long t1 = clock64();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    rt[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r1 - r0;
  }
}

typedef void (*kern_t)(float*, int, float, long long*, long long*);

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 100000;
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int ncu = prop.multiProcessorCount;
  printf("# device %s  CUs %d  clock %d kHz  wave %d\n", prop.gcnArchName, ncu, prop.clockRate, prop.warpSize);
  kern_t kerns[T_NTESTS] = {ubench<0>, ubench<1>, ubench<2>, ubench<3>, ubench<4>, ubench<5>, ubench<6>,
                            ubench<7>, ubench<8>, ubench<9>, ubench<10>, ubench<11>, ubench<12>, ubench<13>};
  const int maxwaves = ncu * 32;
  float* out; long long *cyc, *rt;
  CHECK(hipMalloc(&out, sizeof(float) * maxwaves * 64));
  CHECK(hipMalloc(&cyc, sizeof(long long) * maxwaves));
  CHECK(hipMalloc(&rt, sizeof(long long) * maxwaves));
  std::vector<long long> hc(maxwaves), hr(maxwaves);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int t = 0; t < T_NTESTS; ++t)
    CHECK(hipFuncSetAttribute((const void*)kerns[t], hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  printf("%-34s %6s %9s %11s %11s %11s %9s\n", "test", "w/SIMD", "ms", "cyc/instr", "cyc/i(max)", "ns/instr", "clk GHz");
  for (int t = 0; t < T_NTESTS; ++t) {
    for (int wps : {1, 2, 4, 8}) {
      // one block per CU (100 KiB LDS pins it) of 256*wps threads; 8 w/SIMD = two 1024-thread blocks (70 KiB each)
      int threads = wps >= 4 ? 1024 : 256 * wps;
      int blocks = wps == 8 ? 2 * ncu : ncu;
      size_t lds = wps == 8 ? 70 * 1024 : 100 * 1024;
      int nw = blocks * threads / 64;
      for (int rep = 0; rep < 2; ++rep) {  // rep 0 = warm-up of the same length
        CHECK(hipEventRecord(e0));
        kerns[t]<<<blocks, threads, lds>>>(out, iters, 0.999f, cyc, rt);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
      }
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(hc.data(), cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hr.data(), rt, sizeof(long long) * nw, hipMemcpyDeviceToHost));
      double avg = 0, mx = 0, ravg = 0;
      for (int i = 0; i < nw; ++i) { avg += (double)hc[i]; ravg += (double)hr[i]; if (hc[i] > mx) mx = (double)hc[i]; }
      avg /= nw; ravg /= nw;
      double instr_per_wave = (double)iters * kInstr[t];
      double clk = avg / (ravg * 10e-9) / 1e9;  // s_memrealtime ticks at 100 MHz
      printf("%-34s %6d %9.3f %11.3f %11.3f %11.4f %9.3f\n", kNames[t], wps, ms, avg / (instr_per_wave * wps),
             mx / (instr_per_wave * wps), ms * 1e6 / (instr_per_wave * wps), clk);
      fflush(stdout);
    }
  }
  return 0;
}

// VGPR-bank-conflict probe for gfx950: does the operand register assignment change the issue rate of
// v_fma_f32 / v_pk_fma_f32?  Explicit physical registers, one asm block per variant.
// Build+run: hipcc -O3 --offload-arch=gfx950 tools/ubench_banks.hip -o /tmp/ubench_banks && /tmp/ubench_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19", \
  "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \
  "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","scc"

#define INIT \
  "v_mov_b32 v0, 0x3f7fbe77\n v_mov_b32 v1, 0x3f7fbe77\n v_mov_b32 v4, 0x358637bd\n v_mov_b32 v5, 0x358637bd\n" \
  "v_mov_b32 v2, 1.0\n v_mov_b32 v3, 1.0\n v_mov_b32 v6, 1.0\n v_mov_b32 v7, 1.0\n v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n" \
  "v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n" \
  "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n" \
  "v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n" \
  "v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n" \
  "v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n v_mov_b32 v36, 1.0\n v_mov_b32 v37, 1.0\n v_mov_b32 v38, 1.0\n v_mov_b32 v39, 1.0\n" \
  "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 1.0\n v_mov_b32 v44, 1.0\n v_mov_b32 v45, 1.0\n" \
  "v_mov_b32 v46, 1.0\n v_mov_b32 v47, 1.0\n v_mov_b32 v48, 1.0\n v_mov_b32 v49, 1.0\n v_mov_b32 v50, 1.0\n v_mov_b32 v51, 1.0\n" \
  "v_mov_b32 v52, 1.0\n v_mov_b32 v53, 1.0\n"

// 12 pk_fma, D in banks (2,3), src0 = D, src1 = v[0:1], src2 = v[4:5]   (B and C both banks 0,1)
#define PK_A(D) "v_pk_fma_f32 v[" #D ":" #D "+1], v[" #D ":" #D "+1], v[0:1], v[4:5]\n"
// D in banks (0,1) as well: three-way same bank
#define PK_B(D) "v_pk_fma_f32 v[" #D ":" #D "+1], v[" #D ":" #D "+1], v[0:1], v[4:5]\n"
// src2 = D: only two distinct operands, in different banks
#define PK_C(D) "v_pk_fma_f32 v[" #D ":" #D "+1], v[" #D ":" #D "+1], v[0:1], v[" #D ":" #D "+1]\n"
#define F_A(D) "v_fma_f32 v" #D ", v" #D ", v0, v4\n"      // B,C both bank 0; D varies
#define F_C(D) "v_fma_f32 v" #D ", v" #D ", v0, v" #D "\n"
#define F_D(D) "v_fma_f32 v" #D ", v" #D ", v0, v5\n"      // B bank 0, C bank 1

#define LOOP(BODY) \
  asm volatile(INIT "s_mov_b32 s20, %0\n 1:\n" BODY BODY BODY BODY BODY BODY BODY BODY "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" :: "s"(iters) : CLOB, "s20");

template <int V> __global__ void __launch_bounds__(1024) k(int iters, long long* cyc, long long* rt) {
  extern __shared__ float pin[];
  if (iters < 0) pin[threadIdx.x] = 0.f;
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = clock64();
  if (V == 0) LOOP(PK_A(2) PK_A(6) PK_A(10) PK_A(14) PK_A(18) PK_A(22) PK_A(26) PK_A(30) PK_A(34) PK_A(38) PK_A(42) PK_A(46))
  if (V == 1) LOOP(PK_B(8) PK_B(12) PK_B(16) PK_B(20) PK_B(24) PK_B(28) PK_B(32) PK_B(36) PK_B(40) PK_B(44) PK_B(48) PK_B(52))
  if (V == 2) LOOP(PK_C(2) PK_C(6) PK_C(10) PK_C(14) PK_C(18) PK_C(22) PK_C(26) PK_C(30) PK_C(34) PK_C(38) PK_C(42) PK_C(46))
  if (V == 3) LOOP(F_A(1) F_A(2) F_A(3) F_A(6) F_A(7) F_A(9) F_A(10) F_A(11) F_A(13) F_A(14) F_A(15) F_A(17))       // D never bank 0
  if (V == 4) LOOP(F_A(8) F_A(12) F_A(16) F_A(20) F_A(24) F_A(28) F_A(32) F_A(36) F_A(40) F_A(44) F_A(48) F_A(52))   // all bank 0
  if (V == 5) LOOP(F_C(1) F_C(2) F_C(3) F_C(6) F_C(7) F_C(9) F_C(10) F_C(11) F_C(13) F_C(14) F_C(15) F_C(17))       // 2 operands, no conflict
  if (V == 6) LOOP(F_D(2) F_D(3) F_D(6) F_D(7) F_D(10) F_D(11) F_D(14) F_D(15) F_D(18) F_D(19) F_D(22) F_D(23))     // 3 operands, 3 banks
  long long t1 = clock64();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) { int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[w] = t1 - t0; rt[w] = r1 - r0; }
}

int main() {
  const char* names[] = {"pk_fma D(2,3) B(0,1) C(0,1)", "pk_fma all (0,1)", "pk_fma D(2,3) B(0,1) C=D", "fma D!=0 B0 C0", "fma all bank0",
                         "fma 2 operands no conflict", "fma 3 operands 3 banks"};
  typedef void (*kt)(int, long long*, long long*);
  kt ks[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
  int ncu = 256, iters = 25000;
  long long *cyc, *rt; CHECK(hipMalloc(&cyc, 8 * ncu * 32)); CHECK(hipMalloc(&rt, 8 * ncu * 32));
  std::vector<long long> hc(ncu * 32), hr(ncu * 32);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-32s %6s %9s %10s %9s\n", "variant", "w/SIMD", "ms", "cyc/instr", "clk GHz");
  for (int v = 0; v < 7; ++v) {
    CHECK(hipFuncSetAttribute((const void*)ks[v], hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int wps : {1, 2, 3, 4}) {
      int threads = 256 * wps;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        ks[v]<<<ncu, threads, 100 * 1024>>>(iters, cyc, rt);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
      }
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      int nw = ncu * threads / 64;
      CHECK(hipMemcpy(hc.data(), cyc, 8 * nw, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hr.data(), rt, 8 * nw, hipMemcpyDeviceToHost));
      double a = 0, r = 0; for (int i = 0; i < nw; ++i) { a += hc[i]; r += hr[i]; } a /= nw; r /= nw;
      double clk = a / (r * 10e-9) / 1e9;
      // cycles each SIMD spends per wave-instruction = wall ns x measured clock
      printf("%-32s %6d %9.3f %10.3f %9.3f\n", names[v], wps, ms, ms * 1e6 / ((double)iters * 96 * wps) * clk, clk);
    }
  }
  return 0;
}

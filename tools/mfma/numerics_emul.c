// CPU emulation of the numerics of an MFMA-assisted fp32 force kernel (experiment of VERDICT r1 item 8), to decide BEFORE
// writing the kernel whether r^2 = |xi'|^2 + eps^2 + |xj'|^2 - 2 xi'.xj' (an fmaf chain in k order, which is bit-for-bit what
// v_mfma_f32_32x32x2_f32 computes) and a = sum_j s_ij xj' - xi' sum_j s_ij keep the 2e-5 parity gate when particles are
// Morton-sorted and every 32-sink tile works in coordinates relative to its own centre.
//   gcc -O2 -fopenmp -o numerics_emul numerics_emul.c -lm ;  ./numerics_emul N eps ntiles_sampled [tile]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rng_s = 88172645463325252ull;
static double urand(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (rng_s >> 11) * (1.0 / 9007199254740992.0); }

static uint64_t spread21(uint32_t v) {
  uint64_t x = v & 0x1fffff;
  x = (x | x << 32) & 0x1f00000000ffffull; x = (x | x << 16) & 0x1f0000ff0000ffull; x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull; x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
typedef struct { uint64_t key; int idx; } kv;
static int cmpkv(const void* a, const void* b) { uint64_t x = ((const kv*)a)->key, y = ((const kv*)b)->key; return x < y ? -1 : x > y; }

int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 65536;
  float eps = argc > 2 ? (float)atof(argv[2]) : 0.01f;
  int nsample = argc > 3 ? atoi(argv[3]) : 256;
  int T = argc > 4 ? atoi(argv[4]) : 32;
  int sorted = argc > 5 ? atoi(argv[5]) : 1;
  float kfar = argc > 6 ? (float)atof(argv[6]) : 0.0f;   // source chunks (32) closer than kfar*rho to the tile centre take the exact dx path
  long long near_pairs = 0, all_pairs = 0;
  float* x = malloc(sizeof(float) * n), *y = malloc(sizeof(float) * n), *z = malloc(sizeof(float) * n);
  // Plummer sphere, Henon units (a = 3 pi / 16), positions only
  const double a = 3.0 * M_PI / 16.0;
  for (int i = 0; i < n; ++i) {
    double r;
    do { double u = urand(); r = a / sqrt(pow(u, -2.0 / 3.0) - 1.0); } while (r > 10 * a);
    double ct = 2 * urand() - 1, ph = 2 * M_PI * urand(), st = sqrt(1 - ct * ct);
    x[i] = (float)(r * st * cos(ph)); y[i] = (float)(r * st * sin(ph)); z[i] = (float)(r * ct);
  }
  float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
  for (int i = 0; i < n; ++i) { float p[3] = {x[i], y[i], z[i]}; for (int d = 0; d < 3; ++d) { if (p[d] < lo[d]) lo[d] = p[d]; if (p[d] > hi[d]) hi[d] = p[d]; } }
  float ext = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
  kv* k = malloc(sizeof(kv) * n);
  for (int i = 0; i < n; ++i) {
    uint32_t qx = (uint32_t)((x[i] - lo[0]) / ext * 2097151.0f), qy = (uint32_t)((y[i] - lo[1]) / ext * 2097151.0f), qz = (uint32_t)((z[i] - lo[2]) / ext * 2097151.0f);
    k[i].key = sorted ? (spread21(qx) | spread21(qy) << 1 | spread21(qz) << 2) : (uint64_t)i; k[i].idx = i;
  }
  qsort(k, n, sizeof(kv), cmpkv);
  float* sx = malloc(sizeof(float) * n), *sy = malloc(sizeof(float) * n), *sz = malloc(sizeof(float) * n);
  for (int i = 0; i < n; ++i) { sx[i] = x[k[i].idx]; sy[i] = y[k[i].idx]; sz[i] = z[k[i].idx]; }
  const float e2 = eps * eps;
  const int ntile = n / T;
  double worst = 0, worst_plain = 0, sum = 0; double worst_rho = 0; int worst_tile = -1;
  double* errs = malloc(sizeof(double) * nsample * T);
#pragma omp parallel for schedule(dynamic)
  for (int sidx = 0; sidx < nsample; ++sidx) {
    int t = (int)((long long)sidx * ntile / nsample);
    // tile centre: midpoint of the bounding box of its sinks (fp32)
    float bl[3] = {1e30f, 1e30f, 1e30f}, bh[3] = {-1e30f, -1e30f, -1e30f};
    for (int q = 0; q < T; ++q) { float p[3] = {sx[t * T + q], sy[t * T + q], sz[t * T + q]}; for (int d = 0; d < 3; ++d) { if (p[d] < bl[d]) bl[d] = p[d]; if (p[d] > bh[d]) bh[d] = p[d]; } }
    float c[3] = {0.5f * (bl[0] + bh[0]), 0.5f * (bl[1] + bh[1]), 0.5f * (bl[2] + bh[2])};
    double rho = 0.5 * sqrt((double)(bh[0] - bl[0]) * (bh[0] - bl[0]) + (double)(bh[1] - bl[1]) * (bh[1] - bl[1]) + (double)(bh[2] - bl[2]) * (bh[2] - bl[2]));
    for (int q = 0; q < T; ++q) {
      int i = t * T + q;
      float xi = sx[i] - c[0], yi = sy[i] - c[1], zi = sz[i] - c[2];
      float si2 = fmaf(xi, xi, fmaf(yi, yi, fmaf(zi, zi, e2)));
      double AX = 0, AY = 0, AZ = 0, AS = 0;          // fp64 running totals (flushed per 512 sources)
      double PX = 0, PY = 0, PZ = 0;                  // the plain kernel's numerics (dx form), for comparison
      double RX = 0, RY = 0, RZ = 0;                  // fp64 reference on the same fp32 inputs
      float fx = 0, fy = 0, fz = 0, fs = 0, gx = 0, gy = 0, gz = 0, hx = 0, hy = 0, hz = 0; double NX = 0, NY = 0, NZ = 0;
      int near = 0;
      for (int j = 0; j < n; ++j) {
        if ((j & 31) == 0) {   // classify the chunk of 32 sources by the distance of its bounding box from the tile centre
          float d2 = 0;
          float cl[3] = {1e30f, 1e30f, 1e30f}, ch[3] = {-1e30f, -1e30f, -1e30f};
          for (int u = j; u < j + 32 && u < n; ++u) { float p[3] = {sx[u], sy[u], sz[u]}; for (int d = 0; d < 3; ++d) { if (p[d] < cl[d]) cl[d] = p[d]; if (p[d] > ch[d]) ch[d] = p[d]; } }
          for (int d = 0; d < 3; ++d) { float g = c[d] < cl[d] ? cl[d] - c[d] : (c[d] > ch[d] ? c[d] - ch[d] : 0.f); d2 += g * g; }
          near = d2 < kfar * kfar * (float)(rho * rho);
          if (q == 0) {
#pragma omp atomic
            all_pairs += 1;
            if (near) {
#pragma omp atomic
              near_pairs += 1;
            }
          }
        }
        float xj = sx[j] - c[0], yj = sy[j] - c[1], zj = sz[j] - c[2];
        float sj2 = fmaf(xj, xj, fmaf(yj, yj, zj * zj));
        // MFMA chain, k order: [si2*1, 1*sj2, xi*(-2xj), yi*(-2yj), zi*(-2zj)]
        float r2 = fmaf(si2, 1.0f, 0.0f);
        r2 = fmaf(1.0f, sj2, r2);
        r2 = fmaf(xi, -2.0f * xj, r2); r2 = fmaf(yi, -2.0f * yj, r2); r2 = fmaf(zi, -2.0f * zj, r2);
        if (r2 < e2) r2 = e2;                         // rounding can push a coincident pair below eps^2 (or below 0)
        float ri = 1.0f / sqrtf(r2);
        float s = ri * ri * ri;
        if (near) {   // exact path in tile-relative coordinates: dx = xj' - xi'
          float ex = xj - xi, ey = yj - yi, ez = zj - zi;
          float t2 = fmaf(ex, ex, fmaf(ey, ey, fmaf(ez, ez, e2)));
          float ti = 1.0f / sqrtf(t2); s = ti * ti * ti;
          hx = fmaf(s, ex, hx); hy = fmaf(s, ey, hy); hz = fmaf(s, ez, hz);     // classic form: no cancellation against xi' sum s
        } else {
          fx = fmaf(s, xj, fx); fy = fmaf(s, yj, fy); fz = fmaf(s, zj, fz); fs += s;
        }
        // plain kernel
        float dx = sx[j] - sx[i], dy = sy[j] - sy[i], dz = sz[j] - sz[i];
        float q2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, e2)));
        float qi = 1.0f / sqrtf(q2); float qs = qi * qi * qi;
        gx = fmaf(qs, dx, gx); gy = fmaf(qs, dy, gy); gz = fmaf(qs, dz, gz);
        double ddx = (double)sx[j] - sx[i], ddy = (double)sy[j] - sy[i], ddz = (double)sz[j] - sz[i];
        double d2 = ddx * ddx + ddy * ddy + ddz * ddz + (double)eps * eps; double di = 1.0 / sqrt(d2); double ds = di * di * di;
        RX += ds * ddx; RY += ds * ddy; RZ += ds * ddz;
        if ((j & 511) == 511 || j == n - 1) {
          AX += fx; AY += fy; AZ += fz; AS += fs; fx = fy = fz = fs = 0; NX += hx; NY += hy; NZ += hz; hx = hy = hz = 0;
          PX += gx; PY += gy; PZ += gz; gx = gy = gz = 0;
        }
      }
      double ax = AX - (double)xi * AS + NX, ay = AY - (double)yi * AS + NY, az = AZ - (double)zi * AS + NZ;
      double nr = sqrt(RX * RX + RY * RY + RZ * RZ);
      double e = sqrt((ax - RX) * (ax - RX) + (ay - RY) * (ay - RY) + (az - RZ) * (az - RZ)) / nr;
      double ep = sqrt((PX - RX) * (PX - RX) + (PY - RY) * (PY - RY) + (PZ - RZ) * (PZ - RZ)) / nr;
      errs[sidx * T + q] = e;
      if (e > 5e-3) printf("tile %d q %d: e %.3e xi' (%g %g %g) rho %g | AX %.9g xi*AS %.9g ax %.9g RX %.9g | AS %.9g\n", t, q, e, xi, yi, zi, rho, AX, (double)xi * AS, ax, RX, AS);
#pragma omp critical
      { if (e > worst) { worst = e; worst_rho = rho; worst_tile = t; } if (ep > worst_plain) worst_plain = ep; sum += e; }
    }
  }
  // percentiles
  int m = nsample * T;
  for (int i = 1; i < m; ++i) { double v = errs[i]; int j = i - 1; while (j >= 0 && errs[j] > v) { errs[j + 1] = errs[j]; --j; } errs[j + 1] = v; }
  printf("kfar=%g near fraction of (tile,chunk) pairs %.4f\n", kfar, (double)near_pairs / (double)all_pairs);
  printf("N=%d eps=%g tile=%d sorted=%d sinks=%d: MFMA-form max %.3e (tile %d, rho %.3g) mean %.3e p50 %.3e p99 %.3e | plain-form max %.3e\n",
         n, eps, T, sorted, m, worst, worst_tile, worst_rho, sum / m, errs[m / 2], errs[(int)(m * 0.99)], worst_plain);
  return 0;
}

/* nbody_oracle.c — independent scalar-loop C restatement of the direct-summation path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from this file (oracle/libnbody_oracle.so).
 *
 * PARITY UNPINNED against the reference: the reference repository holds no force kernel or integrator
 * (its `pp` runs went through the external, un-vendored CONCEPT code spawned at
 * SnapVsPK/Ucertain.py:89-97). These functions restate SURVEY.md §7.2 and are cross-checked against
 * oracle/nbody_oracle.py (NumPy) and the analytic known-answer tests in tests/test_oracle_kat.py.
 *
 *   a_i = G * sum_j m_j (r_j - r_i) / (|r_j - r_i|^2 + eps^2)^{3/2}   (zero-denominator pairs skipped)
 *   U   = -G/2 * sum_i sum_{j != i} m_i m_j / sqrt(|r_j - r_i|^2 + eps^2)
 *
 * Build: make -C oracle        (gcc -O3 -fopenmp; no -ffast-math)
 */
#include <math.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Strictly sequential fp64, one sink after another, sources in index order: the checker. */
void nbo_accel_f64(int64_t n_snk, const double* xi, const double* yi, const double* zi,
                   int64_t n_src, const double* xs, const double* ys, const double* zs, const double* ms,
                   double eps, double G, double* ax, double* ay, double* az) {
  const double eps2 = eps * eps;
  for (int64_t i = 0; i < n_snk; ++i) {
    double sx = 0.0, sy = 0.0, sz = 0.0;
    const double px = xi[i], py = yi[i], pz = zi[i];
    for (int64_t j = 0; j < n_src; ++j) {
      const double dx = xs[j] - px, dy = ys[j] - py, dz = zs[j] - pz;
      const double r2 = dx * dx + dy * dy + dz * dz + eps2;
      if (r2 > 0.0) {
        const double w = ms[j] / (r2 * sqrt(r2));
        sx += w * dx; sy += w * dy; sz += w * dz;
      }
    }
    ax[i] = G * sx; ay[i] = G * sy; az[i] = G * sz;
  }
}

void nbo_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int nbo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Same arithmetic per pair, OpenMP over sinks and SIMD over sources (summation order differs):
 * the CPU baseline timed by bench.py. target_clones keeps the prebuilt .so runnable on hosts
 * without AVX-512. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define NBO_CLONES __attribute__((target_clones("avx512f", "avx2,fma", "default")))
#else
#define NBO_CLONES
#endif

NBO_CLONES
void nbo_accel_f64_omp(int64_t n_snk, const double* xi, const double* yi, const double* zi,
                       int64_t n_src, const double* xs, const double* ys, const double* zs, const double* ms,
                       double eps, double G, double* ax, double* ay, double* az) {
  const double eps2 = eps * eps;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_snk; ++i) {
    double sx = 0.0, sy = 0.0, sz = 0.0;
    const double px = xi[i], py = yi[i], pz = zi[i];
#pragma omp simd reduction(+ : sx, sy, sz)
    for (int64_t j = 0; j < n_src; ++j) {
      const double dx = xs[j] - px, dy = ys[j] - py, dz = zs[j] - pz;
      const double r2 = dx * dx + dy * dy + dz * dz + eps2;
      const double w = r2 > 0.0 ? ms[j] / (r2 * sqrt(r2)) : 0.0;
      sx += w * dx; sy += w * dy; sz += w * dz;
    }
    ax[i] = G * sx; ay[i] = G * sy; az[i] = G * sz;
  }
}

NBO_CLONES
void nbo_accel_f32_omp(int64_t n_snk, const float* xi, const float* yi, const float* zi,
                       int64_t n_src, const float* xs, const float* ys, const float* zs, const float* ms,
                       float eps, float G, float* ax, float* ay, float* az) {
  const float eps2 = eps * eps;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_snk; ++i) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    const float px = xi[i], py = yi[i], pz = zi[i];
#pragma omp simd reduction(+ : sx, sy, sz)
    for (int64_t j = 0; j < n_src; ++j) {
      const float dx = xs[j] - px, dy = ys[j] - py, dz = zs[j] - pz;
      const float r2 = dx * dx + dy * dy + dz * dz + eps2;
      const float w = r2 > 0.f ? ms[j] / (r2 * sqrtf(r2)) : 0.f;
      sx += w * dx; sy += w * dy; sz += w * dz;
    }
    ax[i] = G * sx; ay[i] = G * sy; az[i] = G * sz;
  }
}

/* sum_i m_i sum_{j != i} m_j / sqrt(r_ij^2 + eps^2), times -G/2; a pair with a zero denominator (coincident particles at eps = 0)
 * is skipped, as it is in the force sum */
double nbo_potential_f64(int64_t n, const double* x, const double* y, const double* z, const double* m,
                         double eps, double G) {
  const double eps2 = eps * eps;
  double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int64_t j = 0; j < n; ++j) {
      if (j == i) continue;
      const double dx = x[j] - x[i], dy = y[j] - y[i], dz = z[j] - z[i];
      const double r2 = dx * dx + dy * dy + dz * dz + eps2;
      if (r2 > 0.0) s += m[j] / sqrt(r2);
    }
    total += m[i] * s;
  }
  return -0.5 * G * total;
}

/* phi_i = sum_j m_j / sqrt(|r_j - r_i|^2 + eps^2) of n_snk sinks over n_src sources, EVERY source counted (a sink that is also a source
 * contributes its own m_i / eps — the convention of the fused force + potential kernels, which remove that term afterwards); pairs with
 * a zero denominator skipped. No G, no sign. */
void nbo_phi_f64_omp(int64_t n_snk, const double* xi, const double* yi, const double* zi,
                     int64_t n_src, const double* xs, const double* ys, const double* zs, const double* ms,
                     double eps, double* phi) {
  const double eps2 = eps * eps;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_snk; ++i) {
    double s = 0.0;
    for (int64_t j = 0; j < n_src; ++j) {
      const double dx = xs[j] - xi[i], dy = ys[j] - yi[i], dz = zs[j] - zi[i];
      const double r2 = dx * dx + dy * dy + dz * dz + eps2;
      if (r2 > 0.0) s += ms[j] / sqrt(r2);
    }
    phi[i] = s;
  }
}

double nbo_kinetic_f64(int64_t n, const double* vx, const double* vy, const double* vz, const double* m) {
  double k = 0.0;
  for (int64_t i = 0; i < n; ++i) k += m[i] * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
  return 0.5 * k;
}

/* One KDK step with the sequential force loop; `a*` hold a(x) on entry and on exit. */
void nbo_kdk_step_f64(int64_t n, double* x, double* y, double* z, double* vx, double* vy, double* vz,
                      const double* m, double* ax, double* ay, double* az, double dt, double eps, double G) {
  const double h = 0.5 * dt;
  for (int64_t i = 0; i < n; ++i) {
    vx[i] += ax[i] * h; vy[i] += ay[i] * h; vz[i] += az[i] * h;
    x[i] += vx[i] * dt; y[i] += vy[i] * dt; z[i] += vz[i] * dt;
  }
  nbo_accel_f64_omp(n, x, y, z, n, x, y, z, m, eps, G, ax, ay, az);
  for (int64_t i = 0; i < n; ++i) { vx[i] += ax[i] * h; vy[i] += ay[i] * h; vz[i] += az[i] * h; }
}

// Snapshot -> density-field chain on gfx950 (C-ABI in include/nbody_field.h).
//
// Kernels (all HBM / atomic bound, fp64 like the reference):
//   cic_kernel        one lane per particle: cell index and CIC weights computed with the reference's exact
//                     expressions (FraP_kTilDelta.py:100-121: int(p/cell) % N, (p % cell)/cell, (wx*wy)*wz), eight
//                     global_atomic_add_f64 per particle. 24 B read + 8 atomics per particle.
//   affine_kernel     grid = (grid/div1)/div2 - sub (compute_delta, FraP_kTilDelta.py:126-127)
//   r2c / window / c2r kernels around hipFFT Z2Z: delta_k /= W, W = ((w[i]*w[j])*w[k])^2 (FraP_kTilDelta.py:73-91)
//   power_bin_kernel  |delta_k|^2 * scale binned by |k| = sqrt((kx^2+ky^2)+kz^2) into host-supplied edges with
//                     np.histogram's semantics (Power_compare.py:176-198); integer counts are exact.
// Compiled with -ffp-contract=off: products and sums round separately, as NumPy's do.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include "../../include/nbody_amd.h"
#include "../../include/nbody_field.h"

struct nbf_plan {
  int device = 0, ngrid = 0;
  double boxsize = 0.0;
  size_t cells = 0;
  double* grid = nullptr;            // [ngrid^3] real
  hipfftDoubleComplex* cgrid = nullptr;   // [ngrid^3] complex
  double* w1d = nullptr;             // [ngrid]
  double* kvals = nullptr;           // [ngrid]
  double* edges = nullptr;           // [nedges]
  int nedges = 0;
  double* psum = nullptr;            // [nedges-1]
  unsigned long long* counts = nullptr;
  double* pos = nullptr;             // device staging for positions
  size_t pos_cap = 0;
  double* part = nullptr;            // reduction partials
  hipfftDoubleComplex* noise = nullptr;   // [ngrid^3] Hermitian-symmetrised R (GRF generator)
  double* amp = nullptr;             // [ngrid^3] sqrt(P)*fac
  double* stage2 = nullptr;          // [2*ngrid^3] upload staging for c1, c2
  double* zpos = nullptr;            // [ngrid^3 * 3] Zel'dovich positions
  bool have_noise = false, have_amp = false;
  hipfftHandle fft = 0;
  bool have_fft = false;
  hipStream_t stream = nullptr;
  std::string err;
};

static thread_local std::string g_field_err;

static int ffail(nbf_plan* p, int code, const std::string& m) {
  if (p) p->err = m; else g_field_err = m;
  return code;
}
#define F_HIP(p, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return ffail((p), NB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define F_FFT(p, call) do { hipfftResult r_ = (call); if (r_ != HIPFFT_SUCCESS) \
  return ffail((p), NB_ERR_HIP, std::string(#call) + ": hipfft error " + std::to_string((int)r_)); } while (0)

namespace {

// Python's float modulo: result takes the sign of the divisor
__device__ inline double py_mod(double x, double y) {
  double r = fmod(x, y);
  if (r != 0.0 && ((r < 0.0) != (y < 0.0))) r += y;
  return r;
}
// int(x) % N with Python semantics (truncate toward zero, non-negative remainder)
__device__ inline int py_cell(double x, double cell, int N) {
  long long i = (long long)(x / cell);
  long long r = i % N;
  if (r < 0) r += N;
  return (int)r;
}

__global__ void __launch_bounds__(256)
cic_kernel(const double* __restrict__ pos, long long n, double cell, int N, double* __restrict__ grid) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
    const double x = pos[3 * p], y = pos[3 * p + 1], z = pos[3 * p + 2];
    const int i = py_cell(x, cell, N), j = py_cell(y, cell, N), k = py_cell(z, cell, N);
    const double dx = py_mod(x, cell) / cell, dy = py_mod(y, cell) / cell, dz = py_mod(z, cell) / cell;
#pragma unroll
    for (int di = 0; di < 2; ++di) {
      const double wx = di ? dx : 1.0 - dx;
      const int ni = (i + di) % N;
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        const double wy = dj ? dy : 1.0 - dy;
        const int nj = (j + dj) % N;
#pragma unroll
        for (int dk = 0; dk < 2; ++dk) {
          const double wz = dk ? dz : 1.0 - dz;
          const int nk = (k + dk) % N;
          atomicAdd(&grid[((size_t)ni * N + nj) * N + nk], (wx * wy) * wz);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256)
sum_partial_kernel(const double* __restrict__ g, size_t n, double* __restrict__ part) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += g[i];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256)
sum_final_kernel(const double* __restrict__ part, int nparts, double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nparts; b += 256) s += part[b];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void __launch_bounds__(256)
affine_kernel(double* __restrict__ g, size_t n, double div1, double div2, double sub) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    g[i] = (g[i] / div1) / div2 - sub;
}

__global__ void __launch_bounds__(256)
r2c_kernel(const double* __restrict__ g, hipfftDoubleComplex* __restrict__ c, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    c[i].x = g[i]; c[i].y = 0.0;
  }
}

__global__ void __launch_bounds__(256)
window_kernel(hipfftDoubleComplex* __restrict__ c, const double* __restrict__ w1d, int N, double clip) {
  const size_t n = (size_t)N * N * N;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % N), j = (int)((idx / N) % N), i = (int)(idx / ((size_t)N * N));
    const double t = (w1d[i] * w1d[j]) * w1d[k];
    double W = t * t;
    if (clip > 0.0) W = W < clip ? clip : W;           // np.clip(W, clip, None): 1LPTvs2LPT/1vs2tlptplotter.py:89-91, Different_Sizes/Plotter.py:159-164
    else if (W == 0.0) W = 1.0;
    c[idx].x = c[idx].x / W;
    c[idx].y = c[idx].y / W;
  }
}

__global__ void __launch_bounds__(256)
c2r_kernel(const hipfftDoubleComplex* __restrict__ c, double* __restrict__ g, size_t n, double scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    g[i] = c[i].x * scale;
}

// Most modes fall into the few highest-|k| bins, so direct global atomics serialise on a handful of addresses
// (measured 42 ms for 256^3 cells). Each block therefore histograms into LDS (ds_add_f64 / ds_add_u64) and flushes
// its non-empty bins once.
__global__ void __launch_bounds__(256)
power_bin_kernel(const hipfftDoubleComplex* __restrict__ c, const double* __restrict__ kv, const double* __restrict__ edges,
                 int nedges, int N, double scale, double* __restrict__ psum, unsigned long long* __restrict__ counts) {
  extern __shared__ double lds_hist[];                       // [nb] sums, then [nb] counts
  const int nb = nedges - 1;
  double* hsum = lds_hist;
  unsigned long long* hcnt = reinterpret_cast<unsigned long long*>(lds_hist + nb);
  for (int b = threadIdx.x; b < nb; b += blockDim.x) { hsum[b] = 0.0; hcnt[b] = 0ull; }
  __syncthreads();
  const size_t n = (size_t)N * N * N;
  const double e0 = edges[0], e1 = edges[nedges - 1];
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % N), j = (int)((idx / N) % N), i = (int)(idx / ((size_t)N * N));
    const double kx = kv[i], ky = kv[j], kz = kv[k];
    const double km = sqrt((kx * kx + ky * ky) + kz * kz);
    if (!(km > 0.0) || km < e0 || km > e1) continue;
    // np.histogram: bin b holds edges[b] <= x < edges[b+1]; the last bin also holds x == edges[-1]
    int lo = 0, hi = nedges;                 // first edge > km
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (edges[mid] <= km) lo = mid + 1; else hi = mid; }
    int b = lo - 1;
    if (b >= nb) b = nb - 1;
    const double re = c[idx].x, im = c[idx].y;
    atomicAdd(&hsum[b], (re * re + im * im) * scale);
    atomicAdd(&hcnt[b], 1ull);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += blockDim.x) {
    if (hcnt[b]) { atomicAdd(&psum[b], hsum[b]); atomicAdd(&counts[b], hcnt[b]); }
  }
}

// R = c1 + i c2 made Hermitian exactly as the reference's lexicographic triple loop leaves it: the member of a mirror
// pair visited first keeps its value, the other becomes its conjugate; a self-conjugate point is conjugated once
// (variant 0) or untouched (variant 1); R[0,0,0] is real.
__global__ void __launch_bounds__(256)
hermitian_kernel(const double* __restrict__ c1, const double* __restrict__ c2, hipfftDoubleComplex* __restrict__ R,
                 int N, int variant) {
  const size_t n = (size_t)N * N * N;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % N), j = (int)((idx / N) % N), i = (int)(idx / ((size_t)N * N));
    const int mi = (N - i) % N, mj = (N - j) % N, mk = (N - k) % N;
    const size_t m = ((size_t)mi * N + mj) * N + mk;
    double re, im;
    if (idx > m) { re = c1[m]; im = -c2[m]; if (m == 0) im = -0.0; }        // conj(original mirror); mirror 0 is real
    else if (idx == m) { re = c1[idx]; im = variant == 0 ? -c2[idx] : c2[idx]; if (idx == 0) im = variant == 0 ? -0.0 : 0.0; }
    else { re = c1[idx]; im = c2[idx]; }
    R[idx].x = re; R[idx].y = im;
  }
}

// cgrid = (amp * mult) * R with mult = 1 (component 0) or i*k_c/|k|^2 (components 1..3), NumPy's evaluation order:
// ((1j*kc) / k_mag**2) * delta_k, then complex * complex with a zero real part.
__global__ void __launch_bounds__(256)
grf_product_kernel(const hipfftDoubleComplex* __restrict__ R, const double* __restrict__ amp, const double* __restrict__ kv,
                   hipfftDoubleComplex* __restrict__ out, int N, int comp) {
  const size_t n = (size_t)N * N * N;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const double rr = R[idx].x, ri = R[idx].y, a = amp[idx];
    if (comp == 0) { out[idx].x = a * rr; out[idx].y = a * ri; continue; }
    const int k = (int)(idx % N), j = (int)((idx / N) % N), i = (int)(idx / ((size_t)N * N));
    const double kx = kv[i], ky = kv[j], kz = kv[k];
    double km = sqrt((kx * kx + ky * ky) + kz * kz);
    if (idx == 0) km = 1e-10;
    const double kc = comp == 1 ? kx : (comp == 2 ? ky : kz);
    const double b = (kc / (km * km)) * a;          // psi_k = 0 + i*b
    out[idx].x = 0.0 * rr - b * ri;
    out[idx].y = 0.0 * ri + b * rr;
  }
}

__global__ void __launch_bounds__(256)
zeldovich_kernel(const double* __restrict__ psi, double* __restrict__ pos, int N, double L, int comp) {
  const size_t n = (size_t)N * N * N;
  const double step = L / N;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % N), j = (int)((idx / N) % N), i = (int)(idx / ((size_t)N * N));
    const int c = comp == 0 ? i : (comp == 1 ? j : k);
    const double x = (double)c * step + psi[idx];       // np.linspace(0, L, N, endpoint=False)[c] + displacement
    pos[3 * idx + comp] = py_mod(x, L);
  }
}

// np.fft.fftfreq index of grid index i; odd derivatives of a real field have no Nyquist component (nyq0: that index -> 0)
__device__ inline int fft_freq(int i, int N, bool nyq0) {
  if (nyq0 && N % 2 == 0 && i == N / 2) return 0;
  return i < (N + 1) / 2 ? i : i - N;
}

// out(k) = sign * i * k_axis * in(k)     (k_axis = 2 pi f / L, Nyquist component dropped)
__global__ void __launch_bounds__(256)
ik_multiply_kernel(const hipfftDoubleComplex* __restrict__ in, hipfftDoubleComplex* __restrict__ out, int N, double twopi_over_L,
                   int axis, double sign) {
  const size_t n = (size_t)N * N * N;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int c[3] = {(int)(idx / ((size_t)N * N)), (int)((idx / N) % N), (int)(idx % N)};
    const double k = sign * twopi_over_L * (double)fft_freq(c[axis], N, true);
    const hipfftDoubleComplex v = in[idx];
    out[idx] = hipfftDoubleComplex{-k * v.y, k * v.x};
  }
}

// phi2(k) = -src(k) / k^2 (all components of k, Nyquist included), phi2(0) = 0; in place
__global__ void __launch_bounds__(256)
inverse_laplace_kernel(hipfftDoubleComplex* __restrict__ c, int N, double twopi_over_L) {
  const size_t n = (size_t)N * N * N;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const double kx = twopi_over_L * fft_freq((int)(idx / ((size_t)N * N)), N, false), ky = twopi_over_L * fft_freq((int)((idx / N) % N), N, false),
                 kz = twopi_over_L * fft_freq((int)(idx % N), N, false);
    const double k2 = kx * kx + ky * ky + kz * kz;
    if (idx == 0) c[idx] = hipfftDoubleComplex{0.0, 0.0};
    else c[idx] = hipfftDoubleComplex{-c[idx].x / k2, -c[idx].y / k2};
  }
}

// mode 0: s = a*b + a*c + b*c;  mode 1: s -= a*a
__global__ void __launch_bounds__(256)
lpt2_source_kernel(double* __restrict__ s, const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c, size_t n,
                   int mode) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (mode == 0) s[i] = a[i] * b[i] + a[i] * c[i] + b[i] * c[i];
    else s[i] = s[i] - a[i] * a[i];
  }
}

// positions -> the same positions seen from a lattice displaced by `shift` along every axis: np.mod(pos - shift, L)
__global__ void __launch_bounds__(256)
shift_positions_kernel(const double* __restrict__ pos, long long n3, double shift, double L, double* __restrict__ out) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n3; q += (long long)gridDim.x * blockDim.x)
    out[q] = py_mod(pos[q] - shift, L);
}

// real grid -> real (part 0) or imaginary (part 1) part of the complex grid
__global__ void __launch_bounds__(256)
pack_part_kernel(const double* __restrict__ g, hipfftDoubleComplex* __restrict__ c, size_t n, int part) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (part == 0) c[i] = hipfftDoubleComplex{g[i], 0.0}; else c[i].y = g[i];
  }
}

// c = FFT(delta_1 + i delta_2): the transforms of the two real fields are separated with the Hermitian symmetry,
// d_1(k) = (c(k) + conj c(-k)) / 2, d_2(k) = (c(k) - conj c(-k)) / (2i), combined as CONCEPT's interlacing does,
// d = (d_1 + e^{-i h (kx+ky+kz)/2} d_2) / 2 (NN.param:30-41 'interlace': True), deconvolved with the CIC window
// W = ((w_i w_j) w_l)^2 and squared: out(k) = |d / W|^2 * scale. interlace = 0: d = d_1 (delta_2 = 0).
__global__ void __launch_bounds__(256)
power_modes_kernel(const hipfftDoubleComplex* __restrict__ c, const double* __restrict__ w1d, int N, int interlace, double scale,
                   double* __restrict__ out) {
  const size_t cells = (size_t)N * N * N;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < cells; idx += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(idx % N), j = (int)((idx / N) % N), i = (int)(idx / ((size_t)N * N));
    const size_t midx = ((size_t)((N - i) % N) * N + (size_t)((N - j) % N)) * N + (size_t)((N - l) % N);
    const hipfftDoubleComplex F = c[idx], Fm = c[midx];
    double dr = 0.5 * (F.x + Fm.x), di = 0.5 * (F.y - Fm.y);
    if (interlace) {
      const double d2r = 0.5 * (F.y + Fm.y), d2i = -0.5 * (F.x - Fm.x);
      const int fi = i < (N + 1) / 2 ? i : i - N, fj = j < (N + 1) / 2 ? j : j - N, fl = l < (N + 1) / 2 ? l : l - N;   // np.fft.fftfreq
      double sn, cs;
      sincos(-3.141592653589793 * (double)(fi + fj + fl) / (double)N, &sn, &cs);      // -h (kx+ky+kz) / 2
      dr = 0.5 * (dr + (cs * d2r - sn * d2i));
      di = 0.5 * (di + (cs * d2i + sn * d2r));
    }
    const double w = (w1d[i] * w1d[j]) * w1d[l], W = w * w;
    const double qr = dr / W, qi = di / W;
    out[idx] = (qr * qr + qi * qi) * scale;
  }
}

inline int grid_for(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int use_dev(nbf_plan* p) { F_HIP(p, hipSetDevice(p->device)); return NB_OK; }

int forward_fft(nbf_plan* p) {
  r2c_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->grid, p->cgrid, p->cells);
  F_HIP(p, hipGetLastError());
  F_FFT(p, hipfftExecZ2Z(p->fft, p->cgrid, p->cgrid, HIPFFT_FORWARD));
  return NB_OK;
}

}  // namespace

extern "C" {

const char* nbf_last_error(const nbf_plan* p) { return p ? p->err.c_str() : g_field_err.c_str(); }

int nbf_create(nbf_plan** out, int ngrid, double boxsize, int device) {
  if (!out) return ffail(nullptr, NB_ERR_ARG, "nbf_create: out is NULL");
  *out = nullptr;
  if (ngrid < 1 || ngrid > 2048) return ffail(nullptr, NB_ERR_ARG, "nbf_create: ngrid must be in [1, 2048]");
  if (!(boxsize > 0.0)) return ffail(nullptr, NB_ERR_ARG, "nbf_create: boxsize must be > 0");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return ffail(nullptr, NB_ERR_NODEVICE, std::string("nbf_create: no HIP device visible (") + hipGetErrorString(e) + ")");
  if (device < 0 || device >= ndev) return ffail(nullptr, NB_ERR_NODEVICE, "nbf_create: device index out of range");
  nbf_plan* p = new (std::nothrow) nbf_plan();
  if (!p) return ffail(nullptr, NB_ERR_ARG, "nbf_create: out of host memory");
  p->device = device; p->ngrid = ngrid; p->boxsize = boxsize; p->cells = (size_t)ngrid * ngrid * ngrid;
  auto bail = [&](const std::string& m) { std::string msg = m; nbf_destroy(p); return ffail(nullptr, NB_ERR_HIP, msg); };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(std::string("hipSetDevice: ") + hipGetErrorString(e));
  if ((e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate failed");
  if ((e = hipMalloc((void**)&p->grid, p->cells * sizeof(double))) != hipSuccess) return bail("hipMalloc(grid) failed");
  if ((e = hipMalloc((void**)&p->cgrid, p->cells * sizeof(hipfftDoubleComplex))) != hipSuccess) return bail("hipMalloc(cgrid) failed");
  if ((e = hipMalloc((void**)&p->w1d, ngrid * sizeof(double))) != hipSuccess) return bail("hipMalloc(w1d) failed");
  if ((e = hipMalloc((void**)&p->kvals, ngrid * sizeof(double))) != hipSuccess) return bail("hipMalloc(kvals) failed");
  if ((e = hipMalloc((void**)&p->part, 4096 * sizeof(double))) != hipSuccess) return bail("hipMalloc(part) failed");
  // on the plan's own (non-blocking) stream, not the NULL stream: later work on p->stream must be ordered after it
  if ((e = hipMemsetAsync(p->grid, 0, p->cells * sizeof(double), p->stream)) != hipSuccess) return bail("hipMemsetAsync failed");
  if ((e = hipStreamSynchronize(p->stream)) != hipSuccess) return bail("hipStreamSynchronize failed");
  hipfftResult r = hipfftPlan3d(&p->fft, ngrid, ngrid, ngrid, HIPFFT_Z2Z);
  if (r != HIPFFT_SUCCESS) return bail("hipfftPlan3d failed: " + std::to_string((int)r));
  p->have_fft = true;
  if ((r = hipfftSetStream(p->fft, p->stream)) != HIPFFT_SUCCESS) return bail("hipfftSetStream failed");
  *out = p;
  return NB_OK;
}

int nbf_destroy(nbf_plan* p) {
  if (!p) return NB_OK;
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  if (p->have_fft) (void)hipfftDestroy(p->fft);
  void* bufs[] = {p->grid, p->cgrid, p->w1d, p->kvals, p->edges, p->psum, p->counts, p->pos, p->part,
                  p->noise, p->amp, p->stage2, p->zpos};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return NB_OK;
}

int nbf_set_window(nbf_plan* p, const double* w1d) {
  if (!p || !w1d) return ffail(p, NB_ERR_ARG, "nbf_set_window: NULL argument");
  int rc = use_dev(p); if (rc) return rc;
  F_HIP(p, hipMemcpyAsync(p->w1d, w1d, p->ngrid * sizeof(double), hipMemcpyHostToDevice, p->stream));
  F_HIP(p, hipStreamSynchronize(p->stream));
  return NB_OK;
}

int nbf_set_kbins(nbf_plan* p, const double* kvals, const double* edges, int nedges) {
  if (!p || !kvals || !edges) return ffail(p, NB_ERR_ARG, "nbf_set_kbins: NULL argument");
  if (nedges < 2 || nedges > 4096) return ffail(p, NB_ERR_ARG, "nbf_set_kbins: need 2..4096 edges");
  int rc = use_dev(p); if (rc) return rc;
  F_HIP(p, hipStreamSynchronize(p->stream));
  if (p->edges) { F_HIP(p, hipFree(p->edges)); p->edges = nullptr; }
  if (p->psum) { F_HIP(p, hipFree(p->psum)); p->psum = nullptr; }
  if (p->counts) { F_HIP(p, hipFree(p->counts)); p->counts = nullptr; }
  F_HIP(p, hipMalloc((void**)&p->edges, nedges * sizeof(double)));
  F_HIP(p, hipMalloc((void**)&p->psum, (nedges - 1) * sizeof(double)));
  F_HIP(p, hipMalloc((void**)&p->counts, (nedges - 1) * sizeof(unsigned long long)));
  p->nedges = nedges;
  F_HIP(p, hipMemcpyAsync(p->kvals, kvals, p->ngrid * sizeof(double), hipMemcpyHostToDevice, p->stream));
  F_HIP(p, hipMemcpyAsync(p->edges, edges, nedges * sizeof(double), hipMemcpyHostToDevice, p->stream));
  F_HIP(p, hipStreamSynchronize(p->stream));
  return NB_OK;
}

int nbf_cic(nbf_plan* p, const double* pos, int64_t n) {
  if (!p) return NB_ERR_ARG;
  if (n < 0 || (n > 0 && !pos)) return ffail(p, NB_ERR_ARG, "nbf_cic: bad positions");
  int rc = use_dev(p); if (rc) return rc;
  const size_t need = (size_t)n * 3 * sizeof(double);
  if (need > p->pos_cap) {
    F_HIP(p, hipStreamSynchronize(p->stream));
    if (p->pos) { F_HIP(p, hipFree(p->pos)); p->pos = nullptr; p->pos_cap = 0; }
    F_HIP(p, hipMalloc((void**)&p->pos, need));
    p->pos_cap = need;
  }
  if (n > 0) F_HIP(p, hipMemcpyAsync(p->pos, pos, need, hipMemcpyHostToDevice, p->stream));
  F_HIP(p, hipMemsetAsync(p->grid, 0, p->cells * sizeof(double), p->stream));
  if (n > 0) {
    cic_kernel<<<grid_for((size_t)n), 256, 0, p->stream>>>(p->pos, (long long)n, p->boxsize / p->ngrid, p->ngrid, p->grid);
    F_HIP(p, hipGetLastError());
  }
  F_HIP(p, hipStreamSynchronize(p->stream));   // the caller may free `pos` after return
  return NB_OK;
}

int nbf_grid_mean(nbf_plan* p, double* mean) {
  if (!p || !mean) return ffail(p, NB_ERR_ARG, "nbf_grid_mean: NULL argument");
  int rc = use_dev(p); if (rc) return rc;
  const int g = grid_for(p->cells);
  sum_partial_kernel<<<g, 256, 0, p->stream>>>(p->grid, p->cells, p->part);
  F_HIP(p, hipGetLastError());
  sum_final_kernel<<<1, 256, 0, p->stream>>>(p->part, g, p->part + g);
  F_HIP(p, hipGetLastError());
  double s = 0.0;
  F_HIP(p, hipMemcpyAsync(&s, p->part + g, sizeof(double), hipMemcpyDeviceToHost, p->stream));
  F_HIP(p, hipStreamSynchronize(p->stream));
  *mean = s / (double)p->cells;
  return NB_OK;
}

int nbf_grid_affine(nbf_plan* p, double div1, double div2, double sub) {
  if (!p) return NB_ERR_ARG;
  int rc = use_dev(p); if (rc) return rc;
  affine_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->grid, p->cells, div1, div2, sub);
  F_HIP(p, hipGetLastError());
  return NB_OK;
}

// the grid rounded to float32 and back: the plotting scripts that deposit into a float32 array (Different_Sizes/Plotter.py:167)
__global__ void __launch_bounds__(256)
round_f32_kernel(double* __restrict__ g, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) g[i] = (double)(float)g[i];
}

int nbf_grid_round_f32(nbf_plan* p) {
  if (!p) return NB_ERR_ARG;
  int rc = use_dev(p); if (rc) return rc;
  round_f32_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->grid, p->cells);
  F_HIP(p, hipGetLastError());
  return NB_OK;
}

int nbf_deconvolve(nbf_plan* p) { return nbf_deconvolve_clip(p, 0.0); }

int nbf_deconvolve_clip(nbf_plan* p, double clip) {
  if (!p) return NB_ERR_ARG;
  if (!(clip >= 0.0)) return ffail(p, NB_ERR_ARG, "nbf_deconvolve_clip: clip must be >= 0");
  int rc = use_dev(p); if (rc) return rc;
  rc = forward_fft(p); if (rc) return rc;
  window_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->cgrid, p->w1d, p->ngrid, clip);
  F_HIP(p, hipGetLastError());
  F_FFT(p, hipfftExecZ2Z(p->fft, p->cgrid, p->cgrid, HIPFFT_BACKWARD));
  c2r_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->cgrid, p->grid, p->cells, 1.0 / (double)p->cells);
  F_HIP(p, hipGetLastError());
  return NB_OK;
}

int nbf_get_grid(nbf_plan* p, double* out) {
  if (!p || !out) return ffail(p, NB_ERR_ARG, "nbf_get_grid: NULL argument");
  int rc = use_dev(p); if (rc) return rc;
  F_HIP(p, hipMemcpyAsync(out, p->grid, p->cells * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  F_HIP(p, hipStreamSynchronize(p->stream));
  return NB_OK;
}

int nbf_set_grid(nbf_plan* p, const double* in) {
  if (!p || !in) return ffail(p, NB_ERR_ARG, "nbf_set_grid: NULL argument");
  int rc = use_dev(p); if (rc) return rc;
  F_HIP(p, hipMemcpyAsync(p->grid, in, p->cells * sizeof(double), hipMemcpyHostToDevice, p->stream));
  F_HIP(p, hipStreamSynchronize(p->stream));
  return NB_OK;
}

int nbf_power_bins(nbf_plan* p, double scale, double* psum, int64_t* counts) {
  if (!p || !psum || !counts) return ffail(p, NB_ERR_ARG, "nbf_power_bins: NULL argument");
  if (!p->edges) return ffail(p, NB_ERR_STATE, "nbf_power_bins: call nbf_set_kbins first");
  int rc = use_dev(p); if (rc) return rc;
  const int nb = p->nedges - 1;
  rc = forward_fft(p); if (rc) return rc;
  F_HIP(p, hipMemsetAsync(p->psum, 0, nb * sizeof(double), p->stream));
  F_HIP(p, hipMemsetAsync(p->counts, 0, nb * sizeof(unsigned long long), p->stream));
  {
    const size_t want = (p->cells + 255) / 256;
    const int g = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));   // few blocks: one global flush per block and bin
    power_bin_kernel<<<g, 256, (size_t)nb * 16, p->stream>>>(p->cgrid, p->kvals, p->edges, p->nedges, p->ngrid, scale,
                                                           p->psum, p->counts);
  }
  F_HIP(p, hipGetLastError());
  F_HIP(p, hipMemcpyAsync(psum, p->psum, nb * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  static_assert(sizeof(unsigned long long) == sizeof(int64_t), "count type");
  F_HIP(p, hipMemcpyAsync(counts, p->counts, nb * sizeof(int64_t), hipMemcpyDeviceToHost, p->stream));
  F_HIP(p, hipStreamSynchronize(p->stream));
  return NB_OK;
}

int nbf_power_modes(nbf_plan* p, const double* pos, int64_t n, int interlace, double* out) {
  if (!p || !out || n <= 0 || !pos) return ffail(p, NB_ERR_ARG, "nbf_power_modes: NULL argument or no particles");
  if (!p->w1d) return ffail(p, NB_ERR_STATE, "nbf_power_modes: call nbf_set_window first");
  const double h = p->boxsize / p->ngrid;
  // lattice 1 -> delta_1 -> real part
  int rc = nbf_cic(p, pos, n); if (rc) return rc;
  double mean = 0.0;
  rc = nbf_grid_mean(p, &mean); if (rc) return rc;
  rc = nbf_grid_affine(p, mean, 1.0, 1.0); if (rc) return rc;
  pack_part_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->grid, p->cgrid, p->cells, 0);
  F_HIP(p, hipGetLastError());
  if (interlace) {   // the lattice displaced by half a cell along every axis -> delta_2 -> imaginary part
    const size_t need = 2 * (size_t)n * 3 * sizeof(double);
    if (need > p->pos_cap) {          // second half of the staging buffer: the displaced positions
      F_HIP(p, hipStreamSynchronize(p->stream));
      double* q = nullptr;
      F_HIP(p, hipMalloc((void**)&q, need));
      F_HIP(p, hipMemcpyAsync(q, p->pos, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
      F_HIP(p, hipStreamSynchronize(p->stream));
      F_HIP(p, hipFree(p->pos));
      p->pos = q; p->pos_cap = need;
    }
    double* pos2 = p->pos + (size_t)n * 3;
    shift_positions_kernel<<<grid_for((size_t)n * 3), 256, 0, p->stream>>>(p->pos, (long long)n * 3, 0.5 * h, p->boxsize, pos2);
    F_HIP(p, hipGetLastError());
    F_HIP(p, hipMemsetAsync(p->grid, 0, p->cells * sizeof(double), p->stream));
    cic_kernel<<<grid_for((size_t)n), 256, 0, p->stream>>>(pos2, (long long)n, h, p->ngrid, p->grid);
    F_HIP(p, hipGetLastError());
    rc = nbf_grid_mean(p, &mean); if (rc) return rc;
    rc = nbf_grid_affine(p, mean, 1.0, 1.0); if (rc) return rc;
    pack_part_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->grid, p->cgrid, p->cells, 1);
    F_HIP(p, hipGetLastError());
  }
  F_FFT(p, hipfftExecZ2Z(p->fft, p->cgrid, p->cgrid, HIPFFT_FORWARD));
  const double N3 = (double)p->cells;
  power_modes_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->cgrid, p->w1d, p->ngrid, interlace ? 1 : 0,
                                                               p->boxsize * p->boxsize * p->boxsize / (N3 * N3), p->grid);
  F_HIP(p, hipGetLastError());
  F_HIP(p, hipMemcpyAsync(out, p->grid, p->cells * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  F_HIP(p, hipStreamSynchronize(p->stream));
  return NB_OK;
}

int nbf_lpt2_gradient(nbf_plan* p, const double* psi, double* out) {
  if (!p || !psi || !out) return ffail(p, NB_ERR_ARG, "nbf_lpt2_gradient: NULL argument");
  int rc = use_dev(p); if (rc) return rc;
  const size_t n = p->cells, rb = n * sizeof(double), cb = n * sizeof(hipfftDoubleComplex);
  const int N = p->ngrid;
  const double tpl = 2.0 * 3.141592653589793 / p->boxsize, inv = 1.0 / (double)n;
  hipfftDoubleComplex* pk[3] = {nullptr, nullptr, nullptr};
  double* diag[3] = {nullptr, nullptr, nullptr};
  double* src = nullptr;
  auto release = [&]() { for (auto& q : pk) if (q) (void)hipFree(q); for (auto& q : diag) if (q) (void)hipFree(q); if (src) (void)hipFree(src); };
#define L_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { release(); \
    return ffail(p, NB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)
#define L_FFT(call) do { hipfftResult r_ = (call); if (r_ != HIPFFT_SUCCESS) { release(); \
    return ffail(p, NB_ERR_HIP, std::string(#call) + ": hipfft error " + std::to_string((int)r_)); } } while (0)
  for (int i = 0; i < 3; ++i) { L_HIP(hipMalloc((void**)&pk[i], cb)); L_HIP(hipMalloc((void**)&diag[i], rb)); }
  L_HIP(hipMalloc((void**)&src, rb));
  const int g = grid_for(n);
  for (int i = 0; i < 3; ++i) {                     // psi_i(k)
    L_HIP(hipMemcpyAsync(p->grid, psi + (size_t)i * n, rb, hipMemcpyHostToDevice, p->stream));
    r2c_kernel<<<g, 256, 0, p->stream>>>(p->grid, pk[i], n);
    L_FFT(hipfftExecZ2Z(p->fft, pk[i], pk[i], HIPFFT_FORWARD));
  }
  // phi1,ij = -d_j psi_i = real(ifft(-(-i k_j) psi_i(k))) ... as the NumPy restatement: ifftn(-1j k_j psi_i).real * -1
  auto second = [&](int i, int j, double* dst) -> int {
    ik_multiply_kernel<<<g, 256, 0, p->stream>>>(pk[i], p->cgrid, N, tpl, j, -1.0);
    F_FFT(p, hipfftExecZ2Z(p->fft, p->cgrid, p->cgrid, HIPFFT_BACKWARD));
    c2r_kernel<<<g, 256, 0, p->stream>>>(p->cgrid, dst, n, -inv);
    return NB_OK;
  };
  for (int i = 0; i < 3; ++i) { rc = second(i, i, diag[i]); if (rc) { release(); return rc; } }
  lpt2_source_kernel<<<g, 256, 0, p->stream>>>(src, diag[0], diag[1], diag[2], n, 0);
  const int off[3][2] = {{0, 1}, {0, 2}, {1, 2}};
  for (int q = 0; q < 3; ++q) {
    rc = second(off[q][0], off[q][1], p->grid); if (rc) { release(); return rc; }
    lpt2_source_kernel<<<g, 256, 0, p->stream>>>(src, p->grid, nullptr, nullptr, n, 1);
  }
  L_HIP(hipGetLastError());
  r2c_kernel<<<g, 256, 0, p->stream>>>(src, pk[0], n);
  L_FFT(hipfftExecZ2Z(p->fft, pk[0], pk[0], HIPFFT_FORWARD));
  inverse_laplace_kernel<<<g, 256, 0, p->stream>>>(pk[0], N, tpl);
  for (int c = 0; c < 3; ++c) {                     // grad(phi2)_c = real(ifft(i k_c phi2(k)))
    ik_multiply_kernel<<<g, 256, 0, p->stream>>>(pk[0], p->cgrid, N, tpl, c, 1.0);
    L_FFT(hipfftExecZ2Z(p->fft, p->cgrid, p->cgrid, HIPFFT_BACKWARD));
    c2r_kernel<<<g, 256, 0, p->stream>>>(p->cgrid, p->grid, n, inv);
    L_HIP(hipGetLastError());
    L_HIP(hipMemcpyAsync(out + (size_t)c * n, p->grid, rb, hipMemcpyDeviceToHost, p->stream));
    L_HIP(hipStreamSynchronize(p->stream));
  }
#undef L_HIP
#undef L_FFT
  release();
  return NB_OK;
}

int nbf_noise_hermitian(nbf_plan* p, const double* c1, const double* c2, int variant) {
  if (!p || !c1 || !c2) return ffail(p, NB_ERR_ARG, "nbf_noise_hermitian: NULL argument");
  if (variant != 0 && variant != 1) return ffail(p, NB_ERR_ARG, "nbf_noise_hermitian: variant must be 0 or 1");
  int rc = use_dev(p); if (rc) return rc;
  const size_t b = p->cells * sizeof(double);
  if (!p->stage2) F_HIP(p, hipMalloc((void**)&p->stage2, 2 * b));
  if (!p->noise) F_HIP(p, hipMalloc((void**)&p->noise, p->cells * sizeof(hipfftDoubleComplex)));
  F_HIP(p, hipMemcpyAsync(p->stage2, c1, b, hipMemcpyHostToDevice, p->stream));
  F_HIP(p, hipMemcpyAsync(p->stage2 + p->cells, c2, b, hipMemcpyHostToDevice, p->stream));
  hermitian_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->stage2, p->stage2 + p->cells, p->noise, p->ngrid, variant);
  F_HIP(p, hipGetLastError());
  F_HIP(p, hipStreamSynchronize(p->stream));
  p->have_noise = true;
  return NB_OK;
}

int nbf_get_noise(nbf_plan* p, double* out) {
  if (!p || !out) return ffail(p, NB_ERR_ARG, "nbf_get_noise: NULL argument");
  if (!p->have_noise) return ffail(p, NB_ERR_STATE, "nbf_get_noise: call nbf_noise_hermitian first");
  int rc = use_dev(p); if (rc) return rc;
  F_HIP(p, hipMemcpyAsync(out, p->noise, p->cells * sizeof(hipfftDoubleComplex), hipMemcpyDeviceToHost, p->stream));
  F_HIP(p, hipStreamSynchronize(p->stream));
  return NB_OK;
}

int nbf_set_amplitude(nbf_plan* p, const double* amp, const double* kvals) {
  if (!p || !amp || !kvals) return ffail(p, NB_ERR_ARG, "nbf_set_amplitude: NULL argument");
  int rc = use_dev(p); if (rc) return rc;
  if (!p->amp) F_HIP(p, hipMalloc((void**)&p->amp, p->cells * sizeof(double)));
  F_HIP(p, hipMemcpyAsync(p->amp, amp, p->cells * sizeof(double), hipMemcpyHostToDevice, p->stream));
  F_HIP(p, hipMemcpyAsync(p->kvals, kvals, p->ngrid * sizeof(double), hipMemcpyHostToDevice, p->stream));
  F_HIP(p, hipStreamSynchronize(p->stream));
  p->have_amp = true;
  return NB_OK;
}

int nbf_grf_field(nbf_plan* p, int component) {
  if (!p) return NB_ERR_ARG;
  if (component < 0 || component > 3) return ffail(p, NB_ERR_ARG, "nbf_grf_field: component must be 0..3");
  if (!p->have_noise || !p->have_amp) return ffail(p, NB_ERR_STATE, "nbf_grf_field: set the noise and the amplitude first");
  int rc = use_dev(p); if (rc) return rc;
  grf_product_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->noise, p->amp, p->kvals, p->cgrid, p->ngrid, component);
  F_HIP(p, hipGetLastError());
  F_FFT(p, hipfftExecZ2Z(p->fft, p->cgrid, p->cgrid, HIPFFT_BACKWARD));
  c2r_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->cgrid, p->grid, p->cells, 1.0 / (double)p->cells);
  F_HIP(p, hipGetLastError());
  return NB_OK;
}

int nbf_zeldovich(nbf_plan* p, double* pos_out) {
  if (!p || !pos_out) return ffail(p, NB_ERR_ARG, "nbf_zeldovich: NULL argument");
  int rc = use_dev(p); if (rc) return rc;
  if (!p->zpos) F_HIP(p, hipMalloc((void**)&p->zpos, p->cells * 3 * sizeof(double)));
  for (int comp = 0; comp < 3; ++comp) {
    rc = nbf_grf_field(p, comp + 1); if (rc) return rc;
    zeldovich_kernel<<<grid_for(p->cells), 256, 0, p->stream>>>(p->grid, p->zpos, p->ngrid, p->boxsize, comp);
    F_HIP(p, hipGetLastError());
  }
  F_HIP(p, hipMemcpyAsync(pos_out, p->zpos, p->cells * 3 * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  F_HIP(p, hipStreamSynchronize(p->stream));
  return NB_OK;
}

}  // extern "C"

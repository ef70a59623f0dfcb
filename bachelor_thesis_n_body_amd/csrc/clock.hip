// Sustained shader clock of a GPU while other kernels run on it (include/nbody_amd.h: nb_clock_probe_begin / _end).
// One wave PER XCD (8 one-wave blocks on a private stream: the dispatcher deals consecutive workgroups to consecutive XCDs; each wave
// records the XCC_ID it ran on) reads the shader-cycle counter (s_memtime) and the constant-rate counter (s_memrealtime) when it
// starts, sleeps in s_sleep until the requested time has passed on the constant-rate counter, and reads both again: cycles per
// second over that window. The wave issues a handful of scalar instructions per ~8 000 cycles, so the kernels it runs beside
// are not disturbed; it always ends (the constant-rate counter advances whatever else happens, and the window is clamped to 5 s).
// nb_clock_probe_stop enqueues a 4-byte memset on the CALLER's stream: the wave polls that word and ends as soon as the work
// enqueued before the memset has finished — the window then is exactly that work, and a device-wide synchronisation after it
// does not wait for the rest of the requested window.
#include "nb_common.h"
#include "../../include/nbody_amd.h"
#include <new>

constexpr int NB_PROBE_WAVES = 8;   // one per XCD of an MI355X
struct nb_clock_probe {
  int device = 0;
  hipStream_t stream = nullptr;
  long long* dev = nullptr;      // [stop word | (cycles, ticks, xcc id) x NB_PROBE_WAVES]
  double tick_hz = 1e8;          // rate of s_memrealtime
  double mhz[NB_PROBE_WAVES] = {0};   // filled by nb_clock_probe_end: per wave, indexed by the XCC it reported
  int nxcc = 0;
};

namespace {

__global__ void __launch_bounds__(64) clock_probe_kernel(long long ticks, long long* out) {
  const int* stop = reinterpret_cast<const int*>(out);
  long long* mine = out + 1 + 3 * blockIdx.x;
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long c0 = clock64();
  long long r;
  do {
    __builtin_amdgcn_s_sleep(127);
    r = __builtin_amdgcn_s_memrealtime();
  } while (r - r0 < ticks && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0);
  const long long c1 = clock64();
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));      // which XCD this wave ran on
  if (threadIdx.x == 0) { mine[0] = c1 - c0; mine[1] = r - r0; mine[2] = (long long)xcc; }
}

}  // namespace

extern "C" {

int nb_clock_probe_begin(int device, double window_ms, nb_clock_probe** out) {
  if (!out || !(window_ms > 0.0)) return nb_fail(nullptr, NB_ERR_ARG, "nb_clock_probe_begin: null handle pointer or window <= 0");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return nb_fail(nullptr, NB_ERR_NODEVICE, "nb_clock_probe_begin: bad device index");
  nb_clock_probe* p = new (std::nothrow) nb_clock_probe;
  if (!p) return nb_fail(nullptr, NB_ERR_ARG, "nb_clock_probe_begin: out of host memory");
  p->device = device;
  int khz = 0;
  bool ok = hipSetDevice(device) == hipSuccess;
  if (ok && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) p->tick_hz = 1e3 * khz;
  ok = ok && hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipMalloc(&p->dev, (1 + 3 * NB_PROBE_WAVES) * sizeof(long long)) == hipSuccess;
  ok = ok && hipMemsetAsync(p->dev, 0, (1 + 3 * NB_PROBE_WAVES) * sizeof(long long), p->stream) == hipSuccess;
  if (ok) {
    if (window_ms > 5000.0) window_ms = 5000.0;
    clock_probe_kernel<<<NB_PROBE_WAVES, 64, 0, p->stream>>>((long long)(window_ms * 1e-3 * p->tick_hz), p->dev);
    ok = hipGetLastError() == hipSuccess;
  }
  if (!ok) {
    if (p->dev) (void)hipFree(p->dev);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return nb_fail(nullptr, NB_ERR_HIP, "nb_clock_probe_begin: HIP call failed");
  }
  *out = p;
  return NB_OK;
}

int nb_clock_probe_stop(nb_clock_probe* p, void* stream) {
  if (!p) return nb_fail(nullptr, NB_ERR_ARG, "nb_clock_probe_stop: null handle");
  if (hipSetDevice(p->device) != hipSuccess || hipMemsetAsync(p->dev, 0xff, sizeof(int), reinterpret_cast<hipStream_t>(stream)) != hipSuccess)
    return nb_fail(nullptr, NB_ERR_HIP, "nb_clock_probe_stop: HIP call failed");
  return NB_OK;
}

// read the waves back; returns false if none ran
static bool probe_collect(nb_clock_probe* p, double* mean_mhz, double* window_ms) {
  long long h[1 + 3 * NB_PROBE_WAVES] = {0};
  bool ok = hipSetDevice(p->device) == hipSuccess;
  ok = ok && hipStreamSynchronize(p->stream) == hipSuccess;
  ok = ok && hipMemcpy(h, p->dev, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(p->dev); p->dev = nullptr;
  (void)hipStreamDestroy(p->stream); p->stream = nullptr;
  if (!ok) return false;
  double sum = 0.0, win = 0.0;
  int got = 0;
  for (int w = 0; w < NB_PROBE_WAVES; ++w) {
    const long long cyc = h[1 + 3 * w], tk = h[2 + 3 * w], xcc = h[3 + 3 * w];
    if (tk <= 0) continue;
    const double seconds = (double)tk / p->tick_hz, f = (double)cyc / seconds * 1e-6;
    sum += f; win += seconds * 1e3; ++got;
    if (xcc >= 0 && xcc < NB_PROBE_WAVES) { p->mhz[xcc] = f; if (xcc + 1 > p->nxcc) p->nxcc = (int)xcc + 1; }
  }
  if (!got) return false;
  *mean_mhz = sum / got; *window_ms = win / got;
  return true;
}

int nb_clock_probe_end(nb_clock_probe* p, double* mhz, double* window_ms) {
  return nb_clock_probe_end_xcd(p, mhz, window_ms, nullptr, 0);
}

int nb_clock_probe_end_xcd(nb_clock_probe* p, double* mhz, double* window_ms, double* xcd_mhz, int cap) {
  if (!p) return nb_fail(nullptr, NB_ERR_ARG, "nb_clock_probe_end: null handle");
  double mean = 0.0, win = 0.0;
  const bool ok = probe_collect(p, &mean, &win);
  int n = 0;
  if (ok && xcd_mhz) for (; n < p->nxcc && n < cap; ++n) xcd_mhz[n] = p->mhz[n];
  delete p;
  if (!ok) return nb_fail(nullptr, NB_ERR_HIP, "nb_clock_probe_end: HIP call failed or the probe never ran");
  if (mhz) *mhz = mean;
  if (window_ms) *window_ms = win;
  return xcd_mhz ? n : NB_OK;
}

}  // extern "C"

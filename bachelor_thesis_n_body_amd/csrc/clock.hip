// Sustained shader clock of a GPU while other kernels run on it (include/nbody_amd.h: nb_clock_probe_begin / _end).
// One wave on a private stream reads the shader-cycle counter (s_memtime) and the constant-rate counter (s_memrealtime) when it
// starts, sleeps in s_sleep until the requested time has passed on the constant-rate counter, and reads both again: cycles per
// second over that window. The wave issues a handful of scalar instructions per ~8 000 cycles, so the kernels it runs beside
// are not disturbed; it always ends (the constant-rate counter advances whatever else happens, and the window is clamped to 5 s).
// nb_clock_probe_stop enqueues a 4-byte memset on the CALLER's stream: the wave polls that word and ends as soon as the work
// enqueued before the memset has finished — the window then is exactly that work, and a device-wide synchronisation after it
// does not wait for the rest of the requested window.
#include "nb_common.h"
#include "../../include/nbody_amd.h"
#include <new>

struct nb_clock_probe {
  int device = 0;
  hipStream_t stream = nullptr;
  long long* dev = nullptr;      // [cycles, ticks, stop word]
  double tick_hz = 1e8;          // rate of s_memrealtime
};

namespace {

__global__ void __launch_bounds__(64) clock_probe_kernel(long long ticks, long long* out) {
  const int* stop = reinterpret_cast<const int*>(out + 2);
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long c0 = clock64();
  long long r;
  do {
    __builtin_amdgcn_s_sleep(127);
    r = __builtin_amdgcn_s_memrealtime();
  } while (r - r0 < ticks && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0);
  const long long c1 = clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r - r0; }
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
  ok = ok && hipMalloc(&p->dev, 3 * sizeof(long long)) == hipSuccess;
  ok = ok && hipMemsetAsync(p->dev, 0, 3 * sizeof(long long), p->stream) == hipSuccess;
  if (ok) {
    if (window_ms > 5000.0) window_ms = 5000.0;
    clock_probe_kernel<<<1, 64, 0, p->stream>>>((long long)(window_ms * 1e-3 * p->tick_hz), p->dev);
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
  if (hipSetDevice(p->device) != hipSuccess || hipMemsetAsync(p->dev + 2, 0xff, sizeof(int), reinterpret_cast<hipStream_t>(stream)) != hipSuccess)
    return nb_fail(nullptr, NB_ERR_HIP, "nb_clock_probe_stop: HIP call failed");
  return NB_OK;
}

int nb_clock_probe_end(nb_clock_probe* p, double* mhz, double* window_ms) {
  if (!p) return nb_fail(nullptr, NB_ERR_ARG, "nb_clock_probe_end: null handle");
  long long h[2] = {0, 0};
  bool ok = hipSetDevice(p->device) == hipSuccess;
  ok = ok && hipStreamSynchronize(p->stream) == hipSuccess;
  ok = ok && hipMemcpy(h, p->dev, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(p->dev);
  (void)hipStreamDestroy(p->stream);
  const double tick_hz = p->tick_hz;
  delete p;
  if (!ok || h[1] <= 0) return nb_fail(nullptr, NB_ERR_HIP, "nb_clock_probe_end: HIP call failed or the probe never ran");
  const double seconds = (double)h[1] / tick_hz;
  if (mhz) *mhz = (double)h[0] / seconds * 1e-6;
  if (window_ms) *window_ms = seconds * 1e3;
  return NB_OK;
}

}  // extern "C"

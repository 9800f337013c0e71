// Context, device memory and error reporting of librlhip.so.
// Replaces the reference's ctypes bindings of libcudart (raleigh/algebra/cuda_wrap.py:139-162)
// and the per-call cudaMalloc/cudaFree of dense_cublas.py:245-299 with a persistent
// stream, a pinned coefficient ring and a reduction workspace.
#include <stdarg.h>

#include "common.h"

namespace rlh {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  set_error("hip error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  return (int)e ? (int)e : 1;
}

int host_threads() {
  int t = env_int("RLH_HOST_THREADS", 0);
  if (t <= 0) {
    t = (int)std::thread::hardware_concurrency();
    if (t > 16) t = 16;
  }
  return t < 1 ? 1 : t;
}

Context &ctx() {
  static Context c;
  return c;
}

int require_ready() {
  if (!ctx().ready) {
    set_error("rlh_init() has not been called");
    return 1;
  }
  return 0;
}

int check_async_error() {
  Context &c = ctx();
  if (c.async_err_h && *(volatile unsigned *)c.async_err_h != 0) {
    set_error("a device kernel gave up waiting for a dependency (persistent triangular solve): its results are invalid");
    // reported once, to the call that synchronised behind the failed launch: the caller can rebuild the operator or fall
    // back, and the library stays usable (the word used to stay set until rlh_finalize)
    *(volatile unsigned *)c.async_err_h = 0;
    return 1;
  }
  return 0;
}

int ring_acquire(size_t bytes, int *slot, void **h, void **d) {
  Context &c = ctx();
  RLH_REQUIRE(bytes <= kRingSlotBytes, "coefficient block of %zu bytes exceeds the %zu-byte staging slot",
              bytes, kRingSlotBytes);
  int s = c.ring_next;
  c.ring_next = (s + 1) % kRingSlots;
  if (c.ring_used[s]) RLH_HIP(hipEventSynchronize(c.ring_ev[s]));
  *slot = s;
  *h = c.ring_h + (size_t)s * kRingSlotBytes;
  *d = c.ring_d + (size_t)s * kRingSlotBytes;
  return 0;
}

int ring_commit(int slot, size_t bytes) {
  Context &c = ctx();
  RLH_HIP(hipMemcpyAsync(c.ring_d + (size_t)slot * kRingSlotBytes, c.ring_h + (size_t)slot * kRingSlotBytes,
                         bytes, hipMemcpyHostToDevice, c.stream));
  return 0;
}

int ring_release(int slot) {
  Context &c = ctx();
  RLH_HIP(hipEventRecord(c.ring_ev[slot], c.stream));
  c.ring_used[slot] = true;
  return 0;
}

int ensure_result(size_t bytes) {
  Context &c = ctx();
  if (bytes > c.result_d_bytes) {
    size_t nb = bytes < (1u << 20) ? (1u << 20) : bytes * 2;
    RLH_HIP(hipStreamSynchronize(c.stream));
    if (c.result_d) RLH_HIP(hipFree(c.result_d));
    if (c.result_h) RLH_HIP(hipHostFree(c.result_h));
    c.result_d = nullptr; c.result_h = nullptr; c.result_hd = nullptr; c.result_d_bytes = c.result_h_bytes = 0;
    RLH_HIP(hipMalloc((void **)&c.result_d, nb));
    RLH_HIP(hipHostMalloc((void **)&c.result_h, nb, hipHostMallocMapped));
    RLH_HIP(hipHostGetDevicePointer((void **)&c.result_hd, c.result_h, 0));
    c.result_d_bytes = c.result_h_bytes = nb;
  }
  return 0;
}

int fetch_result(void *h_out, const void *d_src, size_t bytes) {
  Context &c = ctx();
  if (bytes == 0) return 0;
  if (d_src == c.result_hd && bytes <= c.result_h_bytes) {
    // the finalize kernel has written straight into the mapped pinned buffer: no copy launch
    RLH_HIP(hipStreamSynchronize(c.stream));
    memcpy(h_out, c.result_h, bytes);
  } else if (bytes <= c.result_h_bytes) {
    RLH_HIP(hipMemcpyAsync(c.result_h, d_src, bytes, hipMemcpyDeviceToHost, c.stream));
    RLH_HIP(hipStreamSynchronize(c.stream));
    memcpy(h_out, c.result_h, bytes);
  } else {
    RLH_HIP(hipMemcpyAsync(h_out, d_src, bytes, hipMemcpyDeviceToHost, c.stream));
    RLH_HIP(hipStreamSynchronize(c.stream));
  }
  return check_async_error();
}

}  // namespace rlh

using namespace rlh;

extern "C" {

int rlh_version(void) { return RLH_VERSION; }

const char *rlh_last_error(void) { return g_err; }

int rlh_device_count(int *count) {
  RLH_REQUIRE(count != nullptr, "rlh_device_count: null argument");
  RLH_HIP(hipGetDeviceCount(count));
  return 0;
}

int rlh_init(int device) {
  Context &c = ctx();
  if (c.ready) {
    RLH_REQUIRE(c.device == device, "rlh_init: already bound to device %d (one device per process)", c.device);
    return 0;
  }
  int count = 0;
  RLH_HIP(hipGetDeviceCount(&count));
  RLH_REQUIRE(device >= 0 && device < count, "rlh_init: device %d out of range (%d visible)", device, count);
  RLH_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  RLH_HIP(hipGetDeviceProperties(&prop, device));
  RLH_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
              "rlh_init: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  c.num_cu = prop.multiProcessorCount;
  RLH_HIP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
  c.stream = c.own_stream;
  RLH_HIP(hipHostMalloc((void **)&c.ring_h, kRingSlots * kRingSlotBytes, hipHostMallocDefault));
  RLH_HIP(hipMalloc((void **)&c.ring_d, kRingSlots * kRingSlotBytes));
  for (int i = 0; i < kRingSlots; ++i) {
    RLH_HIP(hipEventCreateWithFlags(&c.ring_ev[i], hipEventDisableTiming));
    c.ring_used[i] = false;
  }
  RLH_HIP(hipMalloc((void **)&c.work, kWorkspaceBytes));
  RLH_HIP(hipEventCreate(&c.t0));
  RLH_HIP(hipEventCreate(&c.t1));
  RLH_HIP(hipHostMalloc((void **)&c.async_err_h, 64, hipHostMallocMapped));
  *c.async_err_h = 0;
  RLH_HIP(hipHostGetDevicePointer((void **)&c.async_err_d, c.async_err_h, 0));
  c.device = device;
  c.ready = true;
  if (int rc = ensure_result(1u << 20)) return rc;
  return 0;
}

int rlh_finalize(void) {
  Context &c = ctx();
  if (!c.ready) return 0;
  (void)hipStreamSynchronize(c.stream);
  for (int i = 0; i < kRingSlots; ++i) (void)hipEventDestroy(c.ring_ev[i]);
  (void)hipEventDestroy(c.t0);
  (void)hipEventDestroy(c.t1);
  (void)hipHostFree(c.ring_h);
  (void)hipFree(c.ring_d);
  (void)hipFree(c.work);
  if (c.result_d) (void)hipFree(c.result_d);
  if (c.result_h) (void)hipHostFree(c.result_h);
  if (c.async_err_h) (void)hipHostFree(c.async_err_h);
  (void)hipStreamDestroy(c.own_stream);
  c = Context();
  return 0;
}

int rlh_set_stream(void *hip_stream) {
  if (int rc = require_ready()) return rc;
  Context &c = ctx();
  RLH_HIP(hipStreamSynchronize(c.stream));
  c.stream = hip_stream ? (hipStream_t)hip_stream : c.own_stream;
  return 0;
}

int rlh_sync(void) {
  if (int rc = require_ready()) return rc;
  RLH_HIP(hipStreamSynchronize(ctx().stream));
  return check_async_error();
}

int rlh_mem_info(int64_t *free_bytes, int64_t *total_bytes) {
  if (int rc = require_ready()) return rc;
  size_t f = 0, t = 0;
  RLH_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return 0;
}

int rlh_malloc(void **dptr, int64_t bytes) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dptr != nullptr && bytes >= 0, "rlh_malloc: bad arguments");
  *dptr = nullptr;
  if (bytes == 0) return 0;
  RLH_HIP(hipMalloc(dptr, (size_t)bytes));
  return 0;
}

int rlh_free(void *dptr) {
  if (!dptr) return 0;
  if (!ctx().ready) return 0;     // interpreter teardown after rlh_finalize
  RLH_HIP(hipStreamSynchronize(ctx().stream));
  RLH_HIP(hipFree(dptr));
  return 0;
}

int rlh_memset(void *dptr, int value, int64_t bytes) {
  if (int rc = require_ready()) return rc;
  if (bytes <= 0) return 0;
  RLH_HIP(hipMemsetAsync(dptr, value, (size_t)bytes, ctx().stream));
  return 0;
}

int rlh_h2d(void *dptr, const void *hptr, int64_t bytes) {
  if (int rc = require_ready()) return rc;
  if (bytes <= 0) return 0;
  RLH_HIP(hipMemcpyAsync(dptr, hptr, (size_t)bytes, hipMemcpyHostToDevice, ctx().stream));
  RLH_HIP(hipStreamSynchronize(ctx().stream));
  return 0;
}

int rlh_d2h(void *hptr, const void *dptr, int64_t bytes) {
  if (int rc = require_ready()) return rc;
  if (bytes <= 0) return 0;
  RLH_HIP(hipMemcpyAsync(hptr, dptr, (size_t)bytes, hipMemcpyDeviceToHost, ctx().stream));
  RLH_HIP(hipStreamSynchronize(ctx().stream));
  return 0;
}

int rlh_fetch(void *hptr, const void *dptr, int64_t bytes) {
  if (int rc = require_ready()) return rc;
  if (bytes <= 0) return 0;
  RLH_REQUIRE(hptr && dptr, "rlh_fetch: null pointer");
  if (int rc = ensure_result((size_t)bytes)) return rc;
  return fetch_result(hptr, dptr, (size_t)bytes);
}

int rlh_d2d(void *dst, const void *src, int64_t bytes) {
  if (int rc = require_ready()) return rc;
  if (bytes <= 0) return 0;
  RLH_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, ctx().stream));
  return 0;
}

int rlh_copy2d(void *dst, int64_t dpitch, const void *src, int64_t spitch, int64_t width_bytes, int64_t rows,
               int kind) {
  if (int rc = require_ready()) return rc;
  if (width_bytes <= 0 || rows <= 0) return 0;
  RLH_REQUIRE(kind >= 0 && kind <= 2, "rlh_copy2d: kind must be 0 (h2d), 1 (d2h) or 2 (d2d)");
  RLH_REQUIRE(dpitch >= width_bytes && spitch >= width_bytes, "rlh_copy2d: pitch smaller than width");
  hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
  RLH_HIP(hipMemcpy2DAsync(dst, (size_t)dpitch, src, (size_t)spitch, (size_t)width_bytes, (size_t)rows, k,
                           ctx().stream));
  if (kind != 2) RLH_HIP(hipStreamSynchronize(ctx().stream));
  return 0;
}

int rlh_timer_start(void) {
  if (int rc = require_ready()) return rc;
  RLH_HIP(hipEventRecord(ctx().t0, ctx().stream));
  return 0;
}

int rlh_timer_stop(float *milliseconds) {
  if (int rc = require_ready()) return rc;
  RLH_HIP(hipEventRecord(ctx().t1, ctx().stream));
  RLH_HIP(hipEventSynchronize(ctx().t1));
  float ms = 0.f;
  RLH_HIP(hipEventElapsedTime(&ms, ctx().t0, ctx().t1));
  if (milliseconds) *milliseconds = ms;
  return 0;
}

}  // extern "C"

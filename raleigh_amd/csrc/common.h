// Shared declarations for librlhip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <string>
#include <thread>
#include <vector>

#include "rlhip.h"

namespace rlh {

// ---------------------------------------------------------------- errors
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define RLH_HIP(call)                                                   \
  do {                                                                  \
    hipError_t e_ = (call);                                             \
    if (e_ != hipSuccess) return rlh::hip_fail(e_, #call, __FILE__, __LINE__); \
  } while (0)

#define RLH_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      rlh::set_error(__VA_ARGS__);    \
      return 1;                       \
    }                                 \
  } while (0)

// ---------------------------------------------------------------- environment, host thread pool
static inline int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}
// threads for the host-side set-up work (ILUT factorisation, SpMM layout build): RLH_HOST_THREADS, else the machine's
// hardware threads, at most 16 (a GPU box hands every GPU's process about that share of its cores)
int host_threads();
// f(t, nthreads) on min(want, host_threads()) threads (the calling thread is one of them)
template <typename F> static inline void host_parallel(int want, F f) {
  int t = host_threads();
  if (t > want) t = want;
  if (t <= 1) { f(0, 1); return; }
  std::vector<std::thread> pool;
  for (int i = 1; i < t; ++i) pool.emplace_back([&f, i, t]() { f(i, t); });
  f(0, t);
  for (auto &th : pool) th.join();
}

// ---------------------------------------------------------------- dtypes
struct c32 { float re, im; };
struct c64 { double re, im; };

template <int DT> struct DType;
template <> struct DType<RLH_S> { using T = float;  using R = float;  static constexpr bool cplx = false; };
template <> struct DType<RLH_D> { using T = double; using R = double; static constexpr bool cplx = false; };
template <> struct DType<RLH_C> { using T = c32;    using R = float;  static constexpr bool cplx = true; };
template <> struct DType<RLH_Z> { using T = c64;    using R = double; static constexpr bool cplx = true; };

static inline int64_t dtype_size(int dt) {
  switch (dt) { case RLH_S: return 4; case RLH_D: return 8; case RLH_C: return 8; case RLH_Z: return 16; }
  return 0;
}
static inline bool dtype_valid(int dt) { return dt >= RLH_S && dt <= RLH_Z; }

// Scalar arithmetic shared by the VALU kernels.
__device__ __forceinline__ float  zero_of(float)  { return 0.f; }
__device__ __forceinline__ double zero_of(double) { return 0.0; }
__device__ __forceinline__ c32 zero_of(c32) { return c32{0.f, 0.f}; }
__device__ __forceinline__ c64 zero_of(c64) { return c64{0.0, 0.0}; }

// acc += a * b
__device__ __forceinline__ void fma_acc(float &acc, float a, float b)   { acc = fmaf(a, b, acc); }
__device__ __forceinline__ void fma_acc(double &acc, double a, double b) { acc = fma(a, b, acc); }
__device__ __forceinline__ void fma_acc(c32 &acc, c32 a, c32 b) {
  acc.re = fmaf(a.re, b.re, acc.re); acc.re = fmaf(-a.im, b.im, acc.re);
  acc.im = fmaf(a.re, b.im, acc.im); acc.im = fmaf(a.im, b.re, acc.im);
}
__device__ __forceinline__ void fma_acc(c64 &acc, c64 a, c64 b) {
  acc.re = fma(a.re, b.re, acc.re); acc.re = fma(-a.im, b.im, acc.re);
  acc.im = fma(a.re, b.im, acc.im); acc.im = fma(a.im, b.re, acc.im);
}
// acc += conj(a) * b
__device__ __forceinline__ void fma_conj_acc(float &acc, float a, float b)   { acc = fmaf(a, b, acc); }
__device__ __forceinline__ void fma_conj_acc(double &acc, double a, double b) { acc = fma(a, b, acc); }
__device__ __forceinline__ void fma_conj_acc(c32 &acc, c32 a, c32 b) {
  acc.re = fmaf(a.re, b.re, acc.re); acc.re = fmaf(a.im, b.im, acc.re);
  acc.im = fmaf(a.re, b.im, acc.im); acc.im = fmaf(-a.im, b.re, acc.im);
}
__device__ __forceinline__ void fma_conj_acc(c64 &acc, c64 a, c64 b) {
  acc.re = fma(a.re, b.re, acc.re); acc.re = fma(a.im, b.im, acc.re);
  acc.im = fma(a.re, b.im, acc.im); acc.im = fma(-a.im, b.re, acc.im);
}
__device__ __forceinline__ float  add_of(float a, float b)   { return a + b; }
__device__ __forceinline__ double add_of(double a, double b) { return a + b; }
__device__ __forceinline__ c32 add_of(c32 a, c32 b) { return c32{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c64 add_of(c64 a, c64 b) { return c64{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ float  mul_of(float a, float b)   { return a * b; }
__device__ __forceinline__ double mul_of(double a, double b) { return a * b; }
__device__ __forceinline__ c32 mul_of(c32 a, c32 b) { return c32{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ c64 mul_of(c64 a, c64 b) { return c64{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// ---------------------------------------------------------------- context
constexpr int kRingSlots = 16;
constexpr size_t kRingSlotBytes = 1u << 20;       // 1 MiB per coefficient slot
constexpr size_t kWorkspaceBytes = 96u << 20;     // reduction partials

struct Context {
  bool ready = false;
  int device = -1;
  int num_cu = 256;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // coefficient staging ring: pinned host slot -> device slot, async
  char *ring_h = nullptr;
  char *ring_d = nullptr;
  hipEvent_t ring_ev[kRingSlots];
  bool ring_used[kRingSlots];
  int ring_next = 0;
  // reduction workspace + result buffers
  char *work = nullptr;
  char *result_d = nullptr;
  size_t result_d_bytes = 0;
  char *result_h = nullptr;      // pinned, mapped into the device address space
  char *result_hd = nullptr;     // device-side address of result_h (zero-copy result delivery)
  size_t result_h_bytes = 0;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // raised by a kernel that gave up on a bounded spin (persistent triangular solve): mapped pinned word
  unsigned *async_err_h = nullptr;
  unsigned *async_err_d = nullptr;
};

Context &ctx();
int require_ready();
// non-zero (and the error text set) once a kernel has reported a failure it could not return: checked by every
// call that synchronises with the stream and by the calls that launch such kernels
int check_async_error();

// Returns a pinned host slot and its device twin; the caller fills `*h`,
// then calls ring_commit(slot, bytes) which enqueues the H2D copy.  After the
// consuming kernel has been launched call ring_release(slot).
int ring_acquire(size_t bytes, int *slot, void **h, void **d);
int ring_commit(int slot, size_t bytes);
int ring_release(int slot);

int ensure_result(size_t bytes);          // result_d / result_h >= bytes
// copy device result to user host memory through the pinned buffer + sync
int fetch_result(void *h_out, const void *d_src, size_t bytes);

static inline bool aligned16(const void *p, int64_t ld, int64_t es) {
  return ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) && (((ld * es) & 15) == 0);
}

}  // namespace rlh

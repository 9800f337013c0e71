// Hand-off latency probe for gfx950 (not part of the library): how long does a 16-byte piece take from one
// workgroup's store to another workgroup's L1-bypassing poll, when both sit on ONE XCD and when they sit on two,
// for the store / load flavours the persistent triangular solve (csrc/sptrsv.hip) can choose from?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/handoff_probe tools/handoff_probe.hip && /tmp/handoff_probe
// Every spin is bounded by the wall clock; a flavour that never becomes visible prints "never seen".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int LD> __device__ __forceinline__ u4 load16(const u4 *p) {
  u4 v;
  if (LD == 0) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  else if (LD == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
template <int ST> __device__ __forceinline__ void store16(u4 *p, u4 v) {
  if (ST == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
  else if (ST == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

constexpr unsigned long long kLimit = 100000000ull;      // 1 s of the 100 MHz wall clock

// ctrl[0..63]: xcc of every block; ctrl[64]: arrivals; box: two pieces 4 KB apart (ping, pong)
template <int ST, int LD>
__global__ __launch_bounds__(64) void pingpong(unsigned *ctrl, u4 *box, int rounds, int cross, int sleep, unsigned long long *out) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 7;
  const int b = blockIdx.x, nb = gridDim.x;
  const unsigned long long t00 = wall_clock64();
  if (threadIdx.x == 0) {
    __hip_atomic_store(ctrl + b, xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(ctrl + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctrl + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nb)
      if (wall_clock64() - t00 > kLimit) break;
  }
  __syncthreads();
  int partner = -1;
  const unsigned x0 = __hip_atomic_load(ctrl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int k = 1; k < nb && partner < 0; ++k) {
    const unsigned xk = __hip_atomic_load(ctrl + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cross ? xk != x0 : xk == x0) partner = k;
  }
  if (partner < 0) { if (b == 0 && threadIdx.x == 0) out[0] = ~0ull; return; }
  if (b != 0 && b != partner) return;
  if (threadIdx.x != 0) return;
  u4 *mine = box + (b == 0 ? 0 : 256), *theirs = box + (b == 0 ? 256 : 0);
  // warm both lines into this CU's caches the way a poller would have them
  (void)load16<LD>(theirs);
  const unsigned long long t0 = wall_clock64();
  bool lost = false;
  for (int r = 1; r <= rounds && !lost; ++r) {
    if (b == 0) store16<ST>(mine, u4{(unsigned)r, 0u, 0u, (unsigned)r});
    for (;;) {                                               // wait for the other side's round r
      const u4 v = load16<LD>(theirs);
      if (v.x == (unsigned)r && v.w == (unsigned)r) break;
      if (sleep) __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > kLimit) { lost = true; break; }
    }
    if (b != 0) store16<ST>(mine, u4{(unsigned)r, 0u, 0u, (unsigned)r});
  }
  if (b == 0) { out[0] = lost ? ~0ull - 1 : wall_clock64() - t0; out[1] = x0 - 1; out[2] = ctrl[partner] - 1; }
}

template <int ST, int LD>
static void run(const char *name, unsigned *ctrl, u4 *box, unsigned long long *out, int sleep) {
  for (int cross = 0; cross < 2; ++cross) {
    const int rounds = 2000;
    CK(hipMemset(ctrl, 0, 4096));
    CK(hipMemset(box, 0, 2 * 4096));
    CK(hipMemset(out, 0, 64));
    hipLaunchKernelGGL((pingpong<ST, LD>), dim3(32), dim3(64), 0, 0, ctrl, box, rounds, cross, sleep, out);
    CK(hipDeviceSynchronize());
    unsigned long long h[3];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    if (h[0] == ~0ull) printf("%-34s %s: no partner\n", name, cross ? "two XCDs" : "one XCD ");
    else if (h[0] == ~0ull - 1) printf("%-34s %s: never seen\n", name, cross ? "two XCDs" : "one XCD ");
    else printf("%-34s %s (xcc %llu -> %llu): %.3f us per hop\n", name, cross ? "two XCDs" : "one XCD ", h[1], h[2],
                (double)h[0] * 0.01 / (2.0 * rounds));
  }
}

int main() {
  unsigned *ctrl; u4 *box; unsigned long long *out;
  CK(hipMalloc(&ctrl, 4096)); CK(hipMalloc(&box, 2 * 4096)); CK(hipMalloc(&out, 64));
  for (int sleep = 0; sleep < 2; ++sleep) {
    printf("--- poll loop %s s_sleep 1\n", sleep ? "with" : "without");
    run<0, 0>("plain store, sc1 load", ctrl, box, out, sleep);
    run<1, 0>("sc1 store, sc1 load", ctrl, box, out, sleep);
    run<2, 1>("sc0 sc1 store, sc0 sc1 load", ctrl, box, out, sleep);
    run<0, 2>("plain store, nt load", ctrl, box, out, sleep);
    run<0, 1>("plain store, sc0 sc1 load", ctrl, box, out, sleep);
  }
  return 0;
}

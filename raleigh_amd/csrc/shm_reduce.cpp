// Sum of a SMALL host array over the processes of one node through POSIX shared memory (host-only translation unit).
//
// The reductions of the hot path (m x m Gram matrices, m dots: a few KB) end on the HOST -- the solver's Rayleigh-Ritz
// step runs there (raleigh/core/solver.py:1117-1187 reads them as NumPy arrays) -- so on a row-sharded run every one of them
// is: partial result -> all ranks' sum -> host.  Through RCCL that is a collective launch on the stream plus a copy back,
// 28-40 us of fixed cost, thirteen times per inner iteration; at eight ranks the per-rank kernels of the headline take
// 1.4 ms and those fixed costs 0.4 ms.  Here every rank fetches ITS partial (one copy + synchronisation it pays anyway),
// stores it in its slot of a shared segment and adds up the slots of all ranks in rank order -- the same bits on every
// rank -- after a flag-per-rank hand-off: 2-4 us for eight ranks.  Large reductions (the N x k blocks of the dense
// transposed product) and the halo exchange stay on RCCL.
//
// Protocol of call k (k = 1, 2, ... per handle; two buffers per rank, parity k & 1):
//   write own data to slot[rank][k & 1]; flag[rank] = k (release);
//   for r = 0 .. nranks - 1: wait until flag[r] >= k (acquire); accumulate slot[r][k & 1] in rank order.
// A rank overwrites parity k & 1 again in call k + 2, which it enters only after it has seen flag[r] >= k + 1 of every r in
// call k + 1, and a rank raises its flag to k + 1 only after it has finished reading in call k: two buffers suffice.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <new>

#include "common.h"

struct rlh_shm {
  int rank, nranks;
  int64_t slot;                     // bytes per buffer
  char *base;
  size_t bytes;
  uint64_t k;
  double timeout_s;
};

namespace {

constexpr size_t kLine = 64;
static inline std::atomic<uint64_t> *flag_of(char *base, int r) { return reinterpret_cast<std::atomic<uint64_t> *>(base + (size_t)r * kLine); }
static inline char *slot_of(const rlh_shm *s, int r, int parity) {
  return s->base + (size_t)s->nranks * kLine + ((size_t)r * 2 + (size_t)parity) * (size_t)s->slot;
}
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

}  // namespace

using namespace rlh;

extern "C" {

int rlh_shm_create(rlh_shm_t *out, const char *name, int rank, int nranks, int64_t slot_bytes) {
  RLH_REQUIRE(out != nullptr, "rlh_shm_create: null handle pointer");
  *out = nullptr;
  RLH_REQUIRE(name && name[0] == '/' && rank >= 0 && nranks >= 1 && nranks <= 1024 && rank < nranks && slot_bytes >= 64 &&
              slot_bytes <= ((int64_t)1 << 28), "rlh_shm_create: bad arguments");
  static_assert(sizeof(std::atomic<uint64_t>) == 8, "flags are 8 bytes");
  const int64_t slot = (slot_bytes + 63) & ~(int64_t)63;
  const size_t bytes = (size_t)nranks * kLine + (size_t)nranks * 2 * (size_t)slot;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    RLH_REQUIRE(fd >= 0, "rlh_shm_create: cannot create the shared segment %s", name);
    if (ftruncate(fd, (off_t)bytes) != 0) {
      close(fd);
      shm_unlink(name);
      set_error("rlh_shm_create: cannot size the shared segment %s to %zu bytes", name, bytes);
      return 1;
    }
  } else {
    const double t0 = now_s();
    for (;;) {                                   // rank 0 may not have created (or sized) it yet
      fd = shm_open(name, O_RDWR, 0600);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
        close(fd);
        fd = -1;
      }
      RLH_REQUIRE(now_s() - t0 < 60.0, "rlh_shm_create: the shared segment %s did not appear", name);
      usleep(1000);
    }
  }
  void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    if (rank == 0) shm_unlink(name);
    set_error("rlh_shm_create: cannot map the shared segment %s", name);
    return 1;
  }
  // (a fresh segment is zero-filled: every flag starts at 0, the first call is k = 1)
  rlh_shm *s = new rlh_shm();
  s->rank = rank; s->nranks = nranks; s->slot = slot; s->base = static_cast<char *>(p); s->bytes = bytes; s->k = 0;
  s->timeout_s = (double)env_int("RLH_SHM_TIMEOUT", 300);
  *out = s;
  return 0;
}

// removes the NAME (every rank keeps its mapping): call on rank 0 once all ranks have created their handles
int rlh_shm_unlink(const char *name) {
  RLH_REQUIRE(name && name[0] == '/', "rlh_shm_unlink: bad name");
  shm_unlink(name);
  return 0;
}

int rlh_shm_allreduce(rlh_shm_t s, int dtype, int64_t count, void *inout) {
  RLH_REQUIRE(s != nullptr && inout != nullptr && count >= 0, "rlh_shm_allreduce: bad arguments");
  RLH_REQUIRE(dtype == RLH_S || dtype == RLH_D, "rlh_shm_allreduce: float32 or float64 elements (complex data as pairs)");
  const size_t es = dtype == RLH_S ? 4 : 8;
  RLH_REQUIRE((int64_t)(count * es) <= s->slot, "rlh_shm_allreduce: %lld bytes exceed the slot of %lld", (long long)(count * es),
              (long long)s->slot);
  const uint64_t k = ++s->k;
  const int parity = (int)(k & 1);
  if (count > 0) memcpy(slot_of(s, s->rank, parity), inout, (size_t)count * es);
  flag_of(s->base, s->rank)->store(k, std::memory_order_release);
  const double t0 = now_s();
  for (int r = 0; r < s->nranks; ++r) {
    std::atomic<uint64_t> *f = flag_of(s->base, r);
    uint64_t spins = 0;
    while (f->load(std::memory_order_acquire) < k) {
      cpu_relax();
      if ((++spins & 0xFFFFF) == 0 && now_s() - t0 > s->timeout_s) {
        set_error("rlh_shm_allreduce: rank %d did not arrive at reduction %llu within %.0f s", r, (unsigned long long)k, s->timeout_s);
        return 1;
      }
    }
    const char *src = slot_of(s, r, parity);
    if (dtype == RLH_D) {
      double *o = static_cast<double *>(inout);
      const double *v = reinterpret_cast<const double *>(src);
      if (r == 0) for (int64_t i = 0; i < count; ++i) o[i] = v[i];
      else for (int64_t i = 0; i < count; ++i) o[i] += v[i];
    } else {
      float *o = static_cast<float *>(inout);
      const float *v = reinterpret_cast<const float *>(src);
      if (r == 0) for (int64_t i = 0; i < count; ++i) o[i] = v[i];
      else for (int64_t i = 0; i < count; ++i) o[i] += v[i];
    }
  }
  return 0;
}

int rlh_shm_destroy(rlh_shm_t s) {
  if (!s) return 0;
  if (s->base) munmap(s->base, s->bytes);
  delete s;
  return 0;
}

}  // extern "C"

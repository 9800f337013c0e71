// Incomplete LU preconditioner on the device: host ILUT factorisation + level-scheduled
// sparse triangular solves on the whole n x m block.
//
// Reference: raleigh/algebra/sparse_mkl.py:122-140 (IncompleteLU) -> raleigh/algebra/mkl_wrap.py:279-347:
// mkl dcsrilut(tol, maxfil) once, then per VECTOR two mkl_dcsrtrsv calls on the host.  Here
//  * rlh_ilut_factor restates the dual-threshold ILUT(p, tau) of Saad that dcsrilut implements (drop
//    entries below tau * ||row||_2, keep at most p = maxfil largest entries in the L and in the U
//    part of every row) on the host, in double / complex double;
//  * rlh_sptrsv_create turns a triangular CSR factor into a device operator: rows are grouped into
//    dependency LEVELS (a row's level is one more than the largest level among the rows it
//    references), the rows of a level are independent;
//  * rlh_sptrsv_solve_chain applies a chain of such operators to a block of m vectors at once in ONE
//    persistent launch: rows wait for the rows they read, not for their level (design notes at the
//    device section below).  The reference applies 2 m sequential mkl_dcsrtrsv on the host.
#include <math.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <complex>
#include <memory>
#include <thread>
#include <vector>

#include "common.h"

struct rlh_factors {
  int dtype;                       // RLH_D or RLH_Z
  int64_t n;
  std::vector<int64_t> lptr, uptr;
  std::vector<int32_t> lidx, uidx;
  std::vector<char> lval, uval;    // L strictly lower (unit diagonal implied), U upper incl. diagonal
};

namespace rlh { struct TrsvUnit; struct TrsvRow; }
struct TrsvPlan {                  // units of an operator for `pl` lanes across the pieces of a row
  int pl = 0;
  int64_t nunits = 0;
  rlh::TrsvUnit *units = nullptr;  // device
  rlh::TrsvRow *rows = nullptr;    // device, by position
};

struct rlh_sptrsv {
  int dtype;
  int64_t n, nnz;                  // nnz: stored off-diagonal entries
  int lower, unit;
  // device arrays in LEVEL order: position p holds one row (pos_of maps rows to positions); its entries are one
  // contiguous run of cols / vals: x entries (column = the POSITION of the row read, ascending: the oldest dependencies
  // come first), then rhs entries (column = original row; see sptrsv_build).  The result image of an operator is
  // indexed by position too: rows of one level that read neighbouring rows of an earlier level -- any structured grid
  // or node-blocked matrix -- then gather neighbouring pieces, which the memory pipeline merges into few requests
  int32_t *cols;
  void *vals;
  int32_t *pos_of;                 // row -> position
  std::vector<int64_t> lev_off;    // host: first position of every level, nlevels + 1
  std::vector<int64_t> rowptr_h;   // host: first entry of every position, n + 1
  std::vector<int32_t> dep_pos_h;  // host: per x entry, the position of the row it reads (plans are built from it)
  std::vector<int32_t> nx_h;       // host: x entries of every position (its rhs entries follow them)
  int64_t entries, nblocks;        // stored entries after the block transform; diagonal blocks
  int64_t levels_plain;            // dependency levels of the factor as given
  TrsvPlan plan[2];
  int plan_lru;
  int64_t device_bytes;
  // scratch (control words + block images) of the chain this operator heads
  void *work;
  int64_t work_bytes;
};

namespace rlh {

// ------------------------------------------------------------------ host ILUT
static inline double mag(double v) { return fabs(v); }
static inline double mag(const std::complex<double> &v) { return std::abs(v); }

// Row i of the factorisation reads the finished U rows k < i it eliminates with, in increasing k: the rows are claimed
// in order by a pool of host threads, and a thread that needs a row still being worked on waits for that row alone (a
// per-row flag; the lowest unfinished row never waits, so there is no deadlock).  Every row performs exactly the
// operations of the sequential algorithm in the same order, so the factors are bit-identical to it whatever the number
// of threads (tests/test_ilu_cpu.py compares with the pure-Python restatement).  A row's recent neighbours come LAST in
// its elimination order, so what stays serial per row is one elimination step and the selection of the kept entries:
// on an FE matrix whose rows eliminate several hundred pivots the pool scales with the thread count.
struct IlutArena {                               // per-thread storage of finished rows (never moved: other threads read it)
  std::vector<std::unique_ptr<char[]>> chunks;
  size_t used = 0, cap = 0;
  char *take(size_t bytes) {
    bytes = (bytes + 15) & ~(size_t)15;
    if (chunks.empty() || used + bytes > cap) {
      cap = std::max<size_t>(bytes, (size_t)8 << 20);
      chunks.emplace_back(new char[cap]);
      used = 0;
    }
    char *p = chunks.back().get() + used;
    used += bytes;
    return p;
  }
};
template <typename S> struct IlutRow { const int32_t *idx; const S *val; int32_t cnt; };   // U: diagonal first, then ascending

template <typename S>
static int ilut_factor(int64_t n, const int64_t *indptr, const int32_t *indices, const S *values, double tol,
                       int64_t maxfil, rlh_factors *f) {
  f->n = n;
  f->lptr.assign((size_t)n + 1, 0);
  f->uptr.assign((size_t)n + 1, 0);
  if (n == 0) return 0;
  std::vector<IlutRow<S>> lrow((size_t)n), urow((size_t)n);
  std::unique_ptr<std::atomic<unsigned char>[]> done(new std::atomic<unsigned char>[(size_t)n]);
  for (int64_t i = 0; i < n; ++i) done[(size_t)i].store(0, std::memory_order_relaxed);
  std::atomic<int64_t> next{0}, bad_row{-1};
  std::atomic<int> failed{0};
  // Threads pay where a row eliminates many pivots (FE matrices: hundreds per row).  The rows of a stencil do a few dozen
  // operations each and read three or four rows another thread has just written: the cache misses cost more than the
  // arithmetic, and the pool is slower than one thread (lap3d 100^3: 0.9 s against 0.35 s) -- such matrices stay serial.
  // (Chunks of consecutive rows per claim were tried for them: no faster than serial, and on the FE matrix the thread of
  // the next chunk waits for the whole chunk before it: 6.5 s against 2.4 s row by row.)
  const int64_t chunk = 1;
  const bool wide = indptr[n] >= 16 * n;
  const int nthreads = wide ? (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, n / 256)) : 1;
  std::vector<IlutArena> arenas((size_t)nthreads);

  const bool stats = env_int("RLH_ILUT_STATS", 0) != 0;
  std::atomic<long long> st_l{0}, st_u{0};
  auto worker = [&](int tid) {
    IlutArena &arena = arenas[(size_t)tid];
    struct Slot { S w; int32_t mark; };                 // value and row stamp of a column side by side: one cache line per touch
    std::vector<Slot> slot((size_t)n, Slot{S(0), -1});
    std::vector<int32_t> lcand, ucand, heap;
    std::vector<std::pair<double, int32_t>> keep;
    const auto later = std::greater<int32_t>();
    int64_t i = 0, chunk_end = 0;
    for (;; ++i) {
      if (i >= chunk_end) {                             // the next chunk of consecutive rows
        i = next.fetch_add(chunk, std::memory_order_relaxed);
        chunk_end = std::min(n, i + chunk);
      }
      if (i >= n || failed.load(std::memory_order_relaxed)) break;
      double nrm = 0.0;
      for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) nrm += mag(values[e]) * mag(values[e]);
      nrm = sqrt(nrm);
      if (!(nrm > 0.0)) {
        bad_row.store(i);
        failed.store(1);
        break;
      }
      const double tau = tol * nrm;
      heap.clear();                                     // columns < i still to eliminate (min-heap)
      ucand.clear();
      lcand.clear();
      slot[i].mark = (int32_t)i;
      slot[i].w = S(0);
      for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
        const int32_t j = indices[e];
        if (slot[j].mark != i || j == i) {
          if (slot[j].mark != i) { slot[j].mark = (int32_t)i; slot[j].w = S(0); }
          if (j < i) { heap.push_back(j); std::push_heap(heap.begin(), heap.end(), later); }
          else if (j > i) ucand.push_back(j);
        }
        slot[j].w += values[e];
      }
      // (duplicate column indices in a row were summed above; a duplicate may have been pushed twice)
      int32_t last = -1;
      bool bail = false;
      while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), later);
        const int32_t k = heap.back();
        heap.pop_back();
        if (k == last) continue;
        last = k;
        if (!heap.empty()) {                              // the next pivot's row was written by another thread: start fetching it
          const int32_t kn = heap.front();
          if (done[(size_t)kn].load(std::memory_order_acquire)) {
            const IlutRow<S> &un = urow[(size_t)kn];
            __builtin_prefetch(un.idx);
            __builtin_prefetch(un.val);
            __builtin_prefetch((const char *)un.val + 64);
            __builtin_prefetch((const char *)un.val + 128);
            __builtin_prefetch((const char *)un.idx + 64);
          }
        }
        if (!done[(size_t)k].load(std::memory_order_acquire)) {          // row k is still being factorised by another thread
          unsigned spins = 0;
          while (!done[(size_t)k].load(std::memory_order_acquire)) {
            if (failed.load(std::memory_order_relaxed)) { bail = true; break; }
            if (++spins > 64) std::this_thread::yield();
          }
          if (bail) break;
        }
        const IlutRow<S> &uk = urow[(size_t)k];
        const S lik = slot[k].w / uk.val[0];
        if (mag(lik) < tau) continue;                       // first dropping rule
        slot[k].w = lik;
        lcand.push_back(k);
        for (int32_t e = 1; e < uk.cnt; ++e) {              // row k of U beyond its diagonal
          const int32_t j = uk.idx[e];
          if (slot[j].mark != i) {
            slot[j].mark = (int32_t)i;
            slot[j].w = S(0);
            if (j < i) { heap.push_back(j); std::push_heap(heap.begin(), heap.end(), later); }
            else ucand.push_back(j);
          }
          slot[j].w -= lik * uk.val[e];
        }
      }
      if (bail) break;
      // second dropping rule: the p largest of each part (all of them already >= tau in L)
      auto select = [&](std::vector<int32_t> &cand, bool threshold) {
        keep.clear();
        for (int32_t j : cand) {
          const double a = mag(slot[j].w);
          if (!threshold || a >= tau) keep.push_back({a, j});
        }
        if ((int64_t)keep.size() > maxfil) {
          std::nth_element(keep.begin(), keep.begin() + maxfil, keep.end(),
                           [](const std::pair<double, int32_t> &x, const std::pair<double, int32_t> &y) {
                             return x.first > y.first || (x.first == y.first && x.second < y.second);
                           });
          keep.resize((size_t)maxfil);
        }
        cand.clear();
        for (auto &kv : keep) cand.push_back(kv.second);
        std::sort(cand.begin(), cand.end());
      };
      if (stats) { st_l += (long long)lcand.size(); st_u += (long long)ucand.size(); }
      // (a column enters ucand once: its stamp is set when it does -- no sort / unique pass over the few hundred
      // candidates here, on the part of a row's work that no other thread can overlap)
      select(lcand, false);
      select(ucand, true);
      {
        const size_t nl = lcand.size(), nu = ucand.size() + 1;
        int32_t *li = (int32_t *)arena.take(nl * sizeof(int32_t));
        S *lv = (S *)arena.take(nl * sizeof(S));
        for (size_t q = 0; q < nl; ++q) { li[q] = lcand[q]; lv[q] = slot[lcand[q]].w; }
        lrow[(size_t)i] = IlutRow<S>{li, lv, (int32_t)nl};
        int32_t *ui = (int32_t *)arena.take(nu * sizeof(int32_t));
        S *uv = (S *)arena.take(nu * sizeof(S));
        S d = slot[i].w;
        if (mag(d) < tau || mag(d) == 0.0) d = S(tau > 0.0 ? tau : 1e-4 * nrm);   // small pivot: replaced, as dcsrilut does
        ui[0] = (int32_t)i;
        uv[0] = d;
        for (size_t q = 1; q < nu; ++q) { ui[q] = ucand[q - 1]; uv[q] = slot[ucand[q - 1]].w; }
        urow[(size_t)i] = IlutRow<S>{ui, uv, (int32_t)nu};
      }
      done[(size_t)i].store(1, std::memory_order_release);
    }
  };
  host_parallel(nthreads, [&](int t, int) { worker(t); });
  RLH_REQUIRE(!failed.load(), "rlh_ilut_factor: row %lld is empty", (long long)bad_row.load());
  if (stats) fprintf(stderr, "ilut: %lld rows, pivots per row %.1f, U candidates per row %.1f\n", (long long)n, (double)st_l.load() / n, (double)st_u.load() / n);
  // rows -> CSR
  for (int64_t i = 0; i < n; ++i) {
    f->lptr[(size_t)i + 1] = f->lptr[(size_t)i] + lrow[(size_t)i].cnt;
    f->uptr[(size_t)i + 1] = f->uptr[(size_t)i] + urow[(size_t)i].cnt;
  }
  f->lidx.resize((size_t)f->lptr[(size_t)n]);
  f->uidx.resize((size_t)f->uptr[(size_t)n]);
  f->lval.resize((size_t)f->lptr[(size_t)n] * sizeof(S));
  f->uval.resize((size_t)f->uptr[(size_t)n] * sizeof(S));
  S *lval = (S *)f->lval.data(), *uval = (S *)f->uval.data();
  auto gather = [&](int64_t r0, int64_t r1) {
    for (int64_t i = r0; i < r1; ++i) {
      const IlutRow<S> &l = lrow[(size_t)i], &u = urow[(size_t)i];
      std::copy(l.idx, l.idx + l.cnt, f->lidx.begin() + f->lptr[(size_t)i]);
      std::copy(l.val, l.val + l.cnt, lval + f->lptr[(size_t)i]);
      std::copy(u.idx, u.idx + u.cnt, f->uidx.begin() + f->uptr[(size_t)i]);
      std::copy(u.val, u.val + u.cnt, uval + f->uptr[(size_t)i]);
    }
  };
  host_parallel(nthreads, [&](int t, int nt) { gather(n * t / nt, n * (t + 1) / nt); });
  return 0;
}

// ------------------------------------------------------------------ device side
//
// One PERSISTENT launch per chain.  The dependency levels are not kernel boundaries
// any more: every row waits for exactly the rows it reads.
//
//  * Scratch.  The n x m block is cut into 16-byte pieces; the pieces of a row are dealt to at most 8
//    GROUPS (ppt pieces per row and group) and every group owns its own image of the block,
//    [slot][group][row][ppt] pieces: slot 0 the right-hand side, slot k + 1 the result of operator k.
//    The columns of a block are independent right-hand sides, so the groups never exchange a byte.
//  * Teams.  A group is solved by the workgroups of ONE XCD, whichever XCD claims it first (the
//    hardware id is read with s_getreg, claims are agent-scope atomics, a late or missing XCD costs
//    time, never correctness): the rows one workgroup finishes reach the others through that XCD's
//    own L2 -- plain 16-byte stores, L1-bypassing (sc1) 16-byte loads -- without a fence and without
//    a trip to the memory side of the fabric.
//  * Hand-off.  The data is its own flag: the result slots are pre-filled with one particular NaN
//    (every 32-bit word 0xFFFFDEAD), a consumer re-reads a piece until none of its 8-byte halves
//    carries that pattern, a producer never stores it (a result that happens to be this NaN is
//    rewritten as the canonical quiet NaN).  A piece is written once, by one lane, in one store.
//  * Queue.  The rows are cut into UNITS of at most 256 (row, lane) tasks inside one level -- enough lanes per row
//    that a lane holds at most four entries: ONE batch, whose indices and values are in registers before the first
//    poll -- listed in level order; workgroups take units from one counter per group (the next unit's number is
//    fetched while the current one is worked on).  A unit only ever waits for units taken earlier, which are held
//    by resident workgroups: no placement or residency assumption, no grid barrier, no deadlock.
//  * Polls.  A lane asks again only for the pieces it is still missing (every 16-byte gather is an L2 request of
//    its own and a CU issues one per clock), with an s_sleep between rounds.  Measured on the box
//    (profiles/r03_trsv_*.txt): gating the polls behind per-unit done words -- so that only the units next to the
//    front poll -- was slower at every window size than letting every resident unit poll from the moment it is
//    taken; what pays is the NUMBER of units in flight, so the kernel is kept small in registers.
//  * Every spin is bounded (wall clock, and an error word another workgroup may have raised): a
//    protocol failure ends the launch with NaNs in the result and rlh_sync / the next call report it.

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float  neg_of(float a)  { return -a; }
__device__ __forceinline__ double neg_of(double a) { return -a; }
__device__ __forceinline__ c32 neg_of(c32 a) { return c32{-a.re, -a.im}; }
__device__ __forceinline__ c64 neg_of(c64 a) { return c64{-a.re, -a.im}; }
__device__ __forceinline__ float  one_of(float)  { return 1.f; }
__device__ __forceinline__ double one_of(double) { return 1.0; }
__device__ __forceinline__ c32 one_of(c32) { return c32{1.f, 0.f}; }
__device__ __forceinline__ c64 one_of(c64) { return c64{1.0, 0.0}; }
template <typename T> struct Halves64 { static constexpr bool value = false; };   // 8-byte real parts: the pattern sits in the high word
template <> struct Halves64<double> { static constexpr bool value = true; };
template <> struct Halves64<c64> { static constexpr bool value = true; };


constexpr unsigned kSentinel = 0xFFFFDEADu;     // as a float, and as the high word of a double: a NaN
constexpr int kMaxXcd = 8;
constexpr int kCtrlHead = 32;                   // words: [0] next group, [1] error, [2] census
constexpr unsigned long long kSpinTicks = 400000000ull;      // 4 s of the 100 MHz wall clock

struct alignas(16) TrsvUnit {                    // 16 bytes
  int32_t p0; uint16_t nr; uint8_t lg_lr, pad;
  int32_t near_unit;            // the last unit any x entry reads (-1: none)
  int32_t pad2;
};
struct alignas(16) TrsvRow { int64_t e0; int32_t nx, nrhs; };   // x entries, then rhs entries

struct TrsvOp {
  const TrsvUnit *units;
  const TrsvRow *rows;
  const int32_t *cols;
  const int32_t *prev_pos;                      // row -> position in the image this operator's rhs entries read (null: identity)
  const void *vals;
  int32_t unit0, nunits;
};

struct TrsvArgs {
  TrsvOp op[8];
  int nops, total_units, ngroups, ppt, pl, nap;
  int64_t n8;
  u32x4 *scratch;
  unsigned *ctrl;
  unsigned *err_host;                           // mapped host word (Context::async_err)
  unsigned long long spin_ticks;                // 100 MHz wall-clock ticks a launch may spend before it gives up
  unsigned long long *trace;                    // diagnostics (RLH_SPTRSV_TRACE): 16 words per unit of group 0, or null
};

template <typename T> struct PieceOf {
  static constexpr int EPL = 16 / (int)sizeof(T);
  union U { u32x4 w; T e[EPL]; };
};

// a piece some lane has stored (true) or the pre-filled pattern (false); 8-byte halves are judged separately
template <typename T> __device__ __forceinline__ bool piece_ready(u32x4 w) {
  if constexpr (Halves64<T>::value) return w.y != kSentinel && w.w != kSentinel;              // double, complex double
  else return w.x != kSentinel && w.y != kSentinel && w.z != kSentinel && w.w != kSentinel;   // float, complex float
}
template <typename T> __device__ __forceinline__ u32x4 piece_clean(u32x4 w) {
  if constexpr (Halves64<T>::value) {
    if (w.y == kSentinel) { w.x = 0u; w.y = 0x7FF80000u; }
    if (w.w == kSentinel) { w.z = 0u; w.w = 0x7FF80000u; }
  } else {
    if (w.x == kSentinel) w.x = 0x7FC00000u;
    if (w.y == kSentinel) w.y = 0x7FC00000u;
    if (w.z == kSentinel) w.z = 0x7FC00000u;
    if (w.w == kSentinel) w.w = 0x7FC00000u;
  }
  return w;
}

// L1-bypassing 16-byte loads (the asm carries its own wait: hipcc does not count an asm load)
__device__ __forceinline__ u32x4 load_sc1(const u32x4 *p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void load4_sc1(const u32x4 *p0, const u32x4 *p1, const u32x4 *p2, const u32x4 *p3, u32x4 &x0,
                                          u32x4 &x1, u32x4 &x2, u32x4 &x3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\t"
      "global_load_dwordx4 %1, %5, off sc1\n\t"
      "global_load_dwordx4 %2, %6, off sc1\n\t"
      "global_load_dwordx4 %3, %7, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}

struct Watch {                                  // bounds every spin of one thread
  unsigned *ctrl, *err_host;
  unsigned long long t0, limit;
  unsigned spins;
  bool dead;
  __device__ __forceinline__ bool expired() {  // call once per failed poll
    if (dead) return true;
    if ((++spins & 255u) == 0) {
      if (__hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) dead = true;
      else if ((unsigned long long)wall_clock64() - t0 > limit) {
        __hip_atomic_store(ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        dead = true;
      }
    }
    return dead;
  }
  // a wait has ended with what it waited for: the limit bounds ONE wait without progress, not the whole launch (a launch
  // serialised by a profiler's counter collection, a GPU shared with another job or a very long chain may legitimately
  // run for longer than any fixed bound).  Only a thread that has spun long enough to have looked at the clock pays
  // for reading it again.
  __device__ __forceinline__ void progress() {
    if (spins > 255u) t0 = (unsigned long long)wall_clock64();
    spins = 0u;
  }
};

// which group does XCD `xcc` solve in its round `round`?  The first of its workgroups to ask takes the next
// unclaimed group for the XCD; -1: none left.
__device__ __forceinline__ int claim_group(unsigned *ctrl, unsigned xcc, int round, int ngroups, Watch &watch) {
  unsigned *slot = ctrl + kCtrlHead + xcc * (unsigned)(ngroups + 1) + (unsigned)round;
  unsigned v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v == 0) {
    unsigned expected = 0;
    if (__hip_atomic_compare_exchange_strong(slot, &expected, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      v = __hip_atomic_fetch_add(ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 2u;
      __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      v = 1;
    }
  }
  while (v == 1) {
    __builtin_amdgcn_s_sleep(2);
    if (watch.expired()) return -1;
    v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  watch.progress();
  const int g = (int)v - 2;
  return g < ngroups ? g : -1;
}

template <typename T> struct Batch { int32_t c[4]; T v[4]; unsigned rhs; };

// entries e, e + sl, e + 2 sl, e + 3 sl of a row (those beyond `end` repeat the first with value 0); bit u of `rhs`:
// entry u lies at or beyond end_x, i.e. it reads the right-hand side image (where the previous operator keeps that row:
// prev_pos), not the result image
template <typename T>
__device__ __forceinline__ Batch<T> load_batch(const int32_t *__restrict__ cols, const T *__restrict__ vals, int64_t e, int64_t end, int sl,
                                               int64_t end_x, const int32_t *__restrict__ prev_pos) {
  Batch<T> b;
  b.rhs = 0u;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t ee = e + (int64_t)u * sl;
    const bool in = ee < end;
    const int64_t ec = in ? ee : e;
    b.c[u] = cols[ec];
    b.v[u] = in ? neg_of(vals[ec]) : zero_of(T{});
    if (ec >= end_x) b.rhs |= 1u << u;
  }
  if (prev_pos && b.rhs) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (b.rhs & (1u << u)) b.c[u] = prev_pos[b.c[u]];
  }
  return b;
}

// L1-bypassing loads of those of four pieces whose bit is set in `want`, one wait (every load is an L2 request of its
// own -- a workgroup's gathers queue at its CU's one request per clock -- so neither padding slots nor pieces that
// have arrived are asked for again).  The four predicated loads AND their wait are ONE asm statement with the
// destinations as read-write operands: the compiler does not track loads issued from asm, so between a load in one
// statement and its wait in another it would be free to copy or spill a destination register -- reading it before the
// data has landed (piece_ready() would then judge stale data).  Lanes whose bit is clear keep what they held: the
// predication is the EXEC mask, saved and restored around each load.
__device__ __forceinline__ void loadm_sc1(unsigned want, const u32x4 *p0, const u32x4 *p1, const u32x4 *p2, const u32x4 *p3, u32x4 &x0,
                                          u32x4 &x1, u32x4 &x2, u32x4 &x3) {
  unsigned long long saved;
  unsigned bit;
  asm volatile(
      "v_and_b32 %5, 1, %6\n\t"
      "v_cmp_ne_u32 vcc, 0, %5\n\t"
      "s_and_saveexec_b64 %4, vcc\n\t"
      "global_load_dwordx4 %0, %7, off sc1\n\t"
      "s_mov_b64 exec, %4\n\t"
      "v_and_b32 %5, 2, %6\n\t"
      "v_cmp_ne_u32 vcc, 0, %5\n\t"
      "s_and_saveexec_b64 %4, vcc\n\t"
      "global_load_dwordx4 %1, %8, off sc1\n\t"
      "s_mov_b64 exec, %4\n\t"
      "v_and_b32 %5, 4, %6\n\t"
      "v_cmp_ne_u32 vcc, 0, %5\n\t"
      "s_and_saveexec_b64 %4, vcc\n\t"
      "global_load_dwordx4 %2, %9, off sc1\n\t"
      "s_mov_b64 exec, %4\n\t"
      "v_and_b32 %5, 8, %6\n\t"
      "v_cmp_ne_u32 vcc, 0, %5\n\t"
      "s_and_saveexec_b64 %4, vcc\n\t"
      "global_load_dwordx4 %3, %10, off sc1\n\t"
      "s_mov_b64 exec, %4\n\t"
      "s_waitcnt vmcnt(0)"
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&s"(saved), "=&v"(bit)
      : "v"(want), "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory", "vcc");
}

// acc -= sum over this lane's entries [e, end) (stride sl) of value * (X or, from end_x on, R)[column]; `cur` holds the
// first batch.  A piece that has not arrived is asked for again (and only it) until it has.
template <typename T>
__device__ __forceinline__ void accumulate(typename PieceOf<T>::U &acc, Batch<T> cur, int64_t e, int64_t end, int64_t end_x, int sl,
                                           const int32_t *__restrict__ cols, const T *__restrict__ vals, const u32x4 *X, const u32x4 *R,
                                           const int32_t *__restrict__ prev_pos, int ppt, int pp, int nap, Watch &watch) {
  constexpr int EPL = PieceOf<T>::EPL;
  while (e < end) {
    const int64_t en = e + 4 * (int64_t)sl;
    const int cnt = en <= end ? 4 : (int)((end - e + sl - 1) / sl);
    const u32x4 *p0 = ((cur.rhs & 1u) ? R : X) + (int64_t)cur.c[0] * ppt + pp, *p1 = ((cur.rhs & 2u) ? R : X) + (int64_t)cur.c[1] * ppt + pp;
    const u32x4 *p2 = ((cur.rhs & 4u) ? R : X) + (int64_t)cur.c[2] * ppt + pp, *p3 = ((cur.rhs & 8u) ? R : X) + (int64_t)cur.c[3] * ppt + pp;
    typename PieceOf<T>::U x0, x1, x2, x3;
    x0.w = x1.w = x2.w = x3.w = u32x4{0u, 0u, 0u, 0u};      // (read-write operands of loadm_sc1: a defined value)
    unsigned want = (1u << cnt) - 1u;
    for (;;) {
      loadm_sc1(want, p0, p1, p2, p3, x0.w, x1.w, x2.w, x3.w);
      if ((want & 1u) && piece_ready<T>(x0.w)) want &= ~1u;
      if ((want & 2u) && piece_ready<T>(x1.w)) want &= ~2u;
      if ((want & 4u) && piece_ready<T>(x2.w)) want &= ~4u;
      if ((want & 8u) && piece_ready<T>(x3.w)) want &= ~8u;
      if (want == 0u) { watch.progress(); break; }
      if (watch.expired()) break;
      if (nap == 1) __builtin_amdgcn_s_sleep(1); else if (nap == 2) __builtin_amdgcn_s_sleep(2); else if (nap >= 4) __builtin_amdgcn_s_sleep(4);
    }
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
      fma_acc(acc.e[q], cur.v[0], x0.e[q]);
      if (cnt > 1) fma_acc(acc.e[q], cur.v[1], x1.e[q]);
      if (cnt > 2) fma_acc(acc.e[q], cur.v[2], x2.e[q]);
      if (cnt > 3) fma_acc(acc.e[q], cur.v[3], x3.e[q]);
    }
    e = en;
    if (e < end) cur = load_batch<T>(cols, vals, e, end, sl, end_x, prev_pos);
  }
}

template <typename T>
__device__ __forceinline__ typename PieceOf<T>::U add_pieces(typename PieceOf<T>::U a, typename PieceOf<T>::U b) {
#pragma unroll
  for (int q = 0; q < PieceOf<T>::EPL; ++q) a.e[q] = add_of(a.e[q], b.e[q]);
  return a;
}
template <int CTRL, typename PU> __device__ __forceinline__ PU dpp_piece(PU a) {
  PU t;
  t.w.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a.w.x, CTRL, 0xF, 0xF, false);
  t.w.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a.w.y, CTRL, 0xF, 0xF, false);
  t.w.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a.w.z, CTRL, 0xF, 0xF, false);
  t.w.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a.w.w, CTRL, 0xF, 0xF, false);
  return t;
}

template <typename T>
__global__ __launch_bounds__(256) void trsv_pipeline_kernel(const TrsvArgs a) {
  constexpr int EPL = PieceOf<T>::EPL;
  using PU = typename PieceOf<T>::U;
  __shared__ int s_pick[2];
  __shared__ __attribute__((aligned(16))) u32x4 s_red[4 * 64];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  Watch watch{a.ctrl, a.err_host, (unsigned long long)wall_clock64(), a.spin_ticks, 0u, false};
  const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & (kMaxXcd - 1);   // HW_REG_XCC_ID
  unsigned *const queues = a.ctrl + kCtrlHead + kMaxXcd * (a.ngroups + 1);
  for (int round = 0; round <= a.ngroups; ++round) {
    if (tid == 0) s_pick[0] = claim_group(a.ctrl, xcc, round, a.ngroups, watch);
    __syncthreads();
    const int g = s_pick[0];
    __syncthreads();
    if (g < 0) break;
    unsigned *const qhead = queues + 32 * g;
    if (tid == 0) s_pick[0] = (int)__hip_atomic_fetch_add(qhead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int it = 0;; ++it) {
      const int U = s_pick[it & 1];
      if (U >= a.total_units || U < 0) break;
      // the NEXT unit is taken now: the atomic's round trip hides behind this unit (a workgroup works through its
      // units in ascending order, so the oldest unfinished unit is always somebody's current one)
      if (tid == 0) s_pick[(it + 1) & 1] = (int)__hip_atomic_fetch_add(qhead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int k = 0;
      while (k + 1 < a.nops && U >= a.op[k + 1].unit0) ++k;
      const TrsvOp &o = a.op[k];
      const TrsvUnit ud = o.units[U - o.unit0];
      unsigned long long *const stamp = (a.trace && g == 0 && tid == 0) ? a.trace + (int64_t)U * 16 : nullptr;
      if (stamp) stamp[0] = wall_clock64();
      const int lg = ud.lg_lr, lr = 1 << lg;                  // lanes per row
      const int PL = a.pl < lr ? a.pl : lr;                    // lanes across the pieces of a row
      const int sl = lr / PL;                                  // slices across its entries
      const int l = tid & (lr - 1);
      const int pl = l & (PL - 1), slice = l / PL;
      const int row_slot = tid >> lg;
      const bool live = row_slot < (int)ud.nr;
      const int64_t p = ud.p0 + (live ? row_slot : 0);
      const TrsvRow rd = o.rows[p];
      const int32_t *__restrict__ cols = o.cols;
      const T *__restrict__ vals = (const T *)o.vals;
      const int ppt = a.ppt;
      const u32x4 *rhs = a.scratch + ((int64_t)k * a.ngroups + g) * a.n8 * ppt;
      u32x4 *X = a.scratch + ((int64_t)(k + 1) * a.ngroups + g) * a.n8 * ppt;
      const int64_t e_first = rd.e0 + slice, end_x = rd.e0 + rd.nx, end_all = end_x + rd.nrhs;
      bool first = true;
      for (int pp0 = 0; pp0 < ppt; pp0 += PL) {               // (one trip unless a group holds more than 64 pieces of a row)
        const int pq = pp0 + pl;
        const bool work = live && pq < ppt;
        const int pp = pq < ppt ? pq : ppt - 1;
        PU acc;
#pragma unroll
        for (int q = 0; q < EPL; ++q) acc.e[q] = zero_of(T{});
        Batch<T> cur;
        if (work && e_first < end_all) cur = load_batch<T>(cols, vals, e_first, end_all, sl, end_x, o.prev_pos);
        const bool owner = work && slice == 0;
        if (owner) {
          // the line this row's result will be stored into, brought into the L2 now: a consumer's poll of a line the L2
          // holds only the freshly stored bytes of would go to memory for the rest of it
          const u32x4 warm = load_sc1(X + p * ppt + pp);
          asm volatile("" :: "v"(warm));
        }
        if (first && stamp) stamp[4] = wall_clock64();
        if (work) accumulate<T>(acc, cur, e_first, end_all, end_x, sl, cols, vals, X, rhs, o.prev_pos, ppt, pp, a.nap, watch);
        if (first && stamp) stamp[5] = wall_clock64();
        first = false;
        // sum of the slices of a row: lanes PL apart inside a wave, then (rows wider than a wave) through the LDS
        {
          const int top = lr < 64 ? lr : 64;
          int off = PL;
          if (PL == 1) {
            // adjacent lanes: the first four steps are data-parallel-primitive moves inside a row of 16 lanes (no LDS
            // crossbar round trip per step): pairs, quads (quad_perm), the two quads of a half (row_half_mirror),
            // the two halves (row_mirror) -- any pairing of distinct partial sums will do for a sum
            if (top > 1) { acc = add_pieces<T>(acc, dpp_piece<0xB1>(acc)); off = 2; }     // quad_perm [1,0,3,2]
            if (top > 2) { acc = add_pieces<T>(acc, dpp_piece<0x4E>(acc)); off = 4; }     // quad_perm [2,3,0,1]
            if (top > 4) { acc = add_pieces<T>(acc, dpp_piece<0x141>(acc)); off = 8; }    // row_half_mirror
            if (top > 8) { acc = add_pieces<T>(acc, dpp_piece<0x140>(acc)); off = 16; }   // row_mirror
          }
          for (; off < top; off <<= 1) {
            PU t;
            t.w.x = __shfl_xor(acc.w.x, off); t.w.y = __shfl_xor(acc.w.y, off);
            t.w.z = __shfl_xor(acc.w.z, off); t.w.w = __shfl_xor(acc.w.w, off);
            acc = add_pieces<T>(acc, t);
          }
        }
        if (lr > 64) {                                         // (workgroup-uniform)
          if (lane < PL) s_red[wave * 64 + lane] = acc.w;
          __syncthreads();
          if (l < PL) {
            PU tot;
            tot.w = s_red[wave * 64 + lane];
            for (int wv = 1; wv < (lr >> 6); ++wv) {
              PU t;
              t.w = s_red[(wave + wv) * 64 + lane];
#pragma unroll
              for (int q = 0; q < EPL; ++q) tot.e[q] = add_of(tot.e[q], t.e[q]);
            }
            acc = tot;
          }
          __syncthreads();
        }
        if (stamp && pp0 == 0) stamp[8] = wall_clock64();
        if (owner) X[p * ppt + pp] = piece_clean<T>(acc.w);
        if (stamp && pp0 == 0) stamp[9] = wall_clock64();
      }
      // the next pick is in the LDS (the stores need not have landed: nobody is told, consumers poll the pieces)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (stamp) stamp[10] = wall_clock64();
      __builtin_amdgcn_s_barrier();
      if (stamp) { stamp[6] = wall_clock64(); stamp[7] = ((unsigned long long)(unsigned)(o.unit0 + ud.near_unit) << 32) | (xcc << 24) | (unsigned)blockIdx.x; }
    }
    __syncthreads();
  }
}

// right-hand side into slot 0 (row r of the scratch = row perm[r] of B, columns >= m zero), the pattern into every result slot
template <typename T>
__global__ __launch_bounds__(256) void trsv_scatter_in(const T *__restrict__ B, int64_t ldb, const int64_t *__restrict__ perm,
                                                       u32x4 *__restrict__ scratch, int64_t n, int64_t n8, int m, int ngroups, int ppt,
                                                       int nslots) {
  constexpr int EPL = PieceOf<T>::EPL;
  const int64_t total = n8 * ngroups * ppt;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int64_t r = t % n8;                                    // rows fastest: coalesced reads of B's columns
  const int64_t gq = t / n8;
  const int q = (int)(gq % ppt), g = (int)(gq / ppt);
  const int64_t dst = ((int64_t)g * n8 + r) * ppt + q;
  typename PieceOf<T>::U out;
  if (r < n) {
    const int64_t src = perm ? perm[r] : r;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int v = (g * ppt + q) * EPL + e;
      out.e[e] = v < m ? B[src + (int64_t)v * ldb] : zero_of(T{});
    }
    out.w = piece_clean<T>(out.w);
  } else {
    out.w = u32x4{0u, 0u, 0u, 0u};
  }
  scratch[dst] = out.w;
  const u32x4 pattern = {kSentinel, kSentinel, kSentinel, kSentinel};
  const int64_t slot = (int64_t)ngroups * n8 * ppt;
  for (int k = 1; k < nslots; ++k) scratch[k * slot + dst] = pattern;
}

template <typename T>
__global__ __launch_bounds__(256) void trsv_gather_out(const u32x4 *__restrict__ res, const int32_t *__restrict__ pos, const int64_t *__restrict__ perm,
                                                       T *__restrict__ X, int64_t ldx, int64_t n, int64_t n8, int m, int ngroups, int ppt) {
  constexpr int EPL = PieceOf<T>::EPL;
  const int64_t total = n * ngroups * ppt;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int64_t r = t % n;
  const int64_t gq = t / n;
  const int q = (int)(gq % ppt), g = (int)(gq / ppt);
  typename PieceOf<T>::U in;
  in.w = res[((int64_t)g * n8 + pos[r]) * ppt + q];                // (the last operator keeps row r at its position)
  const int64_t dst = perm ? perm[r] : r;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int v = (g * ppt + q) * EPL + e;
    if (v < m) X[dst + (int64_t)v * ldx] = in.e[e];
  }
}

// ------------------------------------------------------------------ host side: plans and the launch
static int pow2_ceil(int64_t v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Units and wait words of one operator for PL lanes across the pieces of a row.
static int build_plan(rlh_sptrsv *t, int pl, TrsvPlan *plan) {
  const int64_t n = t->n;
  const int64_t nlev = (int64_t)t->lev_off.size() - 1;
  std::vector<TrsvUnit> units;
  std::vector<int32_t> unit_of((size_t)n);
  for (int64_t lev = 0; lev < nlev; ++lev) {
    const int64_t p0 = t->lev_off[(size_t)lev], p1 = t->lev_off[(size_t)lev + 1];
    const int64_t nrows = p1 - p0;
    if (nrows == 0) continue;
    int64_t longest = 0;
    for (int64_t q = p0; q < p1; ++q) longest = std::max(longest, t->rowptr_h[(size_t)q + 1] - t->rowptr_h[(size_t)q]);
    // at most four entries per lane -- ONE batch, whose column indices and values are in registers before the
    // unit starts to wait: a second batch would fetch them on the critical path
    int sl = pow2_ceil((longest + 3) / 4);
    if (sl > 256 / pl) sl = 256 / pl;
    if (sl < 1) sl = 1;
    const int lr = pl * sl;
    const int64_t cap = 256 / lr;                              // rows per unit
    const int64_t nu = (nrows + cap - 1) / cap;
    int lg = 0;
    while ((1 << lg) < lr) ++lg;
    int64_t p = p0;
    for (int64_t u = 0; u < nu; ++u) {                         // the rows of a level dealt evenly to its units
      const int64_t cnt = nrows / nu + (u < nrows % nu ? 1 : 0);
      TrsvUnit d;
      memset(&d, 0, sizeof(d));
      d.p0 = (int32_t)p; d.nr = (uint16_t)cnt; d.lg_lr = (uint8_t)lg; d.near_unit = -1;
      for (int64_t q = p; q < p + cnt; ++q) unit_of[(size_t)q] = (int32_t)units.size();
      units.push_back(d);
      p += cnt;
    }
  }
  RLH_REQUIRE(units.size() < ((size_t)1 << 30), "rlh_sptrsv: too many units");
  std::vector<TrsvRow> rows((size_t)std::max<int64_t>(n, 1));
  for (int64_t p = 0; p < n; ++p) {
    const int64_t e0 = t->rowptr_h[(size_t)p], e1 = t->rowptr_h[(size_t)p + 1], ex = e0 + t->nx_h[(size_t)p];
    const int32_t u = unit_of[(size_t)p];
    TrsvUnit &d = units[(size_t)u];
    if (ex > e0) d.near_unit = std::max(d.near_unit, unit_of[(size_t)t->dep_pos_h[(size_t)(ex - 1)]]);
    rows[(size_t)p] = TrsvRow{e0, (int32_t)(ex - e0), (int32_t)(e1 - ex)};
  }
  plan->pl = pl;
  plan->nunits = (int64_t)units.size();
  RLH_HIP(hipMalloc((void **)&plan->units, std::max<size_t>(units.size(), 1) * sizeof(TrsvUnit)));
  RLH_HIP(hipMalloc((void **)&plan->rows, rows.size() * sizeof(TrsvRow)));
  if (!units.empty()) RLH_HIP(hipMemcpy(plan->units, units.data(), units.size() * sizeof(TrsvUnit), hipMemcpyHostToDevice));
  RLH_HIP(hipMemcpy(plan->rows, rows.data(), rows.size() * sizeof(TrsvRow), hipMemcpyHostToDevice));
  return 0;
}

static void free_plan(TrsvPlan *p) {
  if (p->units) (void)hipFree(p->units);
  if (p->rows) (void)hipFree(p->rows);
  p->units = nullptr; p->rows = nullptr; p->pl = 0; p->nunits = 0;
}

static int plan_for(rlh_sptrsv *t, int pl, TrsvPlan **out) {
  for (int i = 0; i < 2; ++i)
    if (t->plan[i].pl == pl) { t->plan_lru = i; *out = &t->plan[i]; return 0; }
  const int victim = t->plan[0].pl == 0 ? 0 : (t->plan[1].pl == 0 ? 1 : 1 - t->plan_lru);
  if (t->plan[victim].pl) {
    RLH_HIP(hipStreamSynchronize(ctx().stream));
    free_plan(&t->plan[victim]);
  }
  if (int rc = build_plan(t, pl, &t->plan[victim])) { free_plan(&t->plan[victim]); return rc; }
  t->plan_lru = victim;
  *out = &t->plan[victim];
  return 0;
}

template <int DT>
static int solve_chain_impl(int nops, rlh_sptrsv *const *ops, const int64_t *perm_in, const int64_t *perm_out, int64_t m,
                            const void *B_, int64_t ldb, void *X_, int64_t ldx) {
  using T = typename DType<DT>::T;
  constexpr int EPL = 16 / (int)sizeof(T);
  Context &c = ctx();
  if (int rc = check_async_error()) return rc;
  rlh_sptrsv *head = ops[0];
  const int64_t n = head->n;
  const int64_t n8 = (n + 7) & ~(int64_t)7;
  const int64_t pieces = (m + EPL - 1) / EPL;
  const int ngroups = (int)std::min<int64_t>(pieces, kMaxXcd);
  const int ppt = (int)((pieces + ngroups - 1) / ngroups);
  const int pl = std::min(pow2_ceil(ppt), 64);
  TrsvArgs a;
  memset(&a, 0, sizeof(a));
  int64_t total_units = 0;
  for (int i = 0; i < nops; ++i) {
    TrsvPlan *plan = nullptr;
    if (int rc = plan_for(ops[i], pl, &plan)) return rc;
    a.op[i].units = plan->units; a.op[i].rows = plan->rows; a.op[i].cols = ops[i]->cols;
    a.op[i].prev_pos = i > 0 ? ops[i - 1]->pos_of : nullptr;
    a.op[i].vals = ops[i]->vals;
    a.op[i].unit0 = (int32_t)total_units; a.op[i].nunits = (int32_t)plan->nunits;
    total_units += plan->nunits;
  }
  RLH_REQUIRE(total_units < ((int64_t)1 << 30), "rlh_sptrsv_solve_chain: too many units");
  const int64_t ctrl_words = kCtrlHead + kMaxXcd * (ngroups + 1) + 32 * (int64_t)ngroups;
  const int64_t ctrl_bytes = ((ctrl_words * 4 + 15) / 16) * 16;
  const int64_t slot_pieces = (int64_t)ngroups * n8 * ppt;
  const int64_t scratch_bytes = (int64_t)(nops + 1) * slot_pieces * 16;
  const int64_t need = ctrl_bytes + scratch_bytes;
  if (head->work_bytes < need) {
    RLH_HIP(hipStreamSynchronize(c.stream));
    if (head->work) RLH_HIP(hipFree(head->work));
    head->work = nullptr; head->work_bytes = 0;
    RLH_HIP(hipMalloc(&head->work, (size_t)need));
    head->work_bytes = need;
  }
  a.nops = nops; a.total_units = (int)total_units; a.ngroups = ngroups; a.ppt = ppt; a.pl = pl;
  a.n8 = n8;
  a.ctrl = (unsigned *)head->work;                            // (the control words open the allocation: zeroed per call)
  a.scratch = (u32x4 *)((char *)head->work + ctrl_bytes);
  a.err_host = c.async_err_d;
  a.spin_ticks = kSpinTicks;
  { const char *e = getenv("RLH_SPTRSV_SPIN_TICKS"); if (e && *e && atoll(e) > 0) a.spin_ticks = (unsigned long long)atoll(e); }   // (tests)
  { const char *e = getenv("RLH_SPTRSV_NAP"); a.nap = (e && *e) ? atoi(e) : 1; }      // s_sleep between poll rounds (tunable)
  const char *trace_path = getenv("RLH_SPTRSV_TRACE");      // diagnostics: per-unit stamps of group 0 to this file (synchronises)
  if (trace_path && *trace_path) {
    RLH_HIP(hipMalloc((void **)&a.trace, (size_t)total_units * 128));
    RLH_HIP(hipMemsetAsync(a.trace, 0, (size_t)total_units * 128, c.stream));
  }
  RLH_HIP(hipMemsetAsync(head->work, 0, (size_t)ctrl_bytes, c.stream));
  {
    const int64_t nb = (slot_pieces + 255) / 256;
    RLH_REQUIRE(nb < ((int64_t)1 << 31), "rlh_sptrsv_solve_chain: block too large");
    hipLaunchKernelGGL((trsv_scatter_in<T>), dim3((unsigned)nb), dim3(256), 0, c.stream, (const T *)B_, ldb, perm_in, a.scratch, n, n8,
                       (int)m, ngroups, ppt, nops + 1);
    RLH_HIP(hipGetLastError());
  }
  {
    int occ = 0;                                               // resident workgroups per CU of this kernel
    RLH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, trsv_pipeline_kernel<T>, 256, 0));
    // (round 4's sweep of 1 / 2 / 3 / 4 resident workgroups per CU, one box: ILUT of lap3d 100^3 4.24 / 2.84 / 2.68 / 2.72 ms, config-3
    // surrogate 5.39 / 3.13 / 3.12 / 3.43, lap3d 160^3 - / - / 7.42 / 6.68, SuperLU factors of lap3d 30^3 - / - / 2.06 / 1.96: no
    // setting wins everywhere and nothing separates the two camps beforehand, so what the registers allow (4) stays;
    // the pause between poll rounds, 1 / 2 / 4, changes nothing)
    int per_cu = occ < 1 ? 1 : (occ > 8 ? 8 : occ);
    const char *e = getenv("RLH_SPTRSV_WG_PER_CU");            // tunable
    if (e && *e && atoi(e) > 0) per_cu = atoi(e);
    int64_t grid = (int64_t)c.num_cu * per_cu;
    const int64_t most = total_units * ngroups + kMaxXcd;      // (no more workgroups than units to take)
    if (grid > most) grid = most;
    if (grid < kMaxXcd) grid = kMaxXcd;
    hipLaunchKernelGGL((trsv_pipeline_kernel<T>), dim3((unsigned)grid), dim3(256), 0, c.stream, a);
    RLH_HIP(hipGetLastError());
  }
  {
    const int64_t nb = (n * ngroups * ppt + 255) / 256;
    hipLaunchKernelGGL((trsv_gather_out<T>), dim3((unsigned)nb), dim3(256), 0, c.stream,
                       (const u32x4 *)(a.scratch + (int64_t)nops * slot_pieces), ops[nops - 1]->pos_of, perm_out, (T *)X_, ldx, n, n8, (int)m,
                       ngroups, ppt);
    RLH_HIP(hipGetLastError());
  }
  if (a.trace) {
    std::vector<unsigned long long> h((size_t)total_units * 16);
    RLH_HIP(hipStreamSynchronize(c.stream));
    RLH_HIP(hipMemcpy(h.data(), a.trace, h.size() * 8, hipMemcpyDeviceToHost));
    RLH_HIP(hipFree(a.trace));
    if (FILE *f = fopen(trace_path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
  }
  return 0;
}

// host arithmetic of the block transform: double / complex double whatever the factor's type
template <int DT> struct HostOf { using H = double; };
template <> struct HostOf<RLH_C> { using H = std::complex<double>; };
template <> struct HostOf<RLH_Z> { using H = std::complex<double>; };
static inline double to_h(float v) { return v; }
static inline double to_h(double v) { return v; }
static inline std::complex<double> to_h(c32 v) { return {v.re, v.im}; }
static inline std::complex<double> to_h(c64 v) { return {v.re, v.im}; }
static inline void from_h(double h, float &v) { v = (float)h; }
static inline void from_h(double h, double &v) { v = h; }
static inline void from_h(const std::complex<double> &h, c32 &v) { v = c32{(float)h.real(), (float)h.imag()}; }
static inline void from_h(const std::complex<double> &h, c64 &v) { v = c64{h.real(), h.imag()}; }

// rows per diagonal block of the transform below (RLH_SPTRSV_BLOCK: 1 switches it off)
static int block_limit() {
  const char *e = getenv("RLH_SPTRSV_BLOCK");
  int b = (e && *e) ? atoi(e) : 8;
  return b < 1 ? 1 : (b > 16 ? 16 : b);
}

// Turns the triangular factor T into the operator the device applies.
//
// BLOCK TRANSFORM.  A chain of consecutive rows that each read the one before (the degrees of freedom of a
// finite-element node, the rows of a supernode of a direct factor) is one dependency level per row.  With D the
// (small, triangular) diagonal block of such a chain, T x = b is D^-1 T x = D^-1 b, and D^-1 T has an identity
// diagonal block: the rows of the chain no longer read each other, at the price of every row reading the union
// of the chain's earlier off-block patterns (rows of one node or supernode share their pattern: 10-30 % more
// entries for 4-5 times fewer levels on the FE factors).  Chains are grown greedily in solve order, up to
// block_limit() rows, while the entry count stays within 30 % of the original.  Every row becomes
//     x_r = sum_k w_k rhs[r_k]  -  sum_j m_j x_j
// with the rhs entries (the row of D^-1; a single 1 / diagonal for an unblocked row) stored behind the x entries:
// the inverse diagonal is folded into the values and no row needs a division.
template <int DT>
static int sptrsv_build(rlh_sptrsv *t, const int64_t *indptr, const int32_t *indices, const void *values_) {
  using T = typename DType<DT>::T;
  using H = typename HostOf<DT>::H;
  const T *values = (const T *)values_;
  const int64_t n = t->n;
  const bool verbose = env_int("RLH_SPTRSV_VERBOSE", 0) != 0;      // wall time of the phases of the set-up on stderr
  auto clock0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!verbose) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "sptrsv set-up: %-28s %.3f s\n", what, std::chrono::duration<double>(now - clock0).count());
    clock0 = now;
  };
  std::vector<H> diag((size_t)n, H(1.0));
  int64_t nstrict = 0;
  for (int64_t i = 0; i < n; ++i) {
    bool have_diag = false;
    for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
      const int32_t j = indices[e];
      RLH_REQUIRE(j >= 0 && j < n, "rlh_sptrsv_create: column index out of range in row %lld", (long long)i);
      if (j == i) {
        RLH_REQUIRE(!t->unit, "rlh_sptrsv_create: a unit-diagonal factor must not store its diagonal (row %lld)", (long long)i);
        RLH_REQUIRE(!have_diag, "rlh_sptrsv_create: row %lld stores its diagonal twice", (long long)i);
        diag[(size_t)i] = to_h(values[e]);
        RLH_REQUIRE(std::abs(diag[(size_t)i]) > 0.0, "rlh_sptrsv_create: zero diagonal in row %lld", (long long)i);
        have_diag = true;
      } else {
        RLH_REQUIRE(t->lower ? j < i : j > i, "rlh_sptrsv_create: entry (%lld, %d) lies in the wrong triangle", (long long)i, j);
        ++nstrict;
      }
    }
    RLH_REQUIRE(t->unit || have_diag, "rlh_sptrsv_create: row %lld has no diagonal entry", (long long)i);
  }
  t->nnz = nstrict;
  auto row_at = [&](int64_t s) -> int64_t { return t->lower ? s : n - 1 - s; };      // solve order
  // ---- the transformed rows: x entries (column, m) then rhs entries (row, -w), per ORIGINAL row
  std::vector<int64_t> rp;
  std::vector<int32_t> nx;
  std::vector<int32_t> cols;
  std::vector<T> vals;
  std::vector<int32_t> level((size_t)n, 0);
  int32_t nlev = 0;
  auto transform = [&](const int B) {
    rp.assign((size_t)n + 1, 0);
    nx.assign((size_t)n, 0);
    cols.clear(); vals.clear();
    const double fmax = 0.3;
    // ---- the chains, in solve order (serial: a chain starts where the one before ends)
    std::vector<int64_t> chain_s;
    std::vector<int32_t> chain_k;
    {
      std::vector<int64_t> seen((size_t)n, -1), member((size_t)n, -1);
      int64_t s = 0, nb = 0;
      while (s < n) {
        const int64_t r0 = row_at(s);
        member[(size_t)r0] = nb;
        int64_t acc = 0, old_total = 0, new_total = 0;
        for (int64_t e = indptr[r0]; e < indptr[r0 + 1]; ++e) {
          const int32_t j = indices[e];
          if (j == r0) continue;
          if (seen[(size_t)j] != nb) { seen[(size_t)j] = nb; ++acc; }
          ++old_total;
        }
        new_total = acc + 1;
        int k = 1;
        while (k < B && s + k < n) {
          const int64_t rj = row_at(s + k), rprev = row_at(s + k - 1);
          bool chained = false;
          int64_t fresh = 0, cnt = 0;
          for (int64_t e = indptr[rj]; e < indptr[rj + 1]; ++e) {
            const int32_t j = indices[e];
            if (j == rj) continue;
            ++cnt;
            if (j == rprev) chained = true;
            if (member[(size_t)j] != nb && seen[(size_t)j] != nb) ++fresh;
          }
          if (!chained) break;
          const int64_t old2 = old_total + cnt, new2 = new_total + acc + fresh + (k + 1);
          if ((double)new2 > (1.0 + fmax) * (double)old2 + 8.0) break;
          for (int64_t e = indptr[rj]; e < indptr[rj + 1]; ++e) {
            const int32_t j = indices[e];
            if (j != rj && member[(size_t)j] != nb) seen[(size_t)j] = nb;
          }
          member[(size_t)rj] = nb;
          acc += fresh; old_total = old2; new_total = new2;
          ++k;
        }
        chain_s.push_back(s);
        chain_k.push_back(k);
        s += k;
        ++nb;
      }
      t->nblocks = nb;
    }
    // ---- the rows of every chain: W = D^-1 times the off-block parts.  The chains are independent of each other:
    // dealt to the host threads (every thread with marks of its own, as many threads as a gigabyte of marks allows);
    // a thread parks the rows it produces in a buffer of its own, the rows are copied out in row order afterwards --
    // the result does not depend on the thread count.
    const int64_t nchains = (int64_t)chain_s.size();
    const int want = nstrict > 2000000 ? (int)std::max<int64_t>(1, std::min<int64_t>(16, (int64_t)(1e9 / (24.0 * (double)(n + 1))))) : 1;
    std::vector<std::vector<int32_t>> pcs((size_t)std::max(want, 1));
    std::vector<std::vector<T>> pvs((size_t)std::max(want, 1));
    std::vector<int64_t> row_start((size_t)n, 0), row_len((size_t)n, 0);
    std::vector<int8_t> row_buf((size_t)n, 0);
    std::atomic<int64_t> next_chain(0);
    host_parallel(want, [&](int tid, int) {
      std::vector<int64_t> mark((size_t)n, -1);
      std::vector<H> wv((size_t)n, H(0.0));
      std::vector<int32_t> touched;
      std::vector<H> D, W;
      std::vector<int32_t> &pc = pcs[(size_t)tid];
      std::vector<T> &pv = pvs[(size_t)tid];
      for (;;) {
        const int64_t c0 = next_chain.fetch_add(256);
        if (c0 >= nchains) break;
        for (int64_t c = c0; c < std::min(nchains, c0 + 256); ++c) {
          const int64_t s = chain_s[(size_t)c];
          const int k = chain_k[(size_t)c];
          auto local_of = [&](int32_t j) -> int64_t {        // index of row j in this chain, or -1
            const int64_t q = (t->lower ? (int64_t)j : n - 1 - (int64_t)j) - s;
            return (q >= 0 && q < k) ? q : -1;
          };
          // D (k x k, solve order) and W = D^-1
          D.assign((size_t)k * k, H(0.0));
          W.assign((size_t)k * k, H(0.0));
          for (int i = 0; i < k; ++i) {
            const int64_t ri = row_at(s + i);
            D[(size_t)i * k + i] = diag[(size_t)ri];
            for (int64_t e = indptr[ri]; e < indptr[ri + 1]; ++e) {
              const int32_t j = indices[e];
              const int64_t q = j != ri ? local_of(j) : -1;
              if (q >= 0) D[(size_t)i * k + q] = D[(size_t)i * k + q] + to_h(values[e]);
            }
          }
          for (int cc = 0; cc < k; ++cc)                          // column cc of the inverse by forward substitution
            for (int i = cc; i < k; ++i) {
              H v = (i == cc) ? H(1.0) : H(0.0);
              for (int l = cc; l < i; ++l) v -= D[(size_t)i * k + l] * W[(size_t)l * k + cc];
              W[(size_t)i * k + cc] = v / D[(size_t)i * k + i];
            }
          for (int i = 0; i < k; ++i) {
            const int64_t ri = row_at(s + i);
            touched.clear();
            for (int l = 0; l <= i; ++l) {
              const H w = W[(size_t)i * k + l];
              if (w == H(0.0)) continue;
              const int64_t rl = row_at(s + l);
              for (int64_t e = indptr[rl]; e < indptr[rl + 1]; ++e) {
                const int32_t j = indices[e];
                if (j == rl || local_of(j) >= 0) continue;
                if (mark[(size_t)j] != ri) { mark[(size_t)j] = ri; wv[(size_t)j] = H(0.0); touched.push_back(j); }
                wv[(size_t)j] += w * to_h(values[e]);
              }
            }
            row_buf[(size_t)ri] = (int8_t)tid;
            row_start[(size_t)ri] = (int64_t)pc.size();
            for (int32_t j : touched) {
              T v;
              from_h(wv[(size_t)j], v);
              pc.push_back(j); pv.push_back(v);
            }
            nx[(size_t)ri] = (int32_t)touched.size();
            for (int l = 0; l <= i; ++l) {
              const H w = W[(size_t)i * k + l];
              if (w == H(0.0) && l != i) continue;
              T v;
              from_h(-w, v);
              pc.push_back((int32_t)row_at(s + l)); pv.push_back(v);
            }
            row_len[(size_t)ri] = (int64_t)pc.size() - row_start[(size_t)ri];
          }
        }
      }
    });
    for (int64_t i = 0; i < n; ++i) rp[(size_t)i + 1] = rp[(size_t)i] + row_len[(size_t)i];
    cols.resize((size_t)rp[(size_t)n]);
    vals.resize((size_t)rp[(size_t)n]);
    host_parallel(want, [&](int tid, int nt) {
      for (int64_t i = n * tid / nt; i < n * (tid + 1) / nt; ++i) {
        const std::vector<int32_t> &pc = pcs[(size_t)row_buf[(size_t)i]];
        const std::vector<T> &pv = pvs[(size_t)row_buf[(size_t)i]];
        std::copy(pc.begin() + row_start[(size_t)i], pc.begin() + row_start[(size_t)i] + row_len[(size_t)i], cols.begin() + rp[(size_t)i]);
        std::copy(pv.begin() + row_start[(size_t)i], pv.begin() + row_start[(size_t)i] + row_len[(size_t)i], vals.begin() + rp[(size_t)i]);
      }
    });
    // dependency levels of the transformed rows (x entries only)
    nlev = 0;
    for (int64_t s = 0; s < n; ++s) {
      const int64_t i = row_at(s);
      int32_t l = 0;
      for (int64_t e = rp[(size_t)i]; e < rp[(size_t)i] + nx[(size_t)i]; ++e) l = std::max(l, level[(size_t)cols[(size_t)e]] + 1);
      level[(size_t)i] = l;
      nlev = std::max(nlev, l + 1);
    }
  };
  // what the transform is for is fewer levels: where it removes less than a tenth of them the extra entries are not worth
  // it (measured with the final kernel: ILUT factors of lap3d 100^3, 1 820 -> 1 417 levels, 2.74 -> 2.54 ms; lap3d 64^3 1.61 ->
  // 1.30 ms; the FE and SuperLU factors, 5-7 times fewer levels, 3-4 times faster)
  {
    int32_t plain = 0;
    for (int64_t s = 0; s < n; ++s) {
      const int64_t i = row_at(s);
      int32_t l = 0;
      for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e)
        if (indices[e] != i) l = std::max(l, level[(size_t)indices[e]] + 1);
      level[(size_t)i] = l;
      plain = std::max(plain, l + 1);
    }
    t->levels_plain = plain;
    lap("checks, plain levels");
    const int B = block_limit();
    transform(B);
    lap("block transform");
    if (B > 1 && !getenv("RLH_SPTRSV_BLOCK") && (double)nlev > 0.9 * (double)plain) { transform(1); lap("transform undone"); }
  }
  t->lev_off.assign((size_t)nlev + 1, 0);
  for (int64_t i = 0; i < n; ++i) t->lev_off[(size_t)level[(size_t)i] + 1]++;
  for (int32_t l = 0; l < nlev; ++l) t->lev_off[(size_t)l + 1] += t->lev_off[(size_t)l];
  std::vector<int32_t> order((size_t)n);
  {
    std::vector<int64_t> next(t->lev_off.begin(), t->lev_off.end() - 1);
    for (int64_t i = 0; i < n; ++i) order[(size_t)next[(size_t)level[(size_t)i]]++] = (int32_t)i;   // ascending rows inside a level
  }
  std::vector<int32_t> pos_of((size_t)n);
  for (int64_t p = 0; p < n; ++p) pos_of[(size_t)order[(size_t)p]] = (int32_t)p;
  {
    // re-store the entries in level order -- the rows of a level are then one contiguous range of positions, their
    // entries one contiguous run -- every row's x entries sorted by the position of the row they read (oldest
    // dependency first), its rhs entries behind them
    std::vector<int64_t> rp2((size_t)n + 1, 0);
    std::vector<int32_t> cols2(cols.size());
    std::vector<T> vals2(vals.size());
    t->dep_pos_h.assign(cols.size(), -1);
    t->nx_h.resize((size_t)n);
    for (int64_t p = 0; p < n; ++p) {
      const int64_t r = order[(size_t)p];
      rp2[(size_t)p + 1] = rp2[(size_t)p] + (rp[(size_t)r + 1] - rp[(size_t)r]);
      t->nx_h[(size_t)p] = nx[(size_t)r];
    }
    // (the positions are independent of each other: dealt to the host threads in chunks; a direct factor has hundreds of
    // entries per row to sort)
    std::atomic<int64_t> next_chunk(0);
    const int64_t chunk = 512;
    host_parallel(cols.size() > 2000000 ? 16 : 1, [&](int, int) {
      std::vector<std::pair<int32_t, int64_t>> key;
      for (;;) {
        const int64_t p0 = next_chunk.fetch_add(chunk);
        if (p0 >= n) break;
        for (int64_t p = p0; p < std::min(n, p0 + chunk); ++p) {
          const int64_t r = order[(size_t)p];
          const int64_t e0 = rp[(size_t)r], ex = e0 + nx[(size_t)r], e1 = rp[(size_t)r + 1];
          int64_t w = rp2[(size_t)p];
          key.clear();
          for (int64_t e = e0; e < ex; ++e) key.push_back({pos_of[(size_t)cols[(size_t)e]], e});
          std::sort(key.begin(), key.end());
          for (auto &ke : key) {
            cols2[(size_t)w] = ke.first;
            vals2[(size_t)w] = vals[(size_t)ke.second];
            t->dep_pos_h[(size_t)w] = ke.first;
            ++w;
          }
          for (int64_t e = ex; e < e1; ++e, ++w) { cols2[(size_t)w] = cols[(size_t)e]; vals2[(size_t)w] = vals[(size_t)e]; }
        }
      }
    });
    rp.swap(rp2); cols.swap(cols2); vals.swap(vals2);
  }
  t->rowptr_h = rp;
  t->entries = (int64_t)cols.size();
  lap("level order");
  RLH_HIP(hipMalloc((void **)&t->cols, std::max<size_t>(cols.size(), 1) * sizeof(int32_t)));
  RLH_HIP(hipMalloc((void **)&t->vals, std::max<size_t>(vals.size(), 1) * sizeof(T)));
  if (!cols.empty()) {
    RLH_HIP(hipMemcpy(t->cols, cols.data(), cols.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    RLH_HIP(hipMemcpy(t->vals, vals.data(), vals.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  RLH_HIP(hipMalloc((void **)&t->pos_of, std::max<size_t>((size_t)n, 1) * sizeof(int32_t)));
  if (n > 0) RLH_HIP(hipMemcpy(t->pos_of, pos_of.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
  t->device_bytes = (int64_t)cols.size() * (4 + (int64_t)sizeof(T)) + n * 4 + n * 16;
  lap("upload");
  return 0;
}


// ------------------------------------------------------------------ block diagonal of an L D L^H factorisation
// X <- D^-1 X in place: the thread of the FIRST row of a 2 x 2 pivot writes both of its rows (it is the only one that
// reads them), the thread of the second row leaves.
template <typename T>
__global__ void __launch_bounds__(256) bdiag_solve_kernel(int64_t n, const T *__restrict__ coef, const int32_t *__restrict__ shift,
                                                          T *X, int64_t ldx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = shift[i];
  if (s < 0) return;
  T *x = X + (int64_t)blockIdx.y * ldx;
  const T xi = x[i];
  if (s == 0) {
    x[i] = mul_of(coef[2 * i], xi);
    return;
  }
  const T xj = x[i + 1];
  x[i] = add_of(mul_of(coef[2 * i], xi), mul_of(coef[2 * i + 1], xj));
  x[i + 1] = add_of(mul_of(coef[2 * i + 2], xj), mul_of(coef[2 * i + 3], xi));
}

template <int DT>
static int bdiag_solve_impl(int64_t n, const void *coef, const int32_t *shift, int64_t m, void *X, int64_t ldx) {
  using T = typename DType<DT>::T;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)m);
  hipLaunchKernelGGL(bdiag_solve_kernel<T>, grid, dim3(256), 0, ctx().stream, n, (const T *)coef, shift, (T *)X, ldx);
  RLH_HIP(hipGetLastError());
  return 0;
}

}  // namespace rlh

using namespace rlh;

extern "C" {

int rlh_ilut_factor(rlh_factors_t *out, int dtype, int64_t n, const int64_t *indptr, const int32_t *indices,
                    const void *values, double tol, int64_t maxfil) {
  RLH_REQUIRE(out != nullptr, "rlh_ilut_factor: null handle pointer");
  *out = nullptr;
  RLH_REQUIRE(dtype == RLH_D || dtype == RLH_Z, "rlh_ilut_factor: the factorisation runs in double / complex double");
  RLH_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && indptr && (indptr[n] == 0 || (indices && values)), "rlh_ilut_factor: bad matrix");
  RLH_REQUIRE(tol >= 0.0 && maxfil >= 0, "rlh_ilut_factor: tol and maxfil must not be negative");
  for (int64_t i = 0; i < n; ++i) {
    RLH_REQUIRE(indptr[i + 1] >= indptr[i], "rlh_ilut_factor: indptr decreases at row %lld", (long long)i);
    for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e)
      RLH_REQUIRE(indices[e] >= 0 && indices[e] < n, "rlh_ilut_factor: column index out of range in row %lld", (long long)i);
  }
  rlh_factors *f = new rlh_factors();
  f->dtype = dtype;
  int rc = dtype == RLH_D ? ilut_factor<double>(n, indptr, indices, (const double *)values, tol, maxfil, f)
                          : ilut_factor<std::complex<double>>(n, indptr, indices, (const std::complex<double> *)values, tol, maxfil, f);
  if (rc) { delete f; return rc; }
  *out = f;
  return 0;
}

int rlh_factors_nnz(rlh_factors_t f, int64_t *nnz_l, int64_t *nnz_u) {
  RLH_REQUIRE(f != nullptr, "rlh_factors_nnz: null handle");
  if (nnz_l) *nnz_l = (int64_t)f->lidx.size();
  if (nnz_u) *nnz_u = (int64_t)f->uidx.size();
  return 0;
}

int rlh_factors_get(rlh_factors_t f, int which, int64_t *indptr, int32_t *indices, void *values) {
  RLH_REQUIRE(f != nullptr && (which == 0 || which == 1), "rlh_factors_get: bad arguments");
  const std::vector<int64_t> &p = which ? f->uptr : f->lptr;
  const std::vector<int32_t> &i = which ? f->uidx : f->lidx;
  const std::vector<char> &v = which ? f->uval : f->lval;
  if (indptr) memcpy(indptr, p.data(), p.size() * sizeof(int64_t));
  if (indices && !i.empty()) memcpy(indices, i.data(), i.size() * sizeof(int32_t));
  if (values && !v.empty()) memcpy(values, v.data(), v.size());
  return 0;
}

int rlh_factors_destroy(rlh_factors_t f) {
  delete f;
  return 0;
}

int rlh_sptrsv_create(rlh_sptrsv_t *out, int dtype, int64_t n, const int64_t *indptr, const int32_t *indices,
                      const void *values, int lower, int unit_diag) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(out != nullptr, "rlh_sptrsv_create: null handle pointer");
  *out = nullptr;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_sptrsv_create: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && indptr && indptr[0] == 0, "rlh_sptrsv_create: bad matrix");
  RLH_REQUIRE(indptr[n] == 0 || (indices && values), "rlh_sptrsv_create: null indices/values");
  rlh_sptrsv *t = new rlh_sptrsv();
  t->dtype = dtype; t->n = n; t->nnz = 0; t->lower = lower ? 1 : 0; t->unit = unit_diag ? 1 : 0;
  t->cols = nullptr; t->vals = nullptr; t->pos_of = nullptr; t->plan_lru = 0; t->entries = 0; t->nblocks = 0; t->levels_plain = 0;
  t->device_bytes = 0; t->work = nullptr; t->work_bytes = 0;
  int rc = 1;
  switch (dtype) {
    case RLH_S: rc = sptrsv_build<RLH_S>(t, indptr, indices, values); break;
    case RLH_D: rc = sptrsv_build<RLH_D>(t, indptr, indices, values); break;
    case RLH_C: rc = sptrsv_build<RLH_C>(t, indptr, indices, values); break;
    case RLH_Z: rc = sptrsv_build<RLH_Z>(t, indptr, indices, values); break;
  }
  if (rc) { rlh_sptrsv_destroy(t); return rc; }
  *out = t;
  return 0;
}

int rlh_sptrsv_info(rlh_sptrsv_t t, int64_t *nnz, int64_t *levels, int64_t *device_bytes) {
  RLH_REQUIRE(t != nullptr, "rlh_sptrsv_info: null handle");
  if (nnz) *nnz = t->nnz;
  if (levels) *levels = (int64_t)t->lev_off.size() - 1;
  if (device_bytes) *device_bytes = t->device_bytes;
  return 0;
}

int rlh_sptrsv_solve_chain(int nops, const rlh_sptrsv_t *ops, const int64_t *d_perm_in, const int64_t *d_perm_out,
                           int64_t m, const void *B, int64_t ldb, void *X, int64_t ldx) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(nops >= 1 && nops <= 8 && ops != nullptr, "rlh_sptrsv_solve_chain: 1 to 8 operators");
  for (int i = 0; i < nops; ++i)
    RLH_REQUIRE(ops[i] != nullptr && ops[i]->dtype == ops[0]->dtype && ops[i]->n == ops[0]->n,
                "rlh_sptrsv_solve_chain: the operators must have one type and one size");
  RLH_REQUIRE(m >= 0 && m <= 65536, "rlh_sptrsv_solve_chain: bad block size");
  if (m == 0 || ops[0]->n == 0) return 0;
  RLH_REQUIRE(B && X && ldb >= ops[0]->n && ldx >= ops[0]->n, "rlh_sptrsv_solve_chain: bad block arguments");
  switch (ops[0]->dtype) {
    case RLH_S: return solve_chain_impl<RLH_S>(nops, ops, d_perm_in, d_perm_out, m, B, ldb, X, ldx);
    case RLH_D: return solve_chain_impl<RLH_D>(nops, ops, d_perm_in, d_perm_out, m, B, ldb, X, ldx);
    case RLH_C: return solve_chain_impl<RLH_C>(nops, ops, d_perm_in, d_perm_out, m, B, ldb, X, ldx);
    case RLH_Z: return solve_chain_impl<RLH_Z>(nops, ops, d_perm_in, d_perm_out, m, B, ldb, X, ldx);
  }
  return 1;
}

int rlh_sptrsv_destroy(rlh_sptrsv_t t) {
  if (!t) return 0;
  if (ctx().ready) {
    (void)hipStreamSynchronize(ctx().stream);
    free_plan(&t->plan[0]);
    free_plan(&t->plan[1]);
    if (t->cols) (void)hipFree(t->cols);
    if (t->vals) (void)hipFree(t->vals);
    if (t->pos_of) (void)hipFree(t->pos_of);
    if (t->work) (void)hipFree(t->work);
  }
  delete t;
  return 0;
}

int rlh_bdiag_solve(int dtype, int64_t n, const void *d_coef, const int32_t *d_shift, int64_t m, void *X, int64_t ldx) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_bdiag_solve: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && m >= 0 && m <= 65535, "rlh_bdiag_solve: bad sizes");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(d_coef && d_shift && X && ldx >= n, "rlh_bdiag_solve: bad arguments");
  switch (dtype) {
    case RLH_S: return bdiag_solve_impl<RLH_S>(n, d_coef, d_shift, m, X, ldx);
    case RLH_D: return bdiag_solve_impl<RLH_D>(n, d_coef, d_shift, m, X, ldx);
    case RLH_C: return bdiag_solve_impl<RLH_C>(n, d_coef, d_shift, m, X, ldx);
    case RLH_Z: return bdiag_solve_impl<RLH_Z>(n, d_coef, d_shift, m, X, ldx);
  }
  return 1;
}

}  // extern "C"

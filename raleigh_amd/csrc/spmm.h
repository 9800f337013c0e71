// Shared declarations of the sparse operator (K13): the handle, the host-side window analysis and
// the XCD-aware launch order used by both windowed layouts (spmm.hip, spmm_wide.inc).
#pragma once

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <thread>
#include <vector>

#include "common.h"

// Per-block record of the 1024-row windowed layout (spmm.hip, "Windowed ELL").
struct WellMeta {
  int64_t eoff;            // first entry slot of the block, in units of 1024 entries
  int32_t goff;            // first staging group of the block in `gsrc`
  int32_t width_ng;        // entry slots per row (low 8 bits) | staging groups per vector << 8
};

// Per-block record of the 256-row interleaved layout (spmm_wide.inc).
struct WideMeta {
  int64_t eoff;            // first entry chunk of the block (a chunk = 8 entry slots of each of the 256 rows)
  int32_t goff;            // first staging group of the block in `gsrc`
  int16_t nchunks;         // entry chunks per row
  int16_t ng;              // staging groups (kWideGroup columns each) per vector
};

struct WideSched {         // launch order built for one (resident workgroups, part, n_own) combination
  int slots, grid, part;
  int64_t n_own;
  int32_t *sched;          // device
  int64_t len;
};

struct rlh_csr {
  int dtype;
  int64_t n_rows, n_cols, nnz;
  int64_t n_slices;
  int64_t padded;          // stored entries incl. padding
  int64_t *slice_ptr;      // device, n_slices + 1 (entry offsets)
  int32_t *cols;           // device, padded
  void *vals;              // device, padded
  int64_t device_bytes;
  // 1024-row windowed layout (rows of at most 8 entries, real types)
  int64_t well_blocks;     // 0: not built
  int well_wmax;           // register slots per row the kernel is instantiated for
  WellMeta *well_meta;     // device, well_blocks
  int32_t *well_gsrc;      // device: first column of every 64-entry staging group
  uint16_t *well_idx;      // device: position of the entry's column in the staged image
  void *well_vals;         // device
  double well_ratio;       // staged elements per stored entry slot (diagnostic)
  int32_t *well_sched;     // device: block processed at launch position p (-1: none)
  int64_t well_sched_len;
  int well_grid;           // workgroups the schedule was laid out for
  // split of the blocks by "references a column >= n_own" (overlap of the halo exchange with the
  // interior rows, rlh_spmm_part): host-side order / largest referenced column per block, and the
  // two launch orders built on first use for a given n_own
  std::vector<int32_t> well_order, well_maxcol;
  int64_t well_split_at;   // n_own the split was built for (-1: none)
  int32_t *well_sched_part[2];
  int64_t well_sched_part_len[2];
  int well_grid_part[2];
  int well_inbounds;       // every staging group lies inside [0, n_cols)
  int well_aligned;        // every staging group starts on a multiple of 8 columns
  // the same layout over STACKS of row blocks (spmm.hip, "Stacked blocks"): a workgroup takes kStkR 1024-row blocks whose
  // windows overlap (for a 3-D stencil two blocks one grid plane apart) and stages the union of their windows once
  int64_t stk_blocks;      // stacks; 0: not built
  WellMeta *stk_meta;      // device, per stack (eoff counts slots: member r's 8 slots start at eoff + 8 r)
  int32_t *stk_member;     // device, kStkR pairs per stack: (first row, rows) of the stack's row blocks ((0, 0): none)
  int32_t *stk_gsrc;       // device
  uint16_t *stk_idx;       // device
  void *stk_vals;          // device
  // value dictionary of the stacks (constant-coefficient stencils repeat a few dozen rows of values: 27 for the 7-point
  // Laplacian): per row the index of its 8-slot value tuple, and the table of distinct tuples -- 4 bytes per row
  // instead of 8 values; nullptr when the rows have more than kStkMaxPatterns distinct tuples
  int32_t *stk_pat;        // device, [stack][member][row of the block]: value pattern | position pattern << 16
  uint16_t *stk_dtab;      // device, [stack][member][kStkMaxDeltas][8]: positions minus the row's index in its block, per
                           // position pattern of that member (a stencil's interior rows all share one); nullptr: none
  void *stk_table;         // device, [pattern][8 values]
  int64_t stk_npat;
  int32_t *stk_sched;      // device
  int64_t stk_sched_len;
  int stk_grid;
  int stk_gmax;           // largest number of staging groups of a stack
  // the stacks without / with halo columns (rlh_spmm_part), as well_split keeps them for the blocks
  std::vector<int32_t> stk_order, stk_maxcol;
  int64_t stk_split_at;
  int32_t *stk_sched_part[2];
  int64_t stk_sched_part_len[2];
  int stk_grid_part[2];
  int stk_self;            // slot 7 of every row of the stacks holds the position of the row's own column (fused Chebyshev step)
  int stk_aligned;         // every staging group of the stacks starts on a multiple of 8 columns
  int stk_overhang;        // columns (< 8) the last staging group reaches past n_cols: the callers' leading dimensions must cover them
  double stk_staged;       // staged elements per row and vector (diagnostic; the unstacked layout's: well_staged)
  double well_staged;
  // 256-row interleaved layout (any row length, any type)
  int64_t wide_blocks;     // 0: not built
  WideMeta *wide_meta;     // device
  int32_t *wide_gsrc;      // device
  void *wide_idx;          // device: 16-byte pieces of 8 positions, [chunk][row]
  void *wide_vals;         // device: 16-byte pieces of values, [chunk][piece][row]
  int wide_gmax;           // largest number of staging groups of a block
  int wide_k;              // rows per thread of the interleaved layout: 2 where consecutive rows share their column pattern (FE nodes)
  std::vector<int32_t> wide_order, wide_maxcol;
  std::vector<WideSched> wide_scheds;
};

namespace rlh {

template <typename T>
struct ChebArgs {
  const T *B; int64_t ldb;
  double cy, cp, cb;
};

// ---- element helpers shared by the SpMM kernels
__device__ __forceinline__ float nt_load(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ double nt_load(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ c32 nt_load(const c32 *p) {
  return c32{__builtin_nontemporal_load(&p->re), __builtin_nontemporal_load(&p->im)};
}
__device__ __forceinline__ c64 nt_load(const c64 *p) {
  return c64{__builtin_nontemporal_load(&p->re), __builtin_nontemporal_load(&p->im)};
}
__device__ __forceinline__ void nt_store(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void nt_store(double *p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void nt_store(c32 *p, c32 v) {
  __builtin_nontemporal_store(v.re, &p->re);
  __builtin_nontemporal_store(v.im, &p->im);
}
__device__ __forceinline__ void nt_store(c64 *p, c64 v) {
  __builtin_nontemporal_store(v.re, &p->re);
  __builtin_nontemporal_store(v.im, &p->im);
}
__device__ __forceinline__ float  scale_of(double s, float v)  { return (float)s * v; }
__device__ __forceinline__ double scale_of(double s, double v) { return s * v; }
__device__ __forceinline__ c32 scale_of(double s, c32 v) { return c32{(float)s * v.re, (float)s * v.im}; }
__device__ __forceinline__ c64 scale_of(double s, c64 v) { return c64{s * v.re, s * v.im}; }
__device__ __forceinline__ float  sub_of(float a, float b)   { return a - b; }
__device__ __forceinline__ double sub_of(double a, double b) { return a - b; }
__device__ __forceinline__ c32 sub_of(c32 a, c32 b) { return c32{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ c64 sub_of(c64 a, c64 b) { return c64{a.re - b.re, a.im - b.im}; }

// 16-byte pieces of a block of vectors: natural alignment of T on the global side (a group may
// start on any column), 16 bytes on the LDS side.
template <typename T, int EPL> struct VecU { T e[EPL]; };
template <typename T, int EPL> struct alignas(16) VecA { T e[EPL]; };
typedef unsigned rlh_u32x4e __attribute__((ext_vector_type(4)));
typedef unsigned rlh_u32x4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte piece at 4-byte alignment

// ---- host side: column windows of a block of rows
struct Win { int32_t start, len, off; };        // off: position of the window in the staged image

// The columns referenced by rows [r0, r1) merged into windows: holes of at most `gap` columns are
// bridged, a window starts on a multiple of 8 columns (16-byte staging loads: a piece is 2 to 8
// elements and must not lie across the own / halo boundary of a row shard) and is padded to whole
// staging groups of `gs` columns that stay inside the column range where the matrix is wide enough (a
// window at the far end is moved left instead of being padded past the last column).  Returns the
// number of staged columns (a multiple of `round_groups` * gs); `ws` is left with at least one window.
// (`cols`: the referenced columns in any order, duplicates allowed; sorted in place)
// (`nc` may be the column count rounded up to a multiple of 8 -- the stacked layout: a window at the far end then
// still starts on a multiple of 8 and hangs over the last column by at most 7, which the leading dimension of the
// caller's block must cover; the kernels that take it check that at launch)
static inline int32_t find_windows_of(std::vector<int32_t> &cols, int64_t nc, int gap, int gs, int round_groups,
                                      std::vector<Win> &ws) {
  std::sort(cols.begin(), cols.end());
  cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
  ws.clear();
  auto place = [&](int64_t first, int64_t last, int64_t &start, int64_t &padded) {
    start = first & ~(int64_t)7;
    padded = (last - start + gs) / gs * gs;
    if (start + padded > nc && nc >= padded) start = nc - padded;
  };
  for (size_t i = 0; i < cols.size();) {
    size_t k = i;
    while (k + 1 < cols.size() && cols[k + 1] - cols[k] <= gap) ++k;
    int64_t first = cols[i], last = cols[k], start, padded;
    for (;;) {
      place(first, last, start, padded);
      if (ws.empty() || start >= (int64_t)ws.back().start + ws.back().len) break;
      first = ws.back().start;               // moved onto its predecessor: one window for both
      ws.pop_back();
    }
    ws.push_back(Win{(int32_t)start, (int32_t)(last - start + 1), 0});
    if (ws.size() > 4096) break;               // hopeless: stop counting
    i = k + 1;
  }
  int32_t off = 0;
  for (Win &w : ws) {
    w.off = off;
    off += (w.len + gs - 1) / gs * gs;
    if (off > (1 << 24)) break;
  }
  if (ws.empty()) {                            // a block of empty rows still stages one group
    ws.push_back(Win{0, 1, 0});
    off = gs;
  }
  const int32_t unit = gs * round_groups;
  return (off + unit - 1) / unit * unit;
}

static inline int32_t find_windows(const int64_t *indptr, const int32_t *indices, int64_t r0, int64_t r1, int64_t nc,
                                   int gap, int gs, int round_groups, std::vector<Win> &ws) {
  std::vector<int32_t> cols(indices + indptr[r0], indices + indptr[r1]);
  return find_windows_of(cols, nc, gap, gs, round_groups, ws);
}

// position of column c in the staged image (last window starting at or before c)
static inline int32_t staged_position(const std::vector<Win> &ws, int32_t c) {
  size_t lo = 0, hi = ws.size();
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) / 2;
    if (ws[mid].start <= c) lo = mid; else hi = mid;
  }
  return ws[lo].off + (c - ws[lo].start);
}

// the same for the entries of one row in turn: columns ascend within a (canonical) row, so the window of an entry is the
// last one's or one of the next few -- a walk from `hint` (updated) instead of a binary search per entry; a column below the
// last one's window (an unsorted row) starts over
static inline int32_t staged_position_walk(const std::vector<Win> &ws, int32_t c, size_t &hint) {
  size_t w = hint;
  if (ws[w].start > c) return staged_position(ws, c);
  while (w + 1 < ws.size() && ws[w + 1].start <= c) ++w;
  hint = w;
  return ws[w].off + (c - ws[w].start);
}

// first column of every staging group of a block (surplus groups repeat the last)
static inline void fill_group_sources(const std::vector<Win> &ws, int32_t ngroups, int gs, int32_t *gsrc) {
  int32_t filled = 0, last = 0;
  for (const Win &w : ws)
    for (int32_t g = 0; g < (w.len + gs - 1) / gs; ++g, ++filled) gsrc[w.off / gs + g] = last = w.start + gs * g;
  for (; filled < ngroups; ++filled) gsrc[filled] = last;
}

// host buffer whose elements are NOT zeroed at allocation (std::vector would fill 0.6 GB serially before the host threads
// overwrite every element of it; left alone, the pages are first touched by the threads that fill them)
template <typename E> struct RawBuf {
  std::unique_ptr<E[]> p;
  size_t n;
  explicit RawBuf(size_t count) : p(new E[count > 0 ? count : 1]), n(count) {}
  E &operator[](size_t i) { return p[i]; }
  const E &operator[](size_t i) const { return p[i]; }
  E *data() { return p.get(); }
  const E *data() const { return p.get(); }
  size_t size() const { return n; }
};

// RLH_SPMM_VERBOSE=1: wall time of the phases of a layout build on stderr
struct PhaseClock {
  bool on;
  std::chrono::steady_clock::time_point t;
  PhaseClock() : on(env_int("RLH_SPMM_VERBOSE", 0) != 0), t(std::chrono::steady_clock::now()) {}
  void lap(const char *what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "csr layout: %-34s %.3f s\n", what, std::chrono::duration<double>(now - t).count());
    t = now;
  }
};

template <typename F>
static void parallel_blocks(int64_t nblocks, F fn) {
  unsigned nt = (unsigned)host_threads();
  if ((int64_t)nt > nblocks) nt = (unsigned)(nblocks > 0 ? nblocks : 1);
  std::atomic<int64_t> next(0);
  auto worker = [&]() {
    for (;;) {
      const int64_t b0 = next.fetch_add(64);
      if (b0 >= nblocks) break;
      const int64_t b1 = b0 + 64 < nblocks ? b0 + 64 : nblocks;
      for (int64_t b = b0; b < b1; ++b) fn(b);
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(worker);
  worker();
  for (auto &t : pool) t.join();
}

// Launch order of a windowed layout's blocks.  Workgroup w of the persistent grid processes
// sched[w], sched[w + grid], ...; workgroups w and w + 8 share an XCD and its L2 under the
// observed round-robin placement, so the positions {r * grid + 8 i + x, i < per_xcd} are blocks that
// XCD x works on at the same time.  A block's windows are mostly the own rows of other blocks
// (the z planes of a 3-D stencil are the rows of the blocks n_y n_x / rows-per-block further on): the
// schedule makes such blocks concurrent on one XCD, so that the window one of them stages is an
// L2 hit left by the block that owns those rows, instead of a second and third trip over the fabric
// (speed only: any order gives the same result).
// position r * grid + 8 i + x  <-  member i of group g = 8 r + x of the ordered block list
static inline void well_layout(const std::vector<int32_t> &order, int slots, std::vector<int32_t> &sched, int &grid) {
  const int xcds = 8;
  int per_xcd = slots / xcds;
  if (per_xcd < 1) per_xcd = 1;
  const int64_t nb = (int64_t)order.size();
  if (nb <= (int64_t)xcds * per_xcd) {
    grid = (int)nb;
    sched = order;
    return;
  }
  grid = xcds * per_xcd;
  const int64_t ngroups = (nb + per_xcd - 1) / per_xcd;
  const int64_t rounds = (ngroups + xcds - 1) / xcds;
  sched.assign((size_t)(rounds * grid), -1);
  for (int64_t k = 0; k < nb; ++k) {
    const int64_t g = k / per_xcd, i = k % per_xcd;
    sched[(size_t)((g / xcds) * grid + i * xcds + g % xcds)] = order[(size_t)k];
  }
}

// Launch groups on the block-overlap graph.  A group is the `per_xcd` units one XCD works on at the same time: what it
// fetches over the fabric is the UNION of its units' windows, so a good group is a compact tile of the graph.
//  * greedy: grown from the lowest unscheduled unit by repeatedly adding the unscheduled unit that overlaps most with
//    the group -- on a 3-D stencil a column of units one plane apart (the heaviest edge every time): no re-read of the
//    z planes inside the column, but every unit's y neighbours come from outside it;
//  * tiles: the two id strides that carry the most overlap weight (for a stencil: one plane, one row block) span a
//    lattice; a group is a w x (per_xcd / w) patch of it, topped up greedily where the lattice has holes.  For the 7-point
//    stencil on 215^3 a 4 x 8 patch of two-plane stacks fetches 1.23 elements per row instead of the column's 1.45.
// The candidate with the smallest total union is taken (RLH_SPMM_TILE=0: greedy only).  Speed only: any order is correct.
// (`owner`: the schedule unit that owns row block c, when a unit is not simply its own row block)
typedef std::vector<std::vector<std::pair<int32_t, int32_t>>> WellGraph;

static inline void well_group_fill(const WellGraph &adj, std::vector<char> &done, std::vector<int64_t> &weight,
                                   int64_t &seed, int per_xcd, std::vector<int32_t> &order,
                                   const std::vector<int32_t> &first) {
  const int64_t n = (int64_t)adj.size();
  std::vector<int32_t> touched;
  int members = 0;
  auto add = [&](int32_t b) {
    done[(size_t)b] = 1;
    order.push_back(b);
    ++members;
    for (const auto &e : adj[(size_t)b])
      if (!done[(size_t)e.first]) {
        if (weight[(size_t)e.first] == 0) touched.push_back(e.first);
        weight[(size_t)e.first] += e.second;
      }
  };
  for (int32_t b : first)
    if (members < per_xcd && !done[(size_t)b]) add(b);
  while (members < per_xcd && (int64_t)order.size() < n) {
    int32_t best = -1;
    for (int32_t c : touched)
      if (!done[(size_t)c] && (best < 0 || weight[(size_t)c] > weight[(size_t)best] ||
                               (weight[(size_t)c] == weight[(size_t)best] && c < best)))
        best = c;
    if (best < 0) {
      while (seed < n && done[(size_t)seed]) ++seed;
      best = (int32_t)seed;
    }
    add(best);
  }
  for (int32_t c : touched) weight[(size_t)c] = 0;
}

// columns the groups of `order` fetch: per group the merged length of its units' windows
static inline int64_t well_union_cost(const std::vector<std::vector<Win>> &wins, const std::vector<int32_t> &order,
                                      int per_xcd) {
  int64_t total = 0;
  std::vector<std::pair<int32_t, int32_t>> iv;
  for (size_t g = 0; g < order.size(); g += (size_t)per_xcd) {
    iv.clear();
    for (size_t k = g; k < order.size() && k < g + (size_t)per_xcd; ++k)
      for (const Win &w : wins[(size_t)order[k]]) iv.push_back({w.start, w.start + w.len});
    std::sort(iv.begin(), iv.end());
    int32_t lo = 0, hi = -1;
    for (const auto &p : iv) {
      if (hi < 0) { lo = p.first; hi = p.second; }
      else if (p.first <= hi) hi = std::max(hi, p.second);
      else { total += hi - lo; lo = p.first; hi = p.second; }
    }
    if (hi >= 0) total += hi - lo;
  }
  return total;
}

static inline void well_schedule(const std::vector<std::vector<Win>> &wins, int64_t nblocks, int64_t n_rows,
                                 int rows_per_block, int slots, std::vector<int32_t> &order,
                                 const std::vector<int32_t> *owner = nullptr) {
  const int xcds = 8;
  int per_xcd = slots / xcds;
  if (per_xcd < 1) per_xcd = 1;
  order.clear();
  order.reserve((size_t)nblocks);
  if (nblocks <= xcds * per_xcd || env_int("RLH_SPMM_SCHED", 1) == 0) {
    for (int64_t b = 0; b < nblocks; ++b) order.push_back((int32_t)b);
    return;
  }
  // overlap graph: weight = rows of unit c that unit b stages (both directions)
  WellGraph adj((size_t)nblocks);
  for (int64_t b = 0; b < nblocks; ++b)
    for (const auto &w : wins[b]) {
      int64_t lo = w.start, hi = (int64_t)w.start + w.len;
      if (hi > n_rows) hi = n_rows;                 // halo columns are not rows of this shard
      for (int64_t c = lo / rows_per_block; c * rows_per_block < hi; ++c) {
        const int64_t u = owner ? (*owner)[(size_t)c] : c;
        if (u == b || u < 0) continue;
        const int64_t ov = std::min<int64_t>(hi, (c + 1) * rows_per_block) - std::max<int64_t>(lo, c * rows_per_block);
        if (ov <= 0) continue;
        adj[b].push_back({(int32_t)u, (int32_t)ov});
        adj[u].push_back({(int32_t)b, (int32_t)ov});
      }
    }
  std::vector<char> done((size_t)nblocks, 0);
  std::vector<int64_t> weight((size_t)nblocks, 0);
  int64_t seed = 0;
  const std::vector<int32_t> none;
  while ((int64_t)order.size() < nblocks) well_group_fill(adj, done, weight, seed, per_xcd, order, none);
  if (env_int("RLH_SPMM_TILE", 1) == 0) return;
  // the two strides that carry the most weight
  std::vector<std::pair<int64_t, int64_t>> stride;          // (id difference, weight)
  {
    std::vector<std::pair<int64_t, int64_t>> all;
    for (int64_t b = 0; b < nblocks; ++b)
      for (const auto &e : adj[(size_t)b])
        if (e.first > b) all.push_back({(int64_t)e.first - b, (int64_t)e.second});
    std::sort(all.begin(), all.end());
    for (size_t i = 0; i < all.size();) {
      size_t k = i;
      int64_t wsum = 0;
      for (; k < all.size() && all[k].first == all[i].first; ++k) wsum += all[k].second;
      stride.push_back({all[i].first, wsum});
      i = k;
    }
    std::sort(stride.begin(), stride.end(), [](const std::pair<int64_t, int64_t> &a, const std::pair<int64_t, int64_t> &b) {
      return a.second > b.second;
    });
  }
  if (stride.size() < 2) return;
  const int64_t d1 = stride[0].first, d2 = stride[1].first;
  int64_t best_cost = well_union_cost(wins, order, per_xcd);
  const bool verbose = env_int("RLH_SPMM_VERBOSE", 0) != 0;
  if (verbose)
    fprintf(stderr, "well_schedule: %lld units of %d rows-per-block, strides %lld (weight %lld) and %lld (%lld); greedy groups fetch %lld columns\n",
            (long long)nblocks, rows_per_block, (long long)d1, (long long)stride[0].second, (long long)d2,
            (long long)stride[1].second, (long long)best_cost);
  for (int w = 2; w <= 8 && w < per_xcd; w *= 2) {
    if (per_xcd % w) continue;
    const int len = per_xcd / w;
    std::vector<int32_t> cand, first;
    cand.reserve((size_t)nblocks);
    std::fill(done.begin(), done.end(), 0);
    seed = 0;
    while ((int64_t)cand.size() < nblocks) {
      while (seed < nblocks && done[(size_t)seed]) ++seed;
      first.clear();
      for (int k = 0; k < len; ++k)
        for (int i = 0; i < w; ++i) {
          const int64_t u = seed + k * d1 + i * d2;
          if (u < nblocks && !done[(size_t)u]) first.push_back((int32_t)u);
        }
      well_group_fill(adj, done, weight, seed, per_xcd, cand, first);
    }
    const int64_t cost = well_union_cost(wins, cand, per_xcd);
    if (verbose) fprintf(stderr, "well_schedule: %d x %d tiles fetch %lld\n", w, len, (long long)cost);
    if (cost < best_cost) {
      best_cost = cost;
      order.swap(cand);
    }
  }
}

// ---- the interleaved layout (spmm_wide_build.hip, spmm_wide_{s,d,c,z}.hip)
constexpr int kWideRows = 256;                 // rows per block = threads per workgroup
constexpr int kWideLdsBytes = 160 * 1024;      // LDS of a CU
constexpr int kWideGroup = 16;                 // columns per staging group (windows are padded to whole groups)
constexpr int kWideGroupShift = 4;
constexpr int kWideHeader = 1024;              // a block's staging-group columns (<= 256); two of them in front of the image
constexpr int kWidePadChunks = 4;              // entry chunks allocated past the last block (the prefetch runs ahead)
int wide_stride(int nv, int es);
int wide_min_nv(int dtype);
int wide_sched(rlh_csr *h, int slots, int part, int64_t n_own, const int32_t **sched, int64_t *len, int *grid);
int wide_build(rlh_csr *h, const int64_t *indptr, const int32_t *indices, const void *values, bool force);
void wide_destroy(rlh_csr *h);
// Y = A X (cheb == nullptr) or the fused Chebyshev step on part 0 / 1 / 2 of the rows
int wide_spmm_s(rlh_csr *h, int part, int64_t m, const void *X, int64_t ldx, int64_t n_own, const void *H, int64_t ldh,
                void *Y, int64_t ldy, const void *B, int64_t ldb, double cy, double cp, double cb);
int wide_spmm_d(rlh_csr *h, int part, int64_t m, const void *X, int64_t ldx, int64_t n_own, const void *H, int64_t ldh,
                void *Y, int64_t ldy, const void *B, int64_t ldb, double cy, double cp, double cb);
int wide_spmm_c(rlh_csr *h, int part, int64_t m, const void *X, int64_t ldx, int64_t n_own, const void *H, int64_t ldh,
                void *Y, int64_t ldy, const void *B, int64_t ldb, double cy, double cp, double cb);
int wide_spmm_z(rlh_csr *h, int part, int64_t m, const void *X, int64_t ldx, int64_t n_own, const void *H, int64_t ldh,
                void *Y, int64_t ldy, const void *B, int64_t ldb, double cy, double cp, double cb);

}  // namespace rlh

// Shared declarations of the sparse operator (K13): the handle, the host-side window analysis and
// the XCD-aware launch order used by both windowed layouts (spmm.hip, spmm_wide.inc).
#pragma once

#include <stdlib.h>

#include <algorithm>
#include <atomic>
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
  // 256-row interleaved layout (any row length, any type)
  int64_t wide_blocks;     // 0: not built
  WideMeta *wide_meta;     // device
  int32_t *wide_gsrc;      // device
  void *wide_idx;          // device: 16-byte pieces of 8 positions, [chunk][row]
  void *wide_vals;         // device: 16-byte pieces of values, [chunk][piece][row]
  int wide_gmax;           // largest number of staging groups of a block
  std::vector<int32_t> wide_order, wide_maxcol;
  std::vector<WideSched> wide_scheds;
};

namespace rlh {

static inline int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

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
static inline int32_t find_windows(const int64_t *indptr, const int32_t *indices, int64_t r0, int64_t r1, int64_t nc,
                                   int gap, int gs, int round_groups, std::vector<Win> &ws) {
  std::vector<int32_t> cols(indices + indptr[r0], indices + indptr[r1]);
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

// position of column c in the staged image (last window starting at or before c)
static inline int32_t staged_position(const std::vector<Win> &ws, int32_t c) {
  size_t lo = 0, hi = ws.size();
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) / 2;
    if (ws[mid].start <= c) lo = mid; else hi = mid;
  }
  return ws[lo].off + (c - ws[lo].start);
}

// first column of every staging group of a block (surplus groups repeat the last)
static inline void fill_group_sources(const std::vector<Win> &ws, int32_t ngroups, int gs, int32_t *gsrc) {
  int32_t filled = 0, last = 0;
  for (const Win &w : ws)
    for (int32_t g = 0; g < (w.len + gs - 1) / gs; ++g, ++filled) gsrc[w.off / gs + g] = last = w.start + gs * g;
  for (; filled < ngroups; ++filled) gsrc[filled] = last;
}

template <typename F>
static void parallel_blocks(int64_t nblocks, F fn) {
  unsigned nt = std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > 16) nt = 16;
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

// Greedy grouping on the block-overlap graph: a group of `per_xcd` blocks is grown from the lowest
// unscheduled block by repeatedly adding the unscheduled block that overlaps most with the group.
static inline void well_schedule(const std::vector<std::vector<Win>> &wins, int64_t nblocks, int64_t n_rows,
                                 int rows_per_block, int slots, std::vector<int32_t> &order) {
  const int xcds = 8;
  int per_xcd = slots / xcds;
  if (per_xcd < 1) per_xcd = 1;
  order.clear();
  order.reserve((size_t)nblocks);
  if (nblocks <= xcds * per_xcd || env_int("RLH_SPMM_SCHED", 1) == 0) {
    for (int64_t b = 0; b < nblocks; ++b) order.push_back((int32_t)b);
    return;
  }
  // overlap graph: weight = rows of block c that block b stages (both directions)
  std::vector<std::vector<std::pair<int32_t, int32_t>>> adj((size_t)nblocks);
  for (int64_t b = 0; b < nblocks; ++b)
    for (const auto &w : wins[b]) {
      int64_t lo = w.start, hi = (int64_t)w.start + w.len;
      if (hi > n_rows) hi = n_rows;                 // halo columns are not rows of this shard
      for (int64_t c = lo / rows_per_block; c * rows_per_block < hi; ++c) {
        if (c == b) continue;
        const int64_t ov = std::min<int64_t>(hi, (c + 1) * rows_per_block) - std::max<int64_t>(lo, c * rows_per_block);
        if (ov <= 0) continue;
        adj[b].push_back({(int32_t)c, (int32_t)ov});
        adj[c].push_back({(int32_t)b, (int32_t)ov});
      }
    }
  std::vector<char> done((size_t)nblocks, 0);
  std::vector<int64_t> weight((size_t)nblocks, 0);
  int64_t seed = 0;
  while ((int64_t)order.size() < nblocks) {
    std::vector<int32_t> touched;
    int members = 0;
    auto add = [&](int32_t b) {
      done[b] = 1;
      order.push_back(b);
      ++members;
      for (const auto &e : adj[b])
        if (!done[e.first]) {
          if (weight[e.first] == 0) touched.push_back(e.first);
          weight[e.first] += e.second;
        }
    };
    while (members < per_xcd && (int64_t)order.size() < nblocks) {
      int32_t best = -1;
      for (int32_t c : touched)
        if (!done[c] && (best < 0 || weight[c] > weight[best] || (weight[c] == weight[best] && c < best))) best = c;
      if (best < 0) {
        while (seed < nblocks && done[seed]) ++seed;
        best = (int32_t)seed;
      }
      add(best);
    }
    for (int32_t c : touched) weight[c] = 0;
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

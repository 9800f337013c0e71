// Host side of the 256-row interleaved windowed layout of the sparse operator (kernel: spmm_wide.inc).
//
// Rows are cut into blocks of 256 (one row per thread of a 256-thread workgroup).  Per block the
// referenced columns are merged into windows of whole 16-column staging groups (spmm.h
// find_windows); an entry stores its value and the 16-bit position of its column in the block's
// staged image.  Entries are stored in CHUNKS of 8 slots per row, as 16-byte pieces per thread:
//   values    [chunk][piece k][row]   piece k = slots k * VPG .. + VPG - 1,  VPG = 16 / sizeof(T)
//   positions [chunk][row]            8 x uint16
// so a wave fetches a chunk with es / 2 + 1 coalesced 16-byte loads per lane and the kernel streams
// a row of ANY length through a fixed number of registers.  The value bytes are copied as they
// are, so this file is independent of the element type.
#include "spmm.h"

namespace rlh {

// LDS bytes per staged column for NV vectors of es bytes: NV elements + one 16-byte pad, and an odd
// number of 16-byte slots so that consecutive columns fall into different bank groups
// (ds_read_b128 / ds_write_b128 of consecutive columns are then conflict-free).
int wide_stride(int nv, int es) {
  int s = nv * es + 16;
  if (((s / 16) & 1) == 0) s += 16;
  return s;
}

int wide_min_nv(int dtype) { return (dtype == RLH_S || dtype == RLH_D) ? 8 : 4; }

int wide_build(rlh_csr *h, const int64_t *indptr, const int32_t *indices, const void *values_, bool force) {
  const int es = (int)dtype_size(h->dtype);
  const int VP = es / 2;                         // 16-byte value pieces per chunk of 8 slots
  const int VPG = 16 / es;                       // values per piece
  const char *values = (const char *)values_;
  const int64_t n = h->n_rows;
  const int64_t nblocks = (n + kWideRows - 1) / kWideRows;
  h->wide_blocks = 0;
  if (nblocks == 0 || h->nnz == 0) return 0;
  std::vector<std::vector<Win>> wins((size_t)nblocks);
  std::vector<int32_t> width((size_t)nblocks, 0), ngroups((size_t)nblocks, 0);
  parallel_blocks(nblocks, [&](int64_t b) {
    const int64_t r0 = b * kWideRows, r1 = (r0 + kWideRows < n) ? r0 + kWideRows : n;
    int64_t w = 0;
    for (int64_t r = r0; r < r1; ++r) w = std::max<int64_t>(w, indptr[r + 1] - indptr[r]);
    width[b] = (int32_t)std::min<int64_t>(w, 1 << 20);
    ngroups[b] = find_windows(indptr, indices, r0, r1, h->n_cols, 32, kWideGroup, 1, wins[b]) / kWideGroup;
  });
  int32_t wmax = 0, gmax = 0;
  int64_t staged = 0, slots = 0;
  for (int64_t b = 0; b < nblocks; ++b) {
    wmax = std::max(wmax, width[b]);
    gmax = std::max(gmax, ngroups[b]);
    staged += (int64_t)ngroups[b] * kWideGroup;
    slots += (int64_t)width[b] * kWideRows;
  }
  h->well_ratio = slots > 0 ? (double)staged / (double)slots : 0.0;
  // every block's image must fit the LDS with the smallest number of vectors per pass, positions
  // are 16 bits, a row has at most 8 * 32767 slots, and the group list must fit the header
  if ((int64_t)gmax * kWideGroup * wide_stride(wide_min_nv(h->dtype), es) + 2 * kWideHeader > kWideLdsBytes) return 0;
  if (gmax * 4 > kWideHeader || wmax > 8 * 32767) return 0;
  if (!force && staged * 10 > slots * 9) return 0;      // too little column locality for the staging to pay
  // Rows that share their column pattern with the row after them -- the unknowns of a node of a finite-element model: every
  // degree of freedom of a node couples to every degree of freedom of its neighbours -- are given to ONE thread in pairs: the
  // pair's entries are one position and two values, and the staged x values of an entry, which are what bounds the kernel
  // on long rows (every stored entry reads its NV staged values out of the LDS: profiles/r02_spmm_wide.txt), are read once
  // for both rows.  All complete pairs (2 i, 2 i + 1) of the matrix must qualify; real types; RLH_WIDE_PAIR=0: never.
  int K = 1;
  if ((h->dtype == RLH_S || h->dtype == RLH_D) && env_int("RLH_WIDE_PAIR", 1) != 0 && h->nnz >= 16 * n) {
    std::atomic<int> differ{0};
    host_parallel(64, [&](int t, int nt) {
      const int64_t pairs = n / 2;
      for (int64_t q = pairs * t / nt; q < pairs * (t + 1) / nt && !differ.load(std::memory_order_relaxed); ++q) {
        const int64_t r = 2 * q, la = indptr[r + 1] - indptr[r], lb = indptr[r + 2] - indptr[r + 1];
        if (la != lb || memcmp(indices + indptr[r], indices + indptr[r + 1], (size_t)la * sizeof(int32_t)) != 0) differ.store(1);
      }
    });
    if (!differ.load()) K = 2;
  }
  h->wide_k = K;
  const int TPS = kWideRows / K;                  // threads per set of vectors = row groups of a block
  std::vector<WideMeta> meta((size_t)nblocks);
  int64_t eoff = 0, goff = 0;
  for (int64_t b = 0; b < nblocks; ++b) {
    const int nch = std::max(1, (width[b] + 7) / 8);
    meta[b] = WideMeta{eoff, (int32_t)goff, (int16_t)nch, (int16_t)ngroups[b]};
    eoff += nch;
    goff += ngroups[b];
  }
  RLH_REQUIRE(goff < ((int64_t)1 << 31), "rlh_csr_create: too many staging groups");
  std::vector<int32_t> gsrc((size_t)goff);
  const int64_t nchunks = eoff + kWidePadChunks;  // padding: the kernel's prefetch runs ahead of the last block's chunks
  // K = 2: positions [chunk][row pair], values [chunk][row of the pair][piece][row pair]
  std::vector<char> idx((size_t)nchunks * TPS * 16, 0);
  std::vector<char> vals((size_t)nchunks * VP * kWideRows * 16, 0);
  parallel_blocks(nblocks, [&](int64_t b) {
    const std::vector<Win> &ws = wins[b];
    fill_group_sources(ws, ngroups[b], kWideGroup, gsrc.data() + meta[b].goff);
    const int64_t r0 = b * kWideRows;
    for (int l = 0; l < kWideRows; ++l) {
      const int64_t r = r0 + l;
      const int lt = l / K, lk = l % K;             // thread of the set, row of its group
      // (a row past the end of the matrix: the slots of its pair partner's positions with value 0)
      const int64_t rp_ = r < n ? r : (K > 1 && r - lk < n ? r - lk : -1);
      const int64_t p = rp_ >= 0 ? indptr[rp_] : 0, len = rp_ >= 0 ? indptr[rp_ + 1] - p : 0;
      const bool real_row = r < n;
      // padding slots carry value 0 and the position of the row's own first entry, so that they
      // only ever touch a column the row references (0 * Inf of a foreign column would be NaN)
      const uint16_t padpos = len > 0 ? (uint16_t)staged_position(ws, indices[p]) : 0;
      for (int t = 0; t < meta[b].nchunks * 8; ++t) {
        const int64_t q = meta[b].eoff + t / 8;
        const int tt = t % 8;
        if (lk == 0) {
          uint16_t *pi = reinterpret_cast<uint16_t *>(idx.data() + ((size_t)q * TPS + lt) * 16) + tt;
          *pi = t < len ? (uint16_t)staged_position(ws, indices[p + t]) : padpos;
        }
        if (t < len && real_row) {
          char *pv = vals.data() + ((((size_t)q * K + lk) * VP + tt / VPG) * TPS + lt) * 16 + (size_t)(tt % VPG) * es;
          memcpy(pv, values + (size_t)(p + t) * es, (size_t)es);
        }
      }
    }
  });
  h->well_inbounds = 1;
  h->well_aligned = 1;
  for (int64_t g = 0; g < goff; ++g) {
    if ((int64_t)gsrc[g] + kWideGroup > h->n_cols) h->well_inbounds = 0;
    if (gsrc[g] & 7) h->well_aligned = 0;
  }
  well_schedule(wins, nblocks, n, kWideRows, ctx().num_cu * 2, h->wide_order);
  h->wide_maxcol.resize((size_t)nblocks);
  for (int64_t b = 0; b < nblocks; ++b) h->wide_maxcol[(size_t)b] = wins[b].back().start + wins[b].back().len - 1;
  h->wide_gmax = gmax;
  RLH_HIP(hipMalloc((void **)&h->wide_meta, (size_t)nblocks * sizeof(WideMeta)));
  RLH_HIP(hipMemcpy(h->wide_meta, meta.data(), (size_t)nblocks * sizeof(WideMeta), hipMemcpyHostToDevice));
  RLH_HIP(hipMalloc((void **)&h->wide_gsrc, (size_t)std::max<int64_t>(goff, 1) * sizeof(int32_t)));
  RLH_HIP(hipMemcpy(h->wide_gsrc, gsrc.data(), (size_t)goff * sizeof(int32_t), hipMemcpyHostToDevice));
  RLH_HIP(hipMalloc((void **)&h->wide_idx, idx.size()));
  RLH_HIP(hipMemcpy(h->wide_idx, idx.data(), idx.size(), hipMemcpyHostToDevice));
  RLH_HIP(hipMalloc((void **)&h->wide_vals, vals.size()));
  RLH_HIP(hipMemcpy(h->wide_vals, vals.data(), vals.size(), hipMemcpyHostToDevice));
  h->padded = eoff * 8 * kWideRows;
  h->device_bytes = nblocks * (int64_t)sizeof(WideMeta) + goff * 4 + (int64_t)idx.size() + (int64_t)vals.size();
  h->wide_blocks = nblocks;
  return 0;
}

void wide_destroy(rlh_csr *h) {
  if (h->wide_meta) (void)hipFree(h->wide_meta);
  if (h->wide_gsrc) (void)hipFree(h->wide_gsrc);
  if (h->wide_idx) (void)hipFree(h->wide_idx);
  if (h->wide_vals) (void)hipFree(h->wide_vals);
  for (WideSched &s : h->wide_scheds)
    if (s.sched) (void)hipFree(s.sched);
  h->wide_scheds.clear();
  h->wide_meta = nullptr; h->wide_gsrc = nullptr; h->wide_idx = nullptr; h->wide_vals = nullptr;
}

// The launch order for `slots` resident workgroups over all blocks (part 0), the blocks that
// reference only columns < n_own (part 1) or the others (part 2); built on first use and kept.
int wide_sched(rlh_csr *h, int slots, int part, int64_t n_own, const int32_t **sched, int64_t *len, int *grid) {
  if (part == 0) n_own = 0;
  for (const WideSched &s : h->wide_scheds)
    if (s.slots == slots && s.part == part && s.n_own == n_own) {
      *sched = s.sched; *len = s.len; *grid = s.grid;
      return 0;
    }
  std::vector<int32_t> list, sch;
  for (int32_t b : h->wide_order) {
    const bool interior = h->wide_maxcol[(size_t)b] < n_own;
    if (part == 0 || (part == 1) == interior) list.push_back(b);
  }
  WideSched s{slots, 0, part, n_own, nullptr, 0};
  if (!list.empty()) {
    well_layout(list, slots, sch, s.grid);
    s.len = (int64_t)sch.size();
    RLH_HIP(hipMalloc((void **)&s.sched, sch.size() * sizeof(int32_t)));
    RLH_HIP(hipMemcpy(s.sched, sch.data(), sch.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (h->wide_scheds.size() >= 16) {             // a bounded cache: drop the oldest
    if (h->wide_scheds.front().sched) {
      (void)hipStreamSynchronize(ctx().stream);
      (void)hipFree(h->wide_scheds.front().sched);
    }
    h->wide_scheds.erase(h->wide_scheds.begin());
  }
  h->wide_scheds.push_back(s);
  *sched = s.sched; *len = s.len; *grid = s.grid;
  return 0;
}

}  // namespace rlh

// K13: sparse symmetric/Hermitian operator Y = A X on a block of vectors.
//
// The reference keeps triu(A) in 1-based CSR and calls mkl_?csrmm with the
// 'SUNF'/'HUNF' descriptor on the column-major n x m block
// (raleigh/algebra/sparse_mkl.py:16-48, raleigh/algebra/mkl_wrap.py:204-276).
// Here the caller hands over the FULL 0-based CSR (both triangles); at creation it is re-laid-out
// on the host in one of three device layouts (rlh_csr_layout reports which):
//  * 1024-row windowed ELL (this file, well_spmm_kernel): rows of at most 8 entries of a real
//    type with column locality -- stencils.  A block's column windows are staged through the LDS,
//    the row's entries stay in registers for all vectors.  Measured 54.6 % of the 8 TB/s peak on
//    the 7-point Laplacian at n = 10^7, m = 32 (1.21 x the algorithmic traffic).
//  * 256-row interleaved windowed layout (spmm_wide.inc): any row length, any type, with column
//    locality -- FE matrices, wide bands, all complex operators.  Entries stream through registers
//    in chunks of 8, the LDS image is [column][vector].
//  * sliced ELLPACK, slice height 64 (sell_spmm_kernel): no column locality.  One 8-byte gather of
//    X per stored entry and vector; limited by the rate at which a CU retires per-lane gathers
//    (profiles/r01_spmm_sweep2.txt), not by HBM: 34 % on the 7-point Laplacian.
#include <unordered_map>

#include "spmm.h"

namespace rlh {

// Work mapping (speed only; any placement gives the same result).  Workgroups b and b + 8
// share an XCD and its 4 MiB L2 under the observed round-robin placement.
//  * The slices are cut into chunks of `chunk` consecutive slices; chunk c belongs to XCD
//    c % 8 and the workgroups of an XCD walk its chunks in order, so rows that reference each
//    other are processed at about the same time on the same L2, and the eight XCDs advance
//    through adjacent chunks (one contiguous window of X for the Infinity Cache).
//  * The m vectors are cut into tiles of JT.  The waves of a workgroup take DIFFERENT tiles of
//    the SAME slice (tiles_per_block of them), which keeps the set of rows in flight per XCD --
//    and with it the L2 footprint of the gathered X lines -- small; the matrix entries of the
//    slice are fetched once per workgroup through L1.
// Fused Chebyshev step (CHEB), three-term form: with t = A y the row's results update
//   p = cy y + cp p + cb (b - t)
// in the same pass (device polynomial preconditioner: y = y_k, p = y_{k-1} on entry and y_{k+1} on
// return, b the right-hand side).  p is read and written by the row's own lane only, so it is
// updated in place; it must not alias y, which other rows are still gathering.  Per element the
// step reads y, p, b and writes p (the two-term form with a residual and a direction block read
// three and wrote three).
template <typename T, int JT, int TT, bool CHEB>
__global__ __launch_bounds__(256, (JT * TT >= 64 ? 1 : 2)) void sell_spmm_kernel(const int64_t *__restrict__ slice_ptr,
                                                        const int32_t *__restrict__ cols,
                                                        const T *__restrict__ vals, int64_t n_rows,
                                                        int64_t n_slices, const T *__restrict__ X, int64_t ldx,
                                                        int64_t n_own, const T *__restrict__ H, int64_t ldh,
                                                        T *__restrict__ Y, int64_t ldy, int m, int chunk,
                                                        int ntiles, int tiles_per_block, ChebArgs<T> cheb) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7;                                  // gridDim.x is a multiple of 8
  const int slices_per_block = 4 / tiles_per_block;
  const int64_t q0 = (int64_t)(blockIdx.x >> 3) * slices_per_block + wave / tiles_per_block;
  const int64_t q_stride = (int64_t)(gridDim.x >> 3) * slices_per_block;
  const int64_t nchunks = (n_slices + chunk - 1) / chunk;
  for (int tile = wave % tiles_per_block; tile < ntiles; tile += tiles_per_block) {
    const int j0 = tile * JT;
    const int jv = (m - j0) < JT ? (m - j0) : JT;
    const T *__restrict__ Xp = X + (int64_t)j0 * ldx;
    const T *__restrict__ Hp = H ? H + (int64_t)j0 * ldh : nullptr;
    T *__restrict__ Yp = Y + (int64_t)j0 * ldy;
    for (int64_t q = q0;; q += q_stride) {
      const int64_t c = (q / chunk) * 8 + xcd;
      if (c >= nchunks) break;
      const int64_t slice = c * chunk + q % chunk;
      if (slice >= n_slices) continue;
      const int64_t base = slice_ptr[slice];
      const int width = (int)((slice_ptr[slice + 1] - base) >> 6);
      const int64_t row = slice * 64 + lane;
      T acc[JT];
#pragma unroll
      for (int j = 0; j < JT; ++j) acc[j] = zero_of(T{});
      // entries in groups of TT: the group's column indices and values stay in registers while
      // the vectors of the tile are walked, so the TT gathers of one vector (neighbouring
      // entries of X for a stencil row) are issued back to back and share L1 lines
      for (int t0 = 0; t0 < width; t0 += TT) {
        const T *xc[TT];
        int64_t ldc[TT];
        T v[TT];
#pragma unroll
        for (int u = 0; u < TT; ++u) {
          const int t = (t0 + u) < width ? (t0 + u) : (width - 1);
          const int64_t e = base + (int64_t)t * 64 + lane;
          const int64_t cidx = __builtin_nontemporal_load(cols + e);   // streamed once per sweep
          v[u] = nt_load(vals + e);
          if (t0 + u >= width) v[u] = zero_of(T{});
          const bool own = cidx < n_own;               // off-shard rows live in the halo block
          xc[u] = own ? Xp + cidx : Hp + (cidx - n_own);
          ldc[u] = own ? ldx : ldh;
        }
        // all TT x JT gathers are issued before the first FMA needs one (memory-level
        // parallelism comes from the wave itself, not from occupancy)
        T xv[TT][JT];
#pragma unroll
        for (int u = 0; u < TT; ++u)
#pragma unroll
          for (int j = 0; j < JT; ++j) {
            const int jj = j < jv ? j : 0;             // clamp: columns >= m re-read column j0
            xv[u][j] = xc[u][(int64_t)jj * ldc[u]];
          }
#pragma unroll
        for (int u = 0; u < TT; ++u)
#pragma unroll
          for (int j = 0; j < JT; ++j) fma_acc(acc[j], v[u], xv[u][j]);
      }
      if (row < n_rows) {
        if constexpr (!CHEB) {
#pragma unroll
          for (int j = 0; j < JT; ++j)
            if (j < jv) nt_store(Yp + row + (int64_t)j * ldy, acc[j]);
        } else {
#pragma unroll
          for (int j = 0; j < JT; ++j)
            if (j < jv) {
              const int64_t jc = j0 + j;
              T *pp = Yp + row + (int64_t)j * ldy;
              const T res = sub_of(cheb.B[row + jc * cheb.ldb], acc[j]);      // b - A y
              *pp = add_of(add_of(scale_of(cheb.cy, Xp[row + (int64_t)j * ldx]), scale_of(cheb.cp, *pp)),
                           scale_of(cheb.cb, res));
            }
        }
      }
    }
  }
}

template <typename T, int JT, int TT>
static int launch_spmm_t(const rlh_csr *h, int64_t m, const T *X, int64_t ldx, int64_t n_own, const T *H, int64_t ldh,
                       T *Y, int64_t ldy, const ChebArgs<T> *cheb) {
  Context &c = ctx();
  static int fit = 0;                                        // resident workgroups per CU of this instantiation
  if (fit == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sell_spmm_kernel<T, JT, TT, false>, 256, 0) != hipSuccess || nb < 1)
      nb = 1;
    fit = nb > 8 ? 8 : nb;
  }
  int per_cu = fit;
  const int cap = env_int("RLH_SPMM_WG_PER_CU", 2);         // 0: as many as fit (tunable; 2 measured best)
  if (cap > 0 && per_cu > cap) per_cu = cap;
  const int chunk = env_int("RLH_SPMM_CHUNK", 64);      // slices per row chunk (tunable)
  const int tpb_cap = env_int("RLH_SPMM_TPB", 1);       // waves of a workgroup on one slice (tunable)
  const int ntiles = (int)((m + JT - 1) / JT);
  int tiles_per_block = ntiles >= 4 ? 4 : (ntiles >= 2 ? 2 : 1);
  if (tiles_per_block > tpb_cap) tiles_per_block = tpb_cap >= 2 ? 2 : 1;
  const int slices_per_block = 4 / tiles_per_block;
  int64_t nb = (int64_t)c.num_cu * per_cu;
  const int64_t need = ((h->n_slices + slices_per_block - 1) / slices_per_block + 7) / 8 * 8;
  if (nb > need) nb = need;
  nb = (nb + 7) / 8 * 8;
  if (cheb) {
    if constexpr (JT < 32)         // (32 accumulators + the fused epilogue spill: the fused step takes tiles of 16)
      hipLaunchKernelGGL((sell_spmm_kernel<T, JT, TT, true>), dim3((unsigned)nb), dim3(256), 0, c.stream, h->slice_ptr,
                         h->cols, (const T *)h->vals, h->n_rows, h->n_slices, X, ldx, n_own, H, ldh, Y, ldy, (int)m,
                         chunk < 1 ? 1 : chunk, ntiles, tiles_per_block, *cheb);
  } else
    hipLaunchKernelGGL((sell_spmm_kernel<T, JT, TT, false>), dim3((unsigned)nb), dim3(256), 0, c.stream, h->slice_ptr,
                       h->cols, (const T *)h->vals, h->n_rows, h->n_slices, X, ldx, n_own, H, ldh, Y, ldy, (int)m,
                       chunk < 1 ? 1 : chunk, ntiles, tiles_per_block, ChebArgs<T>{});
  RLH_HIP(hipGetLastError());
  return 0;
}

template <typename T, int JT>
static int launch_spmm(const rlh_csr *h, int64_t m, const T *X, int64_t ldx, int64_t n_own, const T *H, int64_t ldh,
                       T *Y, int64_t ldy, const ChebArgs<T> *cheb) {
  const int tt = env_int("RLH_SPMM_TT", 1);             // entries per register group (tunable)
  if (tt >= 8) return launch_spmm_t<T, JT, 8>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
  if (tt >= 4) return launch_spmm_t<T, JT, 4>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
  return launch_spmm_t<T, JT, 1>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
}


// ------------------------------------------------------------------ Windowed ELL
// The sliced kernel above is limited by the rate at which a CU's vector-memory path retires
// per-lane gathers (one 8-byte gather per stored entry and vector: measured ~19 cycles per
// 512-byte wave gather whatever the locality, profiles/r01_spmm_sweep2.txt), not by HBM.  For
// matrices with column locality (banded, stencil, FE after a bandwidth-reducing ordering) the
// gathers of a block of rows fall into a few contiguous column windows, so this layout moves
// them to the LDS, whose read rate is 8x higher:
//  * rows are cut into blocks of 1024 (one row per thread of a 1024-thread workgroup, 16 waves);
//  * per block the referenced columns are merged into windows (gaps <= 32 columns are bridged),
//    each padded to a multiple of 64 columns: a "staging group" is 64 consecutive columns that
//    one wave copies from the block of vectors into the LDS with one coalesced load;
//  * an entry stores its value and the 16-bit position of its column in the staged image;
//    the entries of the thread's row stay in registers for all the vectors of the block;
//  * per step the workgroup computes as many vectors as one LDS buffer holds while the loads that
//    fill the other buffer with the next ones are in flight (register-staged double buffer,
//    one barrier per step).
// Each element of X is then loaded once per window it lies in (3.4x for the 7-point stencil on
// 215^3: centre + y neighbours in one window, the two z planes in one each) instead of once per
// entry (7x), and the seven reads per row come out of the LDS.
template <typename T> struct WellCfg { static constexpr int SMAX = sizeof(T) >= 16 ? 4 : 8; };   // <= 64 KiB per buffer

// Entry storage of a block: slot t of thread l (row l of the block) lives in 16-byte pieces per
// thread -- values [t / VPG][l][t % VPG] with VPG = 16 / sizeof(T) values per piece, positions
// [t / 8][l][t % 8] -- so that a thread fetches its row's entries with a few 16-byte loads (one
// 4- or 8-byte and one 2-byte load per entry cost as many vector-memory instructions as all the
// staging of the block).  A block's slots are padded to a multiple of 8; eoff counts slots.
template <typename T>
__host__ __device__ inline int64_t well_val_index(int64_t eoff, int t, int l) {
  constexpr int VPG = 16 / (int)sizeof(T) > 0 ? 16 / (int)sizeof(T) : 1;
  return (eoff + (t / VPG) * VPG) * 1024 + (int64_t)l * VPG + (t % VPG);
}
__host__ __device__ inline int64_t well_idx_index(int64_t eoff, int t, int l) {
  return (eoff + (t / 8) * 8) * 1024 + (int64_t)l * 8 + (t % 8);
}
// the row's WMAX values and positions (the host pads every row to WMAX slots: value 0 at the position of
// the row's own first entry, so a padding slot never touches a column the row does not reference)
template <typename T, int WMAX>
__device__ __forceinline__ void well_load_entries(const T *__restrict__ vals, const uint16_t *__restrict__ idx,
                                                  int64_t eoff, int tid, int width, T (&v)[WMAX], unsigned (&pos)[WMAX]) {
  constexpr int VPG = 16 / (int)sizeof(T) > 0 ? 16 / (int)sizeof(T) : 1;
  static_assert(WMAX % 8 == 0 && WMAX % VPG == 0, "entry slots come in 16-byte pieces");
#pragma unroll
  for (int g = 0; g < WMAX / VPG; ++g) {
    union { rlh_u32x4e u; T t[VPG]; } piece;
    piece.u = __builtin_nontemporal_load(reinterpret_cast<const rlh_u32x4e *>(vals + well_val_index<T>(eoff, g * VPG, tid)));
#pragma unroll
    for (int k = 0; k < VPG; ++k) v[g * VPG + k] = piece.t[k];
  }
#pragma unroll
  for (int g = 0; g < WMAX / 8; ++g) {
    union { rlh_u32x4e u; unsigned short h[8]; } piece;
    piece.u = __builtin_nontemporal_load(reinterpret_cast<const rlh_u32x4e *>(idx + well_idx_index(eoff, g * 8, tid)));
#pragma unroll
    for (int k = 0; k < 8; ++k) pos[g * 8 + k] = (unsigned)piece.h[k];
  }
}
constexpr int kWellRows = 1024;
constexpr int kStkMaxPatterns = 4096;
constexpr int kStkMaxDeltas = 64;                  // position patterns per member of a stack

// The same with the values out of the stacks' dictionary: the row's pattern index (4 bytes) and the 8 values of that
// pattern (a few dozen 16-byte lines that stay in the L1 / L2 for the whole launch) instead of 8 values per row.
// Positions likewise where the handle has position patterns (dtab): within a member of a stack the staged position of a
// stencil row's k-th entry is the row's index in the block plus a constant, so the row stores which of the member's
// few tuples of constants it uses (upper half of its pattern word) instead of 8 positions.
template <typename T, int WMAX>
__device__ __forceinline__ void well_load_entries_pat(const T *__restrict__ table, const int32_t *__restrict__ pat,
                                                      const uint16_t *__restrict__ dtab, const uint16_t *__restrict__ idx,
                                                      int64_t eoff, int64_t pblock, int tid, T (&v)[WMAX],
                                                      unsigned (&pos)[WMAX]) {
  constexpr int VPG = 16 / (int)sizeof(T) > 0 ? 16 / (int)sizeof(T) : 1;
  static_assert(WMAX == 8, "one 16-byte piece of positions per row");
  const unsigned word = (unsigned)pat[pblock * kWellRows + tid];
  const int32_t pid = (int32_t)(word & 0xffffu);
#pragma unroll
  for (int g = 0; g < WMAX / VPG; ++g) {
    union { rlh_u32x4e u; T t[VPG]; } piece;
    piece.u = *reinterpret_cast<const rlh_u32x4e *>(table + (int64_t)pid * WMAX + g * VPG);
#pragma unroll
    for (int k = 0; k < VPG; ++k) v[g * VPG + k] = piece.t[k];
  }
  union { rlh_u32x4e u; unsigned short h[8]; } piece;
  if (dtab) {
    piece.u = *reinterpret_cast<const rlh_u32x4e *>(dtab + (pblock * kStkMaxDeltas + (int64_t)(word >> 16)) * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) pos[k] = ((unsigned)piece.h[k] + (unsigned)tid) & 0xffffu;
  } else {
    piece.u = __builtin_nontemporal_load(reinterpret_cast<const rlh_u32x4e *>(idx + well_idx_index(eoff, 0, tid)));
#pragma unroll
    for (int k = 0; k < 8; ++k) pos[k] = (unsigned)piece.h[k];
  }
}

// Exchange within a quad of lanes (DPP quad_perm): CTRL 0xB1 = lane ^ 1, 0x4E = lane ^ 2.
template <typename T, int CTRL>
__device__ __forceinline__ T dpp_quad(T v) {
  constexpr int W = (int)(sizeof(T) / 4);
  union { T t; int w[W]; } a, b;
  a.t = v;
#pragma unroll
  for (int k = 0; k < W; ++k) b.w[k] = __builtin_amdgcn_update_dpp(0, a.w[k], CTRL, 0xf, 0xf, true);
  return b.t;
}

template <typename T, int WMAX, int EPL, bool CHEB>
__global__ __launch_bounds__(1024) void well_spmm_kernel(const WellMeta *__restrict__ meta,
                                                         const int32_t *__restrict__ gsrc,
                                                         const uint16_t *__restrict__ idx,
                                                         const T *__restrict__ vals, int64_t n_rows, int64_t n_cols,
                                                         const int32_t *__restrict__ sched, int64_t sched_len,
                                                         const T *__restrict__ X, int64_t ldx,
                                                         int64_t n_own, const T *__restrict__ H, int64_t ldh,
                                                         T *__restrict__ Y, int64_t ldy, int m, int cps_cap,
                                                         ChebArgs<T> cheb) {
  constexpr int SMAX = WellCfg<T>::SMAX;      // staging loads per thread and step
  constexpr int BUF = kWellRows * SMAX;       // elements per LDS buffer
  __shared__ T lds[2 * BUF];
  char *ldsb = reinterpret_cast<char *>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const unsigned lane8 = (unsigned)lane * (unsigned)sizeof(T);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T *Hs = H ? H - n_own : X;            // halo row of column c: Hs[c]
  const int last_col = (int)(n_cols - 1);
  // The step bodies below are straight-line on purpose: every load and LDS write of a step is
  // issued unconditionally (surplus staging slots repeat the step's last group, entry slots past
  // the block's width are zeroed), so the compiler keeps counted vmcnt waits and the loads of two
  // steps really are in flight together; with per-load predicates it waited for vmcnt(0) before
  // every load.
  for (int64_t pos = blockIdx.x; pos < sched_len; pos += gridDim.x) {
    const int64_t b = sched[pos];
    if (b < 0) continue;
    const WellMeta mt = meta[b];
    const int width = mt.width_ng & 255;
    const int ng = mt.width_ng >> 8;          // >= 1
    const int F = ng * 64;                    // staged elements per vector
    // vectors per step: as many as the LDS buffer holds, evened out over the steps
    int cps = (16 * SMAX) / ng;
    if (cps > cps_cap) cps = cps_cap;
    if (cps > m) cps = m;
    const int nsteps = (m + cps - 1) / cps;
    cps = (m + nsteps - 1) / nsteps;
    const int G = ng * cps;                   // staging groups per step
    const int64_t row = b * kWellRows + tid;
    const bool whole_block = (b + 1) * kWellRows <= n_rows;     // workgroup-uniform: no row past the end
    // the row's entries: registers for the whole block of vectors (slots past the block's width
    // read the next block's entries -- the arrays are padded -- and are zeroed)
    // Positions: one register per entry holding the byte offset, or (PACK: 32 slots of an 8-byte
    // type, where 32 more registers would spill) two 16-bit positions per register, unpacked at use.
    constexpr bool PACK = WMAX * sizeof(T) >= 256;
    T v[WMAX];
    unsigned ixb[PACK ? WMAX / 2 : WMAX];
    {
      unsigned px[WMAX];
      well_load_entries<T, WMAX>(vals, idx, mt.eoff, tid, width, v, px);
#pragma unroll
      for (int t = 0; t < WMAX; ++t) {
        if constexpr (!PACK) ixb[t] = px[t] * (unsigned)sizeof(T);
        else if (t & 1) ixb[t / 2] |= px[t] << 16;
        else ixb[t / 2] = px[t];
      }
    }
    auto entry_offset = [&](int t) -> unsigned {       // byte offset of entry t's column in a staged vector
      if constexpr (!PACK) return ixb[t];
      else return ((ixb[t / 2] >> (16 * (t & 1))) & 0xffffu) * (unsigned)sizeof(T);
    };
    // this thread's share of a step's staging: one load moves EPL elements per lane, i.e. EPL
    // consecutive 64-column groups per wave; slot i of the wave covers groups q .. q + EPL - 1,
    // q = EPL (wave + 16 i), of the step's G (vector cc = q / ng of the step, groups q % ng .. of
    // the block: ng is a multiple of 8, so a slot never straddles two vectors)
    constexpr int SLOTS = SMAX / EPL;
    constexpr int LPG = 64 / EPL;             // lanes per group
    int sbase[SLOTS];                         // byte offset of the slot's groups in an LDS buffer (wave-uniform)
    int scc[SLOTS];                           // vector within the step (wave-uniform)
    int scol[SLOTS];                          // first column this lane reads
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      int q = (wave + 16 * i) * EPL;
      if (q > G - EPL) q = G - EPL;           // surplus slots repeat the last groups
      const int cc = q / ng, g = q - cc * ng;
      scc[i] = __builtin_amdgcn_readfirstlane(cc);
      sbase[i] = __builtin_amdgcn_readfirstlane((cc * F + g * 64) * (int)sizeof(T));
      if constexpr (EPL == 1)
        scol[i] = __builtin_amdgcn_readfirstlane(gsrc[mt.goff + g]);
      else
        scol[i] = gsrc[mt.goff + g + lane / LPG] + (lane % LPG) * EPL;
    }
    VecU<T, EPL> stA[SLOTS];
    auto stage_load = [&](int s, VecU<T, EPL> (&st)[SLOTS]) {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        int j = s * cps + scc[i];
        if (j > m - 1) j = m - 1;             // a short last step re-reads the last vector
        int c;
        if constexpr (EPL == 1) {
          c = scol[i] + lane;
          if (c > last_col) c = last_col;     // group padding past the last column re-reads it
        } else {
          c = scol[i];                        // the host checked: groups in range, pieces on one side of n_own
        }
        const T *src = c < n_own ? X + (int64_t)j * ldx : Hs + (int64_t)j * ldh;
        st[i] = *reinterpret_cast<const VecU<T, EPL> *>(src + c);
      }
    };
    // LDS addresses are formed at the point of use from an opaque scalar (the byte offset of the
    // buffer, or of the vector in it) plus a per-thread register: left to itself the compiler
    // hoists one address register per (buffer, vector, entry) out of the step loop and spills.
    auto stage_write = [&](int s, const VecU<T, EPL> (&st)[SLOTS]) {
      unsigned boff = (unsigned)(s & 1) * (unsigned)(BUF * sizeof(T));
      asm volatile("" : "+s"(boff));
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        VecA<T, EPL> val;
#pragma unroll
        for (int k = 0; k < EPL; ++k) val.e[k] = st[i].e[k];
        *reinterpret_cast<VecA<T, EPL> *>(ldsb + (lane8 * EPL + (boff + (unsigned)sbase[i]))) = val;
      }
    };
    auto row_product = [&](unsigned bo) -> T {          // (A y)[row] for the vector staged at byte offset bo
      asm volatile("" : "+s"(bo));
      T acc = zero_of(T{});
#pragma unroll
      for (int t = 0; t < WMAX; ++t) fma_acc(acc, v[t], *reinterpret_cast<const T *>(ldsb + (entry_offset(t) + bo)));
      return acc;
    };
    auto compute = [&](int s) {
      unsigned boff = (unsigned)(s & 1) * (unsigned)(BUF * sizeof(T));
      int j = s * cps;
      const int jend = (j + cps < m) ? j + cps : m;
      if constexpr (CHEB && sizeof(T) <= 8) {
        // Two vectors at a time with the two results of a lane pair exchanged (DPP), so that a lane
        // holds rows rb, rb + 1 of ONE vector: y, p, b are then read and p written as 2-element
        // pieces (8 bytes for float, 16 for double) -- half the vector-memory instructions of the
        // element-wise epilogue, which is what bounds the fused step at small block sizes.
        if (whole_block) {
          const int q = lane & 1;
          const int64_t rb = row - q;
          struct alignas(sizeof(T)) V2 { T e[2]; };
          for (; j + 1 < jend; j += 2) {
            const T a0 = row_product(boff), a1 = row_product(boff + (unsigned)F * (unsigned)sizeof(T));
            boff += 2u * (unsigned)F * (unsigned)sizeof(T);
            const T got = dpp_quad<T, 0xB1>(q ? a0 : a1);           // the neighbour's result for MY vector
            const T t0 = q ? got : a0, t1 = q ? a1 : got;           // (A y) at rows rb, rb + 1 of vector j + q
            const int64_t jq = j + q;
            V2 *pp = reinterpret_cast<V2 *>(Y + rb + jq * ldy);
            const V2 pv = *pp;
            const V2 bv = *reinterpret_cast<const V2 *>(cheb.B + rb + jq * cheb.ldb);
            const V2 yv = *reinterpret_cast<const V2 *>(X + rb + jq * ldx);
            V2 out;
            out.e[0] = add_of(add_of(scale_of(cheb.cy, yv.e[0]), scale_of(cheb.cp, pv.e[0])),
                              scale_of(cheb.cb, sub_of(bv.e[0], t0)));
            out.e[1] = add_of(add_of(scale_of(cheb.cy, yv.e[1]), scale_of(cheb.cp, pv.e[1])),
                              scale_of(cheb.cb, sub_of(bv.e[1], t1)));
            *pp = out;
          }
        }
      }
      for (; j < jend; ++j) {
        asm volatile("" : "+s"(boff));
        T acc = zero_of(T{});
#pragma unroll
        for (int t = 0; t < WMAX; ++t) fma_acc(acc, v[t], *reinterpret_cast<const T *>(ldsb + (entry_offset(t) + boff)));
        if (row < n_rows) {
          if constexpr (!CHEB) {
            nt_store(Y + row + (int64_t)j * ldy, acc);
          } else {
            T *pp = Y + row + (int64_t)j * ldy;
            const T res = sub_of(cheb.B[row + (int64_t)j * cheb.ldb], acc);     // b - A y
            *pp = add_of(add_of(scale_of(cheb.cy, X[row + (int64_t)j * ldx]), scale_of(cheb.cp, *pp)),
                         scale_of(cheb.cb, res));
          }
        }
        boff += (unsigned)F * (unsigned)sizeof(T);
      }
    };
    if constexpr (PACK || EPL == 1 || (CHEB && sizeof(T) == 8)) {     // (the fp64 fused step spilled 52 bytes with two sets)
      // one register set only (the element-wise staging of unaligned layouts needs SMAX slots per
      // set): the loads of step s + 1 are in flight during the arithmetic of step s
      stage_load(0, stA);
      stage_write(0, stA);
      __syncthreads();
      for (int s = 0; s + 1 < nsteps; ++s) {
        stage_load(s + 1, stA);
        compute(s);
        stage_write(s + 1, stA);
        __syncthreads();
      }
      compute(nsteps - 1);
    } else {
      // Two register sets: the loads of step s + 2 are issued at the start of step s and written
      // to the LDS at the end of step s + 1, so they have two steps (and two barriers, which plain
      // loads survive) to arrive.
      VecU<T, EPL> stB[SLOTS];
      stage_load(0, stA);
      stage_load(nsteps > 1 ? 1 : 0, stB);
      stage_write(0, stA);
      __syncthreads();
      int s = 0;
      for (; s + 3 < nsteps; s += 2) {          // both prefetch targets exist
        stage_load(s + 2, stA);                 // B holds step s + 1
        compute(s);
        stage_write(s + 1, stB);
        __syncthreads();
        stage_load(s + 3, stB);                 // A holds step s + 2
        compute(s + 1);
        stage_write(s + 2, stA);
        __syncthreads();
      }
      const int left = nsteps - s;              // 1, 2 or 3 steps, nothing beyond them to prefetch
      if (left == 3) {
        stage_load(s + 2, stA);
        compute(s);
        stage_write(s + 1, stB);
        __syncthreads();
        compute(s + 1);
        stage_write(s + 2, stA);
        __syncthreads();
        compute(s + 2);
      } else if (left == 2) {
        compute(s);
        stage_write(s + 1, stB);
        __syncthreads();
        compute(s + 1);
      } else {
        compute(s);
      }
    }
    __syncthreads();                          // the next block's first stage overwrites the buffers
  }
}

// ------------------------------------------------------------------ Stacked blocks
// What bounds the windowed kernel on a 3-D stencil is the volume it stages, not the arithmetic: a 1024-row block of the
// 7-point Laplacian on 215^3 stages 3.4 elements per row and vector (its own rows + the y neighbours, and the two z planes
// in a window each), the 5-point one on 3152^2 3.25, a band matrix 1.06 -- and the three run at 1.36, 1.34 and 1.04 ms
// (profiles/r01_spmm_windowed.txt).  The z-plane windows of block b are the OWN rows of the blocks one grid plane further
// on: a workgroup that takes R such blocks together (thread l = row l of each of them) stages the shared windows once,
// (2 + R * 1.42) / R elements per row instead of 3.42.  The host picks the stacks on the window-overlap graph (greedy
// matching: the unstacked block that overlaps a block's windows most), so nothing here knows about grids; blocks without a
// partner run as stacks of one.  Same entry storage, staging plan and step pipeline as above (16-byte staging only: the
// caller checks that every group is in range); the two staging buffers are the CU's whole LDS.
constexpr int kStkR = 2;                           // row blocks per stack
constexpr int kStkBufBytes = 80 * 1024;            // per staging buffer

template <typename T, int R>
__global__ __launch_bounds__(1024) void well_stack_kernel(const WellMeta *__restrict__ meta,
                                                          const int32_t *__restrict__ member,
                                                          const int32_t *__restrict__ gsrc,
                                                          const uint16_t *__restrict__ idx,
                                                          const T *__restrict__ vals, int64_t n_rows,
                                                          const int32_t *__restrict__ sched, int64_t sched_len,
                                                          const T *__restrict__ X, int64_t ldx,
                                                          T *__restrict__ Y, int64_t ldy, int m, int cps_cap) {
  constexpr int WMAX = 8;
  constexpr int EPL = 16 / (int)sizeof(T);         // elements per 16-byte piece = 64-column groups per wave load
  constexpr int GPS = 16 * EPL;                    // groups one staging slot of the 16 waves covers
  constexpr int BUFG = kStkBufBytes / (64 * (int)sizeof(T));     // groups per buffer
  constexpr int SLOTS = BUFG / GPS;
  static_assert(SLOTS * GPS == BUFG, "a buffer is a whole number of slots");
  extern __shared__ __align__(16) char ldsb[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const unsigned lane16 = (unsigned)lane * 16u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int64_t pos = blockIdx.x; pos < sched_len; pos += gridDim.x) {
    const int64_t sb = sched[pos];
    if (sb < 0) continue;
    const WellMeta mt = meta[sb];
    const int ng = mt.width_ng >> 8;               // >= 8, a multiple of 8
    const int F = ng * 64;
    int cps = BUFG / ng;
    if (cps > cps_cap) cps = cps_cap;
    if (cps > m) cps = m;
    const int nsteps = (m + cps - 1) / cps;
    cps = (m + nsteps - 1) / nsteps;
    const int G = ng * cps;
    // entries: values, and byte offsets in a staged vector -- for 8-byte types two 16-bit positions per register,
    // unpacked at use (16 more registers spill)
    constexpr bool PACK = sizeof(T) >= 8;
    int64_t row[R];
    T v[R][WMAX];
    unsigned ixb[R][PACK ? WMAX / 2 : WMAX];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int2 mr = reinterpret_cast<const int2 *>(member)[sb * R + r];       // first row, rows (0: no such member)
      row[r] = tid < mr.y ? (int64_t)mr.x + tid : n_rows;             // (no such row: its slots hold zeros)
      unsigned px[WMAX];
      well_load_entries<T, WMAX>(vals, idx, mt.eoff + 8 * r, tid, WMAX, v[r], px);
#pragma unroll
      for (int t = 0; t < WMAX; ++t) {
        if constexpr (!PACK) ixb[r][t] = px[t] * (unsigned)sizeof(T);
        else if (t & 1) ixb[r][t / 2] |= px[t] << 16;
        else ixb[r][t / 2] = px[t];
      }
    }
    int sbase[SLOTS], scc[SLOTS], scol[SLOTS];
    constexpr int LPG = 64 / EPL;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      int q = (wave + 16 * i) * EPL;
      if (q > G - EPL) q = G - EPL;                // surplus slots repeat the last groups
      const int cc = q / ng, g = q - cc * ng;
      scc[i] = __builtin_amdgcn_readfirstlane(cc);
      sbase[i] = __builtin_amdgcn_readfirstlane((cc * F + g * 64) * (int)sizeof(T));
      scol[i] = gsrc[mt.goff + g + lane / LPG] + (lane % LPG) * EPL;
    }
    auto stage_load = [&](int s, VecU<T, EPL> (&st)[SLOTS]) {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        int j = s * cps + scc[i];
        if (j > m - 1) j = m - 1;
        st[i] = *reinterpret_cast<const VecU<T, EPL> *>(X + (int64_t)j * ldx + scol[i]);
      }
    };
    auto stage_write = [&](int s, const VecU<T, EPL> (&st)[SLOTS]) {
      unsigned boff = (unsigned)(s & 1) * (unsigned)kStkBufBytes;
      asm volatile("" : "+s"(boff));
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        VecA<T, EPL> val;
#pragma unroll
        for (int k = 0; k < EPL; ++k) val.e[k] = st[i].e[k];
        *reinterpret_cast<VecA<T, EPL> *>(ldsb + (lane16 + (boff + (unsigned)sbase[i]))) = val;
      }
    };
    auto compute = [&](int s) {
      unsigned boff = (unsigned)(s & 1) * (unsigned)kStkBufBytes;
      int j = s * cps;
      const int jend = (j + cps < m) ? j + cps : m;
      for (; j < jend; ++j) {
        asm volatile("" : "+s"(boff));
#pragma unroll
        for (int r = 0; r < R; ++r) {
          T acc = zero_of(T{});
#pragma unroll
          for (int t = 0; t < WMAX; ++t) {
            if constexpr (!PACK) {
              fma_acc(acc, v[r][t], *reinterpret_cast<const T *>(ldsb + (ixb[r][t] + boff)));
            } else if ((t & 1) == 0) {
              unsigned w = ixb[r][t / 2];
              asm volatile("" : "+v"(w));          // (unpacked per vector: hoisted out of the loop the offsets take 16 registers again)
              fma_acc(acc, v[r][t], *reinterpret_cast<const T *>(ldsb + ((w & 0xffffu) * (unsigned)sizeof(T) + boff)));
              fma_acc(acc, v[r][t + 1], *reinterpret_cast<const T *>(ldsb + ((w >> 16) * (unsigned)sizeof(T) + boff)));
            }
          }
          if (row[r] < n_rows) nt_store(Y + row[r] + (int64_t)j * ldy, acc);
        }
        boff += (unsigned)F * (unsigned)sizeof(T);
      }
    };
    if constexpr (sizeof(T) >= 8) {
      // one register set (two spill 36 bytes in fp64): the loads of step s + 1 are in flight during the arithmetic of step s
      VecU<T, EPL> stA[SLOTS];
      stage_load(0, stA);
      stage_write(0, stA);
      __syncthreads();
      for (int s = 0; s + 1 < nsteps; ++s) {
        stage_load(s + 1, stA);
        compute(s);
        stage_write(s + 1, stA);
        __syncthreads();
      }
      compute(nsteps - 1);
      __syncthreads();
      continue;
    }
    VecU<T, EPL> stA[SLOTS], stB[SLOTS];
    stage_load(0, stA);
    stage_load(nsteps > 1 ? 1 : 0, stB);
    stage_write(0, stA);
    __syncthreads();
    int s = 0;
    for (; s + 3 < nsteps; s += 2) {
      stage_load(s + 2, stA);
      compute(s);
      stage_write(s + 1, stB);
      __syncthreads();
      stage_load(s + 3, stB);
      compute(s + 1);
      stage_write(s + 2, stA);
      __syncthreads();
    }
    const int left = nsteps - s;
    if (left == 3) {
      stage_load(s + 2, stA);
      compute(s);
      stage_write(s + 1, stB);
      __syncthreads();
      compute(s + 1);
      stage_write(s + 2, stA);
      __syncthreads();
      compute(s + 2);
    } else if (left == 2) {
      compute(s);
      stage_write(s + 1, stB);
      __syncthreads();
      compute(s + 1);
    } else {
      compute(s);
    }
    __syncthreads();
  }
}

// The same stacks with the staging done by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into the LDS at a
// wave-uniform base + 16 * lane, which is exactly the staged image's layout) into a ring of four slots of one vector each:
// no staging registers (the register-staged kernel above spills at R = 2 in fp64), no ds_write pass, and three vectors in
// flight instead of two steps.  One step = one vector:  wait until its DMAs have landed (counted vmcnt: the DMAs of the
// two later vectors stay in flight);  barrier (every wave's pieces are there, and every
// wave is done with the slot the next DMA overwrites);  issue the DMAs of the vector three steps ahead;  row products
// of this vector out of its slot and the R stores.  The compiler does not count asm loads, so every wait on the DMAs is
// written here, from the number of DMAs this wave has issued after them (LDS-DMAs complete in order among themselves);
// a stack with a ragged member (the last row block of the matrix) waits for vmcnt(0) instead.
// Timing-only builds (DBG, profiles/r03_spmm_stack.txt) put the floor of this access pattern -- DMAs, stores and the
// entries, no LDS reads -- at 1.09 of the kernel's 1.15-1.20 ms on lap3d 215^3 fp64; a version whose ring ran on across
// the stack boundaries, with the next stack's entries prefetched into a second register set, measured the same and is
// not kept.
template <int N> __device__ __forceinline__ void wait_vm_le() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// wait until at most n vector-memory operations of this wave are outstanding (s_waitcnt takes an immediate)
__device__ __forceinline__ void wait_vm_outstanding(int n) {
#define RLH_WC(k) case k: wait_vm_le<k>(); break;
#define RLH_WC8(k) RLH_WC(k) RLH_WC(k + 1) RLH_WC(k + 2) RLH_WC(k + 3) RLH_WC(k + 4) RLH_WC(k + 5) RLH_WC(k + 6) RLH_WC(k + 7)
  switch (n) {
    RLH_WC8(0) RLH_WC8(8) RLH_WC8(16)
    default: wait_vm_le<0>(); break;              // (never too little)
  }
#undef RLH_WC8
#undef RLH_WC
}

constexpr int kStkLdsBytes = 160 * 1024;
// slots of the ring: four of 40 KB, or -- 16-byte elements, whose image of one vector is as large as two of the others'
// -- two of 80 KB (the bytes in flight are what counts, not the vectors)
template <typename T> struct StkRing { static constexpr int NB = sizeof(T) >= 16 ? 2 : 4, SLOT = kStkLdsBytes / NB; };

// "this value is needed now" for the compiler's wait placement
__device__ __forceinline__ void stk_touch(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void stk_touch(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void stk_touch(c32 &v) { asm volatile("" : "+v"(v.re), "+v"(v.im)); }
__device__ __forceinline__ void stk_touch(c64 &v) { asm volatile("" : "+v"(v.re), "+v"(v.im)); }

// one store instruction per value (the waits count instructions)
__device__ __forceinline__ void stk_store(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void stk_store(double *p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void stk_store(c32 *p, c32 v) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __builtin_nontemporal_store(f32x2{v.re, v.im}, reinterpret_cast<f32x2 *>(p));
}
__device__ __forceinline__ void stk_store(c64 *p, c64 v) {
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  __builtin_nontemporal_store(f64x2{v.re, v.im}, reinterpret_cast<f64x2 *>(p));
}

// (DBG: timing-only builds for tools/stack_bench.py -- 1: no DMA, 2: no LDS reads / arithmetic, 4: no stores: wrong
// results; 8: plain instead of non-temporal stores)
// CHEB (16-byte elements): the fused Chebyshev step  P = cy y + cp P + cb (B - A y)  on the same ring -- X is y (staged),
// Y is P (read and written in place), y[row] comes out of the staged image (slot 7 of every row holds its own column's
// position: stk_self).  P and B are ordinary loads, and ordinary loads and LDS-DMAs do not retire in order with each other
// (see the waits below), so this variant never counts: every wait is for everything.  With the two slots of a 16-byte ring
// that costs nothing in depth -- the DMA of vector j + 1 and the operands of vector j are in flight together while the
// products of vector j are formed, one wait, then the update and its stores.
template <typename T, int R, int LD, int DBG = 0, bool CHEB = false>
__global__ __launch_bounds__(1024) void well_stack_dma_kernel(const WellMeta *__restrict__ meta,
                                                              const int32_t *__restrict__ member,
                                                              const int32_t *__restrict__ gsrc,
                                                              const uint16_t *__restrict__ idx,
                                                              const T *__restrict__ vals, const int32_t *__restrict__ pat,
                                                              const uint16_t *__restrict__ dtab, int64_t n_rows,
                                                              const int32_t *__restrict__ sched, int64_t sched_len,
                                                              const T *__restrict__ X, int64_t ldx, int64_t n_own,
                                                              const T *__restrict__ H, int64_t ldh,
                                                              T *__restrict__ Y, int64_t ldy, int m, ChebArgs<T> cheb) {
  constexpr int WMAX = 8;
  constexpr int NB = StkRing<T>::NB, D = NB - 1;   // D vectors ahead
  static_assert(!CHEB || NB == 2, "the fused step drains the ring every vector: two slots only");
  constexpr int SLOT = StkRing<T>::SLOT;
  constexpr int EPL = 16 / (int)sizeof(T);         // elements per 16-byte piece
  constexpr int LPG = 64 / EPL;
  static_assert((D - 1) * LD + D * R < 24, "wait_vm_outstanding's cases");      // (10 + 6)
  extern __shared__ __align__(16) char ldsb[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int64_t pos = blockIdx.x; pos < sched_len; pos += gridDim.x) {
    const int64_t sb = sched[pos];
    if (sb < 0) continue;
    const WellMeta mt = meta[sb];
    const int ng = mt.width_ng >> 8;               // a multiple of 8, <= SLOT / (64 sizeof(T)) and <= 16 EPL LD
    // entries: values, and two 16-bit positions per register as stored (unpacked at use)
    int row[R];                                    // this thread's row of member r (-1: none), 32 bits: n < 2^31
    T v[R][WMAX];
    unsigned ixb[R][WMAX / 2];
    bool whole = true;                             // every member present has all its 1024 rows (workgroup-uniform)
    int nmem = 0;                                  // members present: the first nmem (a stack of one: the plain block)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int2 mr = reinterpret_cast<const int2 *>(member)[sb * R + r];       // first row, rows (0: no such member)
      if (mr.y > 0) {
        nmem = r + 1;
        whole = whole && mr.y == kWellRows;
      }
      row[r] = tid < mr.y ? mr.x + tid : -1;
      unsigned px[WMAX];
      // (not for 16-byte elements: the second path costs complex128 six spilled registers, and scratch traffic would
      // sit in the same counter as the DMAs)
      if (sizeof(T) < 16 && pat) well_load_entries_pat<T, WMAX>(vals, pat, dtab, idx, mt.eoff + 8 * r, sb * R + r, tid, v[r], px);
      else well_load_entries<T, WMAX>(vals, idx, mt.eoff + 8 * r, tid, WMAX, v[r], px);
#pragma unroll
      for (int t = 0; t < WMAX; t += 2) ixb[r][t / 2] = px[t] | (px[t + 1] << 16);
    }
    // a vector's image is ng / EPL pieces of 16 bytes per lane, moved by the first nw waves, LD pieces each at most: wave
    // w takes pieces w, w + nw, ...: `mine` of them (the waits below count per wave; a wave with many pieces keeps more
    // of the later vectors in flight behind the interleaved stores than sixteen waves with one or two each)
    int scol[LD];                                  // first column this lane fetches, per piece
    const int npieces = ng / EPL;
    const int nw = __builtin_amdgcn_readfirstlane((npieces + LD - 1) / LD);
    const int mine = __builtin_amdgcn_readfirstlane(wave < nw ? (npieces - wave + nw - 1) / nw : 0);
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      int q = (wave + nw * i) * EPL;
      if (q > ng - EPL) q = ng - EPL;              // (a piece this wave does not have: any valid address)
      scol[i] = gsrc[mt.goff + q + lane / LPG] + (lane % LPG) * EPL;
    }
    // the entries and the staging plan are in their registers before the first DMA is issued: the compiler's own wait for
    // them would otherwise come at their first use, inside the pipeline, as a vmcnt(0) that drains the ring
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int t = 0; t < WMAX; ++t) stk_touch(v[r][t]);
#pragma unroll
      for (int t = 0; t < WMAX / 2; ++t) asm volatile("" : "+v"(ixb[r][t]));
    }
#pragma unroll
    for (int i = 0; i < LD; ++i) asm volatile("" : "+v"(scol[i]));
    auto issue = [&](int j) {                      // the DMAs of vector j into slot j % NB
      const unsigned slot = (unsigned)(j & (NB - 1)) * (unsigned)SLOT;
      const T *src = X + (int64_t)j * ldx;
      const T *hsrc = H ? H + (int64_t)j * ldh - n_own : src;   // halo row of column c: hsrc[c] (pieces never lie across n_own)
#pragma unroll
      for (int i = 0; i < LD; ++i) {
        if (i >= mine || (DBG & 1)) break;
        const T *g = (scol[i] < n_own ? src : hsrc) + scol[i];
        unsigned dst = slot + (unsigned)(wave + nw * i) * 1024u, keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
      }
    };
    auto compute = [&](int j) {
      unsigned boff = (unsigned)(j & (NB - 1)) * (unsigned)SLOT;
      asm volatile("" : "+s"(boff));
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r >= nmem) break;
        T acc = zero_of(T{});
#pragma unroll
        for (int t = 0; t < WMAX; t += 2) {
          unsigned w = ixb[r][t / 2];
          asm volatile("" : "+v"(w));              // (unpacked per vector: hoisted out of the loop the offsets take 16 registers)
          if constexpr ((DBG & 2) != 0) acc = (t == (j & 6)) ? v[r][t] : acc;
          else {
            fma_acc(acc, v[r][t], *reinterpret_cast<const T *>(ldsb + ((w & 0xffffu) * (unsigned)sizeof(T) + boff)));
            fma_acc(acc, v[r][t + 1], *reinterpret_cast<const T *>(ldsb + ((w >> 16) * (unsigned)sizeof(T) + boff)));
          }
        }
        // (non-temporal stores: 1.107 against 1.140 ms with plain ones, variants taking turns in one process)
        if constexpr ((DBG & 4) != 0) { if (ixb[r][0] == 0xfffffffeu) Y[(int64_t)row[r] + (int64_t)j * ldy] = acc; }
        else if constexpr ((DBG & 8) != 0) { if (whole || row[r] >= 0) Y[(int64_t)row[r] + (int64_t)j * ldy] = acc; }
        else if (whole || row[r] >= 0) stk_store(Y + (int64_t)row[r] + (int64_t)j * ldy, acc);
      }
    };
    // every wave is done with the previous stack's slots (its last vectors were read after the last barrier)
    __builtin_amdgcn_s_barrier();
    if constexpr (CHEB) {
      issue(0);
      for (int j = 0; j < m; ++j) {
        if (j == 0) wait_vm_outstanding(0);        // (later vectors: drained by the wait below)
        // every wave's pieces of vector j have landed, and every wave has formed its products of vector j - 1: the
        // other slot is free
        __builtin_amdgcn_s_barrier();
        if (j + 1 < m) issue(j + 1);
        unsigned boff = (unsigned)(j & (NB - 1)) * (unsigned)SLOT;
        asm volatile("" : "+s"(boff));
        // one member at a time -- products, operands, wait, update: the kernel sits at the 128 registers a wave of a
        // 1024-thread workgroup has (the entries alone are 16 per row), and a spilled operand is reloaded behind a wait of
        // its own.  Slot 7 is the row's own column with value 0: y[row], not part of the product.  The first member's
        // wait covers the ring's DMA, which has been in flight since before its products.
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r >= nmem) break;
          T acc = zero_of(T{});
#pragma unroll
          for (int t = 0; t < WMAX - 2; t += 2) {
            unsigned w = ixb[r][t / 2];
            asm volatile("" : "+v"(w));
            fma_acc(acc, v[r][t], *reinterpret_cast<const T *>(ldsb + ((w & 0xffffu) * (unsigned)sizeof(T) + boff)));
            fma_acc(acc, v[r][t + 1], *reinterpret_cast<const T *>(ldsb + ((w >> 16) * (unsigned)sizeof(T) + boff)));
          }
          unsigned w = ixb[r][WMAX / 2 - 1];
          asm volatile("" : "+v"(w));
          fma_acc(acc, v[r][WMAX - 2], *reinterpret_cast<const T *>(ldsb + ((w & 0xffffu) * (unsigned)sizeof(T) + boff)));
          const T yv = *reinterpret_cast<const T *>(ldsb + ((w >> 16) * (unsigned)sizeof(T) + boff));
          T out = sub_of(scale_of(cheb.cy, yv), scale_of(cheb.cb, acc));
          const bool live = whole || row[r] >= 0;
          T *pp = Y + (int64_t)row[r] + (int64_t)j * ldy;
          T pv = zero_of(T{}), bv = zero_of(T{});
          if (live) {
            pv = *pp;
            bv = cheb.B[(int64_t)row[r] + (int64_t)j * cheb.ldb];
          }
          wait_vm_outstanding(0);
          stk_touch(pv);
          stk_touch(bv);
          out = add_of(add_of(out, scale_of(cheb.cp, pv)), scale_of(cheb.cb, bv));
          if (live) stk_store(pp, out);
        }
      }
      continue;
    }
    for (int j = 0; j < D && j < m; ++j) issue(j);
    for (int j = 0; j < m; ++j) {
      // DMAs issued after those of vector j: the later vectors in flight.  The stores of the last steps are younger too,
      // but they are NOT counted as allowed: LDS-DMAs and ordinary vector-memory operations do not complete in order with
      // each other (seen with loads in the bfloat16 kernel below), and a store that has completed early would let an
      // unfinished DMA through the count; if they are still outstanding this waits for one more vector than necessary
      // (RLH_SPMM_STACK_DBG & 16 counts them as before: the same time to within the noise).
      const int later = m - 1 - j < D - 1 ? m - 1 - j : D - 1;
      // (a short member -- the last block of the matrix, or the plane-aligned blocks -- changes nothing here: the count
      // rests on the DMAs alone, whatever stores a wave does or does not issue)
      wait_vm_outstanding(((DBG & 1) ? 0 : later * mine) + ((DBG & 16) && whole ? (j < D ? j : D) * nmem : 0));
      __builtin_amdgcn_s_barrier();
      if (j + D < m) issue(j + D);
      compute(j);
    }
  }
}

// ------------------------------------------------------------------ bfloat16 Chebyshev step
// The fused three-term step with the three blocks y, p, b stored as bfloat16 (the preconditioner
// only steers the search directions: on lap3d 64^3, degree 24, the iteration count goes from 27 to
// 28 with bfloat16 storage) and all arithmetic in float32 against the float32 windowed operator.
// Per element the step then moves 2-byte values: 4.5 x 2 bytes instead of 4.5 x 4.
//  * staging: a 16-byte piece is 8 elements, one wave load covers 8 groups (1 KiB); the LDS image
//    is bfloat16 (an entry's position times 2 is its byte offset), 8 vectors per step;
//  * the row results of 8 vectors go through a wave-private LDS tile [vector][row] and come back
//    as 8 consecutive rows of ONE vector per lane, so y, p, b are read and p is written as 16-byte
//    pieces (one wave access = the wave's 64 rows x 8 vectors).
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {           // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
// p' = cy y + cp p + cb (b - t) with the roundings pinned (left to the compiler, the two kernels that share this step
// contracted it differently: one bfloat16 ulp apart in 3 of 10^5 elements)
__device__ __forceinline__ float cheb_update(float cy, float y, float cp, float p, float cb, float b, float t) {
  return __fmaf_rn(cy, y, __fmaf_rn(cp, p, __fmul_rn(cb, __fsub_rn(b, t))));
}
struct alignas(16) Bf8 { unsigned short e[8]; };
struct alignas(8) Bf8U { unsigned short e[8]; };                           // staging piece: groups start on multiples of 4 columns

constexpr int kBfCps = 8;                          // vectors per step (fewer when 8 staged images exceed a buffer)
constexpr int kBfBufBytes = 56 * 1024;             // per staging buffer (2 of them + 32 KiB of wave tiles)

template <int WMAX>
__global__ __launch_bounds__(1024) void well_cheb_bf16_kernel(const WellMeta *__restrict__ meta,
                                                              const int32_t *__restrict__ gsrc,
                                                              const uint16_t *__restrict__ idx,
                                                              const float *__restrict__ vals, int64_t n_rows,
                                                              const int32_t *__restrict__ sched, int64_t sched_len,
                                                              const unsigned short *__restrict__ Yk, int64_t ldy,
                                                              int64_t n_own, const unsigned short *__restrict__ H,
                                                              int64_t ldh,
                                                              unsigned short *__restrict__ P, int64_t ldp,
                                                              const unsigned short *__restrict__ B, int64_t ldb,
                                                              int m, float cy, float cp, float cb) {
  constexpr int SLOTS = 4;                         // ceil(56 KiB / 16 B / 1024 threads)
  __shared__ __attribute__((aligned(16))) char ldsb[2 * kBfBufBytes + 16 * kBfCps * 64 * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float *tile = reinterpret_cast<float *>(ldsb + 2 * kBfBufBytes) + wave * (kBfCps * 64);   // [vector][row] of this wave
  for (int64_t pos = blockIdx.x; pos < sched_len; pos += gridDim.x) {
    const int64_t b = sched[pos];
    if (b < 0) continue;
    const WellMeta mt = meta[b];
    const int width = mt.width_ng & 255;
    const int ng = mt.width_ng >> 8;               // multiple of 8, <= 128
    const int F2 = ng * 128;                       // bytes per staged vector
    int cps = kBfBufBytes / F2;                    // >= 3
    if (cps > kBfCps) cps = kBfCps;
    if (cps > m) cps = m;
    const int nsteps = (m + cps - 1) / cps;
    const int G = ng * cps;                        // staging groups per step (G / 8 <= 56 pieces: 4 per wave)
    const int64_t row0 = b * kWellRows + (int64_t)wave * 64;      // first row of this wave
    const int64_t row = row0 + lane;
    float v[WMAX];
    unsigned ixb[WMAX];
    well_load_entries<float, WMAX>(vals, idx, mt.eoff, tid, width, v, ixb);
#pragma unroll
    for (int t = 0; t < WMAX; ++t) ixb[t] *= 2u;        // byte offset in the bfloat16 image
    // staging slots: 8 groups (one 16-byte piece per lane) each
    int sbase[SLOTS], scc[SLOTS], scol[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      int q = (wave + 16 * i) * 8;
      if (q > G - 8) q = G - 8;                    // surplus slots repeat the last groups
      const int cc = q / ng, g = q - cc * ng;
      scc[i] = __builtin_amdgcn_readfirstlane(cc);
      sbase[i] = __builtin_amdgcn_readfirstlane(cc * F2 + g * 128);
      scol[i] = gsrc[mt.goff + g + lane / 8] + (lane % 8) * 8;
    }
    Bf8U stA[SLOTS], stB[SLOTS];
    auto stage_load = [&](int s, Bf8U (&st)[SLOTS]) {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        int j = s * cps + scc[i];
        if (j > m - 1) j = m - 1;
        const int c = scol[i];                      // (pieces never lie across n_own: the host checked)
        const unsigned short *src = c < n_own ? Yk + (int64_t)j * ldy + c : H + (int64_t)j * ldh + (c - n_own);
        st[i] = *reinterpret_cast<const Bf8U *>(src);
      }
    };
    auto stage_write = [&](int s, const Bf8U (&st)[SLOTS]) {
      unsigned boff = (unsigned)(s & 1) * (unsigned)kBfBufBytes;
      asm volatile("" : "+s"(boff));
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        Bf8 val;
#pragma unroll
        for (int k = 0; k < 8; ++k) val.e[k] = st[i].e[k];
        *reinterpret_cast<Bf8 *>(ldsb + ((unsigned)lane * 16u + (boff + (unsigned)sbase[i]))) = val;
      }
    };
    auto compute = [&](int s) {
      unsigned boff = (unsigned)(s & 1) * (unsigned)kBfBufBytes;
      const int j0 = s * cps;
      // after the row products, lane L takes vector j0 + L / 8, rows 8 (L % 8) .. + 7 of the wave's 64;
      // its pieces of p, b, y are requested FIRST (from clamped, always valid addresses) so that
      // their latency passes under the LDS work below instead of after it
      const int c8 = lane >> 3, rg = lane & 7;
      const int j = j0 + c8;
      const int64_t r8 = row0 + rg * 8;
      const bool mine = c8 < cps && j < m && r8 < n_rows;
      const bool whole = mine && r8 + 8 <= n_rows;
      const int jc = j < m ? j : m - 1;
      const int64_t rc = whole ? r8 : 0;
      unsigned short *pp = P + (int64_t)jc * ldp + rc;
      const Bf8 pv = *reinterpret_cast<const Bf8 *>(pp);           // (the host checked 16-byte alignment)
      const Bf8 bv = *reinterpret_cast<const Bf8 *>(B + (int64_t)jc * ldb + rc);
      const Bf8 yv = *reinterpret_cast<const Bf8 *>(Yk + (int64_t)jc * ldy + rc);
      // (A y)[row] for the step's vectors -> the wave's tile
#pragma unroll 1
      for (int c = 0; c < cps; ++c) {
        unsigned bo = boff + (unsigned)c * (unsigned)F2;
        asm volatile("" : "+s"(bo));
        float acc = 0.f;
        if (j0 + c < m) {
#pragma unroll
          for (int t = 0; t < WMAX; ++t)
            acc = fmaf(v[t], bf16_to_f32(*reinterpret_cast<const unsigned short *>(ldsb + (ixb[t] + bo))), acc);
        }
        tile[c * 64 + lane] = acc;
      }
      const int ct = c8 < cps ? c8 : 0;
      float t8[8];
      {
        const float4 lo = *reinterpret_cast<const float4 *>(tile + ct * 64 + rg * 8);
        const float4 hi = *reinterpret_cast<const float4 *>(tile + ct * 64 + rg * 8 + 4);
        t8[0] = lo.x; t8[1] = lo.y; t8[2] = lo.z; t8[3] = lo.w;
        t8[4] = hi.x; t8[5] = hi.y; t8[6] = hi.z; t8[7] = hi.w;
      }
      if (whole) {
        Bf8 out;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          out.e[k] = f32_to_bf16(cheb_update(cy, bf16_to_f32(yv.e[k]), cp, bf16_to_f32(pv.e[k]), cb, bf16_to_f32(bv.e[k]), t8[k]));
        *reinterpret_cast<Bf8 *>(pp) = out;
      } else if (mine) {                           // the last rows of the matrix
        unsigned short *pq = P + (int64_t)j * ldp + r8;
        const unsigned short *bq = B + (int64_t)j * ldb + r8, *yq = Yk + (int64_t)j * ldy + r8;
        for (int k = 0; k < 8 && r8 + k < n_rows; ++k)
          pq[k] = f32_to_bf16(cheb_update(cy, bf16_to_f32(yq[k]), cp, bf16_to_f32(pq[k]), cb, bf16_to_f32(bq[k]), t8[k]));
      }
    };
    stage_load(0, stA);
    stage_load(1, stB);                            // (clamped to the last vector if there is no step 1)
    stage_write(0, stA);
    __syncthreads();
    int s = 0;
    for (; s + 3 < nsteps; s += 2) {
      stage_load(s + 2, stA);
      compute(s);
      stage_write(s + 1, stB);
      __syncthreads();
      stage_load(s + 3, stB);
      compute(s + 1);
      stage_write(s + 2, stA);
      __syncthreads();
    }
    const int left = nsteps - s;
    if (left == 3) {
      stage_load(s + 2, stA);
      compute(s);
      stage_write(s + 1, stB);
      __syncthreads();
      compute(s + 1);
      stage_write(s + 2, stA);
      __syncthreads();
      compute(s + 2);
    } else if (left == 2) {
      compute(s);
      stage_write(s + 1, stB);
      __syncthreads();
      compute(s + 1);
    } else {
      compute(s);
    }
    __syncthreads();
  }
}


template <typename T, int WMAX, int EPL>
static int launch_well_we(const rlh_csr *h, int part, int64_t m, const T *X, int64_t ldx, int64_t n_own, const T *H,
                          int64_t ldh, T *Y, int64_t ldy, const ChebArgs<T> *cheb) {
  Context &c = ctx();
  // launch order: all blocks, or the blocks without / with halo columns (rlh_spmm_part)
  const int32_t *sched = part == 0 ? h->well_sched : h->well_sched_part[part - 1];
  const int64_t sched_len = part == 0 ? h->well_sched_len : h->well_sched_part_len[part - 1];
  const int64_t nb = part == 0 ? h->well_grid : h->well_grid_part[part - 1];   // one workgroup per CU, persistent
  if (nb == 0) return 0;
  const int cps_cap = env_int("RLH_SPMM_CPS", 8);       // vectors per step (tunable)
  if (cheb)
    hipLaunchKernelGGL((well_spmm_kernel<T, WMAX, EPL, true>), dim3((unsigned)nb), dim3(1024), 0, c.stream,
                       h->well_meta, h->well_gsrc, h->well_idx, (const T *)h->well_vals, h->n_rows, h->n_cols,
                       sched, sched_len, X, ldx, n_own, H, ldh, Y, ldy, (int)m,
                       cps_cap < 1 ? 1 : cps_cap, *cheb);
  else
    hipLaunchKernelGGL((well_spmm_kernel<T, WMAX, EPL, false>), dim3((unsigned)nb), dim3(1024), 0, c.stream,
                       h->well_meta, h->well_gsrc, h->well_idx, (const T *)h->well_vals, h->n_rows, h->n_cols,
                       sched, sched_len, X, ldx, n_own, H, ldh, Y, ldy, (int)m,
                       cps_cap < 1 ? 1 : cps_cap, ChebArgs<T>{});
  RLH_HIP(hipGetLastError());
  return 0;
}

template <typename T, int WMAX>
static int launch_well_w(const rlh_csr *h, int part, int64_t m, const T *X, int64_t ldx, int64_t n_own, const T *H,
                         int64_t ldh, T *Y, int64_t ldy, const ChebArgs<T> *cheb) {
  // 16-byte staging loads when every piece is whole: groups inside the column range, and with a
  // halo block no piece across the own / halo boundary (aligned groups, n_own a multiple of the
  // piece); the addresses themselves only need T's alignment
  constexpr int EPL = 16 / (int)sizeof(T);
  if constexpr (EPL > 1) {
    const int vec = env_int("RLH_SPMM_VEC", 1);         // 0: 8-byte staging (tunable)
    const bool whole = h->well_inbounds && (H == nullptr || n_own == h->n_cols || (h->well_aligned && n_own % EPL == 0));
    if (vec && whole) return launch_well_we<T, WMAX, EPL>(h, part, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
  }
  return launch_well_we<T, WMAX, 1>(h, part, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
}

// ------------------------------------------------------------------ bfloat16 Chebyshev step on the stacks
// The fused step of well_cheb_bf16_kernel on the stacked layout, staged by LDS-DMA like well_stack_dma_kernel.  At the
// block size the driver uses for ten eigenpairs (16 vectors) the kernel above spends its time at block boundaries -- two
// steps of eight vectors per block, 1.75 GB of algorithmic traffic in 457 us = 3.8 TB/s -- so here:
//  * two slots of VPS = 2 bfloat16 images (16 KB each: ng <= 128 groups);  one step = VPS vectors: wait for the step's
//    DMAs (issued during the step before), barrier, DMAs of the next step into the other slot, the row products of the
//    stack's two members into registers.  (One vector per step with a ring of 2, 4 or 6 slots: 380 us; two per step: 363.)
//  * every eight vectors the results go through a wave-private tile [vector][row] and come back as eight consecutive
//    rows of one vector per lane, so that y, p, b are read and p is written as 16-byte pieces.  Those operands (three
//    quarters of the step's traffic) are LDS-DMAs too -- into a wave-private area, 1 KB per operand and member, issued at
//    the start of the group of eight: as ordinary loads they did NOT complete in order with the DMAs (a counted wait on
//    a DMA issued after them returned with their registers still in flight: NaNs at 215^3, none in the small tests),
//    and a vmcnt(0) for them would wait for the vectors in flight as well.
//  * a step's DMAs are issued during the step before, so the only DMAs younger than them are, at the first step of a
//    group, the group's 3 nmem operand DMAs (issued in the previous group's epilogue, ahead of its stores): that is what
//    the wait allows.
constexpr int kBfImageBytes = 16 * 1024;           // ng <= 128 groups of 64 two-byte elements
constexpr int kBfRingBytes = 64 * 1024;
constexpr int kBfOperandBytes = 6 * 1024;          // per wave: [member][p, b, y] x 64 lanes x 16 bytes (its first 2 KB double as the tile)

// (DBG: timing-only builds -- 1: no y DMAs, 2: no row products, 4: no operand DMAs / no update: wrong results)
template <int R, int VPS, int DBG = 0>
__global__ __launch_bounds__(1024) void well_stack_cheb_bf16_kernel(const WellMeta *__restrict__ meta,
                                                                    const int32_t *__restrict__ member,
                                                                    const int32_t *__restrict__ gsrc,
                                                                    const uint16_t *__restrict__ idx,
                                                                    const float *__restrict__ vals,
                                                                    const int32_t *__restrict__ pat,
                                                                    const uint16_t *__restrict__ dtab, int64_t n_rows,
                                                                    const int32_t *__restrict__ sched, int64_t sched_len,
                                                                    const unsigned short *__restrict__ Yk, int64_t ldy,
                                                                    int64_t n_own, const unsigned short *__restrict__ H,
                                                                    int64_t ldh, unsigned short *__restrict__ P, int64_t ldp,
                                                                    const unsigned short *__restrict__ B, int64_t ldb,
                                                                    int m, float cy, float cp, float cb) {
  constexpr int WMAX = 8, IMG = kBfImageBytes, SLOT = VPS * IMG;       // two slots of VPS images
  static_assert(R * 3 * 1024 <= kBfOperandBytes && 2 * SLOT <= kBfRingBytes && 8 % VPS == 0, "operand area; ring");
  extern __shared__ __align__(16) char ldsb[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned opbase = (unsigned)kBfRingBytes + (unsigned)wave * (unsigned)kBfOperandBytes;   // this wave's operand area
  const int c8 = lane >> 3, rg = lane & 7;         // epilogue: vector c8 of the group, rows 8 rg .. 8 rg + 7 of the wave's 64
  auto dma16 = [&](const void *g, unsigned dst) {  // 16 bytes per lane to the LDS at dst + 16 lane (dst wave-uniform)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  };
  for (int64_t pos = blockIdx.x; pos < sched_len; pos += gridDim.x) {
    const int64_t sb = sched[pos];
    if (sb < 0) continue;
    const WellMeta mt = meta[sb];
    const int ng = mt.width_ng >> 8;               // a multiple of 8, <= 128
    int64_t row0[R];                               // first of this wave's 64 rows in member r
    float v[R][WMAX];
    unsigned ixb[R][WMAX / 2];
    bool whole = true;
    int nmem = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int2 mr = reinterpret_cast<const int2 *>(member)[sb * R + r];       // first row (a multiple of 8 here), rows
      if (mr.y > 0) {
        nmem = r + 1;
        whole = whole && mr.y == kWellRows;
      }
      // (float32 operators keep the uniform 1024-row blocks: a short member can only be the last block of the matrix, so
      // "row < n_rows" below is "row is in the member")
      row0[r] = mr.y > 0 ? (int64_t)mr.x + (int64_t)wave * 64 : n_rows;
      unsigned px[WMAX];
      if (pat) well_load_entries_pat<float, WMAX>(vals, pat, dtab, idx, mt.eoff + 8 * r, sb * R + r, tid, v[r], px);
      else well_load_entries<float, WMAX>(vals, idx, mt.eoff + 8 * r, tid, WMAX, v[r], px);
#pragma unroll
      for (int t = 0; t < WMAX; t += 2) ixb[r][t / 2] = px[t] | (px[t + 1] << 16);
    }
    // one 16-byte piece per lane = 8 elements: a wave's DMA covers 8 groups; wave w moves piece w of the ng / 8
    const int mine = __builtin_amdgcn_readfirstlane(wave < ng / 8 ? 1 : 0);
    int scol;
    {
      int q = wave * 8;
      if (q > ng - 8) q = ng - 8;
      scol = gsrc[mt.goff + q + lane / 8] + (lane % 8) * 8;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int t = 0; t < WMAX; ++t) asm volatile("" : "+v"(v[r][t]));
#pragma unroll
      for (int t = 0; t < WMAX / 2; ++t) asm volatile("" : "+v"(ixb[r][t]));
    }
    asm volatile("" : "+v"(scol));
    auto issue = [&](int j, unsigned par) {        // the images of vectors j, j + 1 (clamped) into slot `par`
      if (!mine || (DBG & 1)) return;
#pragma unroll
      for (int u = 0; u < VPS; ++u) {
        const int jv = j + u < m ? j + u : m - 1;
        dma16((scol < n_own ? Yk + (int64_t)jv * ldy : H + (int64_t)jv * ldh - n_own) + scol,
              par * (unsigned)SLOT + (unsigned)u * (unsigned)IMG + (unsigned)wave * 1024u);
      }
    };
    __builtin_amdgcn_s_barrier();                  // every wave is done with the previous stack's slots
    unsigned par = 0;                              // slot of the step in hand
    issue(0, 0);
    // this lane's operands of the update of the group that starts at vector j0: rows r8 .. r8 + 7 of vector j0 + c8
    // (clamped where the lane has none), 3 nmem DMAs per wave
    auto operands = [&](int j0) {
      const int jv = j0 + c8;
      const int jc = jv < m ? jv : m - 1;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r >= nmem) break;
        const int64_t r8 = row0[r] + rg * 8;
        const int64_t rc = r8 + 8 <= n_rows ? r8 : 0;
        if constexpr ((DBG & 4) != 0) continue;
        dma16(P + (int64_t)jc * ldp + rc, opbase + (unsigned)(r * 3 + 0) * 1024u);
        dma16(B + (int64_t)jc * ldb + rc, opbase + (unsigned)(r * 3 + 1) * 1024u);
        dma16(Yk + (int64_t)jc * ldy + rc, opbase + (unsigned)(r * 3 + 2) * 1024u);
      }
    };
    operands(0);
    for (int j0 = 0; j0 < m; j0 += 8) {
      const int cnt = m - j0 < 8 ? m - j0 : 8;     // vectors of this group
      const int jv = j0 + c8;
      const bool act = c8 < cnt;
      float acc[R][8];
#pragma unroll
      for (int jj = 0; jj < 8; jj += VPS) {
#pragma unroll
        for (int u = 0; u < VPS; ++u)
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r][jj + u] = 0.f;
        if (jj < cnt) {
          const int j = j0 + jj;
          // (a wave that moves no image has nothing to wait for here -- and must not: a wait would make it, and through
          // the barrier every wave, sit out the acknowledgement of its last stores)
          if (!whole) wait_vm_le<0>();
          else if (mine) wait_vm_outstanding(jj == 0 && !(DBG & 4) ? 3 * nmem : 0);
          __builtin_amdgcn_s_barrier();
          const int jn = jj + VPS < cnt ? j + VPS : j0 + 8;          // first vector of the next step
          if (jn < m) issue(jn, par ^ 1u);
          unsigned boff = par * (unsigned)SLOT;
          asm volatile("" : "+s"(boff));
#pragma unroll
          for (int u = 0; u < VPS; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
              if (r >= nmem) break;
              float a = 0.f;
              if constexpr ((DBG & 2) != 0) a = v[r][(jj + u) & 7];
              else
#pragma unroll
              for (int t = 0; t < WMAX; t += 2) {
                unsigned w = ixb[r][t / 2];
                asm volatile("" : "+v"(w));
                a = fmaf(v[r][t], bf16_to_f32(*reinterpret_cast<const unsigned short *>(ldsb + ((w & 0xffffu) * 2u + boff))), a);
                a = fmaf(v[r][t + 1], bf16_to_f32(*reinterpret_cast<const unsigned short *>(ldsb + ((w >> 16) * 2u + boff))), a);
              }
              acc[r][jj + u] = a;                   // (a vector past the block's last: computed on a repeated image, not used)
            }
            boff += (unsigned)IMG;
          }
          par ^= 1u;
        }
      }
      // the operands have landed with the DMAs of the group's last step, issued after them -- unless the group has one
      // step only (or this wave issues no vector DMAs)
      if (cnt <= VPS || !mine || !whole) wait_vm_le<0>();
      // the update of this group, member by member; the member's operands first (their area doubles as the tile)
      rlh_u32x4e outs[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r >= nmem) break;
        if constexpr ((DBG & 4) != 0) { if (acc[r][0] == 1.2345e30f) P[0] = 0; continue; }
        union { rlh_u32x4e u; unsigned short e[8]; } pu, bu, yu, ou;
        pu.u = *reinterpret_cast<const rlh_u32x4e *>(ldsb + opbase + (unsigned)(r * 3 + 0) * 1024u + (unsigned)lane * 16u);
        bu.u = *reinterpret_cast<const rlh_u32x4e *>(ldsb + opbase + (unsigned)(r * 3 + 1) * 1024u + (unsigned)lane * 16u);
        yu.u = *reinterpret_cast<const rlh_u32x4e *>(ldsb + opbase + (unsigned)(r * 3 + 2) * 1024u + (unsigned)lane * 16u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (read before the tile overwrites them)
        float *tile = reinterpret_cast<float *>(ldsb + opbase + (unsigned)(r * 3) * 1024u);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) tile[jj * 64 + lane] = acc[r][jj];
        float t8[8];
        {
          const float4 lo = *reinterpret_cast<const float4 *>(tile + c8 * 64 + rg * 8);
          const float4 hi = *reinterpret_cast<const float4 *>(tile + c8 * 64 + rg * 8 + 4);
          t8[0] = lo.x; t8[1] = lo.y; t8[2] = lo.z; t8[3] = lo.w;
          t8[4] = hi.x; t8[5] = hi.y; t8[6] = hi.z; t8[7] = hi.w;
        }
        const int64_t r8 = row0[r] + rg * 8;
        if (whole || r8 + 8 <= n_rows) {
#pragma unroll
          for (int k = 0; k < 8; ++k)
            ou.e[k] = f32_to_bf16(cheb_update(cy, bf16_to_f32(yu.e[k]), cp, bf16_to_f32(pu.e[k]), cb, bf16_to_f32(bu.e[k]), t8[k]));
          outs[r] = ou.u;
        } else if (act && r8 < n_rows) {           // the last rows of the matrix
          unsigned short *pq = P + (int64_t)jv * ldp + r8;
          const unsigned short *bq = B + (int64_t)jv * ldb + r8, *yq = Yk + (int64_t)jv * ldy + r8;
          for (int k = 0; k < 8 && r8 + k < n_rows; ++k)
            pq[k] = f32_to_bf16(cheb_update(cy, bf16_to_f32(yq[k]), cp, bf16_to_f32(pq[k]), cb, bf16_to_f32(bq[k]), t8[k]));
        }
      }
      // the next group's operands BEFORE this group's stores (this wave's reads of the area are done: its own LDS
      // operations complete in order, and t8 has been consumed): the stores are then the youngest operations of the wave and
      // no counted wait ever has to sit out their acknowledgement
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (j0 + 8 < m) operands(j0 + 8);
      if constexpr ((DBG & 4) == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r >= nmem) break;
          const int64_t r8 = row0[r] + rg * 8;
          if ((whole || r8 + 8 <= n_rows) && act) *reinterpret_cast<rlh_u32x4e *>(P + (int64_t)jv * ldp + r8) = outs[r];
        }
      }
    }
  }
}

template <typename T>
static int launch_stack(const rlh_csr *h, int part, int64_t m, const T *X, int64_t ldx, int64_t n_own, const T *H,
                        int64_t ldh, T *Y, int64_t ldy, bool dma, const ChebArgs<T> *cheb = nullptr) {
  Context &c = ctx();
  const int32_t *sched = part == 0 ? h->stk_sched : h->stk_sched_part[part - 1];
  const int64_t sched_len = part == 0 ? h->stk_sched_len : h->stk_sched_part_len[part - 1];
  const int grid = part == 0 ? h->stk_grid : h->stk_grid_part[part - 1];
  if (grid == 0) return 0;
  constexpr bool cplx = std::is_same<T, c32>::value || std::is_same<T, c64>::value;
  if (dma) {
    constexpr int EPL = 16 / (int)sizeof(T);
    const int32_t *pat = sizeof(T) < 16 && env_int("RLH_SPMM_STACK_PAT", 1) != 0 ? h->stk_pat : nullptr;   // the value dictionary, where the handle has one
    const int ld = 5;                                                  // 16-byte pieces per issuing wave and vector (<= 80 pieces)
#define RLH_STK_DMA(LD_, ...)                                                                                           \
    do {                                                                                                                \
      static bool attr_dma = false;                                                                                     \
      if (!attr_dma) {                                                                                                  \
        RLH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&well_stack_dma_kernel<T, kStkR, LD_ __VA_ARGS__>),  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kStkLdsBytes));                         \
        attr_dma = true;                                                                                                \
      }                                                                                                                 \
      hipLaunchKernelGGL((well_stack_dma_kernel<T, kStkR, LD_ __VA_ARGS__>), dim3((unsigned)grid), dim3(1024),          \
                         kStkLdsBytes, c.stream, h->stk_meta, h->stk_member, h->stk_gsrc, h->stk_idx,                   \
                         pat ? (const T *)h->stk_table : (const T *)h->stk_vals, pat, pat ? h->stk_dtab : nullptr,      \
                         h->n_rows, sched, sched_len, X, ldx, n_own, H, ldh, Y, ldy, (int)m,                            \
                         cheb ? *cheb : ChebArgs<T>{});                                                                 \
    } while (0)
    const int dbg = env_int("RLH_SPMM_STACK_DBG", 0);
    RLH_REQUIRE(h->stk_gmax <= 80 * EPL, "rlh_spmm: a stack of %d staging groups", h->stk_gmax);   // (a slot holds 40 or 80 pieces)
    if (cheb) {
      if constexpr (sizeof(T) == 16) {
        RLH_STK_DMA(5, , 0, true);
        RLH_HIP(hipGetLastError());
        return 0;
      } else {
        RLH_REQUIRE(false, "rlh_spmm_cheb: the stacked fused step exists for 16-byte elements only");
      }
    }
    if (dbg && std::is_same<T, double>::value) {
      if constexpr (std::is_same<T, double>::value) {
        switch (dbg) {
          case 1: RLH_STK_DMA(5, , 1); break;
          case 2: RLH_STK_DMA(5, , 2); break;
          case 3: RLH_STK_DMA(5, , 3); break;
          case 4: RLH_STK_DMA(5, , 4); break;
          case 5: RLH_STK_DMA(5, , 5); break;
          case 6: RLH_STK_DMA(5, , 6); break;
          case 7: RLH_STK_DMA(5, , 7); break;
          case 8: RLH_STK_DMA(5, , 8); break;
          default: RLH_STK_DMA(5, , 16); break;
        }
      }
    } else RLH_STK_DMA(5);
#undef RLH_STK_DMA
    RLH_HIP(hipGetLastError());
    return 0;
  }
  RLH_REQUIRE(part == 0 && H == nullptr, "rlh_spmm: the register-staged stack kernel takes the whole operator only");
  if constexpr (!cplx) {
    static bool attr = false;
    if (!attr) {
      RLH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&well_stack_kernel<T, kStkR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStkBufBytes));
      attr = true;
    }
    const int cps_cap = env_int("RLH_SPMM_CPS", 8);
    hipLaunchKernelGGL((well_stack_kernel<T, kStkR>), dim3((unsigned)h->stk_grid), dim3(1024), 2 * kStkBufBytes, c.stream,
                       h->stk_meta, h->stk_member, h->stk_gsrc, h->stk_idx, (const T *)h->stk_vals, h->n_rows, h->stk_sched,
                       h->stk_sched_len, X, ldx, Y, ldy, (int)m, cps_cap < 1 ? 1 : cps_cap);
    RLH_HIP(hipGetLastError());
  }
  return 0;
}

template <typename T>
static int launch_well(const rlh_csr *h, int part, int64_t m, const T *X, int64_t ldx, int64_t n_own, const T *H,
                       int64_t ldh, T *Y, int64_t ldy, const ChebArgs<T> *cheb) {
  // (rows of more than 8 entries and the complex types use the interleaved layout, spmm_wide.inc)
  return launch_well_w<T, 8>(h, part, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
}

static int well_split(rlh_csr *h, int64_t n_own);
static int stack_split(rlh_csr *h, int64_t n_own);

// Y = A X on the stacked layout where the handle has one and the call fits it: the whole operator on own columns with
// either kernel; with a halo block, or on the interior / boundary part of the rows (rlh_spmm_part), the LDS-DMA kernel,
// whose 16-byte pieces must then lie on one side of the own / halo boundary.
template <typename T>
static int stack_dispatch(rlh_csr *h, int part, int64_t m, const T *X, int64_t ldx, int64_t n_own, const T *H, int64_t ldh,
                          T *Y, int64_t ldy, bool *done, const ChebArgs<T> *cheb = nullptr) {
  constexpr bool cplx = std::is_same<T, c32>::value || std::is_same<T, c64>::value;
  constexpr int EPL = 16 / (int)sizeof(T);
  *done = false;
  if (h->stk_blocks == 0 || env_int("RLH_SPMM_STACK", 1) == 0) return 0;
  bool dma = h->stk_gmax * 64 * (int)sizeof(T) <= StkRing<T>::SLOT && (cplx || env_int("RLH_SPMM_STACK_DMA", 1) != 0);
  if (cplx && !dma) return 0;
  // float32, whole operator on own columns: the register-staged kernel (0.66 against 0.71 ms on lap3d 215^3 m = 32 -- with
  // one or two DMA pieces per wave the waits, which do not count the interleaved stores as allowed, keep too little in
  // flight; RLH_SPMM_STACK_DMA=2 takes the DMA kernel regardless)
  if (sizeof(T) == 4 && !cplx && part == 0 && H == nullptr && env_int("RLH_SPMM_STACK_DMA", 1) < 2) dma = false;
  // the last staging group may reach up to 7 columns past n_cols: inside the block's leading dimension, or not this way
  if (h->stk_overhang > 0 && (H != nullptr && n_own != h->n_cols ? ldh < h->n_cols - n_own + h->stk_overhang
                                                                  : ldx < h->n_cols + h->stk_overhang)) {
    static bool said = false;
    if (!said && env_int("RLH_SPMM_VERBOSE", 0) != 0) {
      said = true;
      fprintf(stderr, "rlh_spmm_part: stacks not used (overhang %d, ldh %lld, ldx %lld, n_cols %lld, n_own %lld)\n", h->stk_overhang,
              (long long)ldh, (long long)ldx, (long long)h->n_cols, (long long)n_own);
    }
    return 0;
  }
  if (part != 0 || H != nullptr) {
    if (!dma || (H != nullptr && n_own != h->n_cols && !(h->stk_aligned && n_own % EPL == 0))) {
      static bool said = false;
      if (!said && env_int("RLH_SPMM_VERBOSE", 0) != 0) {
        said = true;
        fprintf(stderr, "rlh_spmm_part: stacks not used (dma %d, aligned %d, n_own %lld, gmax %d, overhang %d)\n", (int)dma,
                h->stk_aligned, (long long)n_own, h->stk_gmax, h->stk_overhang);
      }
      return 0;
    }
    if (part != 0)
      if (int rc = stack_split(h, n_own)) return rc;
  }
  if (cheb && !dma) return 0;
  *done = true;
  return launch_stack<T>(h, part, m, X, ldx, n_own, H, ldh, Y, ldy, dma, cheb);
}

template <int DT>
static int spmm_impl(rlh_csr *h, int part, int64_t m, const void *X_, int64_t ldx, int64_t n_own, const void *H_,
                     int64_t ldh, void *Y_, int64_t ldy, const void *B_ = nullptr, int64_t ldb = 0, double cy = 0.0,
                     double cp = 0.0, double cb = 0.0) {
  using T = typename DType<DT>::T;
  constexpr int JTMAX = DType<DT>::cplx ? 16 : 32;
  const T *X = (const T *)X_, *H = (const T *)H_;
  T *Y = (T *)Y_;
  ChebArgs<T> cargs{(const T *)B_, ldb, cy, cp, cb};
  const ChebArgs<T> *cheb = B_ ? &cargs : nullptr;
  if (cheb == nullptr && (DType<DT>::cplx || h->well_blocks > 0)) {     // (a complex operator: the stacks beside its other layout)
    bool done = false;
    const int rc = stack_dispatch<T>(h, part, m, X, ldx, n_own, H, ldh, Y, ldy, &done);
    if (rc || done) return rc;
  }
  // the fused Chebyshev step of a 16-byte operator on the same stacks (rows of at most 7 entries that store their diagonal)
  if (cheb != nullptr && sizeof(T) == 16 && env_int("RLH_SPMM_STACK_CHEB", 1) != 0) {
    bool done = false;
    if (h->stk_self) {
      const int rc = stack_dispatch<T>(h, part, m, X, ldx, n_own, H, ldh, Y, ldy, &done, cheb);
      if (rc || done) return rc;
    }
    {
      static bool said = false;
      if (!said && env_int("RLH_SPMM_VERBOSE", 0) != 0) {
        said = true;
        fprintf(stderr, "rlh_spmm_cheb: stacked fused step not used (stacks %lld, self %d, part %d, halo %d, n_own %lld, n_cols %lld, gmax %d, aligned %d)\n",
                (long long)h->stk_blocks, h->stk_self, part, H != nullptr, (long long)n_own, (long long)h->n_cols, h->stk_gmax, h->stk_aligned);
      }
    }
    // (RLH_SPMM_STACK_CHEB=2: the tests' way of knowing which kernel ran)
    RLH_REQUIRE(env_int("RLH_SPMM_STACK_CHEB", 1) < 2, "rlh_spmm_cheb: the stacked fused step does not take this operator or call");
  }
  if (h->wide_blocks > 0) {
    switch (DT) {
      case RLH_S: return wide_spmm_s(h, part, m, X_, ldx, n_own, H_, ldh, Y_, ldy, B_, ldb, cy, cp, cb);
      case RLH_D: return wide_spmm_d(h, part, m, X_, ldx, n_own, H_, ldh, Y_, ldy, B_, ldb, cy, cp, cb);
      case RLH_C: return wide_spmm_c(h, part, m, X_, ldx, n_own, H_, ldh, Y_, ldy, B_, ldb, cy, cp, cb);
      default:    return wide_spmm_z(h, part, m, X_, ldx, n_own, H_, ldh, Y_, ldy, B_, ldb, cy, cp, cb);
    }
  }
  if constexpr (!DType<DT>::cplx) {
    if (h->well_blocks > 0) {
      if (part != 0)
        if (int rc = well_split(h, n_own)) return rc;
      return launch_well<T>(h, part, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
    }
  }
  if (part == 1) return 0;                    // sliced layout: everything happens in part 2
  const int jt_cap = env_int("RLH_SPMM_JT", 16);        // vectors per lane tile (tunable; 16 measured best at m = 32 fp64)
  if (m <= 4 || jt_cap <= 4) return launch_spmm<T, 4>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
  if (m <= 8 || jt_cap <= 8) return launch_spmm<T, 8>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
  if (m <= 16 || JTMAX == 16 || jt_cap <= 16 || cheb) return launch_spmm<T, 16>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
  return launch_spmm<T, (JTMAX == 32 ? 32 : 16)>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t *__restrict__ idx, int64_t nidx,
                                                          const T *__restrict__ X, int64_t ldx,
                                                          T *__restrict__ Out, int64_t ldo) {
  const T *x = X + (int64_t)blockIdx.y * ldx;
  T *o = Out + (int64_t)blockIdx.y * ldo;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nidx; i += stride) o[i] = x[idx[i]];
}

template <int DT>
static int gather_rows_impl(int64_t nidx, const int64_t *d_idx, int64_t m, const void *X, int64_t ldx, void *Out,
                            int64_t ldo) {
  using T = typename DType<DT>::T;
  Context &c = ctx();
  int64_t nb = (nidx + 255) / 256;
  const int64_t cap = ((int64_t)c.num_cu * 8 + m - 1) / m;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((gather_rows_kernel<T>), dim3((unsigned)nb, (unsigned)m), dim3(256), 0, c.stream, d_idx, nidx,
                     (const T *)X, ldx, (T *)Out, ldo);
  RLH_HIP(hipGetLastError());
  return 0;
}

template <int DT>
static int csr_build(rlh_csr *h, const int64_t *indptr, const int32_t *indices, const void *values_) {
  using T = typename DType<DT>::T;
  const T *values = (const T *)values_;
  const int64_t n = h->n_rows;
  const int64_t ns = (n + 63) / 64;
  std::vector<int64_t> sp(ns + 1, 0);
  for (int64_t s = 0; s < ns; ++s) {
    int64_t w = 0;
    for (int64_t r = s * 64; r < n && r < (s + 1) * 64; ++r) {
      const int64_t len = indptr[r + 1] - indptr[r];
      if (len > w) w = len;
    }
    sp[s + 1] = sp[s] + w * 64;
  }
  const int64_t padded = sp[ns];
  std::vector<int32_t> cols((size_t)padded);
  std::vector<T> vals((size_t)padded);
  for (int64_t s = 0; s < ns; ++s) {
    const int64_t w = (sp[s + 1] - sp[s]) / 64;
    for (int l = 0; l < 64; ++l) {
      const int64_t r = s * 64 + l;
      const int64_t b = r < n ? indptr[r] : 0, len = r < n ? indptr[r + 1] - b : 0;
      // padding entries carry value 0 and a harmless in-range column (the row's first, else 0)
      const int32_t padcol = len > 0 ? indices[b] : 0;
      for (int64_t t = 0; t < w; ++t) {
        const int64_t e = sp[s] + t * 64 + l;
        if (t < len) { cols[e] = indices[b + t]; vals[e] = values[b + t]; }
        else { cols[e] = padcol; memset(&vals[e], 0, sizeof(T)); }
      }
    }
  }
  h->n_slices = ns;
  h->padded = padded;
  RLH_HIP(hipMalloc((void **)&h->slice_ptr, (size_t)(ns + 1) * sizeof(int64_t)));
  RLH_HIP(hipMemcpy(h->slice_ptr, sp.data(), (size_t)(ns + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  if (padded > 0) {
    RLH_HIP(hipMalloc((void **)&h->cols, (size_t)padded * sizeof(int32_t)));
    RLH_HIP(hipMalloc((void **)&h->vals, (size_t)padded * sizeof(T)));
    RLH_HIP(hipMemcpy(h->cols, cols.data(), (size_t)padded * sizeof(int32_t), hipMemcpyHostToDevice));
    RLH_HIP(hipMemcpy(h->vals, vals.data(), (size_t)padded * sizeof(T), hipMemcpyHostToDevice));
  }
  h->device_bytes = (ns + 1) * 8 + padded * (4 + (int64_t)sizeof(T));
  return 0;
}


// The two launch orders of rlh_spmm_part for the own / halo boundary n_own: blocks that reference
// only own columns, and the rest (each keeps the grouped order of the full schedule).
static int well_split(rlh_csr *h, int64_t n_own) {
  if (h->well_split_at == n_own) return 0;
  for (int k = 0; k < 2; ++k) {
    if (h->well_sched_part[k]) (void)hipFree(h->well_sched_part[k]);
    h->well_sched_part[k] = nullptr;
    h->well_sched_part_len[k] = 0;
    h->well_grid_part[k] = 0;
  }
  std::vector<int32_t> part[2];
  for (int32_t b : h->well_order) part[h->well_maxcol[(size_t)b] < n_own ? 0 : 1].push_back(b);
  for (int k = 0; k < 2; ++k) {
    if (part[k].empty()) continue;
    std::vector<int32_t> sched;
    well_layout(part[k], ctx().num_cu, sched, h->well_grid_part[k]);
    h->well_sched_part_len[k] = (int64_t)sched.size();
    RLH_HIP(hipMalloc((void **)&h->well_sched_part[k], sched.size() * sizeof(int32_t)));
    RLH_HIP(hipMemcpy(h->well_sched_part[k], sched.data(), sched.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  h->well_split_at = n_own;
  return 0;
}

template <int DT>
static int stack_build(rlh_csr *h, const int64_t *indptr, const int32_t *indices, const void *values_,
                       const std::vector<std::vector<Win>> &wins, int64_t staged_unstacked);

static int stack_split(rlh_csr *h, int64_t n_own) {       // (well_split for the stacks)
  if (h->stk_split_at == n_own) return 0;
  for (int k = 0; k < 2; ++k) {
    if (h->stk_sched_part[k]) (void)hipFree(h->stk_sched_part[k]);
    h->stk_sched_part[k] = nullptr;
    h->stk_sched_part_len[k] = 0;
    h->stk_grid_part[k] = 0;
  }
  std::vector<int32_t> part[2];
  for (int32_t sb : h->stk_order) part[h->stk_maxcol[(size_t)sb] < n_own ? 0 : 1].push_back(sb);
  for (int k = 0; k < 2; ++k) {
    if (part[k].empty()) continue;
    std::vector<int32_t> sched;
    well_layout(part[k], ctx().num_cu, sched, h->stk_grid_part[k]);
    h->stk_sched_part_len[k] = (int64_t)sched.size();
    RLH_HIP(hipMalloc((void **)&h->stk_sched_part[k], sched.size() * sizeof(int32_t)));
    RLH_HIP(hipMemcpy(h->stk_sched_part[k], sched.data(), sched.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  h->stk_split_at = n_own;
  return 0;
}

// Host side of the 1024-row windowed layout.  Returns 0 with h->well_blocks == 0 when the matrix
// does not qualify (a row longer than 8 entries, a block whose windows do not fit the LDS buffer,
// or too little column locality for the staging to pay).
template <int DT>
static int well_build(rlh_csr *h, const int64_t *indptr, const int32_t *indices, const void *values_, bool force,
                      bool stack_only = false) {
  using T = typename DType<DT>::T;
  constexpr int SMAX = WellCfg<T>::SMAX;
  const T *values = (const T *)values_;
  const int64_t n = h->n_rows;
  const int64_t nblocks = (n + kWellRows - 1) / kWellRows;
  h->well_blocks = 0;
  if (nblocks == 0 || h->nnz == 0) return 0;
  PhaseClock clk;
  std::vector<std::vector<Win>> wins((size_t)nblocks);
  std::vector<int32_t> width((size_t)nblocks, 0), ngroups((size_t)nblocks, 0);
  parallel_blocks(nblocks, [&](int64_t b) {
    const int64_t r0 = b * kWellRows, r1 = (r0 + kWellRows < n) ? r0 + kWellRows : n;
    int64_t w = 0;
    for (int64_t r = r0; r < r1; ++r) w = std::max<int64_t>(w, indptr[r + 1] - indptr[r]);
    width[b] = (int32_t)std::min<int64_t>(w, 1 << 20);
    // groups per block: a multiple of 8 (one 16-byte load covers 2, 4 or -- bfloat16 -- 8 groups)
    ngroups[b] = find_windows(indptr, indices, r0, r1, h->n_cols, 32, 64, 8, wins[b]) / 64;
  });
  clk.lap("well: windows");
  int32_t wmax = 0, gmax = 0;
  int64_t staged = 0, slots = 0;
  for (int64_t b = 0; b < nblocks; ++b) {
    wmax = std::max(wmax, width[b]);
    gmax = std::max(gmax, ngroups[b]);
    staged += (int64_t)ngroups[b] * 64;
    slots += (int64_t)width[b] * kWellRows;
  }
  h->well_ratio = slots > 0 ? (double)staged / (double)slots : 0.0;
  // (the limit on a block's image is the unstacked kernel's staging buffer: a complex operator never runs that kernel -- its
  // stacks have a ring slot of their own size, checked by stack_build)
  if (wmax > 8 || (!stack_only && gmax > 16 * SMAX)) {
    if (clk.on) fprintf(stderr, "csr layout: well: not built (longest row %d, largest block image %d groups of 64 columns, limit %d)\n", wmax, gmax, 16 * SMAX);
    return 0;
  }
  // measured (profiles/r01_spmm_windowed.txt): at 0.65 staged elements per entry slot (5-point
  // stencil) the windowed kernel is 1.27x faster than the sliced one, at 1.06 (a diagonal
  // matrix) 4 % slower
  if (!force && staged * 10 > slots * 9) {
    if (clk.on) fprintf(stderr, "csr layout: well: not built (%lld staged elements for %lld entry slots)\n", (long long)staged, (long long)slots);
    return 0;
  }
  if (stack_only) {          // the complex types: the stacks for rlh_spmm on the whole operator, the interleaved layout for the rest
    h->well_staged = (double)staged / (double)n;
    return stack_build<DT>(h, indptr, indices, values_, wins, staged);
  }
  h->well_wmax = 8;
  std::vector<WellMeta> meta((size_t)nblocks);
  int64_t eoff = 0;
  int64_t goff = 0;
  for (int64_t b = 0; b < nblocks; ++b) {
    meta[b] = WellMeta{eoff, (int32_t)goff, width[b] | (ngroups[b] << 8)};
    eoff += 8;                                      // every row is stored as 8 slots (16-byte pieces per thread)
    goff += ngroups[b];
  }
  RLH_REQUIRE(goff < ((int64_t)1 << 31), "rlh_csr_create: too many staging groups");
  std::vector<int32_t> gsrc((size_t)goff);
  const int64_t epad = 0;
  RawBuf<uint16_t> idx((size_t)(eoff + epad) * kWellRows);      // (every slot of every block is written below)
  RawBuf<T> vals((size_t)(eoff + epad) * kWellRows);
  parallel_blocks(nblocks, [&](int64_t b) {
    const std::vector<Win> &ws = wins[b];
    fill_group_sources(ws, ngroups[b], 64, gsrc.data() + meta[b].goff);
    const int64_t r0 = b * kWellRows;
    for (int l = 0; l < kWellRows; ++l) {
      const int64_t r = r0 + l;
      const int64_t p = r < n ? indptr[r] : 0, len = r < n ? indptr[r + 1] - p : 0;
      // padding slots carry value 0 and the position of the row's own first entry, so that they
      // only ever touch a column the row references (0 * Inf of a foreign column would be NaN)
      const uint16_t padpos = len > 0 ? (uint16_t)staged_position(ws, indices[p]) : 0;
      size_t hint = 0;
      for (int32_t t = 0; t < 8; ++t) {
        const int64_t ev = well_val_index<T>(meta[b].eoff, t, l), ei = well_idx_index(meta[b].eoff, t, l);
        if (t < len) {
          idx[ei] = (uint16_t)staged_position_walk(ws, indices[p + t], hint);
          vals[ev] = values[p + t];
        } else {
          idx[ei] = padpos;
          memset(&vals[ev], 0, sizeof(T));
        }
      }
    }
  });
  clk.lap("well: entries");
  // what the 16-byte staging path may assume
  h->well_inbounds = 1;
  h->well_aligned = 1;
  for (int64_t g = 0; g < goff; ++g) {
    if ((int64_t)gsrc[g] + 64 > h->n_cols) h->well_inbounds = 0;
    if (gsrc[g] & 7) h->well_aligned = 0;
  }
  std::vector<int32_t> sched;
  well_schedule(wins, nblocks, n, kWellRows, ctx().num_cu, h->well_order);
  well_layout(h->well_order, ctx().num_cu, sched, h->well_grid);
  clk.lap("well: schedule");
  h->well_maxcol.resize((size_t)nblocks);
  for (int64_t b = 0; b < nblocks; ++b) h->well_maxcol[(size_t)b] = wins[b].back().start + wins[b].back().len - 1;
  h->well_sched_len = (int64_t)sched.size();
  RLH_HIP(hipMalloc((void **)&h->well_sched, sched.size() * sizeof(int32_t)));
  RLH_HIP(hipMemcpy(h->well_sched, sched.data(), sched.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  RLH_HIP(hipMalloc((void **)&h->well_meta, (size_t)nblocks * sizeof(WellMeta)));
  RLH_HIP(hipMemcpy(h->well_meta, meta.data(), (size_t)nblocks * sizeof(WellMeta), hipMemcpyHostToDevice));
  RLH_HIP(hipMalloc((void **)&h->well_gsrc, (size_t)std::max<int64_t>(goff, 1) * sizeof(int32_t)));
  RLH_HIP(hipMemcpy(h->well_gsrc, gsrc.data(), (size_t)goff * sizeof(int32_t), hipMemcpyHostToDevice));
  const size_t ne = (size_t)(eoff + epad) * kWellRows;
  RLH_HIP(hipMalloc((void **)&h->well_idx, std::max<size_t>(ne, 1) * sizeof(uint16_t)));
  RLH_HIP(hipMalloc((void **)&h->well_vals, std::max<size_t>(ne, 1) * sizeof(T)));
  RLH_HIP(hipMemcpy(h->well_idx, idx.data(), ne * sizeof(uint16_t), hipMemcpyHostToDevice));
  RLH_HIP(hipMemcpy(h->well_vals, vals.data(), ne * sizeof(T), hipMemcpyHostToDevice));
  h->padded = eoff * kWellRows;
  h->device_bytes = nblocks * (int64_t)sizeof(WellMeta) + (int64_t)sched.size() * 4 + goff * 4 + (int64_t)ne * (2 + (int64_t)sizeof(T));
  h->well_blocks = nblocks;
  h->well_staged = (double)staged / (double)n;
  clk.lap("well: upload");
  if (h->well_inbounds) return stack_build<DT>(h, indptr, indices, values_, wins, staged);
  return 0;
}

// The far stride of a grid operator: the offset (largest column - row) most rows share, e.g. one grid plane for a 3-D
// stencil (0: fewer than 60 % of the sampled rows agree).  Row blocks of a stack are partners one such stride apart.
static int64_t far_stride(const int64_t *indptr, const int32_t *indices, int64_t n) {
  std::unordered_map<int64_t, int64_t> count;
  int64_t samples = 0;
  const int64_t step = std::max<int64_t>(1, n / 50000);
  for (int64_t i = 0; i < n; i += step) {
    if (indptr[i + 1] == indptr[i]) continue;
    const int64_t d = (int64_t)indices[indptr[i + 1] - 1] - i;
    ++samples;
    if (d > 0) ++count[d];
  }
  int64_t best = 0, best_count = 0;
  for (const auto &kv : count)
    if (kv.second > best_count || (kv.second == best_count && kv.first < best)) { best = kv.first; best_count = kv.second; }
  return best_count * 10 >= samples * 6 ? best : 0;
}

// Host side of the stacked layout ("Stacked blocks" above): the stacks by greedy matching on the window-overlap graph of
// the 1024-row blocks, then windows / staging groups / entries per stack exactly as well_build lays them out per block.
// Built only where it stages at least 10 % less than the unstacked layout (RLH_SPMM_STACK=2 builds it regardless, 0 never).
template <int DT>
static int stack_build(rlh_csr *h, const int64_t *indptr, const int32_t *indices, const void *values_,
                       const std::vector<std::vector<Win>> &wins, int64_t staged_unstacked) {
  using T = typename DType<DT>::T;
  constexpr int R = kStkR;
  // groups one vector's image may have: a buffer of the register-staged kernel for the real types (the LDS-DMA kernel
  // takes over from it where the image fits a ring slot), a ring slot for the complex ones (LDS-DMA only)
  constexpr int BUFG = (DType<DT>::cplx ? StkRing<T>::SLOT : kStkBufBytes) / (64 * (int)sizeof(T));
  const T *values = (const T *)values_;
  const int64_t n = h->n_rows;
  const int64_t nblocks = (int64_t)wins.size();
  const int mode = env_int("RLH_SPMM_STACK", 1);
  h->stk_blocks = 0;
  // (fewer than 4 row blocks per CU: halving the number of work units costs more than the staging saves -- lap3d 61^3,
  // 222 blocks: 12.4 us as blocks, 14.1 us as stacks)
  // A complex operator that small (the row shards of BASELINE config 5: 977 / 489 / 245 blocks on 2 / 4 / 8 GPUs) still takes
  // the LDS-DMA kernel -- its alternative is the interleaved layout at 0.6-0.7 of the rate: as pairs while every CU gets a
  // stack (measured on one shard of 2 / 4: product 0.44 / 0.24 ms as pairs, 0.54 / 0.29 as stacks of one, 0.77 / 0.40
  // interleaved), as stacks of ONE block below that (one of 8: 0.15 ms, 0.18 as pairs on half the CUs, 0.19 interleaved;
  // tools/c5_shard_bench.py).  RLH_SPMM_STACK_SINGLE=0: no stacks then, as for the real types.
  bool single = false;
  if (mode == 0 || nblocks < 2) return 0;
  if (mode < 2 && nblocks < 4 * (int64_t)ctx().num_cu) {
    // (below three quarters of a block per CU the interleaved layout's 256-row units fill the chip better: 123 blocks, 50^3:
    // product 0.077 ms interleaved, 0.087 as stacks of one; 256 blocks, 64^3: 0.170 against 0.131)
    if (!DType<DT>::cplx || 4 * nblocks < 3 * (int64_t)ctx().num_cu || env_int("RLH_SPMM_STACK_SINGLE", 1) == 0) return 0;
    single = 4 * nblocks < 7 * (int64_t)ctx().num_cu;
  }
  PhaseClock clk;
  // ---- the row ranges of the blocks: uniform 1024-row blocks, paired by greedy matching on the window-overlap graph
  std::vector<int64_t> brow;
  std::vector<int32_t> owner, members;
  for (int64_t b = 0; b <= nblocks; ++b) brow.push_back(std::min(b * kWellRows, n));
  // overlap of block b's windows with the rows of block c, both directions summed
  std::vector<std::vector<std::pair<int32_t, int64_t>>> adj((size_t)nblocks);
  auto bump = [&](int64_t b, int64_t c, int64_t ov) {
    for (auto &e : adj[(size_t)b])
      if (e.first == (int32_t)c) { e.second += ov; return; }
    adj[(size_t)b].push_back({(int32_t)c, ov});
  };
  for (int64_t b = 0; b < nblocks; ++b)
    for (const Win &w : wins[(size_t)b]) {
      const int64_t lo = w.start, hi = std::min<int64_t>((int64_t)w.start + w.len, n);
      for (int64_t c = lo / kWellRows; c * kWellRows < hi; ++c) {
        if (c == b) continue;
        const int64_t ov = std::min<int64_t>(hi, (c + 1) * kWellRows) - std::max<int64_t>(lo, c * kWellRows);
        if (ov <= 0) continue;
        bump(b, c, ov);
        bump(c, b, ov);
      }
    }
  owner.assign((size_t)nblocks, -1);
  int64_t next_free = 0;
  for (int64_t b = 0; b < nblocks; ++b) {
    if (owner[(size_t)b] >= 0) continue;
    const int32_t sb = (int32_t)(members.size() / R);
    owner[(size_t)b] = sb;
    members.push_back((int32_t)b);
    for (int r = 1; r < R; ++r) {
      int32_t best = -1;
      int64_t wbest = 0;
      if (single) { members.push_back(-1); continue; }
      for (size_t k = (size_t)(sb * R); k < members.size(); ++k)          // heaviest free neighbour of the stack so far
        for (const auto &e : adj[(size_t)members[k]])
          if (owner[(size_t)e.first] < 0 && (e.second > wbest || (e.second == wbest && best >= 0 && e.first < best))) {
            best = e.first;
            wbest = e.second;
          }
      if (best < 0) {                                                   // none: the next free block in order
        while (next_free < nblocks && owner[(size_t)next_free] >= 0) ++next_free;
        if (next_free < nblocks) best = (int32_t)next_free;
      }
      if (best >= 0) owner[(size_t)best] = sb;
      members.push_back(best);
    }
  }

  int64_t nst = (int64_t)members.size() / R;
  const int64_t nc8 = (h->n_cols + 7) & ~(int64_t)7;                    // windows may end here: aligned starts at the far end too
  std::vector<std::vector<Win>> swins((size_t)nst);
  std::vector<int32_t> ngroups((size_t)nst, 0);
  auto analyse = [&](int64_t count) {
    swins.assign((size_t)count, std::vector<Win>());
    ngroups.assign((size_t)count, 0);
    parallel_blocks(count, [&](int64_t sb) {
      std::vector<int32_t> cols;
      for (int r = 0; r < R; ++r) {
        const int64_t mb = members[(size_t)(sb * R + r)];
        if (mb < 0) continue;
        const int64_t r0 = brow[(size_t)mb], r1 = brow[(size_t)mb + 1];
        cols.insert(cols.end(), indices + indptr[r0], indices + indptr[r1]);
      }
      ngroups[(size_t)sb] = find_windows_of(cols, nc8, 32, 64, 8, swins[(size_t)sb]) / 64;
    });
  };
  clk.lap("stack: matching");
  analyse(nst);
  // Uniform blocks pair up exactly when the far stride of the operator (a grid plane) is a whole number of them, and well
  // enough when it nearly is (215^2 = 45.14 blocks: 2.49 staged elements per row).  When a block and its partner one plane
  // on are half a block out of step (126^2 = 15.5 blocks), the union of their windows outgrows a slot of the ring and the
  // stacks fall apart into stacks of one (config 5 at 126^3: 1.03 ms against 0.82 ms at 128^3).  Then -- and only then: at
  // 215^3 the uniform blocks are 4 % faster -- the blocks are cut plane by plane instead, ceil(plane / 1024) blocks of
  // (nearly) equal size per plane, so that block j of a plane and block j of the next are exact translates of each other:
  // 126^3 complex128 in 0.865 ms.  Float32 operators keep the uniform blocks (their bfloat16 step moves rows in 16-byte
  // pieces of eight).  RLH_SPMM_STACK_ALIGN=0: never, =2: whenever a far stride exists.
  {
    constexpr int DMAG0 = StkRing<T>::SLOT / (64 * (int)sizeof(T));
    const int align_mode = env_int("RLH_SPMM_STACK_ALIGN", 1);
    int64_t apart = 0, pairs = 0;
    for (int64_t sb = 0; sb < nst; ++sb)
      if (members[(size_t)(sb * R + 1)] >= 0) { ++pairs; apart += ngroups[(size_t)sb] > std::min(BUFG, DMAG0); }
    const int64_t plane = (R == 2 && DT != RLH_S && align_mode != 0 && (apart * 4 > pairs || align_mode == 2))
                              ? far_stride(indptr, indices, n) : 0;
    if (plane > kWellRows && plane < n && plane % kWellRows != 0) {
      brow.clear();
      members.clear();
      std::vector<int64_t> first_of_plane;              // block id of the first block of every plane
      brow.push_back(0);
      for (int64_t p0 = 0; p0 < n; p0 += plane) {
        const int64_t len = std::min(plane, n - p0), k = (len + kWellRows - 1) / kWellRows;
        first_of_plane.push_back((int64_t)brow.size() - 1);
        for (int64_t j = 1; j <= k; ++j) brow.push_back(p0 + len * j / k);
      }
      first_of_plane.push_back((int64_t)brow.size() - 1);
      const int64_t planes = (int64_t)first_of_plane.size() - 1;
      for (int64_t p = 0; p < planes; p += 2) {
        const int64_t b0 = first_of_plane[(size_t)p], k0 = first_of_plane[(size_t)p + 1] - b0;
        const int64_t b1 = p + 1 < planes ? first_of_plane[(size_t)p + 1] : -1, k1 = p + 1 < planes ? first_of_plane[(size_t)p + 2] - b1 : 0;
        for (int64_t j = 0; j < std::max(k0, k1); ++j) {
          if (j < k0) { members.push_back((int32_t)(b0 + j)); members.push_back(j < k1 ? (int32_t)(b1 + j) : -1); }
          else { members.push_back((int32_t)(b1 + j)); members.push_back(-1); }
        }
      }
      nst = (int64_t)members.size() / R;
      owner.assign(brow.size() - 1, -1);
      for (size_t k = 0; k < members.size(); ++k)
        if (members[k] >= 0) owner[(size_t)members[k]] = (int32_t)(k / R);
      analyse(nst);
    }
  }
  const int64_t nblk = (int64_t)brow.size() - 1;
  clk.lap("stack: windows");
  // a stack whose image does not fit (grid planes half a row block out of step with the blocks: 126^2 rows = 15.5
  // blocks) is taken apart into stacks of one
  // ... and so is a stack of a real type whose image fits the register-staged kernel's buffer but not a slot of the LDS-DMA
  // ring, as long as such stacks are few (the blocks of a row shard next to its halo columns: their windows lie in two far
  // apart column ranges): one oversized stack would otherwise take the LDS-DMA kernel -- the only one that serves row
  // shards -- away from the whole operator (the forced one-rank run of lap3d 215^3 fell back to the unstacked kernel that
  // way: 1.33 instead of 1.12 ms per product)
  constexpr int DMAG = StkRing<T>::SLOT / (64 * (int)sizeof(T));
  int64_t over_dma = 0;
  for (int64_t sb = 0; sb < nst; ++sb) over_dma += ngroups[(size_t)sb] > DMAG && members[(size_t)(sb * R + 1)] >= 0;
  const int CAPG = (over_dma > 0 && over_dma * 20 <= nst) ? std::min(BUFG, DMAG) : BUFG;
  bool split = false;
  for (int64_t sb = 0; sb < nst && !split; ++sb) split = ngroups[(size_t)sb] > CAPG && members[(size_t)(sb * R + 1)] >= 0;
  if (split) {
    std::vector<int32_t> again;
    for (int64_t sb = 0; sb < nst; ++sb) {
      if (ngroups[(size_t)sb] <= CAPG || members[(size_t)(sb * R + 1)] < 0) {
        for (int r = 0; r < R; ++r) again.push_back(members[(size_t)(sb * R + r)]);
        continue;
      }
      for (int r = 0; r < R; ++r) {
        again.push_back(members[(size_t)(sb * R + r)]);
        for (int q = 1; q < R; ++q) again.push_back(-1);
      }
    }
    members.swap(again);
    nst = (int64_t)members.size() / R;
    for (int64_t sb = 0; sb < nst; ++sb)
      for (int r = 0; r < R; ++r)
        if (members[(size_t)(sb * R + r)] >= 0) owner[(size_t)members[(size_t)(sb * R + r)]] = (int32_t)sb;
    analyse(nst);
  }
  int64_t staged = 0;
  int32_t gmax = 0;
  for (int64_t sb = 0; sb < nst; ++sb) {
    staged += (int64_t)ngroups[(size_t)sb] * 64;
    gmax = std::max(gmax, ngroups[(size_t)sb]);
  }
  h->stk_staged = (double)staged / (double)(n > 0 ? n : 1);
  h->stk_gmax = gmax;
  if (gmax > BUFG) {                                                     // a stack's image must fit one buffer
    if (clk.on) fprintf(stderr, "csr layout: stack: not built (largest image %d groups of 64 columns, a buffer holds %d)\n", gmax, BUFG);
    return 0;
  }
  // (the complex types: the alternative is the interleaved layout, which re-stages the windows in passes of a few vectors
  // -- config 5's operator at 160^3 in complex64: 1.87 ms there, 0.86 ms here)
  if (mode < 2 && !DType<DT>::cplx && staged * 10 > staged_unstacked * 9) return 0;
  std::vector<WellMeta> meta((size_t)nst);
  int64_t goff = 0;
  for (int64_t sb = 0; sb < nst; ++sb) {
    meta[(size_t)sb] = WellMeta{sb * 8 * R, (int32_t)goff, 8 | (ngroups[(size_t)sb] << 8)};
    goff += ngroups[(size_t)sb];
  }
  RLH_REQUIRE(goff < ((int64_t)1 << 31), "rlh_csr_create: too many staging groups");
  std::vector<int32_t> gsrc((size_t)goff);
  const size_t ne = (size_t)nst * 8 * R * kWellRows;
  RawBuf<uint16_t> idx(ne);                                     // (every slot of every stack is written below)
  RawBuf<T> vals(ne);
  std::atomic<int> self_ok{1};
  parallel_blocks(nst, [&](int64_t sb) {
    const std::vector<Win> &ws = swins[(size_t)sb];
    fill_group_sources(ws, ngroups[(size_t)sb], 64, gsrc.data() + meta[(size_t)sb].goff);
    for (int r = 0; r < R; ++r) {
      const int64_t mb = members[(size_t)(sb * R + r)];
      const int64_t eoff = meta[(size_t)sb].eoff + 8 * r;
      for (int l = 0; l < kWellRows; ++l) {
        const int64_t row = mb >= 0 && brow[(size_t)mb] + l < brow[(size_t)mb + 1] ? brow[(size_t)mb] + l : n;
        const int64_t p = row < n ? indptr[row] : 0, len = row < n ? indptr[row + 1] - p : 0;
        const uint16_t padpos = len > 0 ? (uint16_t)staged_position(ws, indices[p]) : 0;
        size_t hint = 0;
        for (int32_t t = 0; t < 8; ++t) {
          const int64_t ev = well_val_index<T>(eoff, t, l), ei = well_idx_index(eoff, t, l);
          if (t < len) {
            idx[(size_t)ei] = (uint16_t)staged_position_walk(ws, indices[p + t], hint);
            vals[(size_t)ev] = values[p + t];
          } else {
            idx[(size_t)ei] = padpos;
            memset(&vals[(size_t)ev], 0, sizeof(T));
          }
        }
        // the LAST slot of a row with a free one carries (value 0 and) the position of the row's OWN column, where the row
        // stores its diagonal entry: the fused Chebyshev step then finds y[row] in the staged image instead of reading the
        // block a second time (well_stack_dma_kernel<..., CHEB>); one row without makes the handle say so
        if (row < n) {
          bool has_diag = false;
          for (int64_t e = p; e < p + len && !has_diag; ++e) has_diag = indices[e] == row;
          if (len <= 7 && has_diag) idx[(size_t)well_idx_index(eoff, 7, l)] = (uint16_t)staged_position(ws, (int32_t)row);
          else self_ok.store(0, std::memory_order_relaxed);
        }
      }
    }
  });
  h->stk_self = self_ok.load();
  clk.lap("stack: entries");
  h->stk_overhang = 0;
  for (int64_t g = 0; g < goff; ++g) {
    if ((int64_t)gsrc[(size_t)g] + 64 > nc8) return 0;                  // 16-byte staging needs whole groups
    if ((int64_t)gsrc[(size_t)g] + 64 > h->n_cols) h->stk_overhang = (int)(nc8 - h->n_cols);
  }
  h->stk_aligned = 1;
  for (int64_t g = 0; g < goff; ++g)
    if (gsrc[(size_t)g] & 7) h->stk_aligned = 0;
  h->stk_maxcol.resize((size_t)nst);
  for (int64_t sb = 0; sb < nst; ++sb) h->stk_maxcol[(size_t)sb] = swins[(size_t)sb].back().start + swins[(size_t)sb].back().len - 1;
  std::vector<int32_t> &order = h->stk_order;
  std::vector<int32_t> sched;
  std::vector<int32_t> granule_owner((size_t)((n + kWellRows - 1) / kWellRows), -1);      // the stack that owns row 1024 c
  {
    int64_t blk = 0;
    for (size_t c = 0; c < granule_owner.size(); ++c) {
      while (blk + 1 < nblk && brow[(size_t)blk + 1] <= (int64_t)c * kWellRows) ++blk;
      granule_owner[c] = owner[(size_t)blk];
    }
  }
  well_schedule(swins, nst, n, kWellRows, ctx().num_cu, order, &granule_owner);
  well_layout(order, ctx().num_cu, sched, h->stk_grid);
  clk.lap("stack: schedule");
  h->stk_sched_len = (int64_t)sched.size();
  RLH_HIP(hipMalloc((void **)&h->stk_sched, sched.size() * sizeof(int32_t)));
  RLH_HIP(hipMemcpy(h->stk_sched, sched.data(), sched.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  RLH_HIP(hipMalloc((void **)&h->stk_meta, (size_t)nst * sizeof(WellMeta)));
  RLH_HIP(hipMemcpy(h->stk_meta, meta.data(), (size_t)nst * sizeof(WellMeta), hipMemcpyHostToDevice));
  {
    std::vector<int32_t> ranges(members.size() * 2, 0);                   // (first row, rows) of every member; (0, 0): none
    for (size_t k = 0; k < members.size(); ++k)
      if (members[k] >= 0) {
        ranges[2 * k] = (int32_t)brow[(size_t)members[k]];
        ranges[2 * k + 1] = (int32_t)(brow[(size_t)members[k] + 1] - brow[(size_t)members[k]]);
      }
    RLH_HIP(hipMalloc((void **)&h->stk_member, ranges.size() * sizeof(int32_t)));
    RLH_HIP(hipMemcpy(h->stk_member, ranges.data(), ranges.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  RLH_HIP(hipMalloc((void **)&h->stk_gsrc, (size_t)std::max<int64_t>(goff, 1) * sizeof(int32_t)));
  RLH_HIP(hipMemcpy(h->stk_gsrc, gsrc.data(), (size_t)goff * sizeof(int32_t), hipMemcpyHostToDevice));
  // value dictionary: the distinct 8-slot value tuples of the rows (byte-wise), given up beyond kStkMaxPatterns
  {
    std::vector<int32_t> pat((size_t)nst * R * kWellRows, 0);
    std::vector<T> table;
    std::unordered_map<uint64_t, std::vector<int32_t>> seen;
    bool ok = env_int("RLH_SPMM_STACK_PAT", 1) != 0;
    int32_t last_id = -1;
    T last_tup[8];
    for (int64_t sb = 0; sb < nst && ok; ++sb)
      for (int r = 0; r < R && ok; ++r) {
        const int64_t eoff = meta[(size_t)sb].eoff + 8 * r;
        for (int l = 0; l < kWellRows; ++l) {
          T tup[8];
          for (int t = 0; t < 8; ++t) tup[t] = vals[(size_t)well_val_index<T>(eoff, t, l)];
          if (last_id >= 0 && !memcmp(last_tup, tup, sizeof(tup))) {      // (a stencil: the same as the row before)
            pat[(size_t)((sb * R + r) * kWellRows + l)] = last_id;
            continue;
          }
          uint64_t hsh = 1469598103934665603ull;
          const unsigned char *bytes = reinterpret_cast<const unsigned char *>(tup);
          for (size_t k = 0; k < sizeof(tup); ++k) hsh = (hsh ^ bytes[k]) * 1099511628211ull;
          std::vector<int32_t> &cand = seen[hsh];
          int32_t id = -1;
          for (int32_t c : cand)
            if (!memcmp(&table[(size_t)c * 8], tup, sizeof(tup))) { id = c; break; }
          if (id < 0) {
            id = (int32_t)(table.size() / 8);
            if (id >= kStkMaxPatterns) { ok = false; break; }
            table.insert(table.end(), tup, tup + 8);
            cand.push_back(id);
          }
          pat[(size_t)((sb * R + r) * kWellRows + l)] = id;
          last_id = id;
          memcpy(last_tup, tup, sizeof(tup));
        }
      }
    clk.lap("stack: value dictionary");
    // position patterns per member: the positions less the row's index in its block (mod 2^16)
    std::vector<uint16_t> dtab;
    bool dok = ok && env_int("RLH_SPMM_STACK_PAT", 1) >= 1 && env_int("RLH_SPMM_STACK_DPAT", 1) != 0;
    if (dok) {
      dtab.assign((size_t)nst * R * kStkMaxDeltas * 8, 0);
      for (int64_t sb = 0; sb < nst && dok; ++sb)
        for (int r = 0; r < R && dok; ++r) {
          const int64_t eoff = meta[(size_t)sb].eoff + 8 * r, pb = sb * R + r;
          uint16_t *tab = dtab.data() + (size_t)pb * kStkMaxDeltas * 8;
          int used = 0;
          const int64_t mb = members[(size_t)pb];
          for (int l = 0; l < kWellRows && dok; ++l) {
            if (mb < 0 || brow[(size_t)mb] + l >= brow[(size_t)mb + 1]) continue;   // no such row: nothing is stored for it, any pattern will do
            uint16_t tup[8];
            for (int t = 0; t < 8; ++t) tup[t] = (uint16_t)(idx[(size_t)well_idx_index(eoff, t, l)] - (uint16_t)l);
            int id = -1;
            for (int c = used - 1; c >= 0; --c)           // (the last one used is the likeliest)
              if (!memcmp(tab + c * 8, tup, sizeof(tup))) { id = c; break; }
            if (id < 0) {
              if (used >= kStkMaxDeltas) { dok = false; break; }
              id = used++;
              memcpy(tab + id * 8, tup, sizeof(tup));
            }
            pat[(size_t)(pb * kWellRows + l)] |= id << 16;
          }
        }
      if (!dok)
        for (int32_t &w : pat) w &= 0xffff;
    }
    clk.lap("stack: position dictionary");
    h->stk_npat = 0;
    if (ok && !table.empty()) {
      if (dok) {
        RLH_HIP(hipMalloc((void **)&h->stk_dtab, dtab.size() * sizeof(uint16_t)));
        RLH_HIP(hipMemcpy(h->stk_dtab, dtab.data(), dtab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        h->device_bytes += (int64_t)dtab.size() * 2;
      }
      h->stk_npat = (int64_t)table.size() / 8;
      RLH_HIP(hipMalloc((void **)&h->stk_pat, pat.size() * sizeof(int32_t)));
      RLH_HIP(hipMemcpy(h->stk_pat, pat.data(), pat.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      RLH_HIP(hipMalloc((void **)&h->stk_table, table.size() * sizeof(T)));
      RLH_HIP(hipMemcpy(h->stk_table, table.data(), table.size() * sizeof(T), hipMemcpyHostToDevice));
      h->device_bytes += (int64_t)pat.size() * 4 + (int64_t)table.size() * (int64_t)sizeof(T);
    }
  }
  RLH_HIP(hipMalloc((void **)&h->stk_idx, ne * sizeof(uint16_t)));
  RLH_HIP(hipMalloc((void **)&h->stk_vals, ne * sizeof(T)));
  RLH_HIP(hipMemcpy(h->stk_idx, idx.data(), ne * sizeof(uint16_t), hipMemcpyHostToDevice));
  RLH_HIP(hipMemcpy(h->stk_vals, vals.data(), ne * sizeof(T), hipMemcpyHostToDevice));
  h->device_bytes += nst * (int64_t)sizeof(WellMeta) + (int64_t)sched.size() * 4 + goff * 4 + (int64_t)members.size() * 8 +
                     (int64_t)ne * (2 + (int64_t)sizeof(T));
  h->stk_blocks = nst;
  clk.lap("stack: upload");
  return 0;
}

// The Hermitian operator defined by the UPPER triangle of a square CSR matrix (entries below the diagonal, if the caller
// stores them, are ignored): A = U + U^H - diag(U), what mkl_?csrmm computes with the 'SUNF' / 'HUNF' descriptor the
// reference passes (raleigh/algebra/mkl_wrap.py:211-276).  One counting pass and one fill pass over the entries on the
// host (row i of the result = the mirrored entries (j, i), j < i, in ascending j, then the row's own upper entries:
// sorted if the input rows are), then the ordinary rlh_csr_create.
template <typename T> static inline T conj_of(T v) { return v; }
template <> inline c32 conj_of(c32 v) { return c32{v.re, -v.im}; }
template <> inline c64 conj_of(c64 v) { return c64{v.re, -v.im}; }

template <typename T>
static int csr_from_upper(rlh_csr_t *out, int dtype, int64_t n, const int64_t *indptr, const int32_t *indices, const T *values) {
  PhaseClock clk;
  // counting pass, rows dealt to the host threads: own entries per row, mirrored entries per target row (atomic: two
  // threads may mirror into the same row)
  std::unique_ptr<std::atomic<int64_t>[]> cnt(new std::atomic<int64_t>[(size_t)n + 1]);
  std::vector<int64_t> below((size_t)n, 0);
  std::atomic<int64_t> bad_row{-1};
  std::atomic<int> bad_kind{0};
  host_parallel(64, [&](int t, int nt) {
    for (int64_t i = n * t / nt; i < n * (t + 1) / nt; ++i) cnt[(size_t)i + 1].store(0, std::memory_order_relaxed);
    if (t == 0) cnt[0].store(0, std::memory_order_relaxed);
  });
  host_parallel(64, [&](int t, int nt) {
    for (int64_t i = n * t / nt; i < n * (t + 1) / nt; ++i) {
      int32_t prev = -1;
      int64_t own = 0;
      for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
        const int32_t j = indices[e];
        if (j < 0 || j >= n) { bad_row.store(i); bad_kind.store(1); return; }
        if (j <= prev) { bad_row.store(i); bad_kind.store(2); return; }
        prev = j;
        if (j < i) continue;
        ++own;
        if (j > i) cnt[(size_t)j + 1].fetch_add(1, std::memory_order_relaxed);
      }
      below[(size_t)i] = own;                                   // (own entries for now)
    }
  });
  RLH_REQUIRE(bad_kind.load() != 1, "rlh_csr_create_upper: column index out of range in row %lld", (long long)bad_row.load());
  RLH_REQUIRE(bad_kind.load() != 2, "rlh_csr_create_upper: the column indices of row %lld are not sorted (or repeat)",
              (long long)bad_row.load());
  // A caller that stores BOTH triangles of a Hermitian matrix (what SciPy users hold) has handed over the operator itself:
  // if every entry below the diagonal is the conjugate of its mirror image above it -- checked entry by entry, one binary
  // search each, on the host threads -- and the two triangles have equally many entries, nothing needs mirroring and no
  // second copy of the matrix is allocated (lap3d 215^3: 0.2 s and 0.8 GB less).  Any mismatch: the upper triangle rules.
  {
    int64_t own_total = 0, mirrored_total = 0;
    for (int64_t i = 0; i < n; ++i) { own_total += below[(size_t)i]; mirrored_total += cnt[(size_t)i + 1].load(std::memory_order_relaxed); }
    const int64_t lower_total = indptr[n] - own_total;
    if (lower_total > 0 && lower_total == mirrored_total) {
      std::atomic<int> mismatch{0};
      host_parallel(64, [&](int t, int nt) {
        for (int64_t i = n * t / nt; i < n * (t + 1) / nt && !mismatch.load(std::memory_order_relaxed); ++i)
          for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
            const int32_t j = indices[e];
            if (j >= i) break;                                   // (sorted: the rest of the row is the upper part)
            const int32_t *b0 = indices + indptr[j], *b1 = indices + indptr[j + 1];
            const int32_t *f = std::lower_bound(b0, b1, (int32_t)i);
            if (f == b1 || *f != (int32_t)i) { mismatch.store(1); break; }
            const T up = conj_of(values[f - indices]);
            if (memcmp(&up, &values[e], sizeof(T)) != 0) { mismatch.store(1); break; }
          }
      });
      if (!mismatch.load()) {
        clk.lap("both triangles given and consistent");
        return rlh_csr_create(out, dtype, n, n, indptr, indices, values);
      }
    }
  }
  std::vector<int64_t> rp((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; ++i) {
    const int64_t mirrored = cnt[(size_t)i + 1].load(std::memory_order_relaxed);
    rp[(size_t)i + 1] = rp[(size_t)i] + mirrored + below[(size_t)i];
    below[(size_t)i] = mirrored;
    cnt[(size_t)i].store(rp[(size_t)i], std::memory_order_relaxed);     // where the next mirrored entry of row i goes
  }
  std::vector<int32_t> idx((size_t)rp[(size_t)n]);
  std::vector<T> val((size_t)rp[(size_t)n]);
  host_parallel(64, [&](int t, int nt) {
    for (int64_t i = n * t / nt; i < n * (t + 1) / nt; ++i) {
      int64_t w = rp[(size_t)i] + below[(size_t)i];               // the row's own entries follow its mirrored ones
      for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
        const int32_t j = indices[e];
        if (j < i) continue;
        idx[(size_t)w] = j; val[(size_t)w] = values[e]; ++w;
        if (j > i) {
          const int64_t d = cnt[(size_t)j].fetch_add(1, std::memory_order_relaxed);
          idx[(size_t)d] = (int32_t)i; val[(size_t)d] = conj_of(values[e]);
        }
      }
    }
  });
  // the mirrored entries of a row arrive in whatever order the threads reached them: ascending column order restored
  host_parallel(64, [&](int t, int nt) {
    std::vector<std::pair<int32_t, T>> tmp;
    for (int64_t i = n * t / nt; i < n * (t + 1) / nt; ++i) {
      const int64_t b0 = rp[(size_t)i], m = below[(size_t)i];
      bool sorted = true;
      for (int64_t k = 1; k < m && sorted; ++k) sorted = idx[(size_t)(b0 + k - 1)] < idx[(size_t)(b0 + k)];
      if (sorted) continue;
      tmp.resize((size_t)m);
      for (int64_t k = 0; k < m; ++k) tmp[(size_t)k] = {idx[(size_t)(b0 + k)], val[(size_t)(b0 + k)]};
      std::sort(tmp.begin(), tmp.end(), [](const std::pair<int32_t, T> &x, const std::pair<int32_t, T> &y) { return x.first < y.first; });
      for (int64_t k = 0; k < m; ++k) { idx[(size_t)(b0 + k)] = tmp[(size_t)k].first; val[(size_t)(b0 + k)] = tmp[(size_t)k].second; }
    }
  });
  clk.lap("mirror the upper triangle");
  return rlh_csr_create(out, dtype, n, n, rp.data(), idx.data(), val.data());
}

}  // namespace rlh

using namespace rlh;

extern "C" {

int rlh_csr_create(rlh_csr_t *out, int dtype, int64_t n_rows, int64_t n_cols, const int64_t *indptr,
                   const int32_t *indices, const void *values) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(out != nullptr, "rlh_csr_create: null handle pointer");
  *out = nullptr;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_csr_create: unknown dtype %d", dtype);
  RLH_REQUIRE(n_rows >= 0 && n_cols >= 0 && n_cols < ((int64_t)1 << 31), "rlh_csr_create: bad shape");
  RLH_REQUIRE(indptr != nullptr, "rlh_csr_create: null indptr");
  PhaseClock clk;
  const int64_t nnz = indptr[n_rows] - indptr[0];
  RLH_REQUIRE(indptr[0] == 0 && nnz >= 0, "rlh_csr_create: indptr must be 0-based and non-decreasing");
  RLH_REQUIRE(nnz == 0 || (indices && values), "rlh_csr_create: null indices/values");
  for (int64_t r = 0; r < n_rows; ++r)
    RLH_REQUIRE(indptr[r + 1] >= indptr[r], "rlh_csr_create: indptr decreases at row %lld", (long long)r);
  for (int64_t e = 0; e < nnz; ++e)
    RLH_REQUIRE(indices[e] >= 0 && indices[e] < n_cols, "rlh_csr_create: column index %d out of range at entry %lld",
                indices[e], (long long)e);
  clk.lap("argument checks");
  rlh_csr *h = new rlh_csr();
  h->dtype = dtype; h->n_rows = n_rows; h->n_cols = n_cols; h->nnz = nnz;
  h->slice_ptr = nullptr; h->cols = nullptr; h->vals = nullptr;
  h->well_blocks = 0; h->well_meta = nullptr; h->well_gsrc = nullptr; h->well_idx = nullptr; h->well_vals = nullptr;
  h->well_ratio = 0.0; h->well_sched = nullptr; h->well_sched_len = 0; h->well_grid = 0; h->well_inbounds = 0; h->well_aligned = 0;
  h->stk_self = 0;
  h->stk_blocks = 0; h->stk_meta = nullptr; h->stk_member = nullptr; h->stk_gsrc = nullptr; h->stk_idx = nullptr;
  h->stk_split_at = -1; h->stk_sched_part[0] = h->stk_sched_part[1] = nullptr; h->stk_sched_part_len[0] = h->stk_sched_part_len[1] = 0;
  h->stk_grid_part[0] = h->stk_grid_part[1] = 0; h->stk_aligned = 0; h->stk_gmax = 0; h->stk_overhang = 0;
  h->stk_pat = nullptr; h->stk_table = nullptr; h->stk_npat = 0; h->stk_dtab = nullptr;
  h->stk_vals = nullptr; h->stk_sched = nullptr; h->stk_sched_len = 0; h->stk_grid = 0; h->stk_staged = 0.0; h->well_staged = 0.0;
  h->well_split_at = -1; h->well_sched_part[0] = h->well_sched_part[1] = nullptr;
  h->well_sched_part_len[0] = h->well_sched_part_len[1] = 0; h->well_grid_part[0] = h->well_grid_part[1] = 0;
  h->wide_blocks = 0; h->wide_meta = nullptr; h->wide_gsrc = nullptr; h->wide_idx = nullptr; h->wide_vals = nullptr;
  h->wide_gmax = 0; h->wide_k = 1; h->n_slices = 0; h->padded = 0; h->device_bytes = 0; h->well_wmax = 0;
  // Layout (RLH_SPMM_FORMAT=sell|well|wide overrides the choice and the locality test; read per
  // handle so tests can cover all three): rows of at most 8 entries of a real type -> the 1024-row
  // windowed layout; otherwise, or when that one does not qualify -> the 256-row interleaved
  // layout; no column locality at all -> sliced ELL.
  const char *fmt = getenv("RLH_SPMM_FORMAT");
  const bool want_sell = fmt && !strcmp(fmt, "sell"), force_well = fmt && !strcmp(fmt, "well");
  const bool force_wide = fmt && !strcmp(fmt, "wide");
  int rc = 0;
  if (!want_sell && !force_wide) {
    switch (dtype) {
      case RLH_S: rc = well_build<RLH_S>(h, indptr, indices, values, force_well); break;
      case RLH_D: rc = well_build<RLH_D>(h, indptr, indices, values, force_well); break;
      case RLH_C: rc = well_build<RLH_C>(h, indptr, indices, values, false, true); break;
      case RLH_Z: rc = well_build<RLH_Z>(h, indptr, indices, values, false, true); break;
    }
  }
  if (rc == 0 && h->well_blocks == 0 && !want_sell) rc = wide_build(h, indptr, indices, values, force_well || force_wide);
  if (rc == 0 && h->well_blocks == 0 && h->wide_blocks == 0) {
    switch (dtype) {
      case RLH_S: rc = csr_build<RLH_S>(h, indptr, indices, values); break;
      case RLH_D: rc = csr_build<RLH_D>(h, indptr, indices, values); break;
      case RLH_C: rc = csr_build<RLH_C>(h, indptr, indices, values); break;
      case RLH_Z: rc = csr_build<RLH_Z>(h, indptr, indices, values); break;
    }
  }
  if (rc) { rlh_csr_destroy(h); return rc; }
  *out = h;
  return 0;
}

int rlh_csr_create_upper(rlh_csr_t *out, int dtype, int64_t n, const int64_t *indptr, const int32_t *indices,
                         const void *values) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(out != nullptr, "rlh_csr_create_upper: null handle pointer");
  *out = nullptr;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_csr_create_upper: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && indptr != nullptr && indptr[0] == 0, "rlh_csr_create_upper: bad matrix");
  for (int64_t r = 0; r < n; ++r)
    RLH_REQUIRE(indptr[r + 1] >= indptr[r], "rlh_csr_create_upper: indptr decreases at row %lld", (long long)r);
  RLH_REQUIRE(indptr[n] == 0 || (indices && values), "rlh_csr_create_upper: null indices/values");
  switch (dtype) {
    case RLH_S: return csr_from_upper<float>(out, dtype, n, indptr, indices, (const float *)values);
    case RLH_D: return csr_from_upper<double>(out, dtype, n, indptr, indices, (const double *)values);
    case RLH_C: return csr_from_upper<c32>(out, dtype, n, indptr, indices, (const c32 *)values);
    case RLH_Z: return csr_from_upper<c64>(out, dtype, n, indptr, indices, (const c64 *)values);
  }
  return 1;
}

int rlh_csr_destroy(rlh_csr_t h) {
  if (!h) return 0;
  if (ctx().ready) {
    (void)hipStreamSynchronize(ctx().stream);
    if (h->slice_ptr) (void)hipFree(h->slice_ptr);
    if (h->cols) (void)hipFree(h->cols);
    if (h->vals) (void)hipFree(h->vals);
    if (h->well_meta) (void)hipFree(h->well_meta);
    if (h->well_gsrc) (void)hipFree(h->well_gsrc);
    if (h->well_idx) (void)hipFree(h->well_idx);
    if (h->well_vals) (void)hipFree(h->well_vals);
    if (h->well_sched) (void)hipFree(h->well_sched);
    if (h->stk_meta) (void)hipFree(h->stk_meta);
    if (h->stk_member) (void)hipFree(h->stk_member);
    if (h->stk_gsrc) (void)hipFree(h->stk_gsrc);
    if (h->stk_idx) (void)hipFree(h->stk_idx);
    if (h->stk_vals) (void)hipFree(h->stk_vals);
    if (h->stk_pat) (void)hipFree(h->stk_pat);
    if (h->stk_table) (void)hipFree(h->stk_table);
    if (h->stk_dtab) (void)hipFree(h->stk_dtab);
    if (h->stk_sched) (void)hipFree(h->stk_sched);
    for (int k = 0; k < 2; ++k)
      if (h->stk_sched_part[k]) (void)hipFree(h->stk_sched_part[k]);
    for (int k = 0; k < 2; ++k)
      if (h->well_sched_part[k]) (void)hipFree(h->well_sched_part[k]);
    wide_destroy(h);
  }
  delete h;
  return 0;
}

int rlh_csr_info(rlh_csr_t h, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int64_t *device_bytes) {
  RLH_REQUIRE(h != nullptr, "rlh_csr_info: null handle");
  if (n_rows) *n_rows = h->n_rows;
  if (n_cols) *n_cols = h->n_cols;
  if (nnz) *nnz = h->nnz;
  if (device_bytes) *device_bytes = h->device_bytes;
  return 0;
}

}  // extern "C"
// the layout conditions of the bfloat16 Chebyshev step: ONE definition for the launch and for rlh_csr_bf16_ready
static inline bool bf16_layout_ok(const rlh_csr *h) { return h->dtype == RLH_S && h->well_blocks > 0 && h->well_inbounds; }
static inline bool bf16_halo_ok(const rlh_csr *h, int64_t n_own, int64_t ldh) {
  return h->well_aligned && n_own % 8 == 0 && ldh % 8 == 0;
}
extern "C" {

int rlh_csr_layout(rlh_csr_t h, int *layout, int64_t *stored, double *staged_per_slot) {
  RLH_REQUIRE(h != nullptr, "rlh_csr_layout: null handle");
  if (layout) *layout = h->wide_blocks > 0 ? 2 : (h->well_blocks > 0 ? 1 : 0);
  if (stored) *stored = h->padded;
  if (staged_per_slot) *staged_per_slot = h->well_ratio;
  return 0;
}

int rlh_csr_stacks(rlh_csr_t h, int64_t *stacks, double *staged_per_row, double *staged_per_row_stacked) {
  RLH_REQUIRE(h != nullptr, "rlh_csr_stacks: null handle");
  if (stacks) *stacks = h->stk_blocks;
  if (staged_per_row) *staged_per_row = h->well_staged;
  if (staged_per_row_stacked) *staged_per_row_stacked = h->stk_staged;
  return 0;
}

int rlh_csr_bf16_ready(rlh_csr_t h, int64_t n_own, int64_t ldh, int *ok) {
  RLH_REQUIRE(h != nullptr && ok != nullptr, "rlh_csr_bf16_ready: null argument");
  *ok = bf16_layout_ok(h) && h->n_rows <= n_own && n_own <= h->n_cols && (n_own == h->n_cols || bf16_halo_ok(h, n_own, ldh));
  return 0;
}

int rlh_spmm_part(rlh_csr_t h, int part, int64_t m, const void *X, int64_t ldx, int64_t n_own, const void *H,
                  int64_t ldh, void *Y, int64_t ldy) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(h != nullptr, "rlh_spmm: null handle");
  RLH_REQUIRE(part >= 0 && part <= 2, "rlh_spmm: part must be 0 (all rows), 1 (interior) or 2 (boundary)");
  RLH_REQUIRE(m >= 0, "rlh_spmm: negative block size");
  if (m == 0 || h->n_rows == 0) return 0;
  RLH_REQUIRE(Y && (X || n_own == 0), "rlh_spmm: null block pointer");
  RLH_REQUIRE(n_own >= 0 && n_own <= h->n_cols, "rlh_spmm: n_own out of range");
  RLH_REQUIRE(n_own == h->n_cols || H, "rlh_spmm: halo block missing for columns >= n_own");
  RLH_REQUIRE(ldx >= n_own && ldy >= h->n_rows && (!H || ldh >= h->n_cols - n_own),
              "rlh_spmm: leading dimension smaller than the operator size");
  RLH_REQUIRE(X != Y, "rlh_spmm: in-place application is not supported");
  switch (h->dtype) {
    case RLH_S: return spmm_impl<RLH_S>(h, part, m, X, ldx, n_own, H, ldh, Y, ldy);
    case RLH_D: return spmm_impl<RLH_D>(h, part, m, X, ldx, n_own, H, ldh, Y, ldy);
    case RLH_C: return spmm_impl<RLH_C>(h, part, m, X, ldx, n_own, H, ldh, Y, ldy);
    case RLH_Z: return spmm_impl<RLH_Z>(h, part, m, X, ldx, n_own, H, ldh, Y, ldy);
  }
  return 1;
}

int rlh_spmm(rlh_csr_t h, int64_t m, const void *X, int64_t ldx, int64_t n_own, const void *H, int64_t ldh, void *Y,
             int64_t ldy) {
  return rlh_spmm_part(h, 0, m, X, ldx, n_own, H, ldh, Y, ldy);
}

int rlh_spmm_cheb_part(rlh_csr_t h, int part, int64_t m, const void *Y, int64_t ldy, int64_t n_own, const void *H,
                       int64_t ldh, void *P, int64_t ldp, const void *B, int64_t ldb, double cy, double cp, double cb) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(h != nullptr, "rlh_spmm_cheb: null handle");
  RLH_REQUIRE(part >= 0 && part <= 2, "rlh_spmm_cheb: part must be 0 (all rows), 1 (interior) or 2 (boundary)");
  RLH_REQUIRE(m >= 0, "rlh_spmm_cheb: negative block size");
  if (m == 0 || h->n_rows == 0) return 0;
  RLH_REQUIRE(Y && P && B, "rlh_spmm_cheb: null block pointer");
  RLH_REQUIRE(h->n_rows <= n_own && n_own <= h->n_cols, "rlh_spmm_cheb: the operator block must be square in its own rows");
  RLH_REQUIRE(n_own == h->n_cols || H, "rlh_spmm_cheb: halo block missing for columns >= n_own");
  RLH_REQUIRE(ldy >= n_own && ldp >= h->n_rows && ldb >= h->n_rows && (!H || ldh >= h->n_cols - n_own),
              "rlh_spmm_cheb: leading dimension smaller than the operator size");
  RLH_REQUIRE(P != Y && P != B, "rlh_spmm_cheb: P is updated in place and must not alias Y or B");
  switch (h->dtype) {
    case RLH_S: return spmm_impl<RLH_S>(h, part, m, Y, ldy, n_own, H, ldh, P, ldp, B, ldb, cy, cp, cb);
    case RLH_D: return spmm_impl<RLH_D>(h, part, m, Y, ldy, n_own, H, ldh, P, ldp, B, ldb, cy, cp, cb);
    case RLH_C: return spmm_impl<RLH_C>(h, part, m, Y, ldy, n_own, H, ldh, P, ldp, B, ldb, cy, cp, cb);
    case RLH_Z: return spmm_impl<RLH_Z>(h, part, m, Y, ldy, n_own, H, ldh, P, ldp, B, ldb, cy, cp, cb);
  }
  return 1;
}

int rlh_spmm_cheb(rlh_csr_t h, int64_t m, const void *Y, int64_t ldy, int64_t n_own, const void *H, int64_t ldh,
                  void *P, int64_t ldp, const void *B, int64_t ldb, double cy, double cp, double cb) {
  return rlh_spmm_cheb_part(h, 0, m, Y, ldy, n_own, H, ldh, P, ldp, B, ldb, cy, cp, cb);
}

int rlh_spmm_cheb_bf16_part(rlh_csr_t h, int part, int64_t m, const void *Y16, int64_t ldy, int64_t n_own,
                            const void *H16, int64_t ldh, void *P16, int64_t ldp, const void *B16, int64_t ldb,
                            double cy, double cp, double cb) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(h != nullptr, "rlh_spmm_cheb_bf16: null handle");
  RLH_REQUIRE(part >= 0 && part <= 2, "rlh_spmm_cheb_bf16: part must be 0 (all rows), 1 (interior) or 2 (boundary)");
  RLH_REQUIRE(m >= 0, "rlh_spmm_cheb_bf16: negative block size");
  if (m == 0 || h->n_rows == 0) return 0;
  RLH_REQUIRE(bf16_layout_ok(h),
              "rlh_spmm_cheb_bf16: needs a float32 operator in the 1024-row windowed layout (rows of at most 8 "
              "entries) with every staging group inside the column range");
  RLH_REQUIRE(h->n_rows <= n_own && n_own <= h->n_cols, "rlh_spmm_cheb_bf16: the operator block must be square in its own rows");
  RLH_REQUIRE(n_own == h->n_cols || (H16 && bf16_halo_ok(h, n_own, ldh) && ((uintptr_t)H16 % 16) == 0),
              "rlh_spmm_cheb_bf16: a halo block needs n_own and ldh to be multiples of 8, a 16-byte aligned block "
              "and staging groups on multiples of 8 columns");
  RLH_REQUIRE(Y16 && P16 && B16 && P16 != Y16 && P16 != B16, "rlh_spmm_cheb_bf16: bad block pointers");
  RLH_REQUIRE(ldy >= n_own && ldp >= h->n_rows && ldb >= h->n_rows && ldy % 8 == 0 && ldp % 8 == 0 && ldb % 8 == 0 &&
                  ((uintptr_t)Y16 % 16) == 0 && ((uintptr_t)P16 % 16) == 0 && ((uintptr_t)B16 % 16) == 0,
              "rlh_spmm_cheb_bf16: blocks must be 16-byte aligned with leading dimensions that are multiples of 8");
  Context &c = ctx();
  const unsigned short *Y = (const unsigned short *)Y16, *B = (const unsigned short *)B16;
  const unsigned short *H = H16 ? (const unsigned short *)H16 : Y;
  unsigned short *P = (unsigned short *)P16;
  // the stacked layout (8 elements per 16-byte piece: groups on multiples of 8 columns, images of at most 128 groups)
  if (h->stk_blocks > 0 && h->stk_aligned && h->stk_gmax <= kBfImageBytes / 128 && env_int("RLH_SPMM_STACK", 1) != 0 &&
      env_int("RLH_SPMM_STACK_BF16", 1) != 0 &&
      (h->stk_overhang == 0 || (H16 != nullptr && n_own != h->n_cols ? ldh >= h->n_cols - n_own + h->stk_overhang
                                                                      : ldy >= h->n_cols + h->stk_overhang))) {
    if (part != 0)
      if (int rc = stack_split(h, n_own)) return rc;
    const int32_t *ssched = part == 0 ? h->stk_sched : h->stk_sched_part[part - 1];
    const int64_t ssched_len = part == 0 ? h->stk_sched_len : h->stk_sched_part_len[part - 1];
    const int sgrid = part == 0 ? h->stk_grid : h->stk_grid_part[part - 1];
    if (sgrid == 0) return 0;
    constexpr int lds = kBfRingBytes + 16 * kBfOperandBytes;
    const int32_t *bpat = env_int("RLH_SPMM_STACK_PAT", 1) != 0 ? h->stk_pat : nullptr;
#define RLH_BF_STACK(VPS_, ...)                                                                                       \
    do {                                                                                                              \
      static bool attr = false;                                                                                       \
      if (!attr) {                                                                                                    \
        RLH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&well_stack_cheb_bf16_kernel<kStkR, VPS_ __VA_ARGS__>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                \
        attr = true;                                                                                                  \
      }                                                                                                               \
      hipLaunchKernelGGL((well_stack_cheb_bf16_kernel<kStkR, VPS_ __VA_ARGS__>), dim3((unsigned)sgrid), dim3(1024), lds, c.stream, \
                         h->stk_meta, h->stk_member, h->stk_gsrc, h->stk_idx,                                         \
                         bpat ? (const float *)h->stk_table : (const float *)h->stk_vals, bpat,                       \
                         bpat ? h->stk_dtab : nullptr, h->n_rows,                                                     \
                         ssched, ssched_len, Y, ldy, n_own, H, ldh, P, ldp, B, ldb, (int)m, (float)cy, (float)cp,     \
                         (float)cb);                                                                                  \
    } while (0)
    switch (env_int("RLH_SPMM_BF16_DBG", 0)) {
      case 1: RLH_BF_STACK(2, , 1); break;
      case 2: RLH_BF_STACK(2, , 2); break;
      case 3: RLH_BF_STACK(2, , 3); break;
      case 4: RLH_BF_STACK(2, , 4); break;
      case 5: RLH_BF_STACK(2, , 5); break;
      case 6: RLH_BF_STACK(2, , 6); break;
      case 7: RLH_BF_STACK(2, , 7); break;
      default:
        if (env_int("RLH_SPMM_BF16_VPS", 2) >= 2) RLH_BF_STACK(2);
        else RLH_BF_STACK(1);
    }
#undef RLH_BF_STACK
    RLH_HIP(hipGetLastError());
    return 0;
  }
  if (part != 0)
    if (int rc = well_split(h, n_own)) return rc;
  const int32_t *sched = part == 0 ? h->well_sched : h->well_sched_part[part - 1];
  const int64_t sched_len = part == 0 ? h->well_sched_len : h->well_sched_part_len[part - 1];
  const int64_t nb = part == 0 ? h->well_grid : h->well_grid_part[part - 1];
  if (nb == 0) return 0;
#define RLH_BF_LAUNCH(W)                                                                                          \
  hipLaunchKernelGGL((well_cheb_bf16_kernel<W>), dim3((unsigned)nb), dim3(1024), 0, c.stream, h->well_meta,       \
                     h->well_gsrc, h->well_idx, (const float *)h->well_vals, h->n_rows, sched, sched_len, Y, ldy, \
                     n_own, H, ldh, P, ldp, B, ldb, (int)m, (float)cy, (float)cp, (float)cb)
  RLH_BF_LAUNCH(8);
#undef RLH_BF_LAUNCH
  RLH_HIP(hipGetLastError());
  return 0;
}

int rlh_spmm_cheb_bf16(rlh_csr_t h, int64_t m, const void *Y16, int64_t ldy, void *P16, int64_t ldp, const void *B16,
                       int64_t ldb, double cy, double cp, double cb) {
  RLH_REQUIRE(h != nullptr, "rlh_spmm_cheb_bf16: null handle");
  RLH_REQUIRE(h->n_rows == h->n_cols, "rlh_spmm_cheb_bf16: needs a square float32 operator (a row shard goes through "
                                      "rlh_spmm_cheb_bf16_part)");
  return rlh_spmm_cheb_bf16_part(h, 0, m, Y16, ldy, h->n_cols, nullptr, 0, P16, ldp, B16, ldb, cy, cp, cb);
}

int rlh_gather_rows_bf16(int64_t nidx, const int64_t *d_idx, int64_t m, const void *X16, int64_t ldx, void *Out16,
                         int64_t ldo) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(nidx >= 0 && m >= 0, "rlh_gather_rows_bf16: negative size");
  if (nidx == 0 || m == 0) return 0;
  RLH_REQUIRE(d_idx && X16 && Out16 && ldo >= nidx, "rlh_gather_rows_bf16: bad arguments");
  Context &c = ctx();
  int64_t nb = (nidx + 255) / 256;
  const int64_t cap = ((int64_t)c.num_cu * 8 + m - 1) / m;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((gather_rows_kernel<unsigned short>), dim3((unsigned)nb, (unsigned)m), dim3(256), 0, c.stream,
                     d_idx, nidx, (const unsigned short *)X16, ldx, (unsigned short *)Out16, ldo);
  RLH_HIP(hipGetLastError());
  return 0;
}

int rlh_gather_rows(int dtype, int64_t nidx, const int64_t *d_idx, int64_t m, const void *X, int64_t ldx, void *Out,
                    int64_t ldo) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_gather_rows: unknown dtype %d", dtype);
  RLH_REQUIRE(nidx >= 0 && m >= 0, "rlh_gather_rows: negative size");
  if (nidx == 0 || m == 0) return 0;
  RLH_REQUIRE(d_idx && X && Out, "rlh_gather_rows: null pointer");
  RLH_REQUIRE(ldo >= nidx, "rlh_gather_rows: output leading dimension smaller than the index count");
  switch (dtype) {
    case RLH_S: return gather_rows_impl<RLH_S>(nidx, d_idx, m, X, ldx, Out, ldo);
    case RLH_D: return gather_rows_impl<RLH_D>(nidx, d_idx, m, X, ldx, Out, ldo);
    case RLH_C: return gather_rows_impl<RLH_C>(nidx, d_idx, m, X, ldx, Out, ldo);
    case RLH_Z: return gather_rows_impl<RLH_Z>(nidx, d_idx, m, X, ldx, Out, ldo);
  }
  return 1;
}

}  // extern "C"

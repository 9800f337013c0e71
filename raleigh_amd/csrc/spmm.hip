// K13: sparse symmetric/Hermitian operator Y = A X on a block of vectors.
//
// The reference keeps triu(A) in 1-based CSR and calls mkl_?csrmm with the
// 'SUNF'/'HUNF' descriptor on the column-major n x m block
// (raleigh/algebra/sparse_mkl.py:16-48, raleigh/algebra/mkl_wrap.py:204-276).
// Here the caller hands over the FULL 0-based CSR (both triangles); at creation
// it is re-laid-out on the host as sliced ELLPACK with slice height 64 (one
// wavefront per slice, one lane per row): within a slice the entries are stored
// column-major, so the value and column-index loads of a wave are perfectly
// coalesced.  Each lane keeps JT accumulators (one per vector of the block) and,
// per stored entry, gathers X[col, j] for the JT vectors: for banded / stencil /
// FE matrices neighbouring rows reference neighbouring columns, so these gathers
// are coalesced across the wave and re-use lines through L2 / Infinity Cache.
// The kernel is HBM-bound: nnz*(s+4) matrix bytes + one read of X + one write of Y.
#include <stdlib.h>

#include <vector>

#include "common.h"

struct rlh_csr {
  int dtype;
  int64_t n_rows, n_cols, nnz;
  int64_t n_slices;
  int64_t padded;          // stored entries incl. padding
  int64_t *slice_ptr;      // device, n_slices + 1 (entry offsets)
  int32_t *cols;           // device, padded
  void *vals;              // device, padded
  int64_t device_bytes;
};

namespace rlh {

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
// Fused Chebyshev step (CHEB): with t = A d the row's results update r -= t, dn = alpha d + beta r,
// y += dn in the same pass (device polynomial preconditioner); dn must not alias d, which other
// rows are still gathering.
template <typename T>
struct ChebArgs {
  T *R; int64_t ldr;
  T *Dn; int64_t lddn;
  double alpha, beta;
};

__device__ __forceinline__ float  scale_of(double s, float v)  { return (float)s * v; }
__device__ __forceinline__ double scale_of(double s, double v) { return s * v; }
__device__ __forceinline__ c32 scale_of(double s, c32 v) { return c32{(float)s * v.re, (float)s * v.im}; }
__device__ __forceinline__ c64 scale_of(double s, c64 v) { return c64{s * v.re, s * v.im}; }
__device__ __forceinline__ float  sub_of(float a, float b)   { return a - b; }
__device__ __forceinline__ double sub_of(double a, double b) { return a - b; }
__device__ __forceinline__ c32 sub_of(c32 a, c32 b) { return c32{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ c64 sub_of(c64 a, c64 b) { return c64{a.re - b.re, a.im - b.im}; }

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
              T *rp = cheb.R + row + jc * cheb.ldr;
              const T rr = sub_of(*rp, acc[j]);                               // r -= A d
              const T dn = add_of(scale_of(cheb.alpha, Xp[row + (int64_t)j * ldx]), scale_of(cheb.beta, rr));
              *rp = rr;
              cheb.Dn[row + jc * cheb.lddn] = dn;                             // dn = alpha d + beta r
              T *yp = Yp + row + (int64_t)j * ldy;
              *yp = add_of(*yp, dn);                                          // y += dn
            }
        }
      }
    }
  }
}

static int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

template <typename T, int JT, int TT>
static int launch_spmm_t(const rlh_csr *h, int64_t m, const T *X, int64_t ldx, int64_t n_own, const T *H, int64_t ldh,
                       T *Y, int64_t ldy, const ChebArgs<T> *cheb) {
  Context &c = ctx();
  static int per_cu = 0;
  if (per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sell_spmm_kernel<T, JT, TT, false>, 256, 0) != hipSuccess || nb < 1)
      nb = 1;
    per_cu = nb > 8 ? 8 : nb;
    const int cap = env_int("RLH_SPMM_WG_PER_CU", 2);       // 0: as many as fit (tunable; 2 measured best)
    if (cap > 0 && per_cu > cap) per_cu = cap;
  }
  static const int chunk = env_int("RLH_SPMM_CHUNK", 64);      // slices per row chunk (tunable)
  static const int tpb_cap = env_int("RLH_SPMM_TPB", 1);       // waves of a workgroup on one slice (tunable)
  const int ntiles = (int)((m + JT - 1) / JT);
  int tiles_per_block = ntiles >= 4 ? 4 : (ntiles >= 2 ? 2 : 1);
  if (tiles_per_block > tpb_cap) tiles_per_block = tpb_cap >= 2 ? 2 : 1;
  const int slices_per_block = 4 / tiles_per_block;
  int64_t nb = (int64_t)c.num_cu * per_cu;
  const int64_t need = ((h->n_slices + slices_per_block - 1) / slices_per_block + 7) / 8 * 8;
  if (nb > need) nb = need;
  nb = (nb + 7) / 8 * 8;
  if (cheb)
    hipLaunchKernelGGL((sell_spmm_kernel<T, JT, TT, true>), dim3((unsigned)nb), dim3(256), 0, c.stream, h->slice_ptr,
                       h->cols, (const T *)h->vals, h->n_rows, h->n_slices, X, ldx, n_own, H, ldh, Y, ldy, (int)m,
                       chunk < 1 ? 1 : chunk, ntiles, tiles_per_block, *cheb);
  else
    hipLaunchKernelGGL((sell_spmm_kernel<T, JT, TT, false>), dim3((unsigned)nb), dim3(256), 0, c.stream, h->slice_ptr,
                       h->cols, (const T *)h->vals, h->n_rows, h->n_slices, X, ldx, n_own, H, ldh, Y, ldy, (int)m,
                       chunk < 1 ? 1 : chunk, ntiles, tiles_per_block, ChebArgs<T>{});
  RLH_HIP(hipGetLastError());
  return 0;
}

template <typename T, int JT>
static int launch_spmm(const rlh_csr *h, int64_t m, const T *X, int64_t ldx, int64_t n_own, const T *H, int64_t ldh,
                       T *Y, int64_t ldy, const ChebArgs<T> *cheb) {
  static const int tt = env_int("RLH_SPMM_TT", 1);             // entries per register group (tunable)
  if (tt >= 8) return launch_spmm_t<T, JT, 8>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
  if (tt >= 4) return launch_spmm_t<T, JT, 4>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
  return launch_spmm_t<T, JT, 1>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
}

template <int DT>
static int spmm_impl(const rlh_csr *h, int64_t m, const void *X_, int64_t ldx, int64_t n_own, const void *H_,
                     int64_t ldh, void *Y_, int64_t ldy, void *R_ = nullptr, int64_t ldr = 0, void *Dn_ = nullptr,
                     int64_t lddn = 0, double alpha = 0.0, double beta = 0.0) {
  using T = typename DType<DT>::T;
  constexpr int JTMAX = DType<DT>::cplx ? 16 : 32;
  const T *X = (const T *)X_, *H = (const T *)H_;
  T *Y = (T *)Y_;
  ChebArgs<T> cargs{(T *)R_, ldr, (T *)Dn_, lddn, alpha, beta};
  const ChebArgs<T> *cheb = R_ ? &cargs : nullptr;
  static const int jt_cap = env_int("RLH_SPMM_JT", 16);        // vectors per lane tile (tunable; 16 measured best at m = 32 fp64)
  if (m <= 4 || jt_cap <= 4) return launch_spmm<T, 4>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
  if (m <= 8 || jt_cap <= 8) return launch_spmm<T, 8>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
  if (m <= 16 || JTMAX == 16 || jt_cap <= 16) return launch_spmm<T, 16>(h, m, X, ldx, n_own, H, ldh, Y, ldy, cheb);
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
  const int64_t nnz = indptr[n_rows] - indptr[0];
  RLH_REQUIRE(indptr[0] == 0 && nnz >= 0, "rlh_csr_create: indptr must be 0-based and non-decreasing");
  RLH_REQUIRE(nnz == 0 || (indices && values), "rlh_csr_create: null indices/values");
  for (int64_t r = 0; r < n_rows; ++r)
    RLH_REQUIRE(indptr[r + 1] >= indptr[r], "rlh_csr_create: indptr decreases at row %lld", (long long)r);
  for (int64_t e = 0; e < nnz; ++e)
    RLH_REQUIRE(indices[e] >= 0 && indices[e] < n_cols, "rlh_csr_create: column index %d out of range at entry %lld",
                indices[e], (long long)e);
  rlh_csr *h = new rlh_csr();
  h->dtype = dtype; h->n_rows = n_rows; h->n_cols = n_cols; h->nnz = nnz;
  h->slice_ptr = nullptr; h->cols = nullptr; h->vals = nullptr;
  int rc = 0;
  switch (dtype) {
    case RLH_S: rc = csr_build<RLH_S>(h, indptr, indices, values); break;
    case RLH_D: rc = csr_build<RLH_D>(h, indptr, indices, values); break;
    case RLH_C: rc = csr_build<RLH_C>(h, indptr, indices, values); break;
    case RLH_Z: rc = csr_build<RLH_Z>(h, indptr, indices, values); break;
  }
  if (rc) { rlh_csr_destroy(h); return rc; }
  *out = h;
  return 0;
}

int rlh_csr_destroy(rlh_csr_t h) {
  if (!h) return 0;
  if (ctx().ready) {
    (void)hipStreamSynchronize(ctx().stream);
    if (h->slice_ptr) (void)hipFree(h->slice_ptr);
    if (h->cols) (void)hipFree(h->cols);
    if (h->vals) (void)hipFree(h->vals);
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

int rlh_spmm(rlh_csr_t h, int64_t m, const void *X, int64_t ldx, int64_t n_own, const void *H, int64_t ldh, void *Y,
             int64_t ldy) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(h != nullptr, "rlh_spmm: null handle");
  RLH_REQUIRE(m >= 0, "rlh_spmm: negative block size");
  if (m == 0 || h->n_rows == 0) return 0;
  RLH_REQUIRE(Y && (X || n_own == 0), "rlh_spmm: null block pointer");
  RLH_REQUIRE(n_own >= 0 && n_own <= h->n_cols, "rlh_spmm: n_own out of range");
  RLH_REQUIRE(n_own == h->n_cols || H, "rlh_spmm: halo block missing for columns >= n_own");
  RLH_REQUIRE(ldx >= n_own && ldy >= h->n_rows && (!H || ldh >= h->n_cols - n_own),
              "rlh_spmm: leading dimension smaller than the operator size");
  RLH_REQUIRE(X != Y, "rlh_spmm: in-place application is not supported");
  switch (h->dtype) {
    case RLH_S: return spmm_impl<RLH_S>(h, m, X, ldx, n_own, H, ldh, Y, ldy);
    case RLH_D: return spmm_impl<RLH_D>(h, m, X, ldx, n_own, H, ldh, Y, ldy);
    case RLH_C: return spmm_impl<RLH_C>(h, m, X, ldx, n_own, H, ldh, Y, ldy);
    case RLH_Z: return spmm_impl<RLH_Z>(h, m, X, ldx, n_own, H, ldh, Y, ldy);
  }
  return 1;
}

int rlh_spmm_cheb(rlh_csr_t h, int64_t m, const void *D, int64_t ldd, int64_t n_own, const void *H, int64_t ldh,
                  void *R, int64_t ldr, void *Dn, int64_t lddn, void *Y, int64_t ldy, double alpha, double beta) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(h != nullptr, "rlh_spmm_cheb: null handle");
  RLH_REQUIRE(m >= 0, "rlh_spmm_cheb: negative block size");
  if (m == 0 || h->n_rows == 0) return 0;
  RLH_REQUIRE(D && R && Dn && Y, "rlh_spmm_cheb: null block pointer");
  RLH_REQUIRE(h->n_rows <= n_own && n_own <= h->n_cols, "rlh_spmm_cheb: the operator block must be square in its own rows");
  RLH_REQUIRE(n_own == h->n_cols || H, "rlh_spmm_cheb: halo block missing for columns >= n_own");
  RLH_REQUIRE(ldd >= n_own && ldr >= h->n_rows && lddn >= h->n_rows && ldy >= h->n_rows &&
                  (!H || ldh >= h->n_cols - n_own),
              "rlh_spmm_cheb: leading dimension smaller than the operator size");
  RLH_REQUIRE(Dn != D && Dn != R && Dn != Y && D != R && D != Y && R != Y, "rlh_spmm_cheb: the four blocks must be distinct");
  switch (h->dtype) {
    case RLH_S: return spmm_impl<RLH_S>(h, m, D, ldd, n_own, H, ldh, Y, ldy, R, ldr, Dn, lddn, alpha, beta);
    case RLH_D: return spmm_impl<RLH_D>(h, m, D, ldd, n_own, H, ldh, Y, ldy, R, ldr, Dn, lddn, alpha, beta);
    case RLH_C: return spmm_impl<RLH_C>(h, m, D, ldd, n_own, H, ldh, Y, ldy, R, ldr, Dn, lddn, alpha, beta);
    case RLH_Z: return spmm_impl<RLH_Z>(h, m, D, ldd, n_own, H, ldh, Y, ldy, R, ldr, Dn, lddn, alpha, beta);
  }
  return 1;
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

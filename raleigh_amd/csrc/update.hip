// K3/K4 block update Out = beta*Out + alpha * X * Q and the column-wise
// elementwise kernels K5-K8 (axpy, column axpy, copy / gather, scale, conj).
//
// Block update: one lane owns one row.  The k x m coefficient matrix lives in
// device memory (staged through the pinned ring) and is read with wave-uniform
// indices, so it arrives through the scalar cache into SGPRs and feeds
// v_fma_{f32,f64} directly; X columns are read and Out columns written with
// fully coalesced wave accesses.  Arithmetic intensity at m = k = 32 fp64 is
// 4 flop/byte: the VALU needs ~31% utilisation at the HBM rate, so the kernel is
// HBM-bound; no LDS is needed because nothing is shared between lanes.
//
// Replaces: cudaMalloc + H2D + cublas?gemm(NoTrans, Trans|NoTrans) + cudaFree
// (raleigh/algebra/dense_cublas.py:271-342), m axpy / scal / copy calls
// (dense_cublas.py:133-172, 343-350).
#include <algorithm>

#include "common.h"

namespace rlh {

constexpr int kUnrollK = 4;

static int env_flag(const char *name, int dflt) {
  const char *e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

// Two sources: Out = beta*Out + X*Q[0:kpad] + X2*Q[kpad:kpad+k2] in ONE pass (k2 = 0: one source).
// One lane owns RV consecutive rows (RV * sizeof(T) = 16 bytes when the blocks are 16-byte
// aligned and the accumulators fit, else 1): every access to X and Out then moves 16 bytes per
// lane, which the vector-memory path retires at up to twice the bytes per cycle of 8-byte accesses.
template <typename T, int RV> struct alignas(RV * sizeof(T) >= 16 ? 16 : sizeof(T)) RowVec { T e[RV]; };

// 16-byte accesses with or without the non-temporal hint (experiment switch RLH_UPDATE_NT: bit 0 loads, bit 1 stores)
typedef unsigned upd_u4 __attribute__((ext_vector_type(4)));
// (the hint is a TEMPLATE parameter: chosen at run time, every access sits behind a branch and the compiler
// waits for vmcnt(0) where the paths join -- measured on the Gram kernel: no gain from the hint at all)
template <bool NT, typename V> __device__ __forceinline__ V load16(const V *p) {
  static_assert(sizeof(V) == 16, "16-byte pieces");
  union { upd_u4 u; V v; } c;
  if constexpr (NT) c.u = __builtin_nontemporal_load(reinterpret_cast<const upd_u4 *>(p));
  else c.u = *reinterpret_cast<const upd_u4 *>(p);
  return c.v;
}
template <bool NT, typename V> __device__ __forceinline__ void store16(V *p, const V &v) {
  static_assert(sizeof(V) == 16, "16-byte pieces");
  union { upd_u4 u; V v; } c;
  c.v = v;
  if constexpr (NT) __builtin_nontemporal_store(c.u, reinterpret_cast<upd_u4 *>(p));
  else *reinterpret_cast<upd_u4 *>(p) = c.u;
}

template <bool ALIGNED, bool NT, typename V> __device__ __forceinline__ V ldv(const V *p) {
  if constexpr (ALIGNED) return load16<NT>(p);
  else return *p;
}
template <bool ALIGNED, bool NT, typename V> __device__ __forceinline__ void stv(V *p, const V &v) {
  if constexpr (ALIGNED) store16<NT>(p, v);
  else *p = v;
}

template <typename T, int JT, bool BETA, int RV, bool NT>
__global__ __launch_bounds__(256) void block_update_kernel(const T *__restrict__ X, int64_t ldx,
                                                           const T *__restrict__ X2, int64_t ldx2, int k2, int kpad,
                                                           T *__restrict__ Out, int64_t ldo,
                                                           const T *__restrict__ Q, int ldq, int64_t n, int k,
                                                           int m, T *__restrict__ Out2, int64_t ldo2, int msplit) {
  using V = RowVec<T, RV>;
  // workgroup index = row block x panels + panel: the panels of one row block (the two result blocks of
  // rlh_block_update2x2 take two) are dispatched side by side, so the second one finds the rows of X that the first
  // just read in the L2 / Infinity Cache instead of streaming both sources from HBM again
  const int npanels = (m + JT - 1) / JT;
  const int j0 = (int)(blockIdx.x % npanels) * JT;
  const int64_t rowblock = blockIdx.x / npanels;
  const int jv = (m - j0) < JT ? (m - j0) : JT;      // valid output columns of this panel
  const T *__restrict__ Qp = Q + j0;
  // output column j of the panel: columns < msplit live in Out, the others in Out2 (two result
  // blocks from one pass over the sources: rlh_block_update2x2)
  auto out_col = [&](int j) -> T * {
    const int gj = j0 + j;
    return gj < msplit ? Out + (int64_t)gj * ldo : Out2 + (int64_t)(gj - msplit) * ldo2;
  };
  // one row group per lane, no loop: the grid covers every row (launch_update_rv), so the resident workgroups are a
  // compact window sweeping the blocks front to back
  for (int64_t row = (rowblock * 256 + threadIdx.x) * RV, once = 0; once < 1; ++once) {
    // every row group of the wave is complete (wave-uniform): straight-line loads, the columns of X
    // in a two-stage register pipeline -- the loads of the next kUnrollK columns are in flight during
    // the FMAs of the current ones (with one predicate per load the compiler put every load behind a
    // branch and an s_waitcnt vmcnt(0) in front of the FMAs: nothing overlapped inside a wave)
    const int64_t wave_row0 = row - (int64_t)(threadIdx.x & 63) * RV;
    if constexpr (RV > 1) if (wave_row0 + 64 * RV <= n) {
      T acc[RV][JT];
#pragma unroll
      for (int j = 0; j < JT; ++j) {
#pragma unroll
        for (int r = 0; r < RV; ++r) acc[r][j] = zero_of(T{});
        if (BETA && j < jv) {
          const V o = load16<NT>(reinterpret_cast<const V *>(out_col(j) + row));
#pragma unroll
          for (int r = 0; r < RV; ++r) acc[r][j] = o.e[r];
        }
      }
      auto accumulate_fast = [&](const T *__restrict__ S, int64_t lds_, int kk, int qrow0) {
        if (kk <= 0) return;
        V xa[kUnrollK], xb[kUnrollK];
        auto load = [&](V (&x)[kUnrollK], int i) {
#pragma unroll
          for (int u = 0; u < kUnrollK; ++u) {
            const int col = (i + u) < kk ? (i + u) : (kk - 1);       // Q rows >= kk are zero
            x[u] = load16<NT>(reinterpret_cast<const V *>(S + row + (int64_t)col * lds_));
          }
        };
        auto fma = [&](const V (&x)[kUnrollK], int i) {
#pragma unroll
          for (int u = 0; u < kUnrollK; ++u)
#pragma unroll
            for (int j = 0; j < JT; ++j) {
              const T q = Qp[(qrow0 + i + u) * ldq + j];
#pragma unroll
              for (int r = 0; r < RV; ++r) fma_acc(acc[r][j], x[u].e[r], q);
            }
        };
        load(xa, 0);
        for (int i = 0; i < kk; i += 2 * kUnrollK) {
          load(xb, i + kUnrollK < kk ? i + kUnrollK : i);
          fma(xa, i);
          load(xa, i + 2 * kUnrollK < kk ? i + 2 * kUnrollK : i);
          if (i + kUnrollK < kk) fma(xb, i + kUnrollK);
        }
      };
      accumulate_fast(X, ldx, k, 0);
      accumulate_fast(X2, ldx2, k2, kpad);
#pragma unroll
      for (int j = 0; j < JT; ++j)
        if (j < jv) {
          V o;
#pragma unroll
          for (int r = 0; r < RV; ++r) o.e[r] = acc[r][j];
          store16<NT>(reinterpret_cast<V *>(out_col(j) + row), o);
        }
      continue;
    }
    if (row >= n) continue;
    const bool whole = row + RV <= n;                // the last lane group of an n not divisible by RV
    T acc[RV][JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
#pragma unroll
      for (int r = 0; r < RV; ++r) acc[r][j] = zero_of(T{});
      if (BETA && j < jv) {
        if (whole) {
          const V o = *reinterpret_cast<const V *>(out_col(j) + row);
#pragma unroll
          for (int r = 0; r < RV; ++r) acc[r][j] = o.e[r];
        } else {
          for (int r = 0; r < RV; ++r)
            if (row + r < n) acc[r][j] = out_col(j)[row + r];
        }
      }
    }
    auto accumulate = [&](const T *__restrict__ S, int64_t lds_, int kk, int qrow0) {
      for (int i = 0; i < kk; i += kUnrollK) {
        T x[kUnrollK][RV];
#pragma unroll
        for (int u = 0; u < kUnrollK; ++u) {
          const int col = (i + u) < kk ? (i + u) : (kk - 1);     // Q rows >= kk are zero
          if (whole) {
            const V xv = *reinterpret_cast<const V *>(S + row + (int64_t)col * lds_);
#pragma unroll
            for (int r = 0; r < RV; ++r) x[u][r] = xv.e[r];
          } else {
#pragma unroll
            for (int r = 0; r < RV; ++r) x[u][r] = (row + r < n) ? S[row + r + (int64_t)col * lds_] : zero_of(T{});
          }
        }
#pragma unroll
        for (int u = 0; u < kUnrollK; ++u)
#pragma unroll
          for (int j = 0; j < JT; ++j) {
            const T q = Qp[(qrow0 + i + u) * ldq + j];
#pragma unroll
            for (int r = 0; r < RV; ++r) fma_acc(acc[r][j], x[u][r], q);
          }
      }
    };
    accumulate(X, ldx, k, 0);
    accumulate(X2, ldx2, k2, kpad);
#pragma unroll
    for (int j = 0; j < JT; ++j)
      if (j < jv) {
        if (whole) {
          V o;
#pragma unroll
          for (int r = 0; r < RV; ++r) o.e[r] = acc[r][j];
          *reinterpret_cast<V *>(out_col(j) + row) = o;
        } else {
          for (int r = 0; r < RV; ++r)
            if (row + r < n) out_col(j)[row + r] = acc[r][j];
        }
      }
  }
}

template <typename T> struct HostScalar;
template <> struct HostScalar<float>  { static float  mul(const double *a, float q)  { return (float)(a[0] * q); } };
template <> struct HostScalar<double> { static double mul(const double *a, double q) { return a[0] * q; } };
template <> struct HostScalar<c32> {
  static c32 mul(const double *a, c32 q) {
    return c32{(float)(a[0] * q.re - a[1] * q.im), (float)(a[0] * q.im + a[1] * q.re)};
  }
};
template <> struct HostScalar<c64> {
  static c64 mul(const double *a, c64 q) { return c64{a[0] * q.re - a[1] * q.im, a[0] * q.im + a[1] * q.re}; }
};

// non-temporal hint on the block update's loads and stores: RLH_UPDATE_NT = 0 / 1 forces, else for operands beyond
// the Infinity Cache
static int update_nt(int64_t bytes) {
  const char *e = getenv("RLH_UPDATE_NT");
  if (e && *e) return atoi(e) != 0;
  return bytes > ((int64_t)192 << 20);
}

template <typename T, int JT, int RV>
static int launch_update_rv(const T *X, int64_t ldx, T *Out, int64_t ldo, const T *Qd, int ldq, int64_t n, int k, int m,
                            int beta, const T *X2, int64_t ldx2, int k2, int kpad, T *Out2, int64_t ldo2, int msplit) {
  Context &c = ctx();
  int64_t nbx = ((n + RV - 1) / RV + 255) / 256;
  // one row group per lane and no loop (see row_blocks below)
  const int64_t npanels = (m + JT - 1) / JT;
  RLH_REQUIRE(nbx * npanels <= 0x7fffffff, "rlh_block_update: %lld rows exceed the grid", (long long)n);
  dim3 grid((unsigned)(nbx * npanels));
  const int nt = update_nt((int64_t)n * (k + k2 + (beta ? 2 : 1) * m) * (int64_t)sizeof(T));
#define RLH_UPD(BETA_, NT_)                                                                                              \
  hipLaunchKernelGGL((block_update_kernel<T, JT, BETA_, RV, NT_>), grid, dim3(256), 0, c.stream, X, ldx, X2, ldx2, k2, kpad, \
                     Out, ldo, Qd, ldq, n, k, m, Out2, ldo2, msplit)
  if constexpr (RV > 1) {
    if (nt) { if (beta) RLH_UPD(true, true); else RLH_UPD(false, true); }
    else { if (beta) RLH_UPD(true, false); else RLH_UPD(false, false); }
  } else {
    if (beta) RLH_UPD(true, false); else RLH_UPD(false, false);
  }
#undef RLH_UPD
  RLH_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- complex block update on the matrix cores
// Out = beta Out + X Q for the complex types with k, m >= 16: 8 n k m real flops on 32 n m bytes (complex128,
// k = m = 64: 16 flop/byte), i.e. bound by arithmetic, and the SGPR-fed VALU kernel above reaches 42 % of the
// fp64 rate there (2.0 ms at n = 2 x 10^6).  Here the product runs as four real MFMA products
//   Out_r = X_r Q_r - X_i Q_i,   Out_i = X_r Q_i + X_i Q_r
// on v_mfma_{f64,f32}_16x16x4 with the ROWS of X on the lanes: D^T (16 output columns x 16 rows) =
// Q^T (16 x 4) X^T (4 x 16), so a lane holds (re, im) of one row and 4 output columns and stores them as
// 16-byte pieces, 16 lanes = 256 contiguous bytes of a column of Out.
//  * X: each lane loads ITS 16-byte element (row lane & 15, column 4 s + (lane >> 4)) of every k-step
//    straight from global memory (256-byte runs), one row tile ahead of the arithmetic; no LDS staging;
//  * Q: re-laid-out once per workgroup into the LDS in fragment order [column tile][k-step][lane] = (re, im) of
//    Q[4 s + (lane >> 4), 16 t + (lane & 15)], one ds_read_b128 per k-step and column tile;
//  * a wave takes 16 rows and walks all column tiles with the X fragments in registers: X is read once,
//    Out written (and for beta = 1 read) once.
template <typename R> struct Mfma16x4;
template <> struct Mfma16x4<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int out_row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mfma16x4<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int out_row(int lane, int reg) { return (lane >> 4) * 4 + reg; }
};

constexpr int kMfmaUpdKS = 32;               // k-steps (of 4 columns) whose X fragments a wave keeps in registers

template <typename T, typename R, bool BETA, int KS>
__global__ __launch_bounds__(256) void block_update_mfma_kernel(const T *__restrict__ X, int64_t ldx, int k1, int ks1,
                                                                const T *__restrict__ X2, int64_t ldx2, int k2, int ks2,
                                                                T *__restrict__ Out, int64_t ldo, T *__restrict__ Out2,
                                                                int64_t ldo2, int msplit, const T *__restrict__ Q, int ldq,
                                                                int64_t n, int m) {
  using M = Mfma16x4<R>;
  using acc_t = typename M::acc_t;
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  T *qf = reinterpret_cast<T *>(lds_raw);                 // [column tile][k-step][lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int nks = ks1 + ks2, ntile = (m + 15) / 16;
  for (int idx = tid; idx < ntile * nks * 64; idx += 256) {
    const int l = idx & 63, s = (idx >> 6) % nks, t = (idx >> 6) / nks;
    const int col = 16 * t + (l & 15);
    T q = zero_of(T{});
    if (col < m) q = Q[(int64_t)(4 * s + (l >> 4)) * ldq + col];   // (rows past k1 / k2 inside their padded parts are zero)
    qf[idx] = q;
  }
  __syncthreads();
  const int64_t ntiles_rows = (n + 15) / 16;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (tid >> 6), nwaves = (int64_t)gridDim.x * 4;
  const int kk = lane >> 4, rl = lane & 15;
  T xf[KS];
  auto load_x = [&](int64_t rt) {
    int64_t r = rt * 16 + rl;
    r = r < n ? r : n - 1;                                 // rows past the end repeat the last (never stored)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s < nks) {                                       // (workgroup-uniform)
        const bool first = s < ks1;
        int c = first ? 4 * s + kk : 4 * (s - ks1) + kk;
        const int kmax = first ? k1 : k2;
        c = c < kmax ? c : kmax - 1;                       // columns inside the zero padding of Q
        xf[s] = first ? X[r + (int64_t)c * ldx] : X2[r + (int64_t)c * ldx2];
      }
    }
  };
  for (int64_t rt = wave_id; rt < ntiles_rows; rt += nwaves) {
    load_x(rt);
    const int64_t row = rt * 16 + rl;
    // beta = 1: the result fragments of column tile t + 1 are requested before the MFMAs of tile t (unconditional
    // loads of clamped addresses: no branch between a load and its use)
    const int64_t rowc = row < n ? row : n - 1;
    T onext[4];
    auto load_out = [&](int t_, T (&o)[4]) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int j = 16 * t_ + M::out_row(lane, g);
        j = j < m ? j : m - 1;
        o[g] = j < msplit ? Out[rowc + (int64_t)j * ldo] : Out2[rowc + (int64_t)(j - msplit) * ldo2];
      }
    };
    if (BETA) load_out(0, onext);
    for (int t = 0; t < ntile; ++t) {
      acc_t dr = {(R)0, (R)0, (R)0, (R)0}, di = {(R)0, (R)0, (R)0, (R)0};
      if (BETA) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { dr[g] = onext[g].re; di[g] = onext[g].im; }
        load_out(t + 1 < ntile ? t + 1 : t, onext);
      }
      const T *qt = qf + (int64_t)t * nks * 64 + lane;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s < nks) {
          const T q = qt[s * 64];
          const T x = xf[s];
          dr = M::run(q.re, x.re, dr);
          dr = M::run(q.im, -x.im, dr);
          di = M::run(q.im, x.re, di);
          di = M::run(q.re, x.im, di);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = 16 * t + M::out_row(lane, g);
        if (j < m && row < n) {
          T *o = j < msplit ? Out + row + (int64_t)j * ldo : Out2 + row + (int64_t)(j - msplit) * ldo2;
          *o = T{dr[g], di[g]};
        }
      }
    }
  }
}

// The matrix-core path: complex types, every k-step in registers (KS bounds them at compile time: 8, 16 or 32 k-steps,
// so that k <= 64 leaves room for two or three waves per SIMD), Q in the LDS.
template <typename T, int KS>
static int launch_update_mfma_ks(const T *X, int64_t ldx, T *Out, int64_t ldo, const T *Qd, int ldq, int64_t n, int k, int m,
                                 int beta, const T *X2, int64_t ldx2, int k2, int ks1, int ks2, T *Out2, int64_t ldo2,
                                 int msplit, size_t lds) {
  using R = decltype(T{}.re);
  Context &c = ctx();
  static bool attr = false;
  if (!attr) {
    RLH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&block_update_mfma_kernel<T, R, false, KS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RLH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&block_update_mfma_kernel<T, R, true, KS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  int64_t per_cu = lds > 0 ? (int64_t)(160 * 1024 / lds) : 4;
  const int cap = env_flag("RLH_UPDATE_MFMA_WG", 2);      // resident workgroups per CU (tunable)
  if (per_cu > cap) per_cu = cap;
  if (per_cu < 1) per_cu = 1;
  int64_t nb = (int64_t)c.num_cu * per_cu;
  const int64_t need = ((n + 15) / 16 + 3) / 4;
  if (nb > need) nb = need;
  if (nb < 1) nb = 1;
  if (beta)
    hipLaunchKernelGGL((block_update_mfma_kernel<T, R, true, KS>), dim3((unsigned)nb), dim3(256), lds, c.stream, X, ldx, k, ks1, X2,
                       ldx2, k2, ks2, Out, ldo, Out2, ldo2, msplit, Qd, ldq, n, m);
  else
    hipLaunchKernelGGL((block_update_mfma_kernel<T, R, false, KS>), dim3((unsigned)nb), dim3(256), lds, c.stream, X, ldx, k, ks1, X2,
                       ldx2, k2, ks2, Out, ldo, Out2, ldo2, msplit, Qd, ldq, n, m);
  RLH_HIP(hipGetLastError());
  return 0;
}

template <typename T>
static int launch_update_mfma(const T *X, int64_t ldx, T *Out, int64_t ldo, const T *Qd, int ldq, int64_t n, int k, int m,
                              int beta, const T *X2, int64_t ldx2, int k2, int kpad, T *Out2, int64_t ldo2, int msplit) {
  const int ks1 = X2 && k2 > 0 ? kpad / 4 : (k + 3) / 4, ks2 = X2 && k2 > 0 ? (k2 + 3) / 4 : 0;
  const int ntile = (m + 15) / 16;
  const size_t lds = (size_t)ntile * (ks1 + ks2) * 64 * sizeof(T);
  if (ks1 + ks2 <= 8)
    return launch_update_mfma_ks<T, 8>(X, ldx, Out, ldo, Qd, ldq, n, k, m, beta, X2, ldx2, k2, ks1, ks2, Out2, ldo2, msplit, lds);
  if (ks1 + ks2 <= 16)
    return launch_update_mfma_ks<T, 16>(X, ldx, Out, ldo, Qd, ldq, n, k, m, beta, X2, ldx2, k2, ks1, ks2, Out2, ldo2, msplit, lds);
  return launch_update_mfma_ks<T, 32>(X, ldx, Out, ldo, Qd, ldq, n, k, m, beta, X2, ldx2, k2, ks1, ks2, Out2, ldo2, msplit, lds);
}

// ---------------------------------------------------------------- real block update on the matrix cores
// Out = beta Out + X Q (+ X2 Q2) for float / double blocks, rows of X on the MFMA's N side: D^T (16 result columns x 16
// rows) = Q^T (16 x 4) X^T (4 x 16), so lane (kk, rl) loads ITS 16 bytes -- rows V rl .. V rl + V - 1, V = 16 / sizeof(R),
// of column 4 s + kk -- straight from global memory as the B operand of k-step s, and stores 16 bytes of rows
// V rl .. of result column 16 t + out_row(lane, g): one wave instruction moves 4 columns x 256 contiguous bytes,
// which streams as fast as 1-KB runs of one column (5.8 TB/s for the bare 32 + 32 column copy of this shape,
// tools/stream_probe.hip section 14).  A wave walks TPW consecutive tiles of 16 V rows with the X fragments of the
// NEXT tile in flight during the MFMAs and stores of the current one (two register sets); the coefficients sit in
// the LDS in fragment order [column tile][k-step][lane], re-laid-out once per workgroup (4 waves x 8 tiles, so that
// this costs a few percent of the traffic), one ds_read per V MFMAs.  No run-time switch touches a load or store.
// What it replaces: the SGPR-fed VALU kernel above issues 2 k m v_fma per row; for results of more than 32 columns
// its accumulators no longer fit beside 16-byte accesses, it falls back to 8 bytes per lane and is bound by its
// arithmetic (2.8 ms for the driver's 64 -> 64 column update at n = 215^3, where the stream needs 1.75 ms).
// (A first matrix-core version -- one tile per wave, coefficients re-staged by every 128 rows -- measured
// 1.16-1.30 ms against 1.02 ms for m = k = 32 and was dropped; this one amortises both.)
constexpr int kUpdTilesPerWave = 8;
template <typename R, int KS, bool BETA, bool NT>
__global__ __launch_bounds__(256) void block_update_stream_kernel(const R *__restrict__ X, int64_t ldx, int k1, int ks1,
                                                                  const R *__restrict__ X2, int64_t ldx2, int k2, int ks2,
                                                                  R *__restrict__ Out, int64_t ldo, R *__restrict__ Out2,
                                                                  int64_t ldo2, int msplit, const R *__restrict__ Q, int ldq,
                                                                  int64_t n, int m) {
  using M = Mfma16x4<R>;
  using acc_t = typename M::acc_t;
  constexpr int V = 16 / (int)sizeof(R);
  constexpr int TR = 16 * V;                               // rows of one tile
  typedef R vec_t __attribute__((ext_vector_type(V)));
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  R *qf = reinterpret_cast<R *>(lds_raw);                 // [column tile][k-step][lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int nks = ks1 + ks2, ntile = (m + 15) / 16;
  for (int idx = tid; idx < ntile * nks * 64; idx += 256) {
    const int l = idx & 63, s = (idx >> 6) % nks, t = (idx >> 6) / nks;
    const int col = 16 * t + (l & 15);
    qf[idx] = col < m ? Q[(int64_t)(4 * s + (l >> 4)) * ldq + col] : (R)0;   // (rows past k1 / k2 are zero padding)
  }
  __syncthreads();
  const int kk = lane >> 4, rl = lane & 15;
  const int64_t ntiles = (n + TR - 1) / TR;
  // the four waves of a workgroup take CONSECUTIVE tiles in every step (wave w: tiles base + 4 i + w), so a step of
  // the workgroup touches 4 x 256 contiguous bytes of every column at about the same time
  const int64_t tile0 = (int64_t)blockIdx.x * 4 * kUpdTilesPerWave + (tid >> 6);
  if (tile0 >= ntiles) return;
  const int64_t wg_end = ((int64_t)blockIdx.x + 1) * 4 * kUpdTilesPerWave;
  const int64_t tile1 = wg_end < ntiles ? wg_end : ntiles;
  // this lane's column of every k-step (columns inside the zero padding of Q repeat the last one)
  const R *xcol[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const bool first = s < ks1;
    int c = first ? 4 * s + kk : 4 * (s - ks1) + kk;
    const int kmax = first ? k1 : k2;
    c = c < kmax ? c : kmax - 1;
    c = c < 0 ? 0 : c;
    xcol[s] = (first ? X + (int64_t)c * ldx : X2 + (int64_t)c * ldx2) + V * rl;
  }
  vec_t xa[KS], xb[KS];
  auto load_x = [&](int64_t tile, vec_t (&xf)[KS]) {
    const int64_t r0 = tile * TR;
    if (r0 + TR <= n) {                                   // wave-uniform
      // (unconditional: k-steps past the last re-read a valid column -- an L1 hit -- because a branch around a
      // load costs the counted vmcnt waits of the whole pipeline)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if constexpr (NT) xf[s] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(xcol[s] + r0));
        else xf[s] = *reinterpret_cast<const vec_t *>(xcol[s] + r0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s)
        if (s < nks) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const int64_t r = r0 + V * rl + v < n ? r0 + v : n - 1 - V * rl;   // rows past the end repeat the last (never stored)
            xf[s][v] = xcol[s][r];
          }
        }
    }
  };
  auto process = [&](int64_t tile, const vec_t (&xf)[KS]) {
    const int64_t r0 = tile * TR + V * rl;
    const bool full = (tile + 1) * TR <= n;                // wave-uniform
    for (int t = 0; t < ntile; ++t) {
      acc_t d[V];
#pragma unroll
      for (int v = 0; v < V; ++v) d[v] = acc_t{(R)0, (R)0, (R)0, (R)0};
      if (BETA) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j = 16 * t + M::out_row(lane, g);
          if (j < m) {
            const R *o = (j < msplit ? Out + (int64_t)j * ldo : Out2 + (int64_t)(j - msplit) * ldo2) + r0;
            if (full) {
              vec_t ov;
              if constexpr (NT) ov = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(o));
              else ov = *reinterpret_cast<const vec_t *>(o);
#pragma unroll
              for (int v = 0; v < V; ++v) d[v][g] = ov[v];
            } else {
#pragma unroll
              for (int v = 0; v < V; ++v)
                if (r0 + v < n) d[v][g] = o[v];
            }
          }
        }
      }
      const R *qt = qf + (int64_t)t * nks * 64 + lane;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s < nks) {
          const R q = qt[s * 64];
#pragma unroll
          for (int v = 0; v < V; ++v) d[v] = M::run(q, xf[s][v], d[v]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = 16 * t + M::out_row(lane, g);
        if (j < m) {
          R *o = (j < msplit ? Out + (int64_t)j * ldo : Out2 + (int64_t)(j - msplit) * ldo2) + r0;
          if (full) {
            vec_t ov;
#pragma unroll
            for (int v = 0; v < V; ++v) ov[v] = d[v][g];
            if constexpr (NT) __builtin_nontemporal_store(ov, reinterpret_cast<vec_t *>(o));
            else *reinterpret_cast<vec_t *>(o) = ov;
          } else {
#pragma unroll
            for (int v = 0; v < V; ++v)
              if (r0 + v < n) o[v] = d[v][g];
          }
        }
      }
    }
  };
  // two register sets: the loads of tile i + 1 are outstanding while tile i is multiplied and stored
  load_x(tile0, xa);
  int64_t tile = tile0;
  for (; tile + 4 < tile1; tile += 8) {
    load_x(tile + 4, xb);
    process(tile, xa);
    if (tile + 8 < tile1) load_x(tile + 8, xa);
    process(tile + 4, xb);
  }
  if (tile < tile1) process(tile, xa);
}

template <typename R, int KS>
static int launch_update_stream_ks(const R *X, int64_t ldx, R *Out, int64_t ldo, const R *Qd, int ldq, int64_t n, int k,
                                   int m, int beta, const R *X2, int64_t ldx2, int k2, int ks1, int ks2, R *Out2,
                                   int64_t ldo2, int msplit, size_t lds) {
  Context &c = ctx();
  const int nt = update_nt((int64_t)n * (k + k2 + (beta ? 2 : 1) * m) * (int64_t)sizeof(R));
  constexpr int TR = 16 * (16 / (int)sizeof(R));
  const int64_t rows_per_wg = (int64_t)4 * kUpdTilesPerWave * TR;
  const int64_t nb = (n + rows_per_wg - 1) / rows_per_wg;
  RLH_REQUIRE(nb >= 1 && nb <= 0x7fffffff, "rlh_block_update: %lld rows exceed the grid", (long long)n);
  if (X2 == nullptr || k2 <= 0) { X2 = X; ldx2 = ldx; }
#define RLH_UPDS(BETA_, NT_)                                                                                              \
  hipLaunchKernelGGL((block_update_stream_kernel<R, KS, BETA_, NT_>), dim3((unsigned)nb), dim3(256), lds, c.stream, X, ldx, \
                     k, ks1, X2, ldx2, k2, ks2, Out, ldo, Out2, ldo2, msplit, Qd, ldq, n, m)
  if (nt) { if (beta) RLH_UPDS(true, true); else RLH_UPDS(false, true); }
  else { if (beta) RLH_UPDS(true, false); else RLH_UPDS(false, false); }
#undef RLH_UPDS
  RLH_HIP(hipGetLastError());
  return 0;
}

template <typename R>
static int launch_update_stream(const R *X, int64_t ldx, R *Out, int64_t ldo, const R *Qd, int ldq, int64_t n, int k, int m,
                                int beta, const R *X2, int64_t ldx2, int k2, int kpad, R *Out2, int64_t ldo2, int msplit) {
  const bool two = X2 && k2 > 0;
  const int ks1 = two ? kpad / 4 : (k + 3) / 4, ks2 = two ? (k2 + 3) / 4 : 0;
  const size_t lds = (size_t)((m + 15) / 16) * (ks1 + ks2) * 64 * sizeof(R);
  if (!two) k2 = 0;
  if (ks1 + ks2 <= 8)
    return launch_update_stream_ks<R, 8>(X, ldx, Out, ldo, Qd, ldq, n, k, m, beta, X2, ldx2, k2, ks1, ks2, Out2, ldo2, msplit, lds);
  if (ks1 + ks2 <= 16)
    return launch_update_stream_ks<R, 16>(X, ldx, Out, ldo, Qd, ldq, n, k, m, beta, X2, ldx2, k2, ks1, ks2, Out2, ldo2, msplit, lds);
  return launch_update_stream_ks<R, 32>(X, ldx, Out, ldo, Qd, ldq, n, k, m, beta, X2, ldx2, k2, ks1, ks2, Out2, ldo2, msplit, lds);
}

template <typename T> struct IsComplex { static constexpr bool value = false; };
template <> struct IsComplex<c32> { static constexpr bool value = true; };
template <> struct IsComplex<c64> { static constexpr bool value = true; };

// can the streaming matrix-core kernel take this real update?  (RLH_UPDATE_STREAM=0: the VALU kernel; tunable)
template <typename T>
static bool update_stream_ok(const T *X, int64_t ldx, const T *Out, int64_t ldo, const T *Out2, int64_t ldo2, const T *X2,
                             int64_t ldx2, int k, int k2, int kpad, int m) {
  const int ks = (X2 && k2 > 0 ? kpad / 4 + (k2 + 3) / 4 : (k + 3) / 4);
  const size_t lds = (size_t)((m + 15) / 16) * ks * 64 * sizeof(T);
  return env_flag("RLH_UPDATE_STREAM", 1) && k + k2 >= 8 && m >= 8 && ks <= kMfmaUpdKS && lds <= 64 * 1024 &&
         aligned16(X, ldx, sizeof(T)) && aligned16(Out, ldo, sizeof(T)) && aligned16(Out2, ldo2, sizeof(T)) &&
         (!X2 || k2 == 0 || aligned16(X2, ldx2, sizeof(T)));
}

template <typename T, int JT>
static int launch_update(const T *X, int64_t ldx, T *Out, int64_t ldo, const T *Qd, int ldq, int64_t n, int k, int m,
                         int beta, const T *X2 = nullptr, int64_t ldx2 = 0, int k2 = 0, int kpad = 0,
                         T *Out2 = nullptr, int64_t ldo2 = 0, int msplit = -1) {
  if (msplit < 0 || !Out2) { msplit = m; Out2 = Out; ldo2 = ldo; }
  if constexpr (IsComplex<T>::value) {
    // the matrix cores for complex blocks of at least 16 x 16 coefficients whose k-steps fit the register
    // set and whose coefficients fit the LDS (RLH_UPDATE_MFMA=0: VALU kernel, tunable)
    const int ks = (X2 && k2 > 0 ? kpad / 4 + (k2 + 3) / 4 : (k + 3) / 4);
    const size_t lds = (size_t)((m + 15) / 16) * ks * 64 * sizeof(T);
    if (env_flag("RLH_UPDATE_MFMA", 1) && k + k2 >= 16 && m >= 16 && ks <= kMfmaUpdKS && lds <= 160 * 1024 &&
        aligned16(X, ldx, sizeof(T)) && (!X2 || k2 == 0 || aligned16(X2, ldx2, sizeof(T))))
      return launch_update_mfma<T>(X, ldx, Out, ldo, Qd, ldq, n, k, m, beta, X2, ldx2, k2, kpad, Out2, ldo2, msplit);
  }
  if constexpr (!IsComplex<T>::value) {
    // the matrix cores for real blocks: results of at least 8 columns from at least 8, coefficients within 64 KB of
    // LDS, k-steps within the register sets
    if (update_stream_ok<T>(X, ldx, Out, ldo, Out2, ldo2, X2, ldx2, k, k2, kpad, m))
      return launch_update_stream<T>(X, ldx, Out, ldo, Qd, ldq, n, k, m, beta, X2, ldx2, k2, kpad, Out2, ldo2, msplit);
  }
  // 16-byte row groups where RV * JT accumulators of T fit (<= 128 registers) and every block is
  // 16-byte aligned (RLH_UPDATE_RV=0: one row per lane, tunable; 2 x 64 fp64 accumulators per lane spill: 6.2 ms
  // against 2.8 ms with one row per lane for the 64-column panels)
  constexpr int RVMAX = 16 / (int)sizeof(T);
  if constexpr (RVMAX > 1 && RVMAX * JT * sizeof(T) <= 512) {
    static const int rv = env_flag("RLH_UPDATE_RV", 1);
    if (rv && aligned16(X, ldx, sizeof(T)) && aligned16(Out, ldo, sizeof(T)) && aligned16(Out2, ldo2, sizeof(T)) &&
        (!X2 || k2 == 0 || aligned16(X2, ldx2, sizeof(T))))
      return launch_update_rv<T, JT, RVMAX>(X, ldx, Out, ldo, Qd, ldq, n, k, m, beta, X2, ldx2, k2, kpad, Out2, ldo2, msplit);
  }
  return launch_update_rv<T, JT, 1>(X, ldx, Out, ldo, Qd, ldq, n, k, m, beta, X2, ldx2, k2, kpad, Out2, ldo2, msplit);
}

template <int DT>
static int block_update_impl(int64_t n, int64_t k, const void *X_, int64_t ldx, int64_t m, void *Out_, int64_t ldo,
                             const void *q_, int64_t q_rs, int64_t q_cs, const double *alpha, int beta) {
  using T = typename DType<DT>::T;
  constexpr int JTMAX = DType<DT>::cplx ? 16 : 32;
  const T *X = (const T *)X_;
  T *Out = (T *)Out_;
  const T *q = (const T *)q_;
  // (fp64 results of more than 32 columns: panels of 64 with one row per lane -- two panels of 32 read every source
  // twice: 3.0 -> 2.0 ms for m = k = 64 at n = 215^3)
  const int JT = (m <= 8) ? 8 : ((m <= 16 || JTMAX == 16) ? 16 : ((DT == RLH_D && m > 32) ? 64 : 32));
  const int64_t mpad = (m + JT - 1) / JT * JT;
  // rows of Q per launch so that the padded coefficient block fits one ring slot
  int64_t kmax = (int64_t)(kRingSlotBytes / (mpad * sizeof(T))) / kUnrollK * kUnrollK;
  RLH_REQUIRE(kmax >= kUnrollK, "rlh_block_update: %lld output vectors exceed the staging slot", (long long)m);
  if (kmax > 1024) kmax = 1024;
  // (Folding the add into the product -- Out += X Q as [X | Out] [Q; I], so that the result block rides the fragment
  // pipeline of the sources -- was measured and dropped: 1.446 vs 1.443 ms.)
  for (int64_t k0 = 0; k0 < k || k0 == 0; k0 += kmax) {
    const int64_t kk = (k - k0) < kmax ? (k - k0) : kmax;
    const int64_t kpad = (kk + kUnrollK - 1) / kUnrollK * kUnrollK;
    int slot; void *h, *d;
    if (int rc = ring_acquire((size_t)(kpad * mpad) * sizeof(T), &slot, &h, &d)) return rc;
    T *qh = (T *)h;
    memset(qh, 0, (size_t)(kpad * mpad) * sizeof(T));
    for (int64_t i = 0; i < kk; ++i)
      for (int64_t j = 0; j < m; ++j) qh[i * mpad + j] = HostScalar<T>::mul(alpha, q[(k0 + i) * q_rs + j * q_cs]);
    if (int rc = ring_commit(slot, (size_t)(kpad * mpad) * sizeof(T))) return rc;
    const int b = (k0 == 0) ? beta : 1;
    int rc;
    if (kk == 0) {       // k == 0: Out = beta * Out
      if (!b) {
        for (int64_t j = 0; j < m; ++j)
          RLH_HIP(hipMemsetAsync(Out + j * ldo, 0, (size_t)n * sizeof(T), ctx().stream));
      }
      rc = 0;
    } else if (JT == 8) {
      rc = launch_update<T, 8>(X + k0 * ldx, ldx, Out, ldo, (const T *)d, (int)mpad, n, (int)kk, (int)m, b);
    } else if (JT == 16) {
      rc = launch_update<T, 16>(X + k0 * ldx, ldx, Out, ldo, (const T *)d, (int)mpad, n, (int)kk, (int)m, b);
    } else if (JT == 64) {
      if constexpr (DT == RLH_D)
        rc = launch_update<T, 64>(X + k0 * ldx, ldx, Out, ldo, (const T *)d, (int)mpad, n, (int)kk, (int)m, b);
      else
        rc = 1;
    } else {
      rc = launch_update<T, (JTMAX == 32 ? 32 : 16)>(X + k0 * ldx, ldx, Out, ldo, (const T *)d, (int)mpad, n, (int)kk,
                                                     (int)m, b);
    }
    if (rc) return rc;
    if (int rc2 = ring_release(slot)) return rc2;
    if (k == 0) break;
  }
  return 0;
}

template <int DT>
static int block_update2_impl(int64_t n, int64_t k1, const void *X1_, int64_t ldx1, const void *q1_, int64_t q1_rs,
                              int64_t q1_cs, int64_t k2, const void *X2_, int64_t ldx2, const void *q2_,
                              int64_t q2_rs, int64_t q2_cs, int64_t m, void *Out_, int64_t ldo, const double *alpha,
                              int beta, void *OutB_ = nullptr, int64_t ldob = 0, int64_t ma = -1) {
  using T = typename DType<DT>::T;
  constexpr int JTMAX = DType<DT>::cplx ? 16 : 32;
  const T *q1 = (const T *)q1_, *q2 = (const T *)q2_;
  const int JT = (m <= 8) ? 8 : ((m <= 16 || JTMAX == 16) ? 16 : ((DT == RLH_D && m > 32) ? 64 : 32));
  const int64_t mpad = (m + JT - 1) / JT * JT;
  const int64_t kp1 = (k1 + kUnrollK - 1) / kUnrollK * kUnrollK, kp2 = (k2 + kUnrollK - 1) / kUnrollK * kUnrollK;
  const size_t bytes = (size_t)((kp1 + kp2) * mpad) * sizeof(T);
  RLH_REQUIRE(bytes <= kRingSlotBytes, "rlh_block_update2: coefficient blocks of %zu bytes exceed the staging slot",
              bytes);
  int slot; void *h, *d;
  if (int rc = ring_acquire(bytes, &slot, &h, &d)) return rc;
  T *qh = (T *)h;
  memset(qh, 0, bytes);
  for (int64_t i = 0; i < k1; ++i)
    for (int64_t j = 0; j < m; ++j) qh[i * mpad + j] = HostScalar<T>::mul(alpha, q1[i * q1_rs + j * q1_cs]);
  for (int64_t i = 0; i < k2; ++i)
    for (int64_t j = 0; j < m; ++j) qh[(kp1 + i) * mpad + j] = HostScalar<T>::mul(alpha, q2[i * q2_rs + j * q2_cs]);
  if (int rc = ring_commit(slot, bytes)) return rc;
  const T *X1 = (const T *)X1_, *X2 = (const T *)X2_;
  T *Out = (T *)Out_;
  int rc;
  T *OutB = (T *)OutB_;
  if (JT == 8)
    rc = launch_update<T, 8>(X1, ldx1, Out, ldo, (const T *)d, (int)mpad, n, (int)k1, (int)m, beta, X2, ldx2, (int)k2,
                             (int)kp1, OutB, ldob, (int)ma);
  else if (JT == 16)
    rc = launch_update<T, 16>(X1, ldx1, Out, ldo, (const T *)d, (int)mpad, n, (int)k1, (int)m, beta, X2, ldx2,
                              (int)k2, (int)kp1, OutB, ldob, (int)ma);
  else if (JT == 64) {
    if constexpr (DT == RLH_D)
      rc = launch_update<T, 64>(X1, ldx1, Out, ldo, (const T *)d, (int)mpad, n, (int)k1, (int)m, beta, X2, ldx2,
                                (int)k2, (int)kp1, OutB, ldob, (int)ma);
    else
      rc = 1;
  } else
    rc = launch_update<T, (JTMAX == 32 ? 32 : 16)>(X1, ldx1, Out, ldo, (const T *)d, (int)mpad, n, (int)k1, (int)m,
                                                   beta, X2, ldx2, (int)k2, (int)kp1, OutB, ldob, (int)ma);
  if (rc) return rc;
  return ring_release(slot);
}

// Out[:, i] = a[i] * A[:, i] + b[i] * B[:, i]   (residual W = AX - X diag(lmd) in one pass)
template <typename T, bool ALIGNED, bool NT>
__global__ __launch_bounds__(256) void lincomb_cols_kernel(const T *A, int64_t lda,
                                                           const T *B, int64_t ldb, T *Out,
                                                           int64_t ldo, const T *__restrict__ coef, int m, int64_t n) {
  constexpr int VEC = ALIGNED ? 16 / (int)sizeof(T) : 1;
  struct alignas(ALIGNED ? 16 : alignof(T)) V { T v[VEC]; };
  const int col = blockIdx.y;
  const T ca = coef[col], cb = coef[m + col];
  const T *a = A + (int64_t)col * lda;
  const T *b = B + (int64_t)col * ldb;
  T *o = Out + (int64_t)col * ldo;
  const int64_t stride = (int64_t)gridDim.x * 256 * VEC;
  for (int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; r < n; r += stride) {
    if (r + VEC <= n) {
      const V av = ldv<ALIGNED, NT>(reinterpret_cast<const V *>(a + r));
      const V bv = ldv<ALIGNED, NT>(reinterpret_cast<const V *>(b + r));
      V ov;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        T t = mul_of(ca, av.v[e]);
        fma_acc(t, cb, bv.v[e]);
        ov.v[e] = t;
      }
      stv<ALIGNED, NT>(reinterpret_cast<V *>(o + r), ov);
    } else {
      for (int e = 0; e < VEC && r + e < n; ++e) {
        T t = mul_of(ca, a[r + e]);
        fma_acc(t, cb, b[r + e]);
        o[r + e] = t;
      }
    }
  }
}

// ---------------------------------------------------------------- column-wise elementwise kernels
// grid = (row blocks, columns); VEC elements (16 bytes) per lane when aligned.

template <typename T, bool ALIGNED, bool NT>
__global__ __launch_bounds__(256) void axpy_cols_kernel(const T *__restrict__ X, int64_t ldx, T *__restrict__ Y,
                                                        int64_t ldy, const T *__restrict__ s, int64_t n) {
  constexpr int VEC = ALIGNED ? 16 / (int)sizeof(T) : 1;
  struct alignas(ALIGNED ? 16 : alignof(T)) V { T v[VEC]; };
  const int col = blockIdx.y;
  const T a = s[col];
  const T *x = X + (int64_t)col * ldx;
  T *y = Y + (int64_t)col * ldy;
  const int64_t stride = (int64_t)gridDim.x * 256 * VEC;
  for (int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; r < n; r += stride) {
    if (r + VEC <= n) {
      const V xv = ldv<ALIGNED, NT>(reinterpret_cast<const V *>(x + r));
      V yv = ldv<ALIGNED, NT>(reinterpret_cast<const V *>(y + r));
#pragma unroll
      for (int e = 0; e < VEC; ++e) fma_acc(yv.v[e], a, xv.v[e]);
      stv<ALIGNED, NT>(reinterpret_cast<V *>(y + r), yv);
    } else {
      for (int e = 0; e < VEC && r + e < n; ++e) {
        T t = y[r + e];
        fma_acc(t, a, x[r + e]);
        y[r + e] = t;
      }
    }
  }
}

template <typename T, bool ALIGNED, bool NT>
__global__ __launch_bounds__(256) void scale_cols_kernel(T *__restrict__ X, int64_t ldx, const T *__restrict__ s,
                                                         int64_t n) {
  constexpr int VEC = ALIGNED ? 16 / (int)sizeof(T) : 1;
  struct alignas(ALIGNED ? 16 : alignof(T)) V { T v[VEC]; };
  const int col = blockIdx.y;
  const T a = s[col];
  T *x = X + (int64_t)col * ldx;
  const int64_t stride = (int64_t)gridDim.x * 256 * VEC;
  for (int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; r < n; r += stride) {
    if (r + VEC <= n) {
      V xv = ldv<ALIGNED, NT>(reinterpret_cast<const V *>(x + r));
#pragma unroll
      for (int e = 0; e < VEC; ++e) xv.v[e] = mul_of(xv.v[e], a);
      stv<ALIGNED, NT>(reinterpret_cast<V *>(x + r), xv);
    } else {
      for (int e = 0; e < VEC && r + e < n; ++e) x[r + e] = mul_of(x[r + e], a);
    }
  }
}

// Copies columns as raw words of W bytes (16 when everything is 16-byte aligned, else the
// element's real size); ind == nullptr copies column j to column j.
// `tail4` 4-byte words follow the n_w whole words of a column (a column of an odd number of
// 8-byte elements still moves as 16-byte words plus two dwords).  16-byte words go through
// non-temporal loads and stores (measured +3-6 % on a 2.5 GB block: 5.15-5.25 -> 5.30-5.55 TB/s; the
// same hint on the block-update stores changed nothing).
template <typename W>
__global__ __launch_bounds__(256) void copy_cols_kernel(const W *__restrict__ X, int64_t ldx_w, W *__restrict__ Y,
                                                        int64_t ldy_w, const int64_t *__restrict__ ind,
                                                        int64_t n_w, int tail4, int nt) {
  const int col = blockIdx.y;
  const int64_t src = ind ? ind[col] : col;
  const W *x = X + src * ldx_w;
  W *y = Y + (int64_t)col * ldy_w;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if constexpr (sizeof(W) == 16) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *xv = reinterpret_cast<const u32x4 *>(x);
    u32x4 *yv = reinterpret_cast<u32x4 *>(y);
    if (nt) {
      for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_w; r += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(xv + r), yv + r);
    } else {
      for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_w; r += stride) yv[r] = xv[r];
    }
  } else {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_w; r += stride) y[r] = x[r];
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < tail4)
    reinterpret_cast<uint32_t *>(y + n_w)[threadIdx.x] = reinterpret_cast<const uint32_t *>(x + n_w)[threadIdx.x];
}

template <typename R>
__global__ __launch_bounds__(256) void conj_kernel(R *X, int64_t ldx_r, int64_t n) {
  R *x = X + (int64_t)blockIdx.y * ldx_r;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) x[2 * r + 1] = -x[2 * r + 1];
}

// precision conversion of a block (mixed-precision preconditioning): Y = (TD) X
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_kernel(const TS *__restrict__ X, int64_t ldx, TD *__restrict__ Y,
                                                      int64_t ldy, int64_t n) {
  const TS *x = X + (int64_t)blockIdx.y * ldx;
  TD *y = Y + (int64_t)blockIdx.y * ldy;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) y[r] = (TD)x[r];
}

// Counter-based uniform numbers in [-1, 1) (device side of Vectors.fill_random for large blocks):
// element (row, col) is a pure function of (seed, col, row) -- splitmix64 of the combined counter --
// so a row shard generates exactly its rows of the global block whatever the number of ranks.
// Real part only for the complex types, as the reference's numpy.random.rand fill.
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <typename R, int COMPONENTS>
__global__ __launch_bounds__(256) void fill_random_kernel(R *X, int64_t ldx_r, int64_t n, uint64_t seed, int64_t row0,
                                                          int64_t col0) {
  R *x = X + (int64_t)blockIdx.y * ldx_r;
  const uint64_t colkey = seed + (uint64_t)(col0 + blockIdx.y) * 0x632BE59BD9B4E019ull;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
    const uint64_t z = splitmix64(colkey + ((uint64_t)(row0 + r) + 1) * 0x9E3779B97F4A7C15ull);
    R u;
    if constexpr (sizeof(R) == 8) u = (R)(z >> 11) * (R)(1.0 / 9007199254740992.0);   // 53 bits
    else u = (R)(z >> 40) * (R)(1.0 / 16777216.0);                                     // 24 bits
    x[COMPONENTS * r] = (R)2 * u - (R)1;
    if constexpr (COMPONENTS == 2) x[2 * r + 1] = (R)0;
  }
}

// bfloat16 storage of a real block (device polynomial preconditioner): Y16 = bf16(scale * X),
// round to nearest even, and back.
template <typename TS>
__global__ __launch_bounds__(256) void bf16_pack_kernel(const TS *__restrict__ X, int64_t ldx, float scale,
                                                        unsigned short *__restrict__ Y, int64_t ldy, int64_t n) {
  const TS *x = X + (int64_t)blockIdx.y * ldx;
  unsigned short *y = Y + (int64_t)blockIdx.y * ldy;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
    unsigned u = __float_as_uint(scale * (float)x[r]);
    u += 0x7FFFu + ((u >> 16) & 1u);
    y[r] = (unsigned short)(u >> 16);
  }
}

template <typename TD>
__global__ __launch_bounds__(256) void bf16_unpack_kernel(const unsigned short *__restrict__ X, int64_t ldx,
                                                          TD *__restrict__ Y, int64_t ldy, int64_t n) {
  const unsigned short *x = X + (int64_t)blockIdx.y * ldx;
  TD *y = Y + (int64_t)blockIdx.y * ldy;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride)
    y[r] = (TD)__uint_as_float((unsigned)x[r] << 16);
}

// One item per lane and NO loop for the column-wise kernels: workgroups are dispatched in index order, so the
// set of resident ones is a compact window that sweeps each column front to back -- the access order HBM likes
// best.  Measured on a 2.5 GB fp64 block (tools/stream_probe.hip): copy 6.0 TB/s (6.6 with the non-temporal
// hint) against 5.4-5.6 for a capped grid whose workgroups stride through the column, 4.3-5.0 for a
// grid-stride loop over the whole block.  RLH_ROW_BLOCKS_PER_CU > 0 restores a capped grid (experiments).
static inline unsigned row_blocks(int64_t items, int64_t m) {
  int64_t nb = (items + 255) / 256;
  const int per_cu = env_flag("RLH_ROW_BLOCKS_PER_CU", 0);
  if (per_cu > 0) {
    const int64_t cap = ((int64_t)ctx().num_cu * per_cu + m - 1) / m;
    if (nb > cap) nb = cap;
  }
  if (nb > 0x7fffffff) nb = 0x7fffffff;
  if (nb < 1) nb = 1;
  return (unsigned)nb;
}

// Non-temporal loads / stores for blocks that cannot stay in the 256 MB Infinity Cache until their next use
// (RLH_STREAM_NT = 0 / 1 forces): +5-8 % on every streaming kernel at the roofline block size, while small
// blocks keep the default policy and are found cached by the next kernel.
static inline int stream_nt(int64_t bytes) {
  const char *e = getenv("RLH_STREAM_NT");
  if (e && *e) return atoi(e) != 0;
  return bytes > (int64_t)192 << 20;
}

// stages m per-column coefficients of type T into the ring
template <typename T, typename F>
static int stage_coeffs(int64_t m, F fill, int *slot, const T **d) {
  void *h, *dv;
  if (int rc = ring_acquire((size_t)m * sizeof(T), slot, &h, &dv)) return rc;
  T *p = (T *)h;
  for (int64_t i = 0; i < m; ++i) p[i] = fill(i);
  if (int rc = ring_commit(*slot, (size_t)m * sizeof(T))) return rc;
  *d = (const T *)dv;
  return 0;
}

template <int DT>
static int axpy_cols_impl(int64_t n, int64_t m, const void *s_, bool broadcast, const void *X, int64_t ldx, void *Y,
                          int64_t ldy) {
  using T = typename DType<DT>::T;
  const T *s = (const T *)s_;
  int slot; const T *sd;
  if (int rc = stage_coeffs<T>(m, [&](int64_t i) { return broadcast ? s[0] : s[i]; }, &slot, &sd)) return rc;
  const bool al = aligned16(X, ldx, sizeof(T)) && aligned16(Y, ldy, sizeof(T));
  const int vec = al ? 16 / (int)sizeof(T) : 1;
  dim3 grid(row_blocks((n + vec - 1) / vec, m), (unsigned)m);
  if (al && stream_nt(3 * n * m * (int64_t)sizeof(T)))
    hipLaunchKernelGGL((axpy_cols_kernel<T, true, true>), grid, dim3(256), 0, ctx().stream, (const T *)X, ldx, (T *)Y,
                       ldy, sd, n);
  else if (al)
    hipLaunchKernelGGL((axpy_cols_kernel<T, true, false>), grid, dim3(256), 0, ctx().stream, (const T *)X, ldx, (T *)Y,
                       ldy, sd, n);
  else
    hipLaunchKernelGGL((axpy_cols_kernel<T, false, false>), grid, dim3(256), 0, ctx().stream, (const T *)X, ldx, (T *)Y,
                       ldy, sd, n);
  RLH_HIP(hipGetLastError());
  return ring_release(slot);
}

template <int DT>
static int lincomb_cols_impl(int64_t n, int64_t m, const void *a_, const void *A, int64_t lda, const void *b_,
                             const void *B, int64_t ldb, void *Out, int64_t ldo) {
  using T = typename DType<DT>::T;
  const T *a = (const T *)a_, *b = (const T *)b_;
  int slot; const T *cd;
  if (int rc = stage_coeffs<T>(2 * m, [&](int64_t i) { return i < m ? a[i] : b[i - m]; }, &slot, &cd)) return rc;
  const bool al = aligned16(A, lda, sizeof(T)) && aligned16(B, ldb, sizeof(T)) && aligned16(Out, ldo, sizeof(T));
  const int vec = al ? 16 / (int)sizeof(T) : 1;
  dim3 grid(row_blocks((n + vec - 1) / vec, m), (unsigned)m);
  if (al && stream_nt(3 * n * m * (int64_t)sizeof(T)))
    hipLaunchKernelGGL((lincomb_cols_kernel<T, true, true>), grid, dim3(256), 0, ctx().stream, (const T *)A, lda,
                       (const T *)B, ldb, (T *)Out, ldo, cd, (int)m, n);
  else if (al)
    hipLaunchKernelGGL((lincomb_cols_kernel<T, true, false>), grid, dim3(256), 0, ctx().stream, (const T *)A, lda,
                       (const T *)B, ldb, (T *)Out, ldo, cd, (int)m, n);
  else
    hipLaunchKernelGGL((lincomb_cols_kernel<T, false, false>), grid, dim3(256), 0, ctx().stream, (const T *)A, lda,
                       (const T *)B, ldb, (T *)Out, ldo, cd, (int)m, n);
  RLH_HIP(hipGetLastError());
  return ring_release(slot);
}

template <int DT>
static int scale_cols_impl(int64_t n, int64_t m, const double *s, int mode, void *X, int64_t ldx) {
  using T = typename DType<DT>::T;
  using R = typename DType<DT>::R;
  constexpr bool CPLX = DType<DT>::cplx;
  int slot; const T *sd;
  auto fill = [&](int64_t i) -> T {
    double re = CPLX ? s[2 * i] : s[i], im = CPLX ? s[2 * i + 1] : 0.0;
    if (mode == 0) {                       // divide, skipping zeros (dense_numpy.py:49-52)
      const double d = re * re + im * im;
      if (d == 0.0) { re = 1.0; im = 0.0; }
      else if (im == 0.0) { re = 1.0 / re; }
      else { const double r2 = re / d, i2 = -im / d; re = r2; im = i2; }
    }
    if constexpr (CPLX) return T{(R)re, (R)im};
    else return (T)re;
  };
  if (int rc = stage_coeffs<T>(m, fill, &slot, &sd)) return rc;
  const bool al = aligned16(X, ldx, sizeof(T));
  const int vec = al ? 16 / (int)sizeof(T) : 1;
  dim3 grid(row_blocks((n + vec - 1) / vec, m), (unsigned)m);
  if (al && stream_nt(2 * n * m * (int64_t)sizeof(T)))
    hipLaunchKernelGGL((scale_cols_kernel<T, true, true>), grid, dim3(256), 0, ctx().stream, (T *)X, ldx, sd, n);
  else if (al)
    hipLaunchKernelGGL((scale_cols_kernel<T, true, false>), grid, dim3(256), 0, ctx().stream, (T *)X, ldx, sd, n);
  else
    hipLaunchKernelGGL((scale_cols_kernel<T, false, false>), grid, dim3(256), 0, ctx().stream, (T *)X, ldx, sd, n);
  RLH_HIP(hipGetLastError());
  return ring_release(slot);
}

static int copy_cols_impl(int dtype, int64_t n, int64_t m, const int64_t *ind, const void *X, int64_t ldx, void *Y,
                          int64_t ldy) {
  const int64_t es = dtype_size(dtype);
  int slot = -1;
  const int64_t *indd = nullptr;
  if (ind) {
    if (int rc = stage_coeffs<int64_t>(m, [&](int64_t i) { return ind[i]; }, &slot, &indd)) return rc;
  }
  const bool al = aligned16(X, ldx, es) && aligned16(Y, ldy, es);
  if (al) {
    typedef struct alignas(16) { uint32_t w[4]; } W16;
    const int64_t nw = n * es / 16;
    dim3 grid(row_blocks(nw, m), (unsigned)m);
    hipLaunchKernelGGL((copy_cols_kernel<W16>), grid, dim3(256), 0, ctx().stream, (const W16 *)X, ldx * es / 16,
                       (W16 *)Y, ldy * es / 16, indd, nw, (int)((n * es % 16) / 4), stream_nt(2 * n * m * es));
  } else if (es % 8 == 0) {
    const int64_t nw = n * es / 8;
    dim3 grid(row_blocks(nw, m), (unsigned)m);
    hipLaunchKernelGGL((copy_cols_kernel<uint64_t>), grid, dim3(256), 0, ctx().stream, (const uint64_t *)X,
                       ldx * es / 8, (uint64_t *)Y, ldy * es / 8, indd, nw, 0, 0);
  } else {
    dim3 grid(row_blocks(n, m), (unsigned)m);
    hipLaunchKernelGGL((copy_cols_kernel<uint32_t>), grid, dim3(256), 0, ctx().stream, (const uint32_t *)X, ldx,
                       (uint32_t *)Y, ldy, indd, n, 0, 0);
  }
  RLH_HIP(hipGetLastError());
  if (slot >= 0) return ring_release(slot);
  return 0;
}

}  // namespace rlh

using namespace rlh;

#define RLH_DISPATCH(dt, fn, ...)                              \
  switch (dt) {                                                \
    case RLH_S: rc = fn<RLH_S>(__VA_ARGS__); break;            \
    case RLH_D: rc = fn<RLH_D>(__VA_ARGS__); break;            \
    case RLH_C: rc = fn<RLH_C>(__VA_ARGS__); break;            \
    case RLH_Z: rc = fn<RLH_Z>(__VA_ARGS__); break;            \
    default: rlh::set_error("unknown dtype %d", dt); rc = 1;   \
  }

static bool overlaps(const void *a, int64_t a_bytes, const void *b, int64_t b_bytes) {
  const char *pa = (const char *)a, *pb = (const char *)b;
  return pa < pb + b_bytes && pb < pa + a_bytes;
}

extern "C" {

int rlh_block_update(int dtype, int64_t n, int64_t k, const void *X, int64_t ldx, int64_t m, void *Out, int64_t ldo,
                     const void *q, int64_t q_rs, int64_t q_cs, const double *alpha, int beta) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_block_update: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && k >= 0 && m >= 0, "rlh_block_update: negative size");
  RLH_REQUIRE(beta == 0 || beta == 1, "rlh_block_update: beta must be 0 or 1");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(Out && alpha, "rlh_block_update: null pointer");
  RLH_REQUIRE(k == 0 || (X && q), "rlh_block_update: null pointer");
  RLH_REQUIRE(ldo >= n && (k == 0 || ldx >= n), "rlh_block_update: leading dimension smaller than n");
  const int64_t es = dtype_size(dtype);
  RLH_REQUIRE(k == 0 || !overlaps(X, ((k - 1) * ldx + n) * es, Out, ((m - 1) * ldo + n) * es),
              "rlh_block_update: output window overlaps the input window");
  int rc = 0;
  RLH_DISPATCH(dtype, block_update_impl, n, k, X, ldx, m, Out, ldo, q, q_rs, q_cs, alpha, beta)
  return rc;
}

int rlh_block_update2(int dtype, int64_t n, int64_t k1, const void *X1, int64_t ldx1, const void *q1, int64_t q1_rs,
                      int64_t q1_cs, int64_t k2, const void *X2, int64_t ldx2, const void *q2, int64_t q2_rs,
                      int64_t q2_cs, int64_t m, void *Out, int64_t ldo, const double *alpha, int beta) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_block_update2: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && k1 >= 1 && k2 >= 1 && m >= 0, "rlh_block_update2: bad size");
  RLH_REQUIRE(beta == 0 || beta == 1, "rlh_block_update2: beta must be 0 or 1");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(Out && alpha && X1 && X2 && q1 && q2, "rlh_block_update2: null pointer");
  RLH_REQUIRE(ldo >= n && ldx1 >= n && ldx2 >= n, "rlh_block_update2: leading dimension smaller than n");
  const int64_t es = dtype_size(dtype);
  RLH_REQUIRE(!overlaps(X1, ((k1 - 1) * ldx1 + n) * es, Out, ((m - 1) * ldo + n) * es) &&
                  !overlaps(X2, ((k2 - 1) * ldx2 + n) * es, Out, ((m - 1) * ldo + n) * es),
              "rlh_block_update2: output window overlaps an input window");
  int rc = 0;
  RLH_DISPATCH(dtype, block_update2_impl, n, k1, X1, ldx1, q1, q1_rs, q1_cs, k2, X2, ldx2, q2, q2_rs, q2_cs, m, Out,
               ldo, alpha, beta)
  return rc;
}

int rlh_block_update2x2(int dtype, int64_t n, int64_t k1, const void *X1, int64_t ldx1, const void *q1, int64_t q1_rs,
                        int64_t q1_cs, int64_t k2, const void *X2, int64_t ldx2, const void *q2, int64_t q2_rs,
                        int64_t q2_cs, int64_t ma, void *OutA, int64_t ldoa, int64_t mb, void *OutB, int64_t ldob) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_block_update2x2: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && k1 >= 1 && k2 >= 1 && ma >= 1 && mb >= 1, "rlh_block_update2x2: bad size");
  if (n == 0) return 0;
  RLH_REQUIRE(OutA && OutB && X1 && X2 && q1 && q2, "rlh_block_update2x2: null pointer");
  RLH_REQUIRE(ldoa >= n && ldob >= n && ldx1 >= n && ldx2 >= n, "rlh_block_update2x2: leading dimension smaller than n");
  const int64_t es = dtype_size(dtype);
  const int64_t ba = ((ma - 1) * ldoa + n) * es, bb = ((mb - 1) * ldob + n) * es;
  RLH_REQUIRE(!overlaps(X1, ((k1 - 1) * ldx1 + n) * es, OutA, ba) && !overlaps(X2, ((k2 - 1) * ldx2 + n) * es, OutA, ba) &&
                  !overlaps(X1, ((k1 - 1) * ldx1 + n) * es, OutB, bb) && !overlaps(X2, ((k2 - 1) * ldx2 + n) * es, OutB, bb) &&
                  !overlaps(OutA, ba, OutB, bb),
              "rlh_block_update2x2: an output window overlaps an input window or the other output");
  const double one[2] = {1.0, 0.0};
  int rc = 0;
  // Complex blocks whose combined coefficients exceed the LDS of the matrix-core kernel (k1 + k2 = 128, ma + mb = 128
  // complex128: 262 KB) would fall to the VALU kernel in panels of 16 columns, each re-reading both sources (5.4 ms at
  // n = 2 x 10^6); each result block alone fits (131 KB): two matrix-core passes, 4.7 ms.
  if (dtype == RLH_C || dtype == RLH_Z) {
    const int64_t ks = (k1 + 3) / 4 + (k2 + 3) / 4;
    const int64_t both = ((ma + mb + 15) / 16) * ks * 64 * es, worst = ((std::max(ma, mb) + 15) / 16) * ks * 64 * es;
    if (ks <= kMfmaUpdKS && both > 160 * 1024 && worst <= 160 * 1024) {
      RLH_DISPATCH(dtype, block_update2_impl, n, k1, X1, ldx1, q1, q1_rs, q1_cs, k2, X2, ldx2, q2, q2_rs, q2_cs, ma, OutA,
                   ldoa, one, 0, nullptr, 0, -1)
      if (rc) return rc;
      RLH_DISPATCH(dtype, block_update2_impl, n, k1, X1, ldx1, (const char *)q1 + ma * q1_cs * es, q1_rs, q1_cs, k2, X2, ldx2,
                   (const char *)q2 + ma * q2_cs * es, q2_rs, q2_cs, mb, OutB, ldob, one, 0, nullptr, 0, -1)
      return rc;
    }
  }
  RLH_DISPATCH(dtype, block_update2_impl, n, k1, X1, ldx1, q1, q1_rs, q1_cs, k2, X2, ldx2, q2, q2_rs, q2_cs, ma + mb,
               OutA, ldoa, one, 0, OutB, ldob, ma)
  return rc;
}

int rlh_lincomb_cols(int dtype, int64_t n, int64_t m, const void *a, const void *A, int64_t lda, const void *b,
                     const void *B, int64_t ldb, void *Out, int64_t ldo) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_lincomb_cols: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && m >= 0, "rlh_lincomb_cols: negative size");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(a && b && A && B && Out, "rlh_lincomb_cols: null pointer");
  RLH_REQUIRE(lda >= n && ldb >= n && ldo >= n, "rlh_lincomb_cols: leading dimension smaller than n");
  int rc = 0;
  RLH_DISPATCH(dtype, lincomb_cols_impl, n, m, a, A, lda, b, B, ldb, Out, ldo)
  return rc;
}

int rlh_axpy(int dtype, int64_t n, int64_t m, const double *alpha, const void *X, int64_t ldx, void *Y, int64_t ldy) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_axpy: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && m >= 0, "rlh_axpy: negative size");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(alpha && X && Y, "rlh_axpy: null pointer");
  RLH_REQUIRE(ldx >= n && ldy >= n, "rlh_axpy: leading dimension smaller than n");
  float sf[2] = {(float)alpha[0], (float)alpha[1]};
  const void *s = (dtype == RLH_S || dtype == RLH_C) ? (const void *)sf : (const void *)alpha;
  int rc = 0;
  RLH_DISPATCH(dtype, axpy_cols_impl, n, m, s, true, X, ldx, Y, ldy)
  return rc;
}

int rlh_axpy_cols(int dtype, int64_t n, int64_t m, const void *s, const void *X, int64_t ldx, void *Y, int64_t ldy) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_axpy_cols: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && m >= 0, "rlh_axpy_cols: negative size");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(s && X && Y, "rlh_axpy_cols: null pointer");
  RLH_REQUIRE(ldx >= n && ldy >= n, "rlh_axpy_cols: leading dimension smaller than n");
  int rc = 0;
  RLH_DISPATCH(dtype, axpy_cols_impl, n, m, s, false, X, ldx, Y, ldy)
  return rc;
}

int rlh_copy(int dtype, int64_t n, int64_t m, const void *X, int64_t ldx, void *Y, int64_t ldy) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_copy: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && m >= 0, "rlh_copy: negative size");
  if (n == 0 || m == 0 || (X == Y && ldx == ldy)) return 0;
  RLH_REQUIRE(X && Y, "rlh_copy: null pointer");
  RLH_REQUIRE(ldx >= n && ldy >= n, "rlh_copy: leading dimension smaller than n");
  return copy_cols_impl(dtype, n, m, nullptr, X, ldx, Y, ldy);
}

int rlh_copy_cols(int dtype, int64_t n, int64_t m, const int64_t *ind, const void *Xall, int64_t ldx, void *Y,
                  int64_t ldy) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_copy_cols: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && m >= 0, "rlh_copy_cols: negative size");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(ind && Xall && Y, "rlh_copy_cols: null pointer");
  RLH_REQUIRE(ldx >= n && ldy >= n, "rlh_copy_cols: leading dimension smaller than n");
  for (int64_t i = 0; i < m; ++i) RLH_REQUIRE(ind[i] >= 0, "rlh_copy_cols: negative index");
  return copy_cols_impl(dtype, n, m, ind, Xall, ldx, Y, ldy);
}

int rlh_scale_cols(int dtype, int64_t n, int64_t m, const double *s, int mode, void *X, int64_t ldx) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_scale_cols: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && m >= 0, "rlh_scale_cols: negative size");
  RLH_REQUIRE(mode == 0 || mode == 1, "rlh_scale_cols: mode must be 0 (divide) or 1 (multiply)");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(s && X, "rlh_scale_cols: null pointer");
  RLH_REQUIRE(ldx >= n, "rlh_scale_cols: leading dimension smaller than n");
  int rc = 0;
  RLH_DISPATCH(dtype, scale_cols_impl, n, m, s, mode, X, ldx)
  return rc;
}

int rlh_convert(int src_dtype, int dst_dtype, int64_t n, int64_t m, const void *X, int64_t ldx, void *Y, int64_t ldy) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(src_dtype) && dtype_valid(dst_dtype), "rlh_convert: unknown dtype");
  const bool cs = (src_dtype == RLH_C || src_dtype == RLH_Z), cd = (dst_dtype == RLH_C || dst_dtype == RLH_Z);
  RLH_REQUIRE(cs == cd, "rlh_convert: real <-> complex conversion is not supported");
  RLH_REQUIRE(n >= 0 && m >= 0, "rlh_convert: negative size");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(X && Y && ldx >= n && ldy >= n, "rlh_convert: bad arguments");
  if (src_dtype == dst_dtype) return rlh_copy(src_dtype, n, m, X, ldx, Y, ldy);
  // complex blocks are converted as real blocks of twice the length
  const int64_t f = cs ? 2 : 1;
  const bool src_single = (src_dtype == RLH_S || src_dtype == RLH_C);
  dim3 grid(row_blocks(n * f, m), (unsigned)m);
  if (src_single)
    hipLaunchKernelGGL((convert_kernel<float, double>), grid, dim3(256), 0, ctx().stream, (const float *)X, ldx * f,
                       (double *)Y, ldy * f, n * f);
  else
    hipLaunchKernelGGL((convert_kernel<double, float>), grid, dim3(256), 0, ctx().stream, (const double *)X, ldx * f,
                       (float *)Y, ldy * f, n * f);
  RLH_HIP(hipGetLastError());
  return 0;
}

int rlh_fill_random(int dtype, int64_t n, int64_t m, void *X, int64_t ldx, uint64_t seed, int64_t row0, int64_t col0) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_fill_random: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && m >= 0 && row0 >= 0 && col0 >= 0, "rlh_fill_random: negative size or offset");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(X && ldx >= n, "rlh_fill_random: bad arguments");
  dim3 grid(row_blocks(n, m), (unsigned)m);
  hipStream_t st = ctx().stream;
  switch (dtype) {
    case RLH_S: hipLaunchKernelGGL((fill_random_kernel<float, 1>), grid, dim3(256), 0, st, (float *)X, ldx, n, seed, row0, col0); break;
    case RLH_D: hipLaunchKernelGGL((fill_random_kernel<double, 1>), grid, dim3(256), 0, st, (double *)X, ldx, n, seed, row0, col0); break;
    case RLH_C: hipLaunchKernelGGL((fill_random_kernel<float, 2>), grid, dim3(256), 0, st, (float *)X, 2 * ldx, n, seed, row0, col0); break;
    case RLH_Z: hipLaunchKernelGGL((fill_random_kernel<double, 2>), grid, dim3(256), 0, st, (double *)X, 2 * ldx, n, seed, row0, col0); break;
  }
  RLH_HIP(hipGetLastError());
  return 0;
}

int rlh_bf16_pack(int src_dtype, int64_t n, int64_t m, const void *X, int64_t ldx, double scale, void *Y16, int64_t ldy) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(src_dtype == RLH_S || src_dtype == RLH_D, "rlh_bf16_pack: real blocks only");
  RLH_REQUIRE(n >= 0 && m >= 0, "rlh_bf16_pack: negative size");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(X && Y16 && ldx >= n && ldy >= n, "rlh_bf16_pack: bad arguments");
  dim3 grid(row_blocks(n, m), (unsigned)m);
  if (src_dtype == RLH_S)
    hipLaunchKernelGGL((bf16_pack_kernel<float>), grid, dim3(256), 0, ctx().stream, (const float *)X, ldx, (float)scale,
                       (unsigned short *)Y16, ldy, n);
  else
    hipLaunchKernelGGL((bf16_pack_kernel<double>), grid, dim3(256), 0, ctx().stream, (const double *)X, ldx, (float)scale,
                       (unsigned short *)Y16, ldy, n);
  RLH_HIP(hipGetLastError());
  return 0;
}

int rlh_bf16_unpack(int dst_dtype, int64_t n, int64_t m, const void *X16, int64_t ldx, void *Y, int64_t ldy) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dst_dtype == RLH_S || dst_dtype == RLH_D, "rlh_bf16_unpack: real blocks only");
  RLH_REQUIRE(n >= 0 && m >= 0, "rlh_bf16_unpack: negative size");
  if (n == 0 || m == 0) return 0;
  RLH_REQUIRE(X16 && Y && ldx >= n && ldy >= n, "rlh_bf16_unpack: bad arguments");
  dim3 grid(row_blocks(n, m), (unsigned)m);
  if (dst_dtype == RLH_S)
    hipLaunchKernelGGL((bf16_unpack_kernel<float>), grid, dim3(256), 0, ctx().stream, (const unsigned short *)X16, ldx,
                       (float *)Y, ldy, n);
  else
    hipLaunchKernelGGL((bf16_unpack_kernel<double>), grid, dim3(256), 0, ctx().stream, (const unsigned short *)X16, ldx,
                       (double *)Y, ldy, n);
  RLH_HIP(hipGetLastError());
  return 0;
}

int rlh_conj(int dtype, int64_t n, int64_t m, void *X, int64_t ldx) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_conj: unknown dtype %d", dtype);
  if (dtype == RLH_S || dtype == RLH_D || n <= 0 || m <= 0) return 0;
  RLH_REQUIRE(X && ldx >= n, "rlh_conj: bad arguments");
  dim3 grid(row_blocks(n, m), (unsigned)m);
  if (dtype == RLH_C)
    hipLaunchKernelGGL((conj_kernel<float>), grid, dim3(256), 0, ctx().stream, (float *)X, 2 * ldx, n);
  else
    hipLaunchKernelGGL((conj_kernel<double>), grid, dim3(256), 0, ctx().stream, (double *)X, 2 * ldx, n);
  RLH_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"

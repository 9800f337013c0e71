// K12: dense operator Y = A X / Y = A^H X on a block of vectors (the PCA A^T A path).
//
// Reference: one gemm(Trans|NoTrans|ConjTrans, NoTrans, ny, k, nx, 1, A, lda, x, nx, 0, y, ny)
// per application (raleigh/algebra/dense_cublas.py:732-776; dense_numpy.py:153-175); complex
// transp on C-ordered A conjugates x and y around the call (dense_cublas.py:750-776).
//
// fp32: LDS-tiled GEMM on v_mfma_f32_32x32x2_f32.  C^T tile of (BN vectors) x (64 output rows)
// per 256-thread workgroup, BK = 32, register-prefetched next tile, LDS rows padded to 33
// floats so the per-lane fragment reads (32 rows x 1 k) are bank-conflict free.  The vector
// index rides the MFMA row and the output-row index rides the MFMA column (= lane & 31), so
// each accumulator register stores as 128-byte contiguous runs of the column-major Y.
// Other dtypes (f64, complex): dense_mfma16_kernel on the 16x16x4 matrix-core shapes (real planes); the generic
// LDS-tiled VALU kernel remains for unaligned layouts.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace rlh {

struct DenseArgs {
  const void *A; int64_t lda;
  int a_kcontig;        // 1: Op(i,k) = A[i*lda + k]; 0: Op(i,k) = A[k*lda + i]
  int conj_a;           // complex only
  int64_t ny, nx;       // Op is ny x nx
  const void *X; int64_t ldx;
  void *Y; int64_t ldy;
  int m;
  // rank-one epilogue Y[i, v] -= u[i] * c[v] (u == nullptr: a vector of ones), folded into the tile
  // store / the split-K reduction: the mean-shift corrections of the PCA operator
  // (raleigh/interfaces/partial_svd.py:258-291) without a further pass over the block
  const void *r1_u, *r1_c;
};

// y - u c for the element types (real: plain product; complex: u[i] * c[v], no conjugation)
__device__ __forceinline__ float r1_sub(float y, float u, float c) { return y - u * c; }
__device__ __forceinline__ double r1_sub(double y, double u, double c) { return y - u * c; }
__device__ __forceinline__ c32 r1_sub(c32 y, c32 u, c32 c) { const c32 p = mul_of(u, c); return c32{y.re - p.re, y.im - p.im}; }
__device__ __forceinline__ c64 r1_sub(c64 y, c64 u, c64 c) { const c64 p = mul_of(u, c); return c64{y.re - p.re, y.im - p.im}; }
__device__ __forceinline__ float one_of(float) { return 1.f; }
__device__ __forceinline__ double one_of(double) { return 1.0; }
__device__ __forceinline__ c32 one_of(c32) { return c32{1.f, 0.f}; }
__device__ __forceinline__ c64 one_of(c64) { return c64{1.0, 0.0}; }
template <typename T>
__device__ __forceinline__ T r1_apply(const DenseArgs &a, T y, int64_t i, int v) {
  if (!a.r1_c) return y;
  const T u = a.r1_u ? ((const T *)a.r1_u)[i] : one_of(T{});
  return r1_sub(y, u, ((const T *)a.r1_c)[v]);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// C^T tile of BN vectors x MR output rows per 256-thread workgroup, BK = 32, LDS double
// buffered (one barrier per K step, global loads of step t+1 in flight during the MFMAs of
// step t).  gridDim.z splits K; with more than one split the partial tiles go to a workspace
// [split][vector][row] that dense_splitk_reduce sums in a fixed order.
// Requires 16-byte aligned A / X with leading dimensions that are multiples of 4 floats.
template <int MR, int BN, bool A_KC>
__global__ __launch_bounds__(256) void dense_mfma_f32_kernel(DenseArgs a, float *__restrict__ part, int64_t kchunk) {
  constexpr int BK = 32, SK = BK + 1;
  constexpr int WGN = (BN >= 64) ? 2 : 1;          // waves along the vector dimension
  constexpr int WGM = 4 / WGN;                     // waves along the output-row dimension
  constexpr int TN = BN / WGN / 32, TM = MR / WGM / 32;
  static_assert(TN >= 1 && TM >= 1, "tile split");
  constexpr int A_UNITS = MR * BK / 4 / 256;       // 16-byte units per thread and K step
  constexpr int B_UNITS = BN * BK / 4 / 256;
  static_assert(A_UNITS >= 1 && B_UNITS >= 1, "tile too small for 256 threads");

  __shared__ float ldsA[2][MR * SK];
  __shared__ float ldsB[2][BN * SK];

  const float *__restrict__ A = (const float *)a.A;
  const float *__restrict__ X = (const float *)a.X;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int64_t i0 = (int64_t)blockIdx.x * MR;
  const int v0 = blockIdx.y * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
  const int64_t kend = (kbeg + kchunk) < a.nx ? (kbeg + kchunk) : a.nx;

  float4 ra[A_UNITS], rb[B_UNITS];

  auto load_tiles = [&](int64_t k0) {
    const bool tail = (k0 + BK > kend);              // wave-uniform: only the last K step masks
#pragma unroll
    for (int q = 0; q < A_UNITS; ++q) {
      const int u = tid + q * 256;
      float4 v;
      if (A_KC) {                  // unit = 4 consecutive k of one output row
        const int r = u / (BK / 4), kq = (u % (BK / 4)) * 4;
        int64_t i = i0 + r;
        i = i < a.ny ? i : a.ny - 1;                 // clamped rows are computed but never stored
        int64_t k = k0 + kq;
        const int64_t kc = (k + 3 < a.lda) ? k : 0;
        v = *(const float4 *)(A + i * a.lda + kc);
        if (tail) {
          if (k + 0 >= kend) v.x = 0.f;
          if (k + 1 >= kend) v.y = 0.f;
          if (k + 2 >= kend) v.z = 0.f;
          if (k + 3 >= kend) v.w = 0.f;
        }
      } else {                     // unit = 4 consecutive output rows at one k
        const int kk = u / (MR / 4), rq = (u % (MR / 4)) * 4;
        int64_t i = i0 + rq;
        i = (i + 3 < a.lda) ? i : 0;
        const int64_t k = k0 + kk;
        const int64_t kc = k < kend ? k : kbeg;
        v = *(const float4 *)(A + kc * a.lda + i);
        if (tail && k >= kend) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      ra[q] = v;
    }
#pragma unroll
    for (int q = 0; q < B_UNITS; ++q) {
      const int u = tid + q * 256;
      const int c = u / (BK / 4), kq = (u % (BK / 4)) * 4;
      int vc = v0 + c;
      vc = vc < a.m ? vc : a.m - 1;
      const int64_t k = k0 + kq;
      const int64_t kc = (k + 3 < a.ldx) ? k : 0;
      float4 v = *(const float4 *)(X + (int64_t)vc * a.ldx + kc);
      if (tail) {
        if (k + 0 >= kend) v.x = 0.f;
        if (k + 1 >= kend) v.y = 0.f;
        if (k + 2 >= kend) v.z = 0.f;
        if (k + 3 >= kend) v.w = 0.f;
      }
      rb[q] = v;
    }
  };

  auto store_tiles = [&](int buf) {
    float *la = ldsA[buf], *lb = ldsB[buf];
#pragma unroll
    for (int q = 0; q < A_UNITS; ++q) {
      const int u = tid + q * 256;
      if (A_KC) {
        const int r = u / (BK / 4), kq = (u % (BK / 4)) * 4;
        float *d = la + r * SK + kq;
        d[0] = ra[q].x; d[1] = ra[q].y; d[2] = ra[q].z; d[3] = ra[q].w;
      } else {
        const int kk = u / (MR / 4), rq = (u % (MR / 4)) * 4;
        float *d = la + rq * SK + kk;
        d[0] = ra[q].x; d[SK] = ra[q].y; d[2 * SK] = ra[q].z; d[3 * SK] = ra[q].w;
      }
    }
#pragma unroll
    for (int q = 0; q < B_UNITS; ++q) {
      const int u = tid + q * 256;
      const int c = u / (BK / 4), kq = (u % (BK / 4)) * 4;
      float *d = lb + c * SK + kq;
      d[0] = rb[q].x; d[1] = rb[q].y; d[2] = rb[q].z; d[3] = rb[q].w;
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;

  const int fr = lane & 31, fk = lane >> 5;
  const int mrow0 = wm * (MR / WGM), vcol0 = wn * (BN / WGN);
  int buf = 0;
  if (kbeg < kend) {
    load_tiles(kbeg);
    store_tiles(0);
  }
  __syncthreads();
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = (k0 + BK < kend);
    if (more) load_tiles(k0 + BK);
    const float *la = ldsA[buf], *lb = ldsB[buf];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float fx[TN], fa[TM];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) fx[tn] = lb[(vcol0 + tn * 32 + fr) * SK + 2 * ks + fk];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) fa[tm] = la[(mrow0 + tm * 32 + fr) * SK + 2 * ks + fk];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(fx[tn], fa[tm], acc[tn][tm], 0, 0, 0);
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // D[v][i]: col (lane & 31) = output row i, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = vector v
  float *__restrict__ out = part ? part + (int64_t)blockIdx.z * a.m * a.ny : (float *)a.Y;
  const int64_t ldo = part ? a.ny : a.ldy;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int v = v0 + vcol0 + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int64_t i = i0 + mrow0 + tm * 32 + (lane & 31);
        if (v < a.m && i < a.ny) out[i + (int64_t)v * ldo] = part ? acc[tn][tm][r] : r1_apply<float>(a, acc[tn][tm][r], i, v);
      }
}

__global__ __launch_bounds__(256) void dense_splitk_reduce(const float *__restrict__ part, int splits, int64_t ny, int m,
                                                           float *__restrict__ Y, int64_t ldy, DenseArgs a) {
  const int v = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ny; i += stride) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += part[((int64_t)z * m + v) * ny + i];
    Y[i + (int64_t)v * ldy] = r1_apply<float>(a, s, i, v);
  }
}

// Second-generation fp32 kernel (RLH_DENSE_KERNEL=2, the default): the same tiling -- C^T tile of BN
// vectors x 128 output rows per 256-thread workgroup, K split over gridDim.z -- with
//  * an LDS image [row][BK floats] whose 8-byte units are XOR-swizzled with the row number
//    (unit u of row r sits at position u ^ ((r >> 1) & (BK / 2 - 1))): the tiles go in with ONE
//    ds_write_b128 per 16-byte global piece (the 33-float padding of the first kernel forced four
//    ds_write_b32) and the MFMA fragments come out with ONE conflict-free ds_read_b64 per operand tile and
//    TWO k-steps: lanes 0-31 read the unit (k, k + 1), lanes 32-63 the unit (k + 2, k + 3) of their row
//    -- the k order inside four consecutive k is permuted identically for both operands, which a sum over
//    k does not see;
//  * BK = 16 as well as 32: half the LDS per workgroup, so four workgroups instead of two share a CU and
//    a SIMD always has a wave with an MFMA ready while another waits at its barrier (PMC of the first
//    kernel at 20000 x 20000 x 128: MFMA pipes 76 % busy, 1.7 waves per SIMD);
//  * column-major tiles (the transposed product of a row-major matrix) transposed 4 x 4 in registers on
//    the way in, so they too are written with ds_write_b128, conflict-free.
template <int BN, int BK, bool A_KC>
__global__ __launch_bounds__(256, (BK == 16 ? 3 : 2)) void dense_mfma2_f32_kernel(DenseArgs a, float *__restrict__ part, int64_t kchunk) {
  constexpr int MR = 128;
  constexpr int WGN = (BN >= 64) ? 2 : 1;          // waves along the vector dimension
  constexpr int WGM = 4 / WGN;                     // waves along the output-row dimension
  constexpr int TN = BN / WGN / 32, TM = MR / WGM / 32;
  constexpr int ROWB = BK * 4;                     // bytes per LDS row
  constexpr int UM = BK / 2 - 1;                   // swizzle mask on 8-byte units
  constexpr int QPR = BK / 4;                      // 16-byte pieces per row
  constexpr int NBLK = (MR / 4) * (BK / 4);        // 4 x 4 blocks of a column-major tile (A_KC false)
  constexpr int A_UNITS = A_KC ? MR * BK / 4 / 256 : 4;   // 16-byte pieces per thread and K step
  constexpr int B_UNITS = (BN * BK / 4 + 255) / 256;
  static_assert(TN >= 1 && TM >= 1 && A_UNITS >= 1 && NBLK <= 256, "tile split");
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));

  __shared__ __attribute__((aligned(16))) char ldsA[2][MR * ROWB];
  __shared__ __attribute__((aligned(16))) char ldsB[2][BN * ROWB];

  const float *__restrict__ A = (const float *)a.A;
  const float *__restrict__ X = (const float *)a.X;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int64_t i0 = (int64_t)blockIdx.x * MR;
  const int v0 = blockIdx.y * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
  const int64_t kend = (kbeg + kchunk) < a.nx ? (kbeg + kchunk) : a.nx;

  // byte offset of the 16-byte piece (k .. k + 3, k a multiple of 4) of row r, and whether its two
  // 8-byte units are stored in swapped order
  auto piece_off = [](int r, int kq4) -> int { return r * ROWB + ((((kq4 >> 1) ^ ((r >> 1) & UM)) >> 1) << 4); };
  auto piece_swap = [](int r) -> bool { return ((r >> 1) & 1) != 0; };

  f4 ra[A_UNITS], rb[B_UNITS];
  auto load_tiles = [&](int64_t k0, bool tail) {
    if constexpr (A_KC) {          // piece = 4 consecutive k of one output row
#pragma unroll
      for (int q = 0; q < A_UNITS; ++q) {
        const int u = tid + q * 256;
        const int r = u / QPR, kq = (u % QPR) * 4;
        int64_t i = i0 + r;
        i = i < a.ny ? i : a.ny - 1;                 // clamped rows are computed but never stored
        const int64_t k = k0 + kq;
        f4 v = *(const f4 *)(A + i * a.lda + ((!tail || k + 3 < a.lda) ? k : 0));
        if (tail) {
          if (k + 0 >= kend) v[0] = 0.f;
          if (k + 1 >= kend) v[1] = 0.f;
          if (k + 2 >= kend) v[2] = 0.f;
          if (k + 3 >= kend) v[3] = 0.f;
        }
        ra[q] = v;
      }
    } else {                       // piece q of the thread = rows rq .. rq + 3 at k = kq + q: a 4 x 4 block
      const int blk = NBLK >= 256 ? tid : tid % NBLK;   // (BK = 16: 128 blocks, the upper half of the threads repeats them)
      const int kq = (blk / (MR / 4)) * 4, rq = (blk % (MR / 4)) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int64_t i = i0 + rq;
        i = (i + 3 < a.lda) ? i : 0;
        const int64_t k = k0 + kq + q;
        f4 v = *(const f4 *)(A + ((!tail || k < kend) ? k : kbeg) * a.lda + i);
        if (tail && k >= kend) v = f4{0.f, 0.f, 0.f, 0.f};
        ra[q] = v;
      }
    }
#pragma unroll
    for (int q = 0; q < B_UNITS; ++q) {
      const int u = tid + q * 256;
      const int c = (u / QPR) % BN, kq = (u % QPR) * 4;
      int vc = v0 + c;
      vc = vc < a.m ? vc : a.m - 1;
      const int64_t k = k0 + kq;
      f4 v = *(const f4 *)(X + (int64_t)vc * a.ldx + ((!tail || k + 3 < a.ldx) ? k : 0));
      if (tail) {
        if (k + 0 >= kend) v[0] = 0.f;
        if (k + 1 >= kend) v[1] = 0.f;
        if (k + 2 >= kend) v[2] = 0.f;
        if (k + 3 >= kend) v[3] = 0.f;
      }
      rb[q] = v;
    }
  };
  auto put = [&](char *base, int r, int kq, f4 v) {
    const f4 w = piece_swap(r) ? f4{v[2], v[3], v[0], v[1]} : v;
    *reinterpret_cast<f4 *>(base + piece_off(r, kq)) = w;
  };
  auto store_tiles = [&](int buf) {
    char *la = ldsA[buf], *lb = ldsB[buf];
    if constexpr (A_KC) {
#pragma unroll
      for (int q = 0; q < A_UNITS; ++q) {
        const int u = tid + q * 256;
        put(la, u / QPR, (u % QPR) * 4, ra[q]);
      }
    } else {
      const int blk = NBLK >= 256 ? tid : tid % NBLK;
      const int kq = (blk / (MR / 4)) * 4, rq = (blk % (MR / 4)) * 4;
      if (NBLK >= 256 || tid < NBLK) {
#pragma unroll
        for (int j = 0; j < 4; ++j)                  // row rq + j: its four k from the four loaded pieces
          put(la, rq + j, kq, f4{ra[0][j], ra[1][j], ra[2][j], ra[3][j]});
      }
    }
#pragma unroll
    for (int q = 0; q < B_UNITS; ++q) {
      const int u = tid + q * 256;
      if (BN * QPR >= 256 || u < BN * QPR) put(lb, (u / QPR) % BN, (u % QPR) * 4, rb[q]);
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][tm][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  const int mrow0 = wm * (MR / WGM), vcol0 = wn * (BN / WGN);
  // byte offsets of this lane's fragment rows (k-independent part) and their swizzle keys
  int offA[TM], keyA[TM], offB[TN], keyB[TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) { const int r = mrow0 + tm * 32 + fr; offA[tm] = r * ROWB; keyA[tm] = (r >> 1) & UM; }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) { const int r = vcol0 + tn * 32 + fr; offB[tn] = r * ROWB; keyB[tn] = (r >> 1) & UM; }

  int buf = 0;
  if (kbeg < kend) {
    load_tiles(kbeg, kbeg + BK > kend);
    store_tiles(0);
  }
  __syncthreads();
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = (k0 + BK < kend);
    if (more) {
      if (k0 + 2 * BK > kend) load_tiles(k0 + BK, true); else load_tiles(k0 + BK, false);
    }
    const char *la = ldsA[buf], *lb = ldsB[buf];
    // Fragments in two register sets: the ds_reads of group g + 1 are issued BEFORE the eight MFMAs of
    // group g (the scheduling barriers keep them there: left alone the compiler put each group's reads
    // behind the previous group's MFMAs and waited for them at once -- one LDS latency per 512 MFMA
    // cycles with nothing else for the wave to issue).  Four k per group: units 2 g (lanes 0-31) and
    // 2 g + 1 (lanes 32-63).
    f2 fxa[TN], faa[TM], fxb[TN], fab[TM];
    auto frag_read = [&](int g, f2 (&fx)[TN], f2 (&fa)[TM]) {
      const int u = 2 * g + fh;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) fx[tn] = *reinterpret_cast<const f2 *>(lb + offB[tn] + ((u ^ keyB[tn]) << 3));
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) fa[tm] = *reinterpret_cast<const f2 *>(la + offA[tm] + ((u ^ keyA[tm]) << 3));
    };
    auto frag_mfma = [&](const f2 (&fx)[TN], const f2 (&fa)[TM]) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(fx[tn][h], fa[tm][h], acc[tn][tm], 0, 0, 0);
    };
    frag_read(0, fxa, faa);
#pragma unroll
    for (int g = 0; g < BK / 4; g += 2) {
      frag_read(g + 1, fxb, fab);
      __builtin_amdgcn_sched_barrier(0);
      frag_mfma(fxa, faa);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 2 < BK / 4) frag_read(g + 2, fxa, faa);
      __builtin_amdgcn_sched_barrier(0);
      frag_mfma(fxb, fab);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // D[v][i]: col (lane & 31) = output row i, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = vector v
  float *__restrict__ out = part ? part + (int64_t)blockIdx.z * a.m * a.ny : (float *)a.Y;
  const int64_t ldo = part ? a.ny : a.ldy;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int v = v0 + vcol0 + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int64_t i = i0 + mrow0 + tm * 32 + (lane & 31);
        if (v < a.m && i < a.ny) out[i + (int64_t)v * ldo] = part ? acc[tn][tm][r] : r1_apply<float>(a, acc[tn][tm][r], i, v);
      }
}

// Generic VALU GEMM for f64 / complex: 64 x 64 output tile, BK = 16, 4 x 4 micro-tile per thread.
template <typename T, bool CONJ>
__global__ __launch_bounds__(256) void dense_valu_kernel(DenseArgs a) {
  constexpr int BM = 64, BN = 64, BK = 16;
  __shared__ T ldsA[BK][BM + 1];
  __shared__ T ldsB[BK][BN + 1];
  const T *A = (const T *)a.A;
  const T *X = (const T *)a.X;
  T *Y = (T *)a.Y;
  const int tid = threadIdx.x;
  const int ti = tid % 16, tv = tid / 16;          // micro-tile: rows ti*4.., vectors tv*4..
  const int64_t i0 = (int64_t)blockIdx.x * BM;
  const int v0 = blockIdx.y * BN;
  T acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[p][q] = zero_of(T{});
  for (int64_t k0 = 0; k0 < a.nx; k0 += BK) {
    for (int u = tid; u < BM * BK; u += 256) {
      int r, kk;
      if (a.a_kcontig) { r = u / BK; kk = u % BK; } else { kk = u / BM; r = u % BM; }
      const int64_t i = i0 + r, k = k0 + kk;
      T v = zero_of(T{});
      if (i < a.ny && k < a.nx) v = a.a_kcontig ? A[i * a.lda + k] : A[k * a.lda + i];
      ldsA[kk][r] = v;
    }
    for (int u = tid; u < BN * BK; u += 256) {
      const int c = u / BK, kk = u % BK;
      const int64_t k = k0 + kk;
      T v = zero_of(T{});
      if (v0 + c < a.m && k < a.nx) v = X[(int64_t)(v0 + c) * a.ldx + k];
      ldsB[kk][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      T fa[4], fx[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) fa[p] = ldsA[kk][ti * 4 + p];
#pragma unroll
      for (int q = 0; q < 4; ++q) fx[q] = ldsB[kk][tv * 4 + q];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (CONJ) fma_conj_acc(acc[p][q], fa[p], fx[q]);
          else fma_acc(acc[p][q], fa[p], fx[q]);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t i = i0 + ti * 4 + p;
      const int v = v0 + tv * 4 + q;
      if (i < a.ny && v < a.m) Y[i + (int64_t)v * a.ldy] = r1_apply<T>(a, acc[p][q], i, v);
    }
}

// ---------------------------------------------------------------- double / complex on the matrix cores
// The reference runs s / d / c / z through the same gemm (dense_cublas.py:732-776); here float has the two tuned
// kernels above, and double, complex double and complex float share this one: v_mfma_f64_16x16x4_f64 (or
// v_mfma_f32_16x16x4_f32: the same shape) on REAL planes.  A complex tile is split into its re and im planes on
// its way into the LDS and the product becomes four real ones per k-step (conj(A): the sign of the im plane).
//  * C^T tile of 64 vectors x 64 output rows per 256-thread workgroup (four waves of 32 x 32 = 2 x 2 MFMA tiles),
//    BK = 16, LDS double buffered, loads of step t + 1 in flight during the MFMAs of step t;
//  * LDS planes [index][18] elements: a fragment read (lane l: index l % 16, k = l / 16) then falls into 32
//    different 8-byte slots per half-wave (row stride = 2 mod 32), and a 16-byte piece of two consecutive k goes
//    in with one ds_write_b128;
//  * the vector index rides the MFMA row and the output row the MFMA column (lane % 16), so an accumulator
//    register stores as 128-byte runs of the column-major Y.
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f64x4 mfma16(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <typename R> struct Acc4;
template <> struct Acc4<double> { using V = f64x4; };
template <> struct Acc4<float> { using V = f32x4; };

template <typename T, typename R, bool CPLX, bool A_KC, bool CONJ, int TR, int TV>
__global__ __launch_bounds__(256, 2) void dense_mfma16_kernel(DenseArgs a, T *__restrict__ part, int64_t kchunk) {
  constexpr int MR = 32 * TR, BN = 32 * TV, BK = 16, SK = BK + 2;     // 2 x 2 waves of TR x TV MFMA tiles
  constexpr int NP = CPLX ? 2 : 1;                  // planes
  constexpr int EPU = 16 / (int)sizeof(T);          // elements of T per 16-byte unit
  constexpr int UA = MR * BK / EPU / 256, UB = BN * BK / EPU / 256;   // units per thread and K step: A tile, X tile
  static_assert(UA >= 1 && UB >= 1, "tile too small");
  using V = typename Acc4<R>::V;
  struct alignas(16) Unit { T e[EPU]; };
  __shared__ __attribute__((aligned(16))) R ldsA[2][NP][MR * SK];
  __shared__ __attribute__((aligned(16))) R ldsB[2][NP][BN * SK];
  const T *__restrict__ A = (const T *)a.A;
  const T *__restrict__ X = (const T *)a.X;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves: output rows x vectors
  const int64_t i0 = (int64_t)blockIdx.x * MR;
  const int v0 = blockIdx.y * BN;
  // gridDim.z splits K (chunks of whole K steps): the M x m output alone gives too few workgroups to fill the chip evenly --
  // 20 000 x 128 in 64 x 64 tiles is 626 workgroups for 512 resident slots, 1.2 rounds, the second one a fifth full
  const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
  const int64_t kend = (kbeg + kchunk) < a.nx ? (kbeg + kchunk) : a.nx;
  Unit ra0[UA], rb0[UB], ra1[UA], rb1[UB];      // two register sets: the tile two K steps ahead is in flight
  auto zero_unit = [](Unit &u) {
#pragma unroll
    for (int e = 0; e < EPU; ++e) u.e[e] = zero_of(T{});
  };
  // (masking in a code path of its own: a per-element test of the loaded values in every step makes the compiler wait
  // for the loads right where they are issued -- the first version of this kernel did, and ran at 42 % MFMA busy)
  auto load_tiles_as = [&](int64_t k0, Unit (&ra)[UA], Unit (&rb)[UB], auto mask) {
    constexpr bool MASK = decltype(mask)::value;
#pragma unroll
    for (int q = 0; q < UA; ++q) {
      const int u = tid + q * 256;
      if (A_KC) {                                   // EPU consecutive k of one output row
        const int r = u / (BK / EPU), kq = (u % (BK / EPU)) * EPU;
        int64_t i = i0 + r;
        i = i < a.ny ? i : a.ny - 1;                // clamped rows are computed but never stored
        const int64_t k = k0 + kq;
        const int64_t kc = (!MASK || k + EPU <= a.lda) ? k : 0;
        ra[q] = *reinterpret_cast<const Unit *>(A + i * a.lda + kc);
        if constexpr (MASK) {
#pragma unroll
          for (int e = 0; e < EPU; ++e)
            if (k + e >= kend) ra[q].e[e] = zero_of(T{});
        }
      } else {                                      // EPU consecutive output rows at one k
        const int kk = u / (MR / EPU), rq = (u % (MR / EPU)) * EPU;
        int64_t i = i0 + rq;
        i = (i + EPU <= a.lda) ? i : 0;
        const int64_t k = k0 + kk;
        const int64_t kc = (!MASK || k < kend) ? k : 0;
        ra[q] = *reinterpret_cast<const Unit *>(A + kc * a.lda + i);
        if constexpr (MASK) { if (k >= kend) zero_unit(ra[q]); }
      }
    }
#pragma unroll
    for (int q = 0; q < UB; ++q) {
      const int u = tid + q * 256;
      const int c = u / (BK / EPU), kq = (u % (BK / EPU)) * EPU;
      int vc = v0 + c;
      vc = vc < a.m ? vc : a.m - 1;
      const int64_t k = k0 + kq;
      const int64_t kc = (!MASK || k + EPU <= a.ldx) ? k : 0;
      rb[q] = *reinterpret_cast<const Unit *>(X + (int64_t)vc * a.ldx + kc);
      if constexpr (MASK) {
#pragma unroll
        for (int e = 0; e < EPU; ++e)
          if (k + e >= kend) rb[q].e[e] = zero_of(T{});
      }
    }
  };
  auto load_tiles = [&](int64_t k0, Unit (&ra)[UA], Unit (&rb)[UB]) {
    if (k0 + BK > kend) load_tiles_as(k0, ra, rb, std::true_type{});       // (wave-uniform: the last K step only)
    else load_tiles_as(k0, ra, rb, std::false_type{});
  };
  auto put = [](R *re, R *im, int at, const T &v) {          // one element into its plane(s)
    if constexpr (CPLX) { re[at] = v.re; im[at] = v.im; } else { re[at] = v; (void)im; }
  };
  auto store_tiles = [&](int buf, const Unit (&ra)[UA], const Unit (&rb)[UB]) {
    R *are = ldsA[buf][0], *aim = ldsA[buf][NP - 1], *bre = ldsB[buf][0], *bim = ldsB[buf][NP - 1];
#pragma unroll
    for (int q = 0; q < UA; ++q) {
      const int u = tid + q * 256;
      if (A_KC) {
        const int r = u / (BK / EPU), kq = (u % (BK / EPU)) * EPU;
#pragma unroll
        for (int e = 0; e < EPU; ++e) put(are, aim, r * SK + kq + e, ra[q].e[e]);
      } else {
        const int kk = u / (MR / EPU), rq = (u % (MR / EPU)) * EPU;
#pragma unroll
        for (int e = 0; e < EPU; ++e) put(are, aim, (rq + e) * SK + kk, ra[q].e[e]);
      }
    }
#pragma unroll
    for (int q = 0; q < UB; ++q) {
      const int u = tid + q * 256;
      const int c = u / (BK / EPU), kq = (u % (BK / EPU)) * EPU;
#pragma unroll
      for (int e = 0; e < EPU; ++e) put(bre, bim, c * SK + kq + e, rb[q].e[e]);
    }
  };
  V acc[NP][TV][TR];                                 // [plane][vector tile][row tile]
#pragma unroll
  for (int pl = 0; pl < NP; ++pl)
#pragma unroll
    for (int tv = 0; tv < TV; ++tv)
#pragma unroll
      for (int tr = 0; tr < TR; ++tr)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[pl][tv][tr][r] = (R)0;
  const int fi = lane & 15, fk = lane >> 4;
  const int row0 = wm * 16 * TR, vec0 = wn * 16 * TV;
  auto compute = [&](int buf) {
    const R *are = ldsA[buf][0], *aim = ldsA[buf][NP - 1], *bre = ldsB[buf][0], *bim = ldsB[buf][NP - 1];
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      R xr[TV], xi[TV], ar[TR], ai[TR];
#pragma unroll
      for (int t = 0; t < TV; ++t) {
        xr[t] = bre[(vec0 + t * 16 + fi) * SK + ks * 4 + fk];
        if constexpr (CPLX) xi[t] = bim[(vec0 + t * 16 + fi) * SK + ks * 4 + fk];
      }
#pragma unroll
      for (int t = 0; t < TR; ++t) {
        ar[t] = are[(row0 + t * 16 + fi) * SK + ks * 4 + fk];
        if constexpr (CPLX) ai[t] = aim[(row0 + t * 16 + fi) * SK + ks * 4 + fk];
      }
#pragma unroll
      for (int tv = 0; tv < TV; ++tv)
#pragma unroll
        for (int tr = 0; tr < TR; ++tr) {
          acc[0][tv][tr] = mfma16(xr[tv], ar[tr], acc[0][tv][tr]);
          if constexpr (CPLX) {
            // op(A) = conj(A) for CONJ: (ar -/+ i ai)(xr + i xi)
            acc[0][tv][tr] = mfma16(xi[tv], CONJ ? ai[tr] : -ai[tr], acc[0][tv][tr]);
            acc[1][tv][tr] = mfma16(xi[tv], ar[tr], acc[1][tv][tr]);
            acc[1][tv][tr] = mfma16(xr[tv], CONJ ? -ai[tr] : ai[tr], acc[1][tv][tr]);
          }
        }
    }
  };
  // K steps in pairs: at step t the LDS buffer t % 2 holds tile t, one register set holds tile t + 1 and the other
  // receives tile t + 2 -- a global load has a whole step, its MFMAs and a barrier to arrive before it is needed
  // (with one set the matrix cores of this kernel sat idle 57 % of the time behind s_waitcnt)
  const int64_t nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  constexpr bool DEEP = !(CPLX && sizeof(R) == 8);   // (complex double: two sets of four 16-byte units per tile spill)
  if constexpr (DEEP) {
    if (nk > 0) {
      load_tiles(kbeg, ra0, rb0);
      store_tiles(0, ra0, rb0);
      if (nk > 1) load_tiles(kbeg + BK, ra1, rb1);
    }
    __syncthreads();
    for (int64_t t = 0; t < nk; t += 2) {
      if (t + 2 < nk) load_tiles(kbeg + (t + 2) * BK, ra0, rb0);
      compute(0);
      if (t + 1 < nk) store_tiles(1, ra1, rb1);
      __syncthreads();
      if (t + 1 >= nk) break;
      if (t + 3 < nk) load_tiles(kbeg + (t + 3) * BK, ra1, rb1);
      compute(1);
      if (t + 2 < nk) store_tiles(0, ra0, rb0);
      __syncthreads();
    }
  } else {
    int buf = 0;
    if (nk > 0) { load_tiles(kbeg, ra0, rb0); store_tiles(0, ra0, rb0); }
    __syncthreads();
    for (int64_t t = 0; t < nk; ++t) {
      const bool more = t + 1 < nk;
      if (more) load_tiles(kbeg + (t + 1) * BK, ra0, rb0);
      if (buf == 0) compute(0); else compute(1);
      if (more) store_tiles(buf ^ 1, ra0, rb0);
      __syncthreads();
      buf ^= 1;
    }
  }
  // D[v][i]: column (lane % 16) = output row; the vector is the D row: 4 (lane / 16) + reg for the f32 shape,
  // (lane / 16) + 4 reg for v_mfma_f64_16x16x4_f64 (its C/D map differs from every other shape's)
  T *__restrict__ Y = (T *)a.Y;
#pragma unroll
  for (int tv = 0; tv < TV; ++tv)
#pragma unroll
    for (int tr = 0; tr < TR; ++tr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int v = v0 + vec0 + tv * 16 + (sizeof(R) == 8 ? (lane >> 4) + 4 * r : 4 * (lane >> 4) + r);
        const int64_t i = i0 + row0 + tr * 16 + (lane & 15);
        if (v < a.m && i < a.ny) {
          T y;
          if constexpr (CPLX) y = T{acc[0][tv][tr][r], acc[1][tv][tr][r]};
          else y = acc[0][tv][tr][r];
          if (part) part[((int64_t)blockIdx.z * a.m + v) * a.ny + i] = y;     // [split][vector][row], summed by dense_splitk_reduce_t
          else Y[i + (int64_t)v * a.ldy] = r1_apply<T>(a, y, i, v);
        }
      }
}

static int env_int_d(const char *name, int dflt);

__device__ __forceinline__ double sum2(double x, double y) { return x + y; }
__device__ __forceinline__ float sum2(float x, float y) { return x + y; }
__device__ __forceinline__ c32 sum2(c32 x, c32 y) { return c32{x.re + y.re, x.im + y.im}; }
__device__ __forceinline__ c64 sum2(c64 x, c64 y) { return c64{x.re + y.re, x.im + y.im}; }

// the K splits of dense_mfma16_kernel summed in a fixed order, the rank-one epilogue applied
template <typename T>
__global__ __launch_bounds__(256) void dense_splitk_reduce_t(const T *__restrict__ part, int splits, int64_t ny, int m,
                                                             T *__restrict__ Y, int64_t ldy, DenseArgs a) {
  const int v = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ny; i += stride) {
    T s = part[(int64_t)v * ny + i];
    for (int z = 1; z < splits; ++z) s = sum2(s, part[((int64_t)z * m + v) * ny + i]);
    Y[i + (int64_t)v * ldy] = r1_apply<T>(a, s, i, v);
  }
}

template <typename T, typename R, bool CPLX, int TR, int TV>
static int launch_mfma16(const DenseArgs &a) {
  Context &c = ctx();
  const int64_t bx = (a.ny + 32 * TR - 1) / (32 * TR), by = (a.m + 32 * TV - 1) / (32 * TV);
  // K splits until every CU has several workgroups (RLH_DENSE_WG_PER_CU, as the float32 kernels), at least 32 K steps each
  static const int target = env_int_d("RLH_DENSE_WG_PER_CU", 8);
  int64_t splits = ((int64_t)c.num_cu * target + bx * by - 1) / (bx * by);
  const int64_t max_by_k = (a.nx + 16 * 32 - 1) / (16 * 32);
  if (splits > max_by_k) splits = max_by_k;
  if (splits > 16) splits = 16;
  while (splits > 1 && (size_t)splits * a.m * a.ny * sizeof(T) > kWorkspaceBytes) --splits;
  if (splits < 1) splits = 1;
  int64_t kchunk = ((a.nx + splits - 1) / splits + 15) / 16 * 16;
  splits = (a.nx + kchunk - 1) / kchunk;
  if (splits < 1) { splits = 1; kchunk = 16; }
  T *part = splits > 1 ? (T *)c.work : nullptr;
  dim3 grid((unsigned)bx, (unsigned)by, (unsigned)splits);
  if (a.a_kcontig) {
    if (a.conj_a) hipLaunchKernelGGL((dense_mfma16_kernel<T, R, CPLX, true, true, TR, TV>), grid, dim3(256), 0, c.stream, a, part, kchunk);
    else hipLaunchKernelGGL((dense_mfma16_kernel<T, R, CPLX, true, false, TR, TV>), grid, dim3(256), 0, c.stream, a, part, kchunk);
  } else {
    if (a.conj_a) hipLaunchKernelGGL((dense_mfma16_kernel<T, R, CPLX, false, true, TR, TV>), grid, dim3(256), 0, c.stream, a, part, kchunk);
    else hipLaunchKernelGGL((dense_mfma16_kernel<T, R, CPLX, false, false, TR, TV>), grid, dim3(256), 0, c.stream, a, part, kchunk);
  }
  RLH_HIP(hipGetLastError());
  if (splits > 1) {
    int64_t nb = (a.ny + 255) / 256;
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL((dense_splitk_reduce_t<T>), dim3((unsigned)nb, (unsigned)a.m), dim3(256), 0, c.stream, part, (int)splits, a.ny,
                       a.m, (T *)a.Y, a.ldy, a);
    RLH_HIP(hipGetLastError());
  }
  return 0;
}

static int env_int_d(const char *name, int dflt) {
  const char *e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

template <int MR, int BN>
static int launch_mfma(const DenseArgs &a) {
  Context &c = ctx();
  const int64_t bx = (a.ny + MR - 1) / MR, by = (a.m + BN - 1) / BN;
  // split K until there are a few workgroups per CU (the M x m output alone gives too few)
  static const int target = env_int_d("RLH_DENSE_WG_PER_CU", 8);
  int64_t splits = ((int64_t)c.num_cu * target + bx * by - 1) / (bx * by);
  const int64_t max_by_k = (a.nx + 32 * 16 - 1) / (32 * 16);           // at least 16 K steps per split
  if (splits > max_by_k) splits = max_by_k;
  if (splits > 16) splits = 16;
  while (splits > 1 && (size_t)splits * a.m * a.ny * sizeof(float) > kWorkspaceBytes) --splits;
  if (splits < 1) splits = 1;
  int64_t kchunk = ((a.nx + splits - 1) / splits + 31) / 32 * 32;
  splits = (a.nx + kchunk - 1) / kchunk;
  if (splits < 1) { splits = 1; kchunk = 32; }
  float *part = splits > 1 ? (float *)c.work : nullptr;
  dim3 grid((unsigned)bx, (unsigned)by, (unsigned)splits);
  if (a.a_kcontig)
    hipLaunchKernelGGL((dense_mfma_f32_kernel<MR, BN, true>), grid, dim3(256), 0, c.stream, a, part, kchunk);
  else
    hipLaunchKernelGGL((dense_mfma_f32_kernel<MR, BN, false>), grid, dim3(256), 0, c.stream, a, part, kchunk);
  RLH_HIP(hipGetLastError());
  if (splits > 1) {
    int64_t nb = (a.ny + 255) / 256;
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(dense_splitk_reduce, dim3((unsigned)nb, (unsigned)a.m), dim3(256), 0, c.stream, part,
                       (int)splits, a.ny, a.m, (float *)a.Y, a.ldy, a);
    RLH_HIP(hipGetLastError());
  }
  return 0;
}

template <int BN, int BK>
static int launch_mfma2(const DenseArgs &a) {
  Context &c = ctx();
  constexpr int MR = 128;
  const int64_t bx = (a.ny + MR - 1) / MR, by = (a.m + BN - 1) / BN;
  // split K until every CU has a few workgroups (the M x m output alone gives too few): BK = 16 leaves
  // room for four resident workgroups per CU, BK = 32 for two
  const int target = env_int_d("RLH_DENSE_WG_PER_CU", BK == 16 ? 8 : 8);
  int64_t splits = ((int64_t)c.num_cu * target + bx * by - 1) / (bx * by);
  const int64_t max_by_k = (a.nx + 32 * 16 - 1) / (32 * 16);           // at least 512 k per split
  if (splits > max_by_k) splits = max_by_k;
  if (splits > 16) splits = 16;
  while (splits > 1 && (size_t)splits * a.m * a.ny * sizeof(float) > kWorkspaceBytes) --splits;
  if (splits < 1) splits = 1;
  int64_t kchunk = ((a.nx + splits - 1) / splits + 31) / 32 * 32;
  splits = (a.nx + kchunk - 1) / kchunk;
  if (splits < 1) { splits = 1; kchunk = 32; }
  float *part = splits > 1 ? (float *)c.work : nullptr;
  dim3 grid((unsigned)bx, (unsigned)by, (unsigned)splits);
  if (a.a_kcontig)
    hipLaunchKernelGGL((dense_mfma2_f32_kernel<BN, BK, true>), grid, dim3(256), 0, c.stream, a, part, kchunk);
  else
    hipLaunchKernelGGL((dense_mfma2_f32_kernel<BN, BK, false>), grid, dim3(256), 0, c.stream, a, part, kchunk);
  RLH_HIP(hipGetLastError());
  if (splits > 1) {
    int64_t nb = (a.ny + 255) / 256;
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(dense_splitk_reduce, dim3((unsigned)nb, (unsigned)a.m), dim3(256), 0, c.stream, part,
                       (int)splits, a.ny, a.m, (float *)a.Y, a.ldy, a);
    RLH_HIP(hipGetLastError());
  }
  return 0;
}

template <int DT>
static int dense_impl(const DenseArgs &a) {
  using T = typename DType<DT>::T;
  Context &c = ctx();
  bool mfma_ok = false;
  if constexpr (DT == RLH_S) {
    mfma_ok = ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.X)) & 15u) == 0 &&
              (a.lda % 4 == 0) && (a.ldx % 4 == 0) && a.lda >= 4 && a.ldx >= 4 && a.nx > 0;
  }
  if constexpr (DT == RLH_S) {
    // measured at 20000 x 20000 x 128, sustained: tiles with k contiguous 0.92 ms (first kernel) / 0.97 ms (second),
    // column-major tiles 1.02 ms / 0.93 ms (RLH_DENSE_KERNEL=1|2 forces one: tunable)
    const int kern = env_int_d("RLH_DENSE_KERNEL", a.a_kcontig ? 1 : 2);
    if (mfma_ok && kern == 2) {
      // tunable: 16 or 32.  Sustained rates (40 back-to-back calls, profiles/r02_gemm_shapes.txt): column-major
      // tiles 110 TF with BK = 32 against 105 with 16 at 20000 x 20000 x 128 (119 against 114 at 40000^2)
      const int bk = env_int_d("RLH_DENSE_BK", a.a_kcontig ? 16 : 32);
      if (a.m > 64) return bk == 32 ? launch_mfma2<128, 32>(a) : launch_mfma2<128, 16>(a);
      if (a.m > 32) return bk == 32 ? launch_mfma2<64, 32>(a) : launch_mfma2<64, 16>(a);
      return bk == 32 ? launch_mfma2<32, 32>(a) : launch_mfma2<32, 16>(a);
    }
    if (mfma_ok) {
      static const int mr = env_int_d("RLH_DENSE_MR", 128);
      if (a.m > 64) return mr == 64 ? launch_mfma<64, 128>(a) : launch_mfma<128, 128>(a);
      if (a.m > 32) return launch_mfma<128, 64>(a);
      return launch_mfma<128, 32>(a);
    }
  }
  if constexpr (DT != RLH_S) {
    // double / complex: the matrix cores whenever 16-byte pieces can be loaded (RLH_DENSE_VALU=1 forces the VALU kernel)
    constexpr int64_t EPU = 16 / (int64_t)sizeof(T);
    const bool ok = ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.X)) & 15u) == 0 && (a.lda % EPU == 0) &&
                    (a.ldx % EPU == 0) && a.lda >= EPU && a.ldx >= EPU && a.nx > 0 && !env_int_d("RLH_DENSE_VALU", 0);
    if (ok) {
      // 64 rows x 128 vectors per workgroup where there are more than 64 vectors -- every element of A is read ONCE -- with K
      // split over gridDim.z until the chip is evenly filled (launch_mfma16): 20000 x 20000 x 128 fp64 in 1.91 / 1.95 ms =
      // 53.7 / 52.5 TF, 62500 x 40000 53.6 / 53.2 TF (64 x 64 tiles, which read A twice: 49.7 / 47.3 TF with the K splits,
      // 42.7 / 41.1 TF without: 626 workgroups on 512 resident slots left the second round a fifth full; 128 x 64 tiles
      // 51.8 / 49.4 TF -- and two of their instantiations spilled: dropped).  RLH_DENSE_D_TILE=22: 64 x 64 tiles regardless.
      if constexpr (DT == RLH_D) {
        const int tile = env_int_d("RLH_DENSE_D_TILE", 24);
        if (tile == 24 && a.m > 64) return launch_mfma16<double, double, false, 2, 4>(a);
        return launch_mfma16<double, double, false, 2, 2>(a);
      }
      if constexpr (DT == RLH_Z) return launch_mfma16<c64, double, true, 2, 2>(a);
      if constexpr (DT == RLH_C) return launch_mfma16<c32, float, true, 2, 2>(a);
    }
  }
  dim3 grid((unsigned)((a.ny + 63) / 64), (unsigned)((a.m + 63) / 64));
  if (a.conj_a)
    hipLaunchKernelGGL((dense_valu_kernel<T, true>), grid, dim3(256), 0, c.stream, a);
  else
    hipLaunchKernelGGL((dense_valu_kernel<T, false>), grid, dim3(256), 0, c.stream, a);
  RLH_HIP(hipGetLastError());
  return 0;
}

}  // namespace rlh

using namespace rlh;

extern "C" int rlh_dense_apply_r1(int dtype, int64_t M, int64_t N, const void *A, int64_t lda, int order, int transp,
                                  int64_t m, const void *X, int64_t ldx, void *Y, int64_t ldy, const void *d_u,
                                  const void *d_c);

extern "C" int rlh_dense_apply(int dtype, int64_t M, int64_t N, const void *A, int64_t lda, int order, int transp,
                               int64_t m, const void *X, int64_t ldx, void *Y, int64_t ldy) {
  return rlh_dense_apply_r1(dtype, M, N, A, lda, order, transp, m, X, ldx, Y, ldy, nullptr, nullptr);
}

extern "C" int rlh_dense_apply_r1(int dtype, int64_t M, int64_t N, const void *A, int64_t lda, int order, int transp,
                                  int64_t m, const void *X, int64_t ldx, void *Y, int64_t ldy, const void *d_u,
                                  const void *d_c) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(d_c || !d_u, "rlh_dense_apply_r1: a vector u without coefficients c");
  RLH_REQUIRE(dtype_valid(dtype), "rlh_dense_apply: unknown dtype %d", dtype);
  RLH_REQUIRE(M >= 0 && N >= 0 && m >= 0, "rlh_dense_apply: negative size");
  RLH_REQUIRE(order == 0 || order == 1, "rlh_dense_apply: order must be 0 (row-major) or 1 (column-major)");
  RLH_REQUIRE(transp == 0 || transp == 1, "rlh_dense_apply: transp must be 0 or 1");
  const int64_t ny = transp ? N : M, nx = transp ? M : N;
  if (m == 0 || ny == 0) return 0;
  RLH_REQUIRE(Y && (nx == 0 || (A && X)), "rlh_dense_apply: null pointer");
  RLH_REQUIRE(lda >= (order == 0 ? N : M), "rlh_dense_apply: lda too small");
  RLH_REQUIRE(ldx >= nx && ldy >= ny, "rlh_dense_apply: Matrix and vectors dimensions incompatible");
  RLH_REQUIRE(m <= 65535 * 64, "rlh_dense_apply: too many vectors");
  DenseArgs a;
  a.A = A; a.lda = lda;
  // Op(i,k): transp 0 -> A(i,k); transp 1 -> conj(A(k,i)).  Row-major A(i,k) = A[i*lda+k].
  a.a_kcontig = ((order == 0) != (transp == 1)) ? 1 : 0;
  a.conj_a = (transp == 1 && (dtype == RLH_C || dtype == RLH_Z)) ? 1 : 0;
  a.ny = ny; a.nx = nx; a.X = X; a.ldx = ldx; a.Y = Y; a.ldy = ldy; a.m = (int)m;
  a.r1_u = d_u; a.r1_c = d_c;
  switch (dtype) {
    case RLH_S: return dense_impl<RLH_S>(a);
    case RLH_D: return dense_impl<RLH_D>(a);
    case RLH_C: return dense_impl<RLH_C>(a);
    case RLH_Z: return dense_impl<RLH_Z>(a);
  }
  return 1;
}

// K1 Gram (X.dot(Y)), K2 dots, K2t transposed dots for gfx950.
//
// Gram: G = Y^H X is a contraction over the n rows of two tall-skinny blocks.
// Each 256-thread workgroup streams 512-byte column segments (coalesced 16-byte
// loads) of a row chunk into an LDS tile [vcol][row], the four waves then run
// v_mfma_{f64,f32}_16x16x4 over disjoint row quarters of the tile with
// conflict-free ds_read (column stride == 2 mod 32), accumulating a
// (PI*16) x (PJ*16) panel of G in registers.  Loads of chunk t+1 are in flight
// while chunk t is multiplied.  Per-workgroup partials are combined by a second,
// deterministic kernel (fixed summation order => bitwise reproducible results).
// Complex blocks are viewed as real blocks with twice the columns (re, im
// de-interleaved on the way into LDS); the finalize kernel recombines
// conj(y)*x = (RR + II) + i (RI - IR).
//
// Replaces: cublas?gemm(ConjTrans, NoTrans) + cudaMalloc/cudaMemcpy/cudaFree per call
// (raleigh/algebra/dense_cublas.py:245-269), m blocking cublas?dot calls
// (dense_cublas.py:233-243) and gemmBatched (dense_cublas.py:175-221).
#include <stdlib.h>

#include "common.h"

namespace rlh {

template <typename R> struct Mfma16;
template <> struct Mfma16<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4*reg
  static __device__ __forceinline__ int out_row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mfma16<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4)*4 + reg
  static __device__ __forceinline__ int out_row(int lane, int reg) { return (lane >> 4) * 4 + reg; }
};

// A window made of several blocks side by side (rlh_gram_multi: [AX | X]^H X in one pass):
// segment k covers the columns [c0, next c0) of the concatenated window.
struct GramSeg { const void *p; int64_t ld; int c0; };
constexpr int kGramSegs = 4;

struct GramArgs {
  const void *X, *Y;
  int64_t ldx, ldy;   // in elements of T
  int64_t n;          // rows
  int mx, my;         // columns of T
  int same;           // X and Y are the same window: read once
  int npj;            // number of X panels
  int64_t nchunks;
  void *partials;     // [npanels][gridDim.x][VY][VX] reals
  int nxs, nys;       // segments of the X / Y window (MULTI kernels only)
  int nt;             // non-temporal loads of the blocks (each byte is read once per launch)
  int xal;            // streaming kernel: the X window IS the columns [xal, xal + mx) of the Y window (or -1)
  GramSeg xs[kGramSegs], ys[kGramSegs];
};

// (three resident workgroups per CU measured faster in sustained back-to-back use than the four that
// __launch_bounds__(256, 4) gives: 5.97 vs 5.53 TB/s)
// MODE 0: general (unaligned layouts, several panels of a self-Gram): loads predicated per piece,
//         one chunk in flight.  MODE 3 / 4: the same loop on row chunks twice / four times as long
//         (1 / 2 KiB per column piece instead of 512 bytes: longer DRAM bursts, +2.5 % on the
//         two-operand fp64 Gram) where the LDS tile and the loads of a chunk still fit.
// MODE 1 / 2: aligned two-operand Gram / aligned single-panel self-Gram: the panel shape is known
//         at compile time, so the loads of a chunk are straight-line code and the loop keeps TWO
//         chunks in flight in two register sets (the compiler emits counted vmcnt waits only for
//         straight-line load groups: with a predicate per load it waits for vmcnt(0)).
// MULTI: the windows are concatenations of up to kGramSegs blocks (general loop, MODE 0, only).
// QUAD (general loop, MODE 0, aligned, PI = 4, PJ = 2, 512 threads): the workgroup's panel is 128 x 128
// real columns -- a complex128 block of 64 vectors against another -- and each of the EIGHT waves owns
// a 64 x 32 part of it (4 x 2 tiles, 64 accumulator registers) over ALL rows of a chunk instead of the
// whole panel over a quarter of the rows.  Wide
// windows otherwise take several 64 x 64 panels, every one of which re-reads its column ranges of both
// operands: at m = k = 64 complex128 that is twice the bytes, and the MFMA pipes idle while the one
// resident workgroup per CU stages (measured 2.08 ms = 40 % of the fp64 MFMA rate, 25 % of the HBM rate;
// the two bounds are 0.83 and 0.82 ms).  Chunks are 256-byte column pieces so that two workgroups fit a CU.
template <int DT, int PI, int PJ, bool ALIGNED, int MODE, bool MULTI = false, bool QUAD = false, bool SYMS = false>
__global__ __launch_bounds__(QUAD ? 512 : 256, QUAD ? 2 : ((MODE == 1 && PI <= 2 && PJ <= 2) ? 3 : 1)) void gram_kernel(GramArgs a) {
  using T = typename DType<DT>::T;
  using R = typename DType<DT>::R;
  using M = Mfma16<R>;
  using acc_t = typename M::acc_t;
  constexpr bool CPLX = DType<DT>::cplx;
  constexpr int NC = CPLX ? 2 : 1;               // reals per element
  constexpr int RM = MODE == 3 ? 2 : (MODE == 4 ? 4 : 1);
  // rows per chunk: 512-byte column pieces, 1 KiB (MODE 3), 2 KiB (MODE 4); QUAD: 256 bytes for the
  // 128 x 128 panel, 512 bytes for the 64 x 64 one (two workgroups per CU either way)
  constexpr int ROWS = QUAD ? (PI * PJ >= 8 ? 256 : 512) / (int)sizeof(R) : RM * 512 / (int)sizeof(R);
  constexpr int RPU = 16 / (int)sizeof(R);       // reals per 16-byte unit
  constexpr int UPC = ROWS * NC / RPU;           // units per T-column per chunk
  constexpr int NT = QUAD ? 512 : 256;           // threads
  constexpr int WQI = QUAD ? 2 : 1, WQJ = QUAD ? 4 : 1;   // waves per panel side
  constexpr int VY = PI * 16 * WQI, VX = PJ * 16 * WQJ;   // real (virtual) columns per panel
  static_assert(!QUAD || (MODE == 0 && ((PI == 4 && PJ == 2) || (PI == 2 && PJ == 1))), "quadrant panels are 128 x 128 or 64 x 64");
  constexpr int CY = VY / NC, CX = VX / NC;      // T columns per panel
  constexpr int S = ROWS + 2;                    // LDS column stride in reals (== 2 mod 32)
  constexpr int UNITS_MAX = (CY + CX) * UPC;
  constexpr int UPT = (UNITS_MAX + NT - 1) / NT; // units per thread

  __shared__ __attribute__((aligned(16))) R lds[(VY + VX) * S];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int panel = blockIdx.y;
  const int pi = panel / a.npj, pj = panel % a.npj;
  // (SYMS: the instantiation for the self-Gram of ONE square quadrant panel)
  const bool same_panel = SYMS ? true : (MODE == 2 ? true : (MODE == 1 ? false : (a.same && (pi == pj) && (PI == PJ))));
  const int cy0 = pi * CY, cx0 = pj * CX;        // first T column of the panels
  const int units_y = same_panel ? 0 : CY * UPC;
  const int nunits = units_y + CX * UPC;
  R *ldsY = same_panel ? (lds + VY * S) : lds;
  R *ldsX = lds + VY * S;

  const R *Xr = reinterpret_cast<const R *>(a.X);
  const R *Yr = reinterpret_cast<const R *>(a.Y);

  typedef R vec_t __attribute__((ext_vector_type(RPU)));   // 16 bytes: one global_load_dwordx4
  vec_t regs[UPT];

  // Thread t handles the 16-byte piece k = t % UPC of column (t / UPC) + q * CPQ for q = 0, 1, ...:
  // the first UY pieces belong to the Y panel (none when the panel is shared with X), the rest to
  // the X panel.  Columns past the window are clamped to a valid one: they only feed Gram
  // entries that are never written out, so no zero fill is needed.
  constexpr int CPQ = NT / UPC;                  // columns covered by one pass of the threads
  constexpr int UYQ = CY / CPQ, UXQ = CX / CPQ;  // passes over the Y / X panel
  static_assert(NT % UPC == 0 && CY % CPQ == 0 && CX % CPQ == 0, "panel / thread mapping");
  const int tcol = tid / UPC, tk = tid % UPC;
  const int uy = same_panel ? 0 : UYQ;           // wave-uniform

  // (MULTI: the row-independent part of every piece address -- segment, column -- is fixed per
  // thread and found once, in front of the chunk loop)
  const R *pbase[MULTI ? UPT : 1];
  if constexpr (MULTI) {
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
      const bool isY = q < uy;
      int gcol = (isY ? cy0 + q * CPQ : cx0 + (q - uy) * CPQ) + tcol;
      const int mcols = isY ? a.my : a.mx;
      gcol = gcol < mcols ? gcol : mcols - 1;
      const GramSeg *segs = isY ? a.ys : a.xs;
      const int ns = isY ? a.nys : a.nxs;
      const void *p = segs[0].p;
      int64_t ld = segs[0].ld;
      int c0 = 0;
#pragma unroll
      for (int k = 1; k < kGramSegs; ++k)
        if (k < ns && gcol >= segs[k].c0) { p = segs[k].p; ld = segs[k].ld; c0 = segs[k].c0; }
      pbase[q] = reinterpret_cast<const R *>(p) + (int64_t)(gcol - c0) * ld * NC + tk * RPU;
    }
  }
  auto piece_ptr = [&](int q, int64_t row0) -> const R * {
    if constexpr (MULTI) {
      return pbase[q] + row0 * NC;
    } else {
      const bool isY = q < uy;
      int gcol = (isY ? cy0 + q * CPQ : cx0 + (q - uy) * CPQ) + tcol;
      const int mcols = isY ? a.my : a.mx;
      gcol = gcol < mcols ? gcol : mcols - 1;
      return (isY ? Yr : Xr) + ((int64_t)gcol * (isY ? a.ldy : a.ldx) + row0) * NC + tk * RPU;
    }
  };

  // (plain loads: a run-time choice of the non-temporal hint would put every load behind a branch; the streaming
  // kernel below, which serves the roofline shapes, takes the hint as a template parameter)
  auto load_piece = [&](const R *p) -> vec_t { return *reinterpret_cast<const vec_t *>(p); };
  auto load_chunk = [&](int64_t chunk) {
    const int64_t row0 = chunk * ROWS;
    const bool full = ALIGNED && (row0 + ROWS <= a.n);          // wave-uniform
    if (full) {
#pragma unroll
      for (int q = 0; q < UPT; ++q)
        if (q < uy + UXQ) regs[q] = load_piece(piece_ptr(q, row0));
    } else {                                                     // last chunk / unaligned layout
      const int64_t rend = (a.n - row0) * NC - (int64_t)tk * RPU;   // valid reals from this piece on
#pragma unroll
      for (int q = 0; q < UPT; ++q) {
        if (q < uy + UXQ) {
          const R *p = piece_ptr(q, row0);
          vec_t val;
#pragma unroll
          for (int e = 0; e < RPU; ++e) val[e] = (e < rend) ? p[e] : (R)0;
          regs[q] = val;
        }
      }
    }
  };

  auto store_chunk = [&](const vec_t (&regs)[UPT]) {
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
      if (q < uy + UXQ) {
        const bool isY = q < uy;
        const int col = (isY ? q : q - uy) * CPQ + tcol, k = tk;
        R *dst = isY ? ldsY : ldsX;
        if constexpr (!CPLX) {
          R *d = dst + col * S + k * RPU;
          if constexpr (sizeof(R) == 8) {
            *reinterpret_cast<vec_t *>(d) = regs[q];     // ds_write_b128 (S*8 % 16 == 0)
          } else {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 lo = {(float)regs[q][0], (float)regs[q][1]};
            f2 hi = {(float)regs[q][RPU > 2 ? 2 : 0], (float)regs[q][RPU > 3 ? 3 : 1]};
            reinterpret_cast<f2 *>(d)[0] = lo;           // 2 x ds_write_b64 (S*4 % 8 == 0)
            reinterpret_cast<f2 *>(d)[1] = hi;
          }
        } else {
          // unit holds RPU/2 complex rows: de-interleave into the re / im virtual columns
          constexpr int CR = RPU / 2;                    // complex rows per unit (1 z, 2 c)
          R *dre = dst + (2 * col) * S + k * CR;
          R *dim = dre + S;
          if constexpr (CR == 1) {
            dre[0] = regs[q][0];
            dim[0] = regs[q][1];
          } else {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 re = {(float)regs[q][0], (float)regs[q][RPU > 2 ? 2 : 0]};
            f2 im = {(float)regs[q][1], (float)regs[q][RPU > 3 ? 3 : 1]};
            *reinterpret_cast<f2 *>(dre) = re;
            *reinterpret_cast<f2 *>(dim) = im;
          }
        }
      }
    }
  };

  acc_t acc[PI][PJ];
#pragma unroll
  for (int i = 0; i < PI; ++i)
#pragma unroll
    for (int j = 0; j < PJ; ++j) acc[i][j] = acc_t{(R)0, (R)0, (R)0, (R)0};

  const int fr = lane & 15, fk = lane >> 4;
  // first tile row / column of this wave in the panel (QUAD: its quadrant)
  const int ti0 = QUAD ? (wave >> 2) * PI : 0, tj0 = QUAD ? (wave & 3) * PJ : 0;
  // Self-Gram of a 128 x 128 quadrant panel: only the 36 tiles on and above the diagonal are needed (gram_finalize
  // mirrors the rest), but with one 4 x 2 quadrant per wave the four waves whose quadrants lie above the diagonal do all
  // 8 of theirs and everybody waits for them.  Instead tile rows r and 7 - r (9 tiles together) go to the wave pair
  // (2 r, 2 r + 1): wave 2 r takes (r, r .. r + 4), wave 2 r + 1 the other 3 - r tiles of row r and the r + 1 tiles of
  // row 7 - r -- five or four tiles per wave instead of eight, still at most six fragment reads per k-step.
  constexpr bool SYMQ = SYMS && QUAD && PI == 4 && PJ == 2;
  constexpr bool sym = SYMQ;
  const int sr = wave >> 1, sh = wave & 1;
  auto sym_tile = [&](int t, int &ti, int &tj) {     // tile t of this wave in the symmetric assignment
    if (sh == 0) { ti = sr; tj = sr + t; }
    else if (t < 3 - sr) { ti = sr; tj = sr + 5 + t; }
    else { ti = 7 - sr; tj = 7 - sr + (t - (3 - sr)); }
  };
  const int sym_cnt = sh == 0 ? 5 : 4;
  auto mfma_phase_sym = [&]() {
    if constexpr (SYMQ) {
#pragma unroll 4
      for (int ks = 0; ks < ROWS / 4; ++ks) {
        const int row = ks * 4 + fk;
        const R fa0 = ldsX[(sr * 16 + fr) * S + row];
        const R fa1 = ldsX[((7 - sr) * 16 + fr) * S + row];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          if (t < sym_cnt) {
            int ti, tj;
            sym_tile(t, ti, tj);
            const R fb = ldsX[(tj * 16 + fr) * S + row];
            acc[t >> 1][t & 1] = M::run(ti == sr ? fa0 : fa1, fb, acc[t >> 1][t & 1]);
          }
        }
      }
    }
  };
  auto mfma_phase = [&]() {
    if (sym) { mfma_phase_sym(); return; }
    const int rbase = QUAD ? 0 : wave * (ROWS / 4);
#pragma unroll 4
    for (int ks = 0; ks < (QUAD ? ROWS / 4 : ROWS / 16); ++ks) {
      const int row = rbase + ks * 4 + fk;
      R fa[PI], fb[PJ];
#pragma unroll
      for (int i = 0; i < PI; ++i) fa[i] = ldsY[((ti0 + i) * 16 + fr) * S + row];
#pragma unroll
      for (int j = 0; j < PJ; ++j) fb[j] = ldsX[((tj0 + j) * 16 + fr) * S + row];
#pragma unroll
      for (int i = 0; i < PI; ++i)
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
          // a panel of a self-Gram is symmetric: only the tiles on and above its diagonal are
          // computed, gram_finalize mirrors the rest
          if (same_panel && ti0 + i > tj0 + j) continue;
          acc[i][j] = M::run(fa[i], fb[j], acc[i][j]);
        }
    }
  };

  if constexpr (MODE == 0 || MODE >= 3) {
    int64_t chunk = blockIdx.x;
    if (chunk < a.nchunks) load_chunk(chunk);
    for (; chunk < a.nchunks; chunk += gridDim.x) {
      __syncthreads();                 // previous MFMA phase has finished reading the tile
      store_chunk(regs);
      __syncthreads();
      const int64_t next = chunk + gridDim.x;
      if (next < a.nchunks) load_chunk(next);   // in flight during the MFMA phase
      mfma_phase();
    }
  } else {
    constexpr int NQ = (MODE == 2 ? 0 : UYQ) + UXQ;       // 16-byte loads per thread and chunk
    vec_t regs2[UPT];
    auto load_full = [&](int64_t chunk, vec_t (&r)[UPT]) {
      const int64_t row0 = chunk * ROWS;
#pragma unroll
      for (int q = 0; q < NQ; ++q) r[q] = load_piece(piece_ptr(q, row0));
    };
    const int64_t nfull = a.n / ROWS;                     // chunks with all ROWS rows
    const int64_t G = gridDim.x;
    int64_t c = blockIdx.x;
    if (c + 3 * G < nfull) {
      // (the loads in front of the loop are unconditional on this path: a conditional one would
      // make the compiler wait for both register sets at the first LDS write of the loop)
      load_full(c, regs);
      __builtin_amdgcn_sched_barrier(0);                  // keep the issue order: regs first
      load_full(c + G, regs2);
      __builtin_amdgcn_sched_barrier(0);
      for (; c + 3 * G < nfull; c += 2 * G) {             // both prefetch targets exist: no branches inside
        __syncthreads();
        store_chunk(regs);
        __syncthreads();
        load_full(c + 2 * G, regs);
        mfma_phase();
        __syncthreads();
        store_chunk(regs2);
        __syncthreads();
        load_full(c + 3 * G, regs2);
        mfma_phase();
      }
    } else {
      if (c < nfull) load_full(c, regs);
      if (c + G < nfull) load_full(c + G, regs2);
    }
    // at most three full chunks left: regs holds c, regs2 holds c + G
    if (c < nfull) {
      __syncthreads();
      store_chunk(regs);
      __syncthreads();
      if (c + 2 * G < nfull) load_full(c + 2 * G, regs);
      mfma_phase();
    }
    if (c + G < nfull) {
      __syncthreads();
      store_chunk(regs2);
      __syncthreads();
      mfma_phase();
    }
    if (c + 2 * G < nfull) {
      __syncthreads();
      store_chunk(regs);
      __syncthreads();
      mfma_phase();
    }
    // the one chunk with fewer than ROWS rows goes to the workgroup whose turn it is
    if (nfull < a.nchunks && nfull % G == blockIdx.x) {
      load_chunk(nfull);
      __syncthreads();
      store_chunk(regs);
      __syncthreads();
      mfma_phase();
    }
  }

  if constexpr (QUAD) {
    // every wave owns its quadrant: straight from the accumulators to the partials
    R *outq = reinterpret_cast<R *>(a.partials) + (int64_t)panel * (VY * VX) * gridDim.x + blockIdx.x;
    if (sym) {
#pragma unroll
      for (int t = 0; t < 5; ++t)
        if (t < sym_cnt) {
          int ti, tj;
          sym_tile(t, ti, tj);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ii = ti * 16 + M::out_row(lane, r), jj = tj * 16 + (lane & 15);
            outq[(int64_t)(ii * VX + jj) * gridDim.x] = acc[t >> 1][t & 1][r];
          }
        }
      return;
    }
#pragma unroll
    for (int i = 0; i < PI; ++i)
#pragma unroll
      for (int j = 0; j < PJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ii = (ti0 + i) * 16 + M::out_row(lane, r), jj = (tj0 + j) * 16 + (lane & 15);
          outq[(int64_t)(ii * VX + jj) * gridDim.x] = acc[i][j][r];
        }
    return;
  }
  // ---- combine the four waves in a fixed order through LDS, then write the partial
  __syncthreads();
  constexpr int TILE = 256;   // reals per 16x16 tile
  static_assert(QUAD || PI * PJ * TILE <= (VY + VX) * S, "epilogue does not fit the staging tile");
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < PI; ++i)
#pragma unroll
        for (int j = 0; j < PJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ii = i * 16 + M::out_row(lane, r), jj = j * 16 + (lane & 15);
            R *p = &lds[ii * VX + jj];
            *p = (w == 0) ? acc[i][j][r] : (*p + acc[i][j][r]);
          }
    }
    __syncthreads();
  }
  // partials are laid out [panel][entry][workgroup] so that the finalize kernel reads the
  // contributions to one entry as one contiguous run
  R *out = reinterpret_cast<R *>(a.partials) + (int64_t)panel * (VY * VX) * gridDim.x + blockIdx.x;
  for (int e = tid; e < VY * VX; e += 256) out[(int64_t)e * gridDim.x] = lds[e];
}

// ---------------------------------------------------------------- complex128, up to 64 x 64 columns, two operands
// G = Y^H X for complex128 blocks of 33 .. 64 vectors (BASELINE config 5: m = 64).  The kernel above reaches 52 TF = 66 %
// of the fp64 matrix-core rate there: every thread de-interleaves its pieces into re / im planes with sixteen scalar
// ds_write_b64 per chunk, through registers, between two workgroup barriers.  Here nothing is de-interleaved: a complex
// column of n elements IS a real vector of 2 n reals (re_0, im_0, re_1, ...), and with J the map (re, im) -> (im, -re)
//     Re G = Y_r^T X_r,     Im G = Y_r^T (J X_r)
// -- a REAL Gram of the 64 columns of Y_r against the 128 columns [X_r | J X_r] over 2 n rows: the same flops (8 n m^2),
// a 64 x 128 panel (32 accumulator registers per wave instead of 64), and J X_r never exists: its MFMA fragment is the
// fragment of X_r read at row ^ 1 with the sign of the odd rows flipped.  So the LDS tile is the memory layout itself,
// and the staging is global_load_lds_dwordx4: 16 bytes per lane straight into the LDS, no staging registers, no ds_write,
// ONE barrier per chunk, the next chunk's DMAs in flight during the whole multiply phase (two 65 KB buffers).
//   * Chunk: 32 complex rows of all 128 columns.  One DMA instruction moves 64 pieces = rows 0 .. 31 of TWO columns, which
//     land 512 bytes apart: columns c and c + 16 -- never part of the same 16-column fragment read -- share a slot, slots
//     are 1040 bytes apart (== 4 dwords mod 64 banks), so the 16 lanes x 4 rows of a ds_read_b64 fall into 64 different
//     banks per half-wave.
//   * 1024 threads, one workgroup per CU: two groups of eight waves take the two halves of a chunk's rows; within a group
//     wave u owns Y tiles {2 (u / 4), 2 (u / 4) + 1} x X tile u % 4 for both X_r and J X_r: 4 MFMAs and 4 fragment reads per
//     k-step.  The groups' accumulators are added through the LDS at the end, one partial per workgroup, summed by
//     gram_z_finalize in a fixed order (bitwise reproducible).
//   * Columns past a window's end repeat its last column (entries never written out); rows past n in the last chunk are
//     zeroed in the LDS by the lane that fetched them.
constexpr int kZgRows = 32;                  // complex rows per chunk
constexpr int kZgSlot = 1040;                // bytes per slot: two columns of 32 x 16 bytes + 16
constexpr int kZgBuf = 64 * kZgSlot;         // one chunk
struct ZGramArgs {
  const c64 *X, *Y;
  int64_t ldx, ldy, n, nchunks;
  int mx, my;
  double *partials;                          // [64 x 128 entries][gridDim.x]
};

template <int DBG>
__global__ __launch_bounds__(1024) void gram_z_dma_kernel(ZGramArgs a) {
  extern __shared__ __align__(16) char zlds[];
  typedef double acc_t __attribute__((ext_vector_type(4)));
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto dma16 = [&](const void *g, unsigned dst) {  // 16 bytes per lane to the LDS at dst + 16 lane (dst wave-uniform)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
  };
  // ---- staging plan: slots 4 wave .. 4 wave + 3; lanes 0 .. 31 the slot's first column, 32 .. 63 its second
  const c64 *gcol[4];
  const int lrow = lane & 31;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int s = wave * 4 + i, q = s >> 4, ii = s & 15;
    const int c = (2 * q + (lane >> 5)) * 16 + ii;              // column of the combined window [Y | X]
    if (c < 64) gcol[i] = a.Y + (int64_t)(c < a.my ? c : a.my - 1) * a.ldy;
    else gcol[i] = a.X + (int64_t)(c - 64 < a.mx ? c - 64 : a.mx - 1) * a.ldx;
  }
  auto issue = [&](int64_t chunk, unsigned buf) {
    int64_t row = chunk * kZgRows + lrow;
    if (row > a.n - 1) row = a.n - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr ((DBG & 1) != 0) break;
      dma16(gcol[i] + row, buf * (unsigned)kZgBuf + (unsigned)(wave * 4 + i) * (unsigned)kZgSlot);
    }
  };
  // ---- multiply plan
  const int grp = wave >> 3, u = wave & 7, yp = u >> 2, xj = u & 3;
  const int fr = lane & 15, fk = lane >> 4;
  const unsigned rbase = (unsigned)grp * 256u;                    // this group's first real row, in bytes
  const unsigned ya0 = (unsigned)(yp * 16 + fr) * (unsigned)kZgSlot + rbase, ya1 = ya0 + 512u;
  const unsigned xo = (unsigned)((2 + (xj >> 1)) * 16 + fr) * (unsigned)kZgSlot + (unsigned)(xj & 1) * 512u + rbase;
  const unsigned k8 = (unsigned)fk * 8u, k8j = (unsigned)(fk ^ 1) * 8u;
  const double sgn = (fk & 1) ? -1.0 : 1.0;
  acc_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = acc_t{0.0, 0.0, 0.0, 0.0};

  int64_t c = blockIdx.x;
  unsigned buf = 0;
  if (c < a.nchunks) issue(c, 0);
  for (; c < a.nchunks; c += gridDim.x, buf ^= 1u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's pieces of chunk c have landed
    if ((c + 1) * kZgRows > a.n && c * kZgRows + lrow >= a.n) {   // (last chunk) rows past the end: zeros
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<d2 *>(zlds + buf * (unsigned)kZgBuf + (unsigned)(wave * 4 + i) * (unsigned)kZgSlot + (unsigned)lane * 16u) = d2{0.0, 0.0};
    }
    __syncthreads();                                              // everybody's pieces; and chunk c - 1 has been multiplied
    const int64_t next = c + gridDim.x;
    if (next < a.nchunks) issue(next, buf ^ 1u);
    const char *base = zlds + buf * (unsigned)kZgBuf;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if constexpr ((DBG & 2) != 0) break;
      const unsigned ro = (unsigned)ks * 32u;
      const double fa0 = *reinterpret_cast<const double *>(base + ya0 + ro + k8);
      const double fa1 = *reinterpret_cast<const double *>(base + ya1 + ro + k8);
      const double fb = *reinterpret_cast<const double *>(base + xo + ro + k8);
      const double fj = sgn * *reinterpret_cast<const double *>(base + xo + ro + k8j);
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0, fb, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1, fb, acc[1][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0, fj, acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1, fj, acc[1][1], 0, 0, 0);
    }
  }
  // ---- the two groups' panels added in a fixed order, one partial per workgroup
  __syncthreads();
  acc_t *red = reinterpret_cast<acc_t *>(zlds);                   // [wave u][tile 0 .. 3][lane]
  if (grp == 1) {
#pragma unroll
    for (int t = 0; t < 4; ++t) red[(u * 4 + t) * 64 + lane] = acc[t >> 1][t & 1];
  }
  __syncthreads();
  if (grp == 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const acc_t o = red[(u * 4 + t) * 64 + lane];
      const acc_t v = acc[t >> 1][t & 1];
      const int ti = 2 * yp + (t >> 1);                            // Y tile
      const int jj = (t & 1) * 64 + xj * 16 + (lane & 15);          // column of [Re | Im]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = ti * 16 + (lane >> 4) + 4 * r;
        a.partials[(int64_t)(ii * 128 + jj) * gridDim.x + blockIdx.x] = v[r] + o[r];
      }
    }
  }
}

// out[i][j] = sum over the workgroups of (Re, Im) partials, fixed order: one wave per entry
__global__ __launch_bounds__(256) void gram_z_finalize(const double *partials, int nb, int my, int mx, c64 *out) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= my * mx) return;
  const int i = e / mx, j = e % mx;
  const double *pr = partials + (int64_t)(i * 128 + j) * nb, *pi = partials + (int64_t)(i * 128 + 64 + j) * nb;
  double sr = 0.0, si = 0.0;
  for (int b = lane; b < nb; b += 64) { sr += pr[b]; si += pi[b]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sr += __shfl_xor(sr, off); si += __shfl_xor(si, off); }
  if (lane == 0) out[e] = c64{sr, si};
}

static int gram_z_dma_launch(int64_t n, int64_t mx, const void *X, int64_t ldx, int64_t my, const void *Y, int64_t ldy,
                             void *d_out) {
  Context &c = ctx();
  ZGramArgs a;
  a.X = (const c64 *)X; a.Y = (const c64 *)Y; a.ldx = ldx; a.ldy = ldy; a.n = n; a.mx = (int)mx; a.my = (int)my;
  a.nchunks = (n + kZgRows - 1) / kZgRows;
  a.partials = (double *)c.work;
  int64_t nb = c.num_cu;
  if (nb > a.nchunks) nb = a.nchunks;
  RLH_REQUIRE((size_t)nb * 64 * 128 * sizeof(double) <= kWorkspaceBytes, "rlh_gram: reduction workspace");
#define RLH_ZG(D_)                                                                                                  \
  do {                                                                                                              \
    static bool attr = false;                                                                                       \
    if (!attr) {                                                                                                    \
      RLH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gram_z_dma_kernel<D_>),                           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kZgBuf));                         \
      attr = true;                                                                                                  \
    }                                                                                                               \
    hipLaunchKernelGGL(gram_z_dma_kernel<D_>, dim3((unsigned)nb), dim3(1024), 2 * kZgBuf, c.stream, a);             \
  } while (0)
  switch (getenv("RLH_GRAM_ZDBG") ? atoi(getenv("RLH_GRAM_ZDBG")) : 0) {   // (timing-only builds: 1 no DMA, 2 no multiply)
    case 1: RLH_ZG(1); break;
    case 2: RLH_ZG(2); break;
    case 3: RLH_ZG(3); break;
    default: RLH_ZG(0); break;
  }
#undef RLH_ZG
  RLH_HIP(hipGetLastError());
  const int total = (int)(my * mx);
  hipLaunchKernelGGL(gram_z_finalize, dim3((total + 3) / 4), dim3(256), 0, c.stream, (const double *)c.work, (int)nb, (int)my,
                     (int)mx, (c64 *)d_out);
  RLH_HIP(hipGetLastError());
  return 0;
}

// Sums the per-workgroup partials of every Gram entry in a fixed order (one wavefront per
// entry: 64 strided partial sums, then a fixed xor-shuffle tree) and writes the (my, mx)
// result; complex outputs recombine conj(y)*x = (RR + II) + i (RI - IR).
template <int DT>
__global__ __launch_bounds__(256) void gram_finalize(const void *partials_, int nbx, int npj, int VY, int VX, int my,
                                                     int mx, void *out_, int same) {
  using T = typename DType<DT>::T;
  using R = typename DType<DT>::R;
  constexpr bool CPLX = DType<DT>::cplx;
  constexpr int NCOMP = CPLX ? 4 : 1;
  const R *partials = reinterpret_cast<const R *>(partials_);
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);      // one wave per output entry
  if (e >= my * mx) return;
  const int i = e / mx, j = e % mx;
  const int64_t slab = (int64_t)VY * VX;
  double t[NCOMP];
#pragma unroll
  for (int comp = 0; comp < NCOMP; ++comp) {
    int vi = CPLX ? 2 * i + (comp >> 1) : i;           // comp: 0 RR, 1 RI, 2 IR, 3 II
    int vj = CPLX ? 2 * j + (comp & 1) : j;
    // self-Gram: diagonal panels hold only the 16 x 16 tiles on and above their diagonal; the
    // real-view Gram matrix is symmetric, so the mirrored entry supplies the rest
    if (same && vi / VY == vj / VX && (vi % VY) / 16 > (vj % VX) / 16) { const int t_ = vi; vi = vj; vj = t_; }
    const int panel = (vi / VY) * npj + (vj / VX);
    const R *p = partials + ((int64_t)panel * slab + (vi % VY) * VX + (vj % VX)) * nbx;
    double s = 0.0;
    for (int b = lane; b < nbx; b += 64) s += (double)p[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    t[comp] = s;
  }
  if (lane == 0) {
    T *out = reinterpret_cast<T *>(out_);
    if constexpr (!CPLX) {
      out[e] = (R)t[0];
    } else {
      out[e].re = (R)(t[0] + t[3]);
      out[e].im = (R)(t[1] - t[2]);
    }
  }
}

// Resident workgroups per CU for one instantiation (registers + LDS), asked once.
template <int DT, int PI, int PJ, bool ALIGNED, int MODE, bool MULTI = false, bool QUAD = false, bool SYMS = false>
static int gram_blocks_per_cu() {
  static int cached = 0;
  if (cached == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gram_kernel<DT, PI, PJ, ALIGNED, MODE, MULTI, QUAD, SYMS>, QUAD ? 512 : 256, 0) != hipSuccess ||
        nb < 1)
      nb = 1;
    cached = nb > 8 ? 8 : nb;
    const char *e = getenv("RLH_GRAM_WG_PER_CU");          // tunable cap on resident workgroups
    if (e && *e && atoi(e) > 0 && atoi(e) < cached) cached = atoi(e);
  }
  return cached;
}

static inline bool env_quad() {         // RLH_GRAM_QUAD=0: no quadrant panels (tunable)
  const char *e = getenv("RLH_GRAM_QUAD");
  return !(e && *e == '0');
}

static inline int pick_tiles(int v) {   // 16x16 tiles per panel side: 1, 2 or 4
  if (v <= 16) return 1;
  if (v <= 32) return 2;
  return 4;
}

template <int DT, int PI, int PJ, bool ALIGNED, int MODE, bool MULTI = false, bool QUAD = false, bool SYMS = false>
static int gram_launch(GramArgs &a, int npi, int npj, int64_t my, int64_t mx, void *d_out) {
  using R = typename DType<DT>::R;
  Context &c = ctx();
  constexpr int VY = PI * 16 * (QUAD ? 2 : 1), VX = PJ * 16 * (QUAD ? 4 : 1);
  const int npanels = npi * npj;
  // the grid is sized to what is resident at once: every workgroup strides over the row
  // chunks, so a second, partially filled round of workgroups would only add a tail
  int64_t nbx = (int64_t)c.num_cu * gram_blocks_per_cu<DT, PI, PJ, ALIGNED, MODE, MULTI, QUAD, SYMS>() / npanels;
  if (nbx < 1) nbx = 1;
  if (nbx > a.nchunks) nbx = a.nchunks;
  const size_t part_bytes = sizeof(R) * VY * VX;
  while (nbx > 1 && (size_t)nbx * npanels * part_bytes > kWorkspaceBytes) nbx /= 2;
  RLH_REQUIRE((size_t)nbx * npanels * part_bytes <= kWorkspaceBytes,
              "rlh_gram: %lld x %lld result exceeds the reduction workspace", (long long)my, (long long)mx);
  hipLaunchKernelGGL((gram_kernel<DT, PI, PJ, ALIGNED, MODE, MULTI, QUAD, SYMS>), dim3((unsigned)nbx, (unsigned)npanels), dim3(QUAD ? 512 : 256), 0,
                     c.stream, a);
  RLH_HIP(hipGetLastError());
  const int total = (int)(my * mx);
  hipLaunchKernelGGL((gram_finalize<DT>), dim3((total + 3) / 4), dim3(256), 0, c.stream, c.work, (int)nbx, npj, VY,
                     VX, (int)my, (int)mx, d_out, a.same);
  RLH_HIP(hipGetLastError());
  return 0;
}

// Blocks that cannot stay in the 256 MB Infinity Cache between two launches are read with the non-temporal
// hint: measured on the access structure of this kernel (tools/stream_probe.hip) 6.1-6.2 -> 6.8-7.0 TB/s.
// Smaller blocks keep the default policy, under which a following launch finds them cached (RLH_GRAM_NT forces).
static int gram_nt(int64_t bytes) {
  const char *e = getenv("RLH_GRAM_NT");
  if (e && *e) return atoi(e);
  return bytes > (int64_t)192 << 20;
}

// ---------------------------------------------------------------- wave-private streaming Gram
// The roofline shapes -- real blocks, 16-byte aligned, at most 32 columns on the X side and 64 (two stacked blocks
// of the solver's fused reductions) on the Y side -- without any workgroup barrier.
// A workgroup is ONE wave.  It stages its own tile of TB bytes per column (32 fp64 rows at TB = 256) of the
// Y and X columns through registers into its private LDS image, and multiplies the tile out of the image while
// the loads of its next TWO tiles are in flight (two register sets): no wave ever waits for another one, and the
// memory system always has work queued.  The 256-thread kernel above spends 12 % of its time around its two
// barriers per chunk -- with the arithmetic AND the LDS traffic switched off it still ran at 5.86 TB/s, against
// 6.6-7.0 TB/s for the bare access pattern (tools/stream_probe.hip); this one measures 6.4-6.7 TB/s.
// LDS image: [column][16-byte piece], piece p of column c in slot p ^ swz(c), swz(c) = (c & 15) * (pieces / 16):
// the 16 lanes of a fragment read (16 columns, the same rows) then fall into 16 different slots, i.e. all 64
// banks once per half-wave (ds_read_b64) or wave (ds_read_b32); the ds_write_b128 of a staged piece covers
// whole columns.  The non-temporal hint is a template parameter: chosen at run time every load sits behind a
// branch, the compiler waits for vmcnt(0) where the paths join, and the hint gains nothing (measured).
// NYC = staged Y columns: 32, 64, or 0 for a self-Gram (A and B fragments both from the X image, tiles on and
// above the diagonal only).  Windows may be concatenations of blocks (GramArgs segments).  Results: one partial
// per wave, combined by gram_finalize in a fixed order.
// Waves per workgroup: the waves are independent (no barrier), but they are launched as ONE workgroup per CU --
// its LDS request is padded beyond half of the CU's LDS so that no two fit -- because the dispatcher does not
// spread one-wave workgroups evenly: right after a kernel with a large grid (dots, copy) some CUs received five
// waves and others three, and the launch took as long as its fullest CU (0.90-0.95 ms instead of 0.76 ms).
constexpr int gram_stream_waves(int ncol) { return ncol <= 16 ? 16 : (ncol <= 32 ? 8 : 4); }

// XAL: the X window is a block that also sits in the Y window ([X | Y]^H Y, [AX | X]^H X of the solver's stacked
// reductions), at columns a.xal .. (a multiple of 16): it is staged ONCE and the B fragments are read out of the Y
// image (16 instead of 24 loads per tile; such calls took 0.92-1.0 ms where two distinct blocks stream in 0.77 ms).
// CPLX: the blocks are complex (NYC, NXC count the REAL-view columns: re and im of column c are the virtual columns 2 c,
// 2 c + 1, as in the workgroup kernel; gram_finalize recombines conj(y) x).  A 16-byte piece then holds one complex128
// row or two complex64 rows of a column; it is split on the way into the image (two 8-byte LDS writes).
template <typename R, int TB, int NYC, int NXC, bool NT, bool XAL = false, bool CPLX = false>
__global__ __launch_bounds__(64 * (XAL ? 8 : gram_stream_waves(NYC + NXC))) void gram_stream_kernel(GramArgs a) {
  using M = Mfma16<R>;
  using acc_t = typename M::acc_t;
  constexpr bool SELF = NYC == 0;
  constexpr int RPU = 16 / (int)sizeof(R);      // rows per 16-byte piece
  constexpr int NP = TB / 16;                   // pieces per column and tile
  constexpr int ROWS = NP * RPU;                // rows per tile
  constexpr int NCOL = XAL ? NYC : NYC + NXC;   // staged columns: Y (NYC) then X (NXC = 16, 32 or 64; none if aliased)
  constexpr int NC = CPLX ? 2 : 1;              // reals per element of the blocks
  constexpr int NPT = NP * NC;                  // pieces per BLOCK column and tile (a complex column is two images)
  constexpr int CPL = 64 / NPT;                 // block columns covered by one load instruction of the wave
  constexpr int NL = NCOL / NC / CPL;           // loads per lane and tile
  static_assert(!(CPLX && XAL) && NPT <= 64, "complex tiles");
  constexpr int KS = ROWS / 4;                  // MFMA k-steps per tile
  constexpr int SW = NP / 16;
  constexpr int PJ = NXC / 16, PI = SELF ? PJ : NYC / 16;
  static_assert(NP >= 16 && NP <= 64 && 64 % NP == 0 && NCOL % CPL == 0, "tile shape");
  // (XAL: two waves per SIMD -- with half the bytes per MFMA the launch is no longer purely memory-bound, and a
  // second wave fills the first one's waits)
  constexpr int WPG = XAL ? 8 : gram_stream_waves(NYC + NXC);
  extern __shared__ __attribute__((aligned(16))) char lds_all[];
  char *lds = lds_all + (threadIdx.x >> 6) * (NCOL * TB);   // this wave's private image
  typedef R vec_t __attribute__((ext_vector_type(RPU)));
  const int lane = threadIdx.x & 63;
  const int p = lane % NPT, cl = lane / NPT;

  // this lane's piece of staged column q * CPL + cl at row 0 (columns past a window repeat its last one: they
  // only feed entries that are never written out)
  const R *colp[NL];
  int tstep[NL];                                // rows a tile advances this lane's piece by: ROWS, or 0 past the window
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    const int sc = q * CPL + cl;                // staged BLOCK column: Y window first, then X
    const bool isY = sc < NYC / NC;
    int c = isY ? sc : sc - NYC / NC;
    const int mc = isY ? a.my : a.mx;
    // staged columns past the window feed only entries that are never written out: their lanes re-read ONE
    // piece (row 0 of the window's last column, an L1 hit) instead of streaming a duplicate of that column
    tstep[q] = c < mc ? ROWS * NC : 0;          // (in reals)
    c = c < mc ? c : mc - 1;
    const GramSeg *segs = isY ? a.ys : a.xs;
    const int ns = isY ? a.nys : a.nxs;
    const void *bp = segs[0].p;
    int64_t ld = segs[0].ld;
    int c0 = 0;
#pragma unroll
    for (int k = 1; k < kGramSegs; ++k)
      if (k < ns && c >= segs[k].c0) { bp = segs[k].p; ld = segs[k].ld; c0 = segs[k].c0; }
    colp[q] = reinterpret_cast<const R *>(bp) + (int64_t)(c - c0) * ld * NC + p * RPU;
  }
  vec_t regsA[NL], regsB[NL];
  auto load_tile = [&](int64_t tile, vec_t (&regs)[NL]) {
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const vec_t *g = reinterpret_cast<const vec_t *>(colp[q] + tile * tstep[q]);
      if constexpr (NT) regs[q] = __builtin_nontemporal_load(g);
      else regs[q] = *g;
    }
  };
  auto load_tail = [&](int64_t tile, vec_t (&regs)[NL]) {   // the tile with fewer than ROWS rows: zero fill
    const int64_t row0 = tile * ROWS;
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const R *g = colp[q] + tile * tstep[q];
      vec_t v;
#pragma unroll
      for (int e = 0; e < RPU; ++e) v[e] = (row0 + (p * RPU + e) / NC < a.n) ? g[e] : (R)0;
      regs[q] = v;
    }
  };
  auto store_tile = [&](const vec_t (&regs)[NL]) {
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const int sc = q * CPL + cl;
      if constexpr (!CPLX) {
        *reinterpret_cast<vec_t *>(lds + sc * TB + ((p ^ ((sc & 15) * SW)) << 4)) = regs[q];
      } else {
        // piece p of the complex column: complex128 row p / complex64 rows 2 p, 2 p + 1 -> the 8-byte half p & 1 of
        // slot p >> 1 of the re image (virtual column 2 sc) and of the im image (2 sc + 1)
        typedef R half_t __attribute__((ext_vector_type(RPU / 2)));
        half_t re, im;
        if constexpr (RPU == 2) { re[0] = regs[q][0]; im[0] = regs[q][1]; }
        else { re[0] = regs[q][0]; re[1] = regs[q][2]; im[0] = regs[q][1]; im[1] = regs[q][3]; }
        const int v0 = 2 * sc, v1 = 2 * sc + 1, off8 = (p & 1) * 8;
        *reinterpret_cast<half_t *>(lds + v0 * TB + (((p >> 1) ^ ((v0 & 15) * SW)) << 4) + off8) = re;
        *reinterpret_cast<half_t *>(lds + v1 * TB + (((p >> 1) ^ ((v1 & 15) * SW)) << 4) + off8) = im;
      }
    }
  };
  acc_t acc[PI][PJ];
#pragma unroll
  for (int i = 0; i < PI; ++i)
#pragma unroll
    for (int j = 0; j < PJ; ++j) acc[i][j] = acc_t{(R)0, (R)0, (R)0, (R)0};
  const int fr = lane & 15, fk = lane >> 4;
  const int xcol0 = XAL ? a.xal : NYC;          // first image column of the X window
  // byte offset of this lane's fragment element of k-step s inside a column image
  auto frag_off = [&](int s) -> int {
    if constexpr (sizeof(R) == 8) return (((2 * s + (fk >> 1)) ^ (fr * SW)) << 4) + (fk & 1) * 8;
    else return ((s ^ (fr * SW)) << 4) + fk * 4;
  };
  auto compute = [&]() {
#pragma unroll 4
    for (int s = 0; s < KS; ++s) {
      const int off = frag_off(s);
      R fa[PI], fb[PJ];
#pragma unroll
      for (int j = 0; j < PJ; ++j) fb[j] = *reinterpret_cast<const R *>(lds + (xcol0 + j * 16 + fr) * TB + off);
#pragma unroll
      for (int i = 0; i < PI; ++i) {
        if constexpr (SELF) fa[i] = fb[i];
        else fa[i] = *reinterpret_cast<const R *>(lds + (i * 16 + fr) * TB + off);
      }
#pragma unroll
      for (int i = 0; i < PI; ++i)
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
          if (SELF && i > j) continue;          // symmetric: gram_finalize mirrors the tile below the diagonal
          acc[i][j] = M::run(fa[i], fb[j], acc[i][j]);
        }
    }
  };

  // Tile j of this wave is t0 + j G; loads past its last tile re-read that tile (an L2 hit) and are dropped, so
  // the loop body is straight-line code and the compiler waits with counted vmcnt for the older set only.
  const int64_t nfull = a.n / ROWS, G = (int64_t)gridDim.x * WPG, t0 = (int64_t)blockIdx.x * WPG + (threadIdx.x >> 6);
  const int64_t count = t0 < nfull ? (nfull - t0 + G - 1) / G : 0;
  auto tile_of = [&](int64_t j) -> int64_t { return t0 + (j < count ? j : count - 1) * G; };
  if constexpr (NCOL >= 128 || XAL) {           // (XAL: two waves per SIMD, 256 registers each: one set)
    // (128 staged columns: one register set of 32 pieces -- two would spill; a tile is 32 KB, so one tile per wave in
    // flight during its compute phase is as many bytes per CU as two tiles of the 64-column shapes)
    if (count > 0) {
      load_tile(tile_of(0), regsA);
      for (int64_t i = 0; i < count; ++i) {
        store_tile(regsA);
        __builtin_amdgcn_wave_barrier();
        load_tile(tile_of(i + 1), regsA);
        compute();
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else if (count > 0) {
    load_tile(tile_of(0), regsA);
    __builtin_amdgcn_sched_barrier(0);
    load_tile(tile_of(1), regsB);
    __builtin_amdgcn_sched_barrier(0);
    int64_t i = 0;
    for (; i + 1 < count; i += 2) {
      store_tile(regsA);
      __builtin_amdgcn_wave_barrier();
      load_tile(tile_of(i + 2), regsA);
      compute();
      __builtin_amdgcn_wave_barrier();
      store_tile(regsB);
      __builtin_amdgcn_wave_barrier();
      load_tile(tile_of(i + 3), regsB);
      compute();
      __builtin_amdgcn_wave_barrier();
    }
    if (i < count) {
      store_tile(regsA);
      __builtin_amdgcn_wave_barrier();
      compute();
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (nfull * ROWS < a.n && nfull % G == t0) {
    load_tail(nfull, regsA);
    store_tile(regsA);
    __builtin_amdgcn_wave_barrier();
    compute();
  }
  // One partial per WORKGROUP: every wave leaves its accumulators in the (now idle) LDS, tile after tile as
  // [lane][register], and the waves then sum the tiles in a fixed order, tile tt by wave tt mod WPG (a quarter of the
  // scattered 8-byte partial writes and of gram_finalize's reads; the order of the sum does not depend on timing).
  constexpr int SCR = PI * PJ * 256 * (int)sizeof(R);      // scratch per wave (may exceed its image: barrier first)
  __syncthreads();
  R *mine = reinterpret_cast<R *>(lds_all + (threadIdx.x >> 6) * SCR);
#pragma unroll
  for (int i = 0; i < PI; ++i)
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      if (SELF && i > j) continue;
      *reinterpret_cast<acc_t *>(mine + ((i * PJ + j) * 64 + lane) * 4) = acc[i][j];
    }
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  R *out = reinterpret_cast<R *>(a.partials) + blockIdx.x;
#pragma unroll
  for (int i = 0; i < PI; ++i)
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
      if (SELF && i > j) continue;
      const int tt = i * PJ + j;
      if (tt % WPG != wave) continue;          // (wave-uniform)
      acc_t sum = acc_t{(R)0, (R)0, (R)0, (R)0};
      for (int w = 0; w < WPG; ++w) {
        const acc_t part = *reinterpret_cast<const acc_t *>(reinterpret_cast<const R *>(lds_all + w * SCR) + (tt * 64 + lane) * 4);
        sum += part;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = i * 16 + M::out_row(lane, r), jj = j * 16 + (lane & 15);
        out[(int64_t)(ii * NXC + jj) * gridDim.x] = sum[r];
      }
    }
}

// my <= 64, mx <= 64, every segment 16-byte aligned; a.xs / a.ys / a.nxs / a.nys describe the
// windows, a.same a self-Gram of one block.
template <int DT>
static int gram_stream_launch(GramArgs &a, int64_t my, int64_t mx, void *d_out) {
  using R = typename DType<DT>::R;
  Context &c = ctx();
  constexpr int TB = 256;
  constexpr int ROWS = TB / (int)sizeof(R);
  const bool self = a.same;
  constexpr bool CPLX = DType<DT>::cplx;
  const int64_t vmx = mx * (CPLX ? 2 : 1), vmy = my * (CPLX ? 2 : 1);       // real-view widths
  const int nxc = vmx <= 16 ? 16 : (vmx <= 32 ? 32 : 64);
  const int nyc = self ? 0 : (vmy <= 16 && nxc < 64 ? 16 : (vmy <= 32 && nxc < 64 ? 32 : 64));
  // One workgroup per CU of 4 waves (one per SIMD) where a tile is 12 KB or more, 8 / 16 waves for narrower
  // tiles: with two register sets in flight per wave more waves only widen the window of rows the chip works on
  // at once (measured 6.3-6.7 TB/s at 4 waves per CU, 6.2 at 8, 4.8 at 2).
  const int wpg = gram_stream_waves(nyc + nxc);
  size_t lds = (size_t)wpg * (nyc + nxc) * TB;
  const size_t scr = (size_t)wpg * ((self ? nxc : nyc) / 16) * (nxc / 16) * 256 * sizeof(R);   // the final reduction's scratch
  if (lds < scr) lds = scr;
  if (lds < 84 * 1024) lds = 84 * 1024;          // more than half of the CU's 160 KB: one workgroup per CU
  const int64_t ntiles = (a.n + ROWS - 1) / ROWS;
  int64_t nbx = c.num_cu;
  if (nbx * wpg > ntiles) nbx = (ntiles + wpg - 1) / wpg;
  if (nbx < 1) nbx = 1;
  const int64_t nparts = nbx;                    // one partial per workgroup
  int64_t nparts_alias = 0;
  const int VY = self ? nxc : nyc;
  RLH_REQUIRE((size_t)nparts * VY * nxc * sizeof(R) <= kWorkspaceBytes, "rlh_gram: reduction workspace");
#define RLH_GS1(NYC_, NXC_, NT_)                                                                                       \
  do {                                                                                                                 \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      RLH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gram_stream_kernel<R, TB, NYC_, NXC_, NT_, false, CPLX>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                            \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((gram_stream_kernel<R, TB, NYC_, NXC_, NT_, false, CPLX>), dim3((unsigned)nbx),                 \
                       dim3(64 * gram_stream_waves(NYC_ + NXC_)), lds, c.stream, a);                                  \
  } while (0)
#define RLH_GS(NYC_, NXC_)                                                                                             \
  do {                                                                                                                 \
    if (a.nt & 1) RLH_GS1(NYC_, NXC_, true); else RLH_GS1(NYC_, NXC_, false);                                          \
  } while (0)
  if (!CPLX && a.xal >= 0 && nyc == 64 && nxc == 32) {
    // (the image and the launch are those of the 64-column Y window alone)
    lds = (size_t)8 * 64 * TB;
    if (lds < (size_t)8 * 8 * 256 * sizeof(R)) lds = (size_t)8 * 8 * 256 * sizeof(R);
    if (lds < 84 * 1024) lds = 84 * 1024;
    nbx = c.num_cu;
    if (nbx * 8 > ntiles) nbx = (ntiles + 7) / 8;
    if (nbx < 1) nbx = 1;
    nparts_alias = nbx;
#define RLH_GSA(NT_)                                                                                                   \
  do {                                                                                                                 \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      RLH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gram_stream_kernel<R, TB, 64, 32, NT_, true>),       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                            \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((gram_stream_kernel<R, TB, 64, 32, NT_, true>), dim3((unsigned)nbx), dim3(64 * 8),                    \
                       lds, c.stream, a);                                                                              \
  } while (0)
    if (a.nt & 1) RLH_GSA(true); else RLH_GSA(false);
#undef RLH_GSA
  } else if (nxc == 16) {
    if (nyc == 0) RLH_GS(0, 16); else if (nyc == 16) RLH_GS(16, 16); else if (nyc == 32) RLH_GS(32, 16); else RLH_GS(64, 16);
  } else if (nxc == 32) {
    if (nyc == 0) RLH_GS(0, 32); else if (nyc == 16) RLH_GS(16, 32); else if (nyc == 32) RLH_GS(32, 32); else RLH_GS(64, 32);
  } else {
    if (nyc == 0) RLH_GS(0, 64); else RLH_GS(64, 64);
  }
#undef RLH_GS
#undef RLH_GS1
  RLH_HIP(hipGetLastError());
  const int total = (int)(my * mx);
  hipLaunchKernelGGL((gram_finalize<DT>), dim3((total + 3) / 4), dim3(256), 0, c.stream, c.work,
                     (int)(nparts_alias ? nparts_alias : nparts), 1, VY, nxc,
                     (int)my, (int)mx, d_out, a.same);
  RLH_HIP(hipGetLastError());
  return 0;
}

template <int DT>
static int gram_impl(int64_t n, int64_t mx, const void *X, int64_t ldx, int64_t my, const void *Y, int64_t ldy,
                     void *d_out) {
  using R = typename DType<DT>::R;
  constexpr int NC = DType<DT>::cplx ? 2 : 1;
  Context &c = ctx();
  const int vx = (int)mx * NC, vy = (int)my * NC;
  const int PI = pick_tiles(vy), PJ = pick_tiles(vx);
  const int npi = (vy + PI * 16 - 1) / (PI * 16), npj = (vx + PJ * 16 - 1) / (PJ * 16);
  GramArgs a;
  a.X = X; a.Y = Y; a.ldx = ldx; a.ldy = ldy; a.n = n; a.mx = (int)mx; a.my = (int)my;
  a.same = (X == Y && ldx == ldy && mx == my) ? 1 : 0;
  a.npj = npj; a.partials = c.work;
  const int64_t es = dtype_size(DT);
  a.nt = gram_nt(n * (a.same ? mx : mx + my) * es);
  const bool aligned = aligned16(X, ldx, es) && aligned16(Y, ldy, es);
  {
    // windows of at most 64 real-view columns on either side (and more than 8 on one of them): the wave-private
    // streaming kernel (RLH_GRAM_STREAM=0: the workgroup kernel)
    const int stream = getenv("RLH_GRAM_STREAM") ? atoi(getenv("RLH_GRAM_STREAM")) : 1;
    // (at every size: with one partial per workgroup it measures 15 vs 14 us at n = 27000, 37 vs 40 us at n = 262144,
    // 91 vs 105 us at n = 10^6 against the workgroup kernel; the non-temporal hint stays a matter of size)
    if (stream && aligned && vx <= 64 && vy <= 64 && (vx > 8 || vy > 8)) {
      a.npj = 1;
      a.xal = -1;
      a.nxs = a.nys = 1;
      a.xs[0] = GramSeg{X, ldx, 0};
      a.ys[0] = GramSeg{Y, ldy, 0};
      for (int k = 1; k < kGramSegs; ++k) a.xs[k] = a.ys[k] = GramSeg{nullptr, 0, 1 << 30};
      return gram_stream_launch<DT>(a, my, mx, d_out);
    }
  }
  // Kernel variant (see gram_kernel).  Measured at n = 10^7, m = 32 fp64: the two-chunk pipeline
  // gains 5 % on the self-Gram (0.50 -> 0.475 ms) and nothing on the two-operand Gram (0.885 vs
  // 0.868 ms in sustained use: RLH_GRAM_PIPE=2 selects it); 1 KiB column pieces gain 2.5 % on the
  // two-operand Gram (0.889 -> 0.868 ms) and 2 KiB pieces 6-9 % at m = 16 (0.46 -> 0.427 ms).
  // RLH_GRAM_PIPE=0: general loop only; RLH_GRAM_ROWS=1|2|4
  // caps the piece length (tunables).
  static const int pipe = getenv("RLH_GRAM_PIPE") ? atoi(getenv("RLH_GRAM_PIPE")) : 1;
  static const int rows_cap = getenv("RLH_GRAM_ROWS") ? atoi(getenv("RLH_GRAM_ROWS")) : 4;
  int mode = 0;
  if (aligned && pipe != 0) {
    if (a.same && npi == 1 && npj == 1 && PI == PJ && PI <= 2) mode = 2;   // (longer chunks instead: 0.48 -> 0.535 ms)
    else if (!a.same && pipe >= 2 && PI * PJ <= 4) mode = 1;
    else if (PI <= 2 && PJ <= 2 && rows_cap >= 2) mode = (PI * PJ == 1 && rows_cap >= 4) ? 4 : 3;
  }
  // complex128, 33 .. 64 vectors on both sides, two operands: the interleaved real view on LDS-DMA staging (gram_z_dma_kernel)
  if constexpr (DT == RLH_Z) {
    static const int zdma = getenv("RLH_GRAM_ZDMA") ? atoi(getenv("RLH_GRAM_ZDMA")) : 1;
    if (zdma && aligned && !a.same && mx > 32 && mx <= 64 && my > 32 && my <= 64 && n >= 64 * (int64_t)kZgRows)
      return gram_z_dma_launch(n, mx, X, ldx, my, Y, ldy, d_out);
  }
  // windows of more than 64 real columns: 128 x 128 panels with one quadrant per wave (see gram_kernel)
  if (aligned && (vx > 64 || vy > 64) && vx > 32 && vy > 32 && env_quad()) {
    a.npj = (vx + 127) / 128;
    a.nchunks = (n + 256 / (int)sizeof(R) - 1) / (256 / (int)sizeof(R));
    if (a.same && vx <= 128)      // one square panel of a self-Gram: the symmetric tile assignment
      return gram_launch<DT, 4, 2, true, 0, false, true, true>(a, 1, 1, my, mx, d_out);
    return gram_launch<DT, 4, 2, true, 0, false, true>(a, (vy + 127) / 128, a.npj, my, mx, d_out);
  }
  // 33 .. 64 real columns on both sides: one 64 x 64 panel, 2 x 1 tiles per wave of a 512-thread workgroup
  if (aligned && vx > 32 && vy > 32 && vx <= 64 && vy <= 64 && env_quad()) {
    a.npj = 1;
    a.nchunks = (n + 512 / (int)sizeof(R) - 1) / (512 / (int)sizeof(R));
    if (a.same) return gram_launch<DT, 2, 1, true, 0, false, true, true>(a, 1, 1, my, mx, d_out);   // staged once
    return gram_launch<DT, 2, 1, true, 0, false, true>(a, 1, 1, my, mx, d_out);
  }
  const int ROWS = (mode == 3 ? 2 : (mode == 4 ? 4 : 1)) * 512 / (int)sizeof(R);
  a.nchunks = (n + ROWS - 1) / ROWS;
#define RLH_GRAM_CASE(pi, pj)                                                              \
  if (PI == pi && PJ == pj) {                                                              \
    if constexpr (pi * pj <= 4)       /* two register sets of a larger panel do not fit */ \
      if (mode == 1) return gram_launch<DT, pi, pj, true, 1>(a, npi, npj, my, mx, d_out);  \
    if constexpr (pi <= 2 && pj <= 2) /* nor does the LDS tile of a longer chunk */        \
      if (mode == 3) return gram_launch<DT, pi, pj, true, 3>(a, npi, npj, my, mx, d_out);  \
    if constexpr (pi * pj == 1)                                                            \
      if (mode == 4) return gram_launch<DT, pi, pj, true, 4>(a, npi, npj, my, mx, d_out);  \
    if constexpr (pi == pj && pi <= 2)                                                     \
      if (mode == 2) return gram_launch<DT, pi, pj, true, 2>(a, npi, npj, my, mx, d_out);  \
    return aligned ? gram_launch<DT, pi, pj, true, 0>(a, npi, npj, my, mx, d_out)          \
                   : gram_launch<DT, pi, pj, false, 0>(a, npi, npj, my, mx, d_out);        \
  }
  RLH_GRAM_CASE(1, 1) RLH_GRAM_CASE(1, 2) RLH_GRAM_CASE(1, 4)
  RLH_GRAM_CASE(2, 1) RLH_GRAM_CASE(2, 2) RLH_GRAM_CASE(2, 4)
  RLH_GRAM_CASE(4, 1) RLH_GRAM_CASE(4, 2) RLH_GRAM_CASE(4, 4)
#undef RLH_GRAM_CASE
  set_error("rlh_gram: no kernel for panel %d x %d", PI, PJ);
  return 1;
}

// Gram of two concatenated windows: G = [Y_0 | Y_1 | ...]^H [X_0 | X_1 | ...] in one pass (every block
// is read once: [AX | X]^H X costs three block reads where two separate Grams cost four).
template <int DT>
static int gram_multi_impl(int64_t n, int nx, const void *const *X, const int64_t *ldx, const int64_t *mx, int ny,
                           const void *const *Y, const int64_t *ldy, const int64_t *my, void *d_out) {
  using R = typename DType<DT>::R;
  constexpr int NC = DType<DT>::cplx ? 2 : 1;
  Context &c = ctx();
  GramArgs a;
  const int64_t es = dtype_size(DT);
  bool aligned = true;
  int64_t mxt = 0, myt = 0;
  for (int k = 0; k < nx; ++k) {
    a.xs[k] = GramSeg{X[k], ldx[k], (int)mxt};
    mxt += mx[k];
    aligned = aligned && aligned16(X[k], ldx[k], es);
  }
  for (int k = 0; k < ny; ++k) {
    a.ys[k] = GramSeg{Y[k], ldy[k], (int)myt};
    myt += my[k];
    aligned = aligned && aligned16(Y[k], ldy[k], es);
  }
  for (int k = nx; k < kGramSegs; ++k) a.xs[k] = GramSeg{nullptr, 0, 1 << 30};
  for (int k = ny; k < kGramSegs; ++k) a.ys[k] = GramSeg{nullptr, 0, 1 << 30};
  a.nxs = nx; a.nys = ny;
  const int vx = (int)mxt * NC, vy = (int)myt * NC;
  const int PI = pick_tiles(vy), PJ = pick_tiles(vx);
  const int npi = (vy + PI * 16 - 1) / (PI * 16), npj = (vx + PJ * 16 - 1) / (PJ * 16);
  a.X = X[0]; a.Y = Y[0]; a.ldx = ldx[0]; a.ldy = ldy[0]; a.n = n; a.mx = (int)mxt; a.my = (int)myt;
  a.same = 0; a.npj = npj; a.partials = c.work;
  a.nt = gram_nt(n * (mxt + myt) * es);
  {
    const int stream = getenv("RLH_GRAM_STREAM") ? atoi(getenv("RLH_GRAM_STREAM")) : 1;
    if (stream && aligned && vx <= 64 && vy <= 64 && (vx > 8 || vy > 8)) {
      a.npj = 1;
      a.xal = -1;
      if (!DType<DT>::cplx && nx == 1 && mxt > 16 && mxt <= 32 && myt > 32)   // the X block is one of the Y window's blocks
        for (int k2 = 0; k2 < ny; ++k2)
          if (Y[k2] == X[0] && ldy[k2] == ldx[0] && my[k2] == mx[0] && a.ys[k2].c0 % 16 == 0) a.xal = a.ys[k2].c0;
      return gram_stream_launch<DT>(a, myt, mxt, d_out);
    }
  }
  // A stacked window that the segment kernel would cut into several 64 x 32 panels (complex blocks of 64 vectors: 16
  // panels, each re-reading its column ranges: 3.7-3.9 ms where two separate 128 x 128-panel calls take 2.3-2.6 ms):
  // one call per left block instead, written to its rows of the result (one right block: those rows are contiguous).
  if (nx == 1 && npi * npj > 1) {
    for (int k = 0; k < ny; ++k)
      if (int rc = gram_impl<DT>(n, mx[0], X[0], ldx[0], my[k], Y[k], ldy[k],
                                 (char *)d_out + (size_t)a.ys[k].c0 * (size_t)mxt * (size_t)es))
        return rc;
    return 0;
  }
  const int ROWS = 512 / (int)sizeof(R);
  a.nchunks = (n + ROWS - 1) / ROWS;
#define RLH_GRAM_MCASE(pi, pj)                                                                        \
  if (PI == pi && PJ == pj)                                                                           \
    return aligned ? gram_launch<DT, pi, pj, true, 0, true>(a, npi, npj, myt, mxt, d_out)             \
                   : gram_launch<DT, pi, pj, false, 0, true>(a, npi, npj, myt, mxt, d_out);
  RLH_GRAM_MCASE(1, 1) RLH_GRAM_MCASE(1, 2) RLH_GRAM_MCASE(1, 4)
  RLH_GRAM_MCASE(2, 1) RLH_GRAM_MCASE(2, 2) RLH_GRAM_MCASE(2, 4)
  RLH_GRAM_MCASE(4, 1) RLH_GRAM_MCASE(4, 2) RLH_GRAM_MCASE(4, 4)
#undef RLH_GRAM_MCASE
  set_error("rlh_gram_multi: no kernel for panel %d x %d", PI, PJ);
  return 1;
}

// ---------------------------------------------------------------- dots (K2)
// One workgroup per (row block, column); deterministic two-stage reduction.
template <typename T, bool ALIGNED, bool NT>
__global__ __launch_bounds__(256) void dots_kernel(const T *X, int64_t ldx, const T *Y, int64_t ldy, int64_t n,
                                                   T *partials, int nbx, int per) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int col = blockIdx.y;
  const T *x = X + (int64_t)col * ldx;
  const T *y = Y + (int64_t)col * ldy;
  T acc = zero_of(T{});
  // a workgroup owns `per` consecutive runs of 256 16-byte pieces: resident workgroups form a compact window that
  // sweeps the column front to back (no grid-stride loop: see row_blocks in update.hip)
  // (`per` is a multiple of 4: groups of four pieces per thread, the eight loads of a complete group issued
  // before the first multiply)
  struct alignas(16) V { T v[VEC]; };
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  auto piece = [&](const T *p) -> V {
    union { u4 u; V v; } c;
    if constexpr (NT) c.u = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(p));
    else c.u = *reinterpret_cast<const u4 *>(p);
    return c.v;
  };
  for (int g = 0; g < per; g += 4) {
    const int64_t base = ((int64_t)blockIdx.x * per + g) * 256 * VEC;
    if (base >= n) break;
    if (ALIGNED && base + 4 * 256 * VEC <= n && x == y) {        // X.dots(X): every piece is loaded once
      V xv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[i] = piece(x + base + ((int64_t)i * 256 + threadIdx.x) * VEC);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) fma_conj_acc(acc, xv[i].v[e], xv[i].v[e]);
    } else if (ALIGNED && base + 4 * 256 * VEC <= n) {
      V xv[4], yv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t r = base + ((int64_t)i * 256 + threadIdx.x) * VEC;
        xv[i] = piece(x + r);
        yv[i] = piece(y + r);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) fma_conj_acc(acc, yv[i].v[e], xv[i].v[e]);
    } else {
      for (int i = 0; i < 4; ++i) {
        const int64_t r = base + ((int64_t)i * 256 + threadIdx.x) * VEC;
        for (int e = 0; e < VEC && r + e < n; ++e) fma_conj_acc(acc, y[r + e], x[r + e]);
      }
    }
  }
  __shared__ T red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = add_of(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[(int64_t)col * nbx + blockIdx.x] = red[0];
}

template <typename T>
__global__ __launch_bounds__(256) void dots_finalize(const T *partials, int nbx, int m, T *out) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);     // one wave per column
  if (col >= m) return;
  T s = zero_of(T{});
  for (int b = lane; b < nbx; b += 64) s = add_of(s, partials[(int64_t)col * nbx + b]);
  // fixed xor-shuffle tree over the 64 lanes (deterministic)
  if constexpr (sizeof(T) == sizeof(float) || sizeof(T) == sizeof(double)) {
    using R = T;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s = s + (R)__shfl_xor(s, off);
  }
  if (lane == 0) out[col] = s;
}

template <typename R2, typename C>
__device__ __forceinline__ C shfl_complex(C v, int off) {
  C r;
  r.re = __shfl_xor(v.re, off);
  r.im = __shfl_xor(v.im, off);
  return r;
}

template <>
__global__ __launch_bounds__(256) void dots_finalize<c32>(const c32 *partials, int nbx, int m, c32 *out) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= m) return;
  c32 s = zero_of(c32{});
  for (int b = lane; b < nbx; b += 64) s = add_of(s, partials[(int64_t)col * nbx + b]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s = add_of(s, shfl_complex<float>(s, off));
  if (lane == 0) out[col] = s;
}

template <>
__global__ __launch_bounds__(256) void dots_finalize<c64>(const c64 *partials, int nbx, int m, c64 *out) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= m) return;
  c64 s = zero_of(c64{});
  for (int b = lane; b < nbx; b += 64) s = add_of(s, partials[(int64_t)col * nbx + b]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s = add_of(s, shfl_complex<double>(s, off));
  if (lane == 0) out[col] = s;
}

template <int DT>
static int dots_impl(int64_t n, int64_t m, const void *X, int64_t ldx, const void *Y, int64_t ldy, void *d_out) {
  using T = typename DType<DT>::T;
  Context &c = ctx();
  constexpr int VEC = 16 / (int)sizeof(T);
  // vector loads per thread (a multiple of 4): at most ~2048 partials per column, so that the one wave per column
  // of dots_finalize stays a few microseconds
  int per = 4;
  int64_t nbx = (n + 256 * VEC * (int64_t)per - 1) / (256 * VEC * (int64_t)per);
  while (nbx > 2048 || (nbx > 1 && (size_t)nbx * m * sizeof(T) > kWorkspaceBytes / 4)) {
    per += 4;
    nbx = (n + 256 * VEC * (int64_t)per - 1) / (256 * VEC * (int64_t)per);
  }
  if (nbx < 1) nbx = 1;
  RLH_REQUIRE((size_t)nbx * m * sizeof(T) <= kWorkspaceBytes, "rlh_dots: too many columns");
  const int nt = gram_nt(n * m * (int64_t)sizeof(T) * (X == Y ? 1 : 2));
  const bool aligned = aligned16(X, ldx, sizeof(T)) && aligned16(Y, ldy, sizeof(T));
  dim3 grid((unsigned)nbx, (unsigned)m);
  if (aligned && (nt & 1))
    hipLaunchKernelGGL((dots_kernel<T, true, true>), grid, dim3(256), 0, c.stream, (const T *)X, ldx, (const T *)Y, ldy,
                       n, (T *)c.work, (int)nbx, per);
  else if (aligned)
    hipLaunchKernelGGL((dots_kernel<T, true, false>), grid, dim3(256), 0, c.stream, (const T *)X, ldx, (const T *)Y, ldy,
                       n, (T *)c.work, (int)nbx, per);
  else
    hipLaunchKernelGGL((dots_kernel<T, false, false>), grid, dim3(256), 0, c.stream, (const T *)X, ldx, (const T *)Y, ldy,
                       n, (T *)c.work, (int)nbx, per);
  RLH_HIP(hipGetLastError());
  hipLaunchKernelGGL((dots_finalize<T>), dim3(((int)m + 3) / 4), dim3(256), 0, c.stream, (const T *)c.work,
                     (int)nbx, (int)m, (T *)d_out);
  RLH_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- largest modulus of the entries
// max_ij max(|re x_ij|, |im x_ij|) of a block viewed as reals (AMatrix.scale(): the reference scans
// the host array with numpy amin/amax, raleigh/algebra/dense_matrix.py:44-49 -- 0.1 s for 20000^2).
template <typename R>
__global__ __launch_bounds__(256) void absmax_kernel(const R *X, int64_t ldx_r, int64_t n_r, double *partials, int nbx) {
  const R *x = X + (int64_t)blockIdx.y * ldx_r;
  double mx = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_r; r += stride) {
    const double v = fabs((double)x[r]);
    mx = v > mx ? v : mx;                      // (a NaN entry is skipped, as numpy's fmax would)
  }
  __shared__ double red[256];
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + s] ? red[threadIdx.x] : red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * nbx + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void absmax_finalize(const double *partials, int64_t count, double *out) {
  double mx = 0.0;
  for (int64_t i = threadIdx.x; i < count; i += 256) mx = partials[i] > mx ? partials[i] : mx;
  __shared__ double red[256];
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + s] ? red[threadIdx.x] : red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

template <int DT>
static int absmax_impl(int64_t n, int64_t m, const void *X, int64_t ldx, void *d_out) {
  using R = typename DType<DT>::R;
  constexpr int NC = DType<DT>::cplx ? 2 : 1;
  Context &c = ctx();
  int64_t nbx = (n * NC + 256 * 8 - 1) / (256 * 8);
  const int64_t cap = ((int64_t)c.num_cu * 8 + m - 1) / m;
  if (nbx > cap) nbx = cap;
  if (nbx < 1) nbx = 1;
  RLH_REQUIRE((size_t)nbx * m * sizeof(double) <= kWorkspaceBytes, "rlh_absmax: too many columns");
  hipLaunchKernelGGL((absmax_kernel<R>), dim3((unsigned)nbx, (unsigned)m), dim3(256), 0, c.stream, (const R *)X,
                     ldx * NC, n * NC, (double *)c.work, (int)nbx);
  RLH_HIP(hipGetLastError());
  hipLaunchKernelGGL(absmax_finalize, dim3(1), dim3(256), 0, c.stream, (const double *)c.work, nbx * m, (double *)d_out);
  RLH_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- transposed dots (K2t)
template <typename T>
__global__ __launch_bounds__(256) void dots_transp_kernel(const T *X, int64_t ldx, const T *Y, int64_t ldy,
                                                          int64_t n, int m, T *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
    T acc = zero_of(T{});
    for (int i = 0; i < m; ++i) fma_conj_acc(acc, Y[r + (int64_t)i * ldy], X[r + (int64_t)i * ldx]);
    out[r] = acc;
  }
}

template <int DT>
static int dots_transp_impl(int64_t n, int64_t m, const void *X, int64_t ldx, const void *Y, int64_t ldy,
                            void *d_out) {
  using T = typename DType<DT>::T;
  Context &c = ctx();
  int64_t nb = (n + 255) / 256;
  if (nb > (int64_t)c.num_cu * 8) nb = (int64_t)c.num_cu * 8;
  hipLaunchKernelGGL((dots_transp_kernel<T>), dim3((unsigned)nb), dim3(256), 0, c.stream, (const T *)X, ldx,
                     (const T *)Y, ldy, n, (int)m, (T *)d_out);
  RLH_HIP(hipGetLastError());
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

extern "C" {

int rlh_gram(int dtype, int64_t n, int64_t mx, const void *X, int64_t ldx, int64_t my, const void *Y, int64_t ldy,
             void *d_out, void *h_out) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_gram: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && mx >= 0 && my >= 0, "rlh_gram: negative size");
  RLH_REQUIRE(mx <= 32768 && my <= 32768, "rlh_gram: more than 32768 vectors in a window");
  if (mx == 0 || my == 0) return 0;
  RLH_REQUIRE(X && Y, "rlh_gram: null block pointer");
  RLH_REQUIRE(ldx >= n && ldy >= n, "rlh_gram: leading dimension smaller than n");
  RLH_REQUIRE(d_out || h_out, "rlh_gram: no output buffer");
  const size_t bytes = (size_t)(mx * my * dtype_size(dtype));
  if (!d_out) {
    if (int rc = ensure_result(bytes)) return rc;
    d_out = ctx().result_hd;          // small result: written by the finalize kernel into mapped host memory
  }
  int rc = 0;
  if (n == 0) {
    RLH_HIP(hipMemsetAsync(d_out, 0, bytes, ctx().stream));
  } else {
    RLH_DISPATCH(dtype, gram_impl, n, mx, X, ldx, my, Y, ldy, d_out)
  }
  if (rc) return rc;
  if (h_out) return fetch_result(h_out, d_out, bytes);
  return 0;
}

int rlh_gram_multi(int dtype, int64_t n, int nx, const void *const *X, const int64_t *ldx, const int64_t *mx, int ny,
                   const void *const *Y, const int64_t *ldy, const int64_t *my, void *d_out, void *h_out) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_gram_multi: unknown dtype %d", dtype);
  RLH_REQUIRE(nx >= 1 && nx <= kGramSegs && ny >= 1 && ny <= kGramSegs, "rlh_gram_multi: 1 to %d blocks per window", kGramSegs);
  RLH_REQUIRE(n >= 0 && X && ldx && mx && Y && ldy && my, "rlh_gram_multi: bad arguments");
  int64_t mxt = 0, myt = 0;
  for (int k = 0; k < nx; ++k) {
    RLH_REQUIRE(mx[k] >= 1 && X[k] && ldx[k] >= n, "rlh_gram_multi: bad block %d of the right window", k);
    mxt += mx[k];
  }
  for (int k = 0; k < ny; ++k) {
    RLH_REQUIRE(my[k] >= 1 && Y[k] && ldy[k] >= n, "rlh_gram_multi: bad block %d of the left window", k);
    myt += my[k];
  }
  RLH_REQUIRE(mxt <= 32768 && myt <= 32768, "rlh_gram_multi: more than 32768 vectors in a window");
  RLH_REQUIRE(d_out || h_out, "rlh_gram_multi: no output buffer");
  const size_t bytes = (size_t)(mxt * myt * dtype_size(dtype));
  if (!d_out) {
    if (int rc = ensure_result(bytes)) return rc;
    d_out = ctx().result_hd;
  }
  int rc = 0;
  if (n == 0) {
    RLH_HIP(hipMemsetAsync(d_out, 0, bytes, ctx().stream));
  } else {
    RLH_DISPATCH(dtype, gram_multi_impl, n, nx, X, ldx, mx, ny, Y, ldy, my, d_out)
  }
  if (rc) return rc;
  if (h_out) return fetch_result(h_out, d_out, bytes);
  return 0;
}

int rlh_dots(int dtype, int64_t n, int64_t m, const void *X, int64_t ldx, const void *Y, int64_t ldy, void *d_out,
             void *h_out) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_dots: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && m >= 0, "rlh_dots: negative size");
  RLH_REQUIRE(m <= 32768, "rlh_dots: more than 32768 vectors in a window");
  if (m == 0) return 0;
  RLH_REQUIRE(X && Y, "rlh_dots: null block pointer");
  RLH_REQUIRE(ldx >= n && ldy >= n, "rlh_dots: leading dimension smaller than n");
  RLH_REQUIRE(d_out || h_out, "rlh_dots: no output buffer");
  const size_t bytes = (size_t)(m * dtype_size(dtype));
  if (!d_out) {
    if (int rc = ensure_result(bytes)) return rc;
    d_out = ctx().result_hd;          // small result: written by the finalize kernel into mapped host memory
  }
  int rc = 0;
  if (n == 0) {
    RLH_HIP(hipMemsetAsync(d_out, 0, bytes, ctx().stream));
  } else {
    RLH_DISPATCH(dtype, dots_impl, n, m, X, ldx, Y, ldy, d_out)
  }
  if (rc) return rc;
  if (h_out) return fetch_result(h_out, d_out, bytes);
  return 0;
}

int rlh_absmax(int dtype, int64_t n, int64_t m, const void *X, int64_t ldx, double *h_out) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_absmax: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && m >= 0 && m <= 65535, "rlh_absmax: bad size");
  RLH_REQUIRE(h_out != nullptr, "rlh_absmax: null output");
  if (n == 0 || m == 0) { *h_out = 0.0; return 0; }
  RLH_REQUIRE(X && ldx >= n, "rlh_absmax: bad block");
  if (int rc = ensure_result(sizeof(double))) return rc;
  void *d_out = ctx().result_hd;      // written by the finalize kernel into mapped host memory
  int rc = 0;
  RLH_DISPATCH(dtype, absmax_impl, n, m, X, ldx, d_out)
  if (rc) return rc;
  return fetch_result(h_out, d_out, sizeof(double));
}

int rlh_dots_transp(int dtype, int64_t n, int64_t m, const void *X, int64_t ldx, const void *Y, int64_t ldy,
                    void *d_out) {
  if (int rc = require_ready()) return rc;
  RLH_REQUIRE(dtype_valid(dtype), "rlh_dots_transp: unknown dtype %d", dtype);
  RLH_REQUIRE(n >= 0 && m >= 0, "rlh_dots_transp: negative size");
  if (n == 0) return 0;
  RLH_REQUIRE(d_out, "rlh_dots_transp: null output");
  RLH_REQUIRE(m == 0 || (X && Y), "rlh_dots_transp: null block pointer");
  RLH_REQUIRE(ldx >= n && ldy >= n, "rlh_dots_transp: leading dimension smaller than n");
  int rc = 0;
  RLH_DISPATCH(dtype, dots_transp_impl, n, m, X, ldx, Y, ldy, d_out)
  return rc;
}

}  // extern "C"

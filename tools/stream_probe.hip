// Streaming-pattern probe for gfx950 (not part of the library): what HBM rate do the access STRUCTURES of the
// column-wise kernels (one column per workgroup) and of the block update (every lane touches all 32 columns of
// its rows) reach, with and without the non-temporal hint, at the roofline point's block size?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe tools/stream_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NTL, bool NTS> __device__ __forceinline__ u4 ld(const u4 *p) { return NTL ? __builtin_nontemporal_load(p) : *p; }
template <bool NTL, bool NTS> __device__ __forceinline__ void st(u4 *p, u4 v) { if (NTS) __builtin_nontemporal_store(v, p); else *p = v; }

// 1. flat copy, one 16-byte word per thread
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_flat(const u4 *x, u4 *y, int64_t nw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nw) st<NTL, NTS>(y + i, ld<NTL, NTS>(x + i));
}
// 2. flat copy, grid-stride, U words in flight per thread
template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void copy_stride(const u4 *x, u4 *y, int64_t nw) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < nw; i += U * stride) {
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NTL, NTS>(x + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NTL, NTS>(y + i + u * stride, v[u]);
  }
  for (; i < nw; i += stride) st<NTL, NTS>(y + i, ld<NTL, NTS>(x + i));
}
// 3. row-wise: a lane owns one 16-byte row group and walks C columns, U at a time (the block update's structure;
//    ADD also reads the destination)
template <bool NTL, bool NTS, int C, int U, bool ADD>
__global__ __launch_bounds__(256) void rowwise(const u4 *x, u4 *y, int64_t ldw, int64_t nw) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nw; r += stride) {
    u4 acc[ADD ? C : 1];
    if (ADD) {
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] = ld<NTL, NTS>(y + r + c * ldw);
    }
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += U) {
      u4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld<NTL, NTS>(x + r + (c0 + u) * ldw);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ADD) v[u] += acc[c0 + u];
        st<NTL, NTS>(y + r + (c0 + u) * ldw, v[u]);
      }
    }
  }
}
// 4. row-wise in two phases: all C loads, then all C stores (what the update does: every store after the last load)
template <bool NTL, bool NTS, int C>
__global__ __launch_bounds__(256) void rowwise_phased(const u4 *x, u4 *y, int64_t ldw, int64_t nw) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nw; r += stride) {
    u4 v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = ld<NTL, NTS>(x + r + c * ldw);
#pragma unroll
    for (int c = 0; c < C; ++c) st<NTL, NTS>(y + r + c * ldw, v[c]);
  }
}
// 5. column-wise with a loop (the library's copy): grid (row blocks, columns)
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_cols(const u4 *x, u4 *y, int64_t ldw, int64_t nw) {
  const u4 *xc = x + blockIdx.y * ldw;
  u4 *yc = y + blockIdx.y * ldw;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nw; r += stride) st<NTL, NTS>(yc + r, ld<NTL, NTS>(xc + r));
}
// 6. read only (sum), row-wise over C columns and flat
template <bool NTL, int C>
__global__ __launch_bounds__(256) void read_rowwise(const u4 *x, u4 *y, int64_t ldw, int64_t nw) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  u4 s = {0, 0, 0, 0};
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nw; r += stride) {
#pragma unroll
    for (int c = 0; c < C; ++c) s += ld<NTL, false>(x + r + c * ldw);
  }
  if (s.x == 0x12345678u) y[0] = s;
}
template <bool NTS, int C>
__global__ __launch_bounds__(256) void write_rowwise(u4 *y, int64_t ldw, int64_t nw) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const u4 s = {1, 2, 3, 4};
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nw; r += stride) {
#pragma unroll
    for (int c = 0; c < C; ++c) st<false, NTS>(y + r + c * ldw, s);
  }
}


// 7. row-wise with CONTIGUOUS chunks: workgroup b owns row groups [b * chunk, (b + 1) * chunk) (chunk a multiple of 256)
template <bool NTL, bool NTS, int C, int U>
__global__ __launch_bounds__(256) void rowwise_chunked(const u4 *x, u4 *y, int64_t ldw, int64_t nw, int64_t chunk) {
  const int64_t r1 = ((int64_t)blockIdx.x + 1) * chunk < nw ? ((int64_t)blockIdx.x + 1) * chunk : nw;
  for (int64_t r = (int64_t)blockIdx.x * chunk + threadIdx.x; r < r1; r += 256) {
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += U) {
      u4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld<NTL, NTS>(x + r + (c0 + u) * ldw);
#pragma unroll
      for (int u = 0; u < U; ++u) st<NTL, NTS>(y + r + (c0 + u) * ldw, v[u]);
    }
  }
}
template <bool NTL, int C>
__global__ __launch_bounds__(256) void read_chunked(const u4 *x, u4 *y, int64_t ldw, int64_t nw, int64_t chunk) {
  const int64_t r1 = ((int64_t)blockIdx.x + 1) * chunk < nw ? ((int64_t)blockIdx.x + 1) * chunk : nw;
  u4 s = {0, 0, 0, 0};
  for (int64_t r = (int64_t)blockIdx.x * chunk + threadIdx.x; r < r1; r += 256) {
#pragma unroll
    for (int c = 0; c < C; ++c) s += ld<NTL, false>(x + r + c * ldw);
  }
  if (s.x == 0x12345678u) y[0] = s;
}


// 8. the Gram kernel's read pattern: a 256-thread workgroup reads NCOL columns x RUN bytes per step; thread t takes
//    16-byte piece t % (RUN/16) of column t / (RUN/16) + q * (256 / (RUN/16)).  grid-stride (CHUNKED = false) or
//    contiguous steps per workgroup.
template <bool NTL, int NCOL, int RUN, bool CHUNKED>
__global__ __launch_bounds__(256) void read_gram_like(const u4 *x, u4 *y, int64_t ldw, int64_t nw, int64_t steps_per_wg) {
  constexpr int UPC = RUN / 16, CPQ = 256 / UPC, NQ = NCOL / CPQ;
  const int tcol = threadIdx.x / UPC, tk = threadIdx.x % UPC;
  const int64_t nsteps = nw / UPC;
  u4 s = {0, 0, 0, 0};
  int64_t st0 = CHUNKED ? (int64_t)blockIdx.x * steps_per_wg : blockIdx.x;
  const int64_t st1 = CHUNKED ? (st0 + steps_per_wg < nsteps ? st0 + steps_per_wg : nsteps) : nsteps;
  const int64_t inc = CHUNKED ? 1 : gridDim.x;
  for (int64_t stp = st0; stp < st1; stp += inc) {
    u4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = ld<NTL, false>(x + (int64_t)(tcol + q * CPQ) * ldw + stp * UPC + tk);
#pragma unroll
    for (int q = 0; q < NQ; ++q) s += v[q];
  }
  if (s.x == 0x12345678u) y[0] = s;
}
// 9. row-wise copy where a thread owns W consecutive 16-byte words per column (a workgroup covers 4 W KB per column)
template <bool NTL, bool NTS, int C, int U, int W>
__global__ __launch_bounds__(256) void rowwise_wide(const u4 *x, u4 *y, int64_t ldw, int64_t nw) {
  const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * W;
  if (r + W > nw) return;
#pragma unroll
  for (int c0 = 0; c0 < C; c0 += U) {
    u4 v[U][W];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int w = 0; w < W; ++w) v[u][w] = ld<NTL, NTS>(x + r + w + (c0 + u) * ldw);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int w = 0; w < W; ++w) st<NTL, NTS>(y + r + w + (c0 + u) * ldw, v[u][w]);
  }
}
// 10. row-wise copy, wave-contiguous variant: a WAVE owns W consecutive KB per column (lane l takes word l of each KB)
template <bool NTL, bool NTS, int C, int U, int W>
__global__ __launch_bounds__(256) void rowwise_wavewide(const u4 *x, u4 *y, int64_t ldw, int64_t nw) {
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t r = wave * 64 * W + (threadIdx.x & 63);
  if (wave * 64 * W + 64 * W > nw) return;
#pragma unroll
  for (int c0 = 0; c0 < C; c0 += U) {
    u4 v[U][W];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int w = 0; w < W; ++w) v[u][w] = ld<NTL, NTS>(x + r + 64 * w + (c0 + u) * ldw);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int w = 0; w < W; ++w) st<NTL, NTS>(y + r + 64 * w + (c0 + u) * ldw, v[u][w]);
  }
}


// 11. the block update's arithmetic on top of the row-wise copy: a lane owns 2 rows (16 bytes), 32 output columns,
//     1024 FMAs per row.  QMODE 0: coefficients resident in SGPRs (8 kernel-argument doubles, reused), 1: read from
//     global memory with uniform indices (s_load, the library kernel's pattern), 2: no arithmetic (copy of column sums)
struct Q8 { double q[8]; };
template <int QMODE, bool NT>
__global__ __launch_bounds__(256) void rowwise_fma(const double *x, double *y, int64_t ld, int64_t n, Q8 qa, const double *__restrict__ Q) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (r + 2 > n) return;
  double acc[2][32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[0][j] = acc[1][j] = 0.0;
#pragma unroll
  for (int c0 = 0; c0 < 32; c0 += 4) {
    d2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2 *>(x + r + (c0 + u) * ld)) : *reinterpret_cast<const d2 *>(x + r + (c0 + u) * ld);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        if (QMODE == 2) {
          if (j == ((c0 + u) & 31)) { acc[0][j] += v[u].x; acc[1][j] += v[u].y; }
        } else {
          const double q = QMODE == 0 ? qa.q[((c0 + u) * (2 * (j & 3) + 1) + (j >> 2)) & 7] : Q[(c0 + u) * 32 + j];
          acc[0][j] = fma(v[u].x, q, acc[0][j]);
          acc[1][j] = fma(v[u].y, q, acc[1][j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    d2 o = {acc[0][j], acc[1][j]};
    if (NT) __builtin_nontemporal_store(o, reinterpret_cast<d2 *>(y + r + j * ld));
    else *reinterpret_cast<d2 *>(y + r + j * ld) = o;
  }
}


// 12. two-block reads: the Gram pattern over 32 columns of x AND 32 columns of y per step (yoff = extra offset of y in
//     16-byte words), and the dots pattern (column c of x and of y, flat grid, PER pieces per thread)
template <bool NTL, int RUN>
__global__ __launch_bounds__(256) void read_gram2(const u4 *x, const u4 *yv, u4 *out, int64_t ldw, int64_t nw) {
  constexpr int UPC = RUN / 16, CPQ = 256 / UPC, NQ = 32 / CPQ;
  const int tcol = threadIdx.x / UPC, tk = threadIdx.x % UPC;
  const int64_t nsteps = nw / UPC;
  u4 s = {0, 0, 0, 0};
  for (int64_t stp = blockIdx.x; stp < nsteps; stp += gridDim.x) {
    u4 v[2 * NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = ld<NTL, false>(yv + (int64_t)(tcol + q * CPQ) * ldw + stp * UPC + tk);
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[NQ + q] = ld<NTL, false>(x + (int64_t)(tcol + q * CPQ) * ldw + stp * UPC + tk);
#pragma unroll
    for (int q = 0; q < 2 * NQ; ++q) s += v[q];
  }
  if (s.x == 0x12345678u) out[0] = s;
}
template <bool NTL, int PER>
__global__ __launch_bounds__(256) void read_dots2(const u4 *x, const u4 *yv, u4 *out, int64_t ldw, int64_t nw) {
  const u4 *xc = x + blockIdx.y * ldw, *yc = yv + blockIdx.y * ldw;
  const int64_t base = (int64_t)blockIdx.x * PER * 256 + threadIdx.x;
  u4 s = {0, 0, 0, 0};
  u4 a[PER], b[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int64_t r = base + i * 256;
    a[i] = r < nw ? ld<NTL, false>(xc + r) : s;
    b[i] = r < nw ? ld<NTL, false>(yc + r) : s;
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) s += a[i] * b[i];
  if (s.x == 0x12345678u) out[0] = s;
}


__global__ void fill_random(u4 *x, int64_t nw, unsigned seed) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nw; i += (int64_t)gridDim.x * 256) {
    unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    // doubles in [-1, 1): random mantissas, exponents near 0x3FE
    double a = (double)(long long)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
    z = z * 0x9E3779B97F4A7C15ull + 12345;
    double b = (double)(long long)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
    union { double d[2]; u4 v; } c; c.d[0] = a; c.d[1] = b;
    x[i] = c.v;
  }
}


// 13. the wave-private streaming Gram's load pattern: one-wave workgroups, a wave reads NCOL columns x TB bytes per
//     tile (lane l: piece l % (TB/16) of column q * (1024/TB) + l / (TB/16)), SETS register sets in flight
template <bool NTL, int TB, int SETS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void read_stream_like(const u4 *x, const u4 *yv, u4 *out, int64_t ldw, int64_t nw) {
  constexpr int NP = TB / 16, CPL = 64 / NP, NL = 64 / CPL;
  const int lane = threadIdx.x & 63;
  const int p = lane % NP, cl = lane / NP;
  const int64_t ntiles = nw / NP;
  const int64_t G = (int64_t)gridDim.x * WAVES, t0 = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  u4 s = {0, 0, 0, 0};
  u4 ra[NL], rb[NL];
  auto load = [&](int64_t tile, u4 (&r)[NL]) {
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const int sc = q * CPL + cl;
      const u4 *base = sc < 32 ? yv : x;
      r[q] = ld<NTL, false>(base + (int64_t)(sc & 31) * ldw + tile * NP + p);
    }
  };
  auto use = [&](u4 (&r)[NL]) {
#pragma unroll
    for (int q = 0; q < NL; ++q) s += r[q];
  };
  if (SETS == 1) {
    for (int64_t t = t0; t < ntiles; t += G) { load(t, ra); use(ra); }
  } else {
    int64_t t = t0;
    if (t < ntiles) load(t, ra);
    if (t + G < ntiles) load(t + G, rb);
    for (; t + 3 * G < ntiles; t += 2 * G) {
      use(ra); load(t + 2 * G, ra);
      use(rb); load(t + 3 * G, rb);
    }
    if (t < ntiles) use(ra);
    if (t + G < ntiles) use(rb);
    if (t + 2 * G < ntiles) { load(t + 2 * G, ra); use(ra); }
  }
  if (s.x == 0x12345678u) out[0] = s;
}


// 14. row-wise copy in the MFMA fragment shape: one wave instruction = 4 columns x 256 contiguous bytes (lane l: piece
//     l & 15 of column 4 q + (l >> 4)); LDRUN / STRUN choose that shape (true) or 1-KB runs of one column (false)
//     separately for the loads and the stores.  A wave owns 128 rows (1 KB per column) either way.
template <bool NTL, bool NTS, bool LDFRAG, bool STFRAG, int C>
__global__ __launch_bounds__(256) void rowwise_frag(const u4 *x, u4 *y, int64_t ldw, int64_t nw) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t r0 = wave * 64;                          // first 16-byte word of this wave's rows
  if (r0 + 64 > nw) return;
#pragma unroll
  for (int c0 = 0; c0 < C; c0 += 4) {
    u4 v[4];
    // 4 columns x 64 words: either column-at-a-time (instruction u = column c0 + u, lane = word) or fragment shape
    // (instruction u = words 16 u .. 16 u + 15 of the four columns, lane l: column c0 + (l >> 4), word 16 u + (l & 15))
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = LDFRAG ? ld<NTL, NTS>(x + r0 + 16 * u + (lane & 15) + (c0 + (lane >> 4)) * ldw)
                    : ld<NTL, NTS>(x + r0 + lane + (c0 + u) * ldw);
    if (LDFRAG != STFRAG) {
      // (a real kernel would transpose 4 x 4 lane groups here: v_permlane16/32_swap; the probe only measures memory)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (STFRAG) st<NTL, NTS>(y + r0 + 16 * u + (lane & 15) + (c0 + (lane >> 4)) * ldw, v[u]);
      else st<NTL, NTS>(y + r0 + lane + (c0 + u) * ldw, v[u]);
    }
  }
}

template <typename F> static double timed(F f, int reps = 9) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char **argv) {
  const int64_t n = 9938400, m = 32;              // doubles per column (the roofline block, padded)
  const int64_t ldw = n / 2, nw = ldw;            // 16-byte words per column
  const size_t bytes = (size_t)n * m * 8;
  u4 *x, *y;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes));
  CK(hipMemset(x, 1, bytes)); CK(hipMemset(y, 2, bytes));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cu = p.multiProcessorCount;
  const double GB = bytes / 1e9;
  auto report = [&](const char *name, double ms, double gb) { printf("%-58s %7.3f ms  %7.1f GB/s\n", name, ms, gb / ms * 1e3); fflush(stdout); };
  const int64_t tot = nw * m;
  report("copy_flat nt ld+st", timed([&] { hipLaunchKernelGGL((copy_flat<true, true>), dim3((tot + 255) / 256), dim3(256), 0, 0, x, y, tot); }), 2 * GB);
  const dim3 gw((nw / 64 + 3) / 4);
  for (int rep = 0; rep < 2; ++rep) {
    report("row-wise copy, loads 1-KB runs, stores 1-KB runs (nt)", timed([&] { hipLaunchKernelGGL((rowwise_frag<true, true, false, false, 32>), gw, dim3(256), 0, 0, x, y, ldw, nw); }), 2 * GB);
    report("row-wise copy, loads 4 x 256 B,  stores 1-KB runs (nt)", timed([&] { hipLaunchKernelGGL((rowwise_frag<true, true, true, false, 32>), gw, dim3(256), 0, 0, x, y, ldw, nw); }), 2 * GB);
    report("row-wise copy, loads 1-KB runs, stores 4 x 256 B  (nt)", timed([&] { hipLaunchKernelGGL((rowwise_frag<true, true, false, true, 32>), gw, dim3(256), 0, 0, x, y, ldw, nw); }), 2 * GB);
    report("row-wise copy, loads 4 x 256 B,  stores 4 x 256 B  (nt)", timed([&] { hipLaunchKernelGGL((rowwise_frag<true, true, true, true, 32>), gw, dim3(256), 0, 0, x, y, ldw, nw); }), 2 * GB);
    report("row-wise copy, loads 4 x 256 B,  stores 4 x 256 B  (plain)", timed([&] { hipLaunchKernelGGL((rowwise_frag<false, false, true, true, 32>), gw, dim3(256), 0, 0, x, y, ldw, nw); }), 2 * GB);
  }
  return 0;
}

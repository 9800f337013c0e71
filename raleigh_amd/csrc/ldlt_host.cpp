// Symmetric indefinite factorisation  P A P^T = L D L^H  on the host, for the direct shift-invert operator
// (SURVEY 8(f).2).  Host-only translation unit (no device code: compiled as plain C++ so that the dense update can
// carry per-ISA variants).
//
// Reference: raleigh/algebra/sparse_mkl.py:51-119 (SparseSymmetricSolver) -> raleigh/algebra/mkl_wrap.py:354-489
// (class ParDiSo: MKL PARDISO with mtype -2 / -4 real symmetric / Hermitian indefinite, 2 / 4 positive definite;
// analyse = phase 11, factorize = phase 22, solve = phase 33, inertia from iparm[21], iparm[22]).  PARDISO is a
// closed library; what is restated here is the published method of that class of solvers:
//  * fill-reducing ordering by approximate minimum degree on the quotient graph (Amestoy, Davis, Duff 1996), with
//    element absorption and indistinguishable variables merged;
//  * elimination tree, postorder, column counts, fundamental supernodes with relaxed amalgamation;
//  * multifrontal numerical factorisation (Duff, Reid 1983): one dense frontal matrix per supernode, children's
//    Schur complements (contribution blocks) extend-added into the parent;
//  * inside a front, 1 x 1 and 2 x 2 pivots chosen among the FULLY SUMMED variables by threshold partial pivoting
//    (|d| >= u * max off-diagonal of the column over the whole front; the 2 x 2 test of Duff, Reid / MA57), the panel
//    organised like LAPACK's xLASYF (columns brought up to date when examined, one rank-nb update of the trailing
//    matrix per panel).  A variable with no acceptable pivot is DELAYED to the parent's front, so the factors of a
//    saddle-point matrix (zero diagonal block) or of an unluckily shifted one exist and are accurate -- PARDISO
//    perturbs such pivots instead; the root front pivots by Bunch-Kaufman and only an exactly / numerically
//    singular remainder is perturbed (and counted);
//  * the inertia is read off D (signs of the 1 x 1 pivots, determinant and trace of the 2 x 2 ones).
// The solves run on the device: L and L^H become sptrsv operators, D^-1 a tiny kernel (rlh_bdiag_solve, sptrsv.hip).
#include <math.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <complex>
#include <memory>
#include <numeric>
#include <vector>

#include "common.h"

struct rlh_ldlt {
  int dtype;                        // RLH_D or RLH_Z
  int64_t n;
  std::vector<int64_t> lptr;        // L strictly lower, unit diagonal implied, CSR in PIVOT order
  std::vector<int32_t> lidx;
  std::vector<char> lval;
  std::vector<char> d, e;           // D: diagonal, and below-diagonal entry of a 2 x 2 block at its first row (else 0)
  std::vector<int8_t> blk;          // 0: 1 x 1 pivot, 1 / 2: first / second row of a 2 x 2 pivot
  std::vector<int64_t> uptr;        // the same entries by COLUMNS of L = rows of L^H (conjugated), indices ascending
  std::vector<int32_t> uidx;
  std::vector<char> uval;
  std::vector<int64_t> order;       // order[k] = row of A eliminated k-th
  int64_t info[RLH_LDLT_INFO];
};

namespace rlh {
namespace {

typedef std::complex<double> zd;

static inline double cj(double x) { return x; }
static inline zd cj(const zd &x) { return std::conj(x); }
static inline double mag(double x) { return fabs(x); }
static inline double mag(const zd &x) { return sqrt(x.real() * x.real() + x.imag() * x.imag()); }
static inline double re(double x) { return x; }
static inline double re(const zd &x) { return x.real(); }

// ------------------------------------------------------------------ ordering
// Approximate minimum degree on the quotient graph.  ap / ai: symmetric adjacency without the diagonal.
static void minimum_degree(int32_t n, const std::vector<int64_t> &ap, const std::vector<int32_t> &ai,
                           std::vector<int32_t> &perm) {
  std::vector<std::vector<int32_t>> av(n), ae(n), el(n);
  std::vector<int32_t> nv(n, 1), deg(n), status(n, 0), mark(n, 0), wst(n, 0), w(n, 0), esize(n, 0);
  std::vector<int32_t> head(n + 1, -1), nxt(n, -1), prv(n, -1), member_next(n, -1), member_tail(n);
  std::vector<int32_t> seq, lp, dropped;
  std::vector<std::pair<uint32_t, int32_t>> hashed;
  seq.reserve(n);
  for (int32_t i = 0; i < n; ++i) {
    av[i].assign(ai.begin() + ap[i], ai.begin() + ap[i + 1]);
    deg[i] = (int32_t)av[i].size();
    member_tail[i] = i;
  }
  auto insert = [&](int32_t i) {
    const int32_t d = deg[i];
    prv[i] = -1; nxt[i] = head[d];
    if (head[d] >= 0) prv[head[d]] = i;
    head[d] = i;
  };
  auto remove = [&](int32_t i) {
    if (prv[i] >= 0) nxt[prv[i]] = nxt[i]; else head[deg[i]] = nxt[i];
    if (nxt[i] >= 0) prv[nxt[i]] = prv[i];
  };
  for (int32_t i = 0; i < n; ++i) insert(i);
  int32_t stamp = 0, mindeg = 0, nel = 0;
  while (nel < n) {
    while (head[mindeg] < 0) ++mindeg;
    const int32_t p = head[mindeg];
    remove(p);
    ++stamp;
    mark[p] = stamp;
    lp.clear();
    int32_t dlp = 0;
    for (int32_t v : av[p])
      if (status[v] == 0 && mark[v] != stamp) { mark[v] = stamp; lp.push_back(v); dlp += nv[v]; }
    for (int32_t e : ae[p]) {
      if (status[e] != 1) continue;
      for (int32_t v : el[e])
        if (status[v] == 0 && mark[v] != stamp) { mark[v] = stamp; lp.push_back(v); dlp += nv[v]; }
      status[e] = 2;                                   // absorbed into the new element
      std::vector<int32_t>().swap(el[e]);
    }
    status[p] = 1;
    esize[p] = dlp;
    std::vector<int32_t>().swap(av[p]);
    std::vector<int32_t>().swap(ae[p]);
    nel += nv[p];
    seq.push_back(p);
    const int32_t nleft = n - nel;
    // |L_e \ L_p| for every element next to a variable of L_p
    for (int32_t i : lp) {
      remove(i);
      size_t keep = 0;
      for (int32_t e : ae[i]) {
        if (status[e] != 1) continue;
        ae[i][keep++] = e;
        if (wst[e] != stamp) { wst[e] = stamp; w[e] = esize[e]; }
        w[e] -= nv[i];
      }
      ae[i].resize(keep);
    }
    dropped.clear();
    hashed.clear();
    for (int32_t i : lp) {
      int64_t d = 0;
      uint32_t h = 0;
      size_t keep = 0;
      for (int32_t v : av[i])
        if (status[v] == 0 && mark[v] != stamp) { av[i][keep++] = v; d += nv[v]; h += (uint32_t)v; }
      av[i].resize(keep);
      keep = 0;
      for (int32_t e : ae[i]) {
        if (w[e] > 0) { ae[i][keep++] = e; d += w[e]; h += (uint32_t)e; }
        else if (wst[e] == stamp) { wst[e] = -stamp; dropped.push_back(e); }   // all of e lies in L_p: absorbed too
      }
      ae[i].resize(keep);
      ae[i].push_back(p);
      h += (uint32_t)p;
      int64_t nd = d + dlp - nv[i];
      nd = std::min<int64_t>(nd, (int64_t)deg[i] + dlp - nv[i]);
      nd = std::min<int64_t>(nd, nleft - nv[i]);
      deg[i] = (int32_t)std::max<int64_t>(nd, 0);
      hashed.emplace_back(h, i);
    }
    for (int32_t e : dropped) { status[e] = 2; std::vector<int32_t>().swap(el[e]); }
    // indistinguishable variables (same variables and elements next to them) become one
    std::sort(hashed.begin(), hashed.end());
    for (size_t a = 0; a < hashed.size(); ++a) {
      const int32_t i = hashed[a].second;
      if (status[i] != 0) continue;
      bool marked = false;
      for (size_t b = a + 1; b < hashed.size() && hashed[b].first == hashed[a].first; ++b) {
        const int32_t j = hashed[b].second;
        if (status[j] != 0 || av[j].size() != av[i].size() || ae[j].size() != ae[i].size()) continue;
        if (!marked) {
          ++stamp;                                     // (marks of L_p are no longer needed below)
          for (int32_t v : av[i]) mark[v] = stamp;
          for (int32_t e : ae[i]) wst[e] = stamp;
          marked = true;
        }
        bool same = true;
        for (int32_t v : av[j]) if (mark[v] != stamp) { same = false; break; }
        if (same) for (int32_t e : ae[j]) if (wst[e] != stamp) { same = false; break; }
        if (!same) continue;
        nv[i] += nv[j];
        deg[i] = std::max(deg[i] - nv[j], 0);
        nv[j] = 0;
        status[j] = 3;
        member_next[member_tail[i]] = j;
        member_tail[i] = member_tail[j];
        std::vector<int32_t>().swap(av[j]);
        std::vector<int32_t>().swap(ae[j]);
      }
    }
    size_t keep = 0;
    for (int32_t i : lp)
      if (status[i] == 0) {
        lp[keep++] = i;
        insert(i);
        if (deg[i] < mindeg) mindeg = deg[i];
      }
    lp.resize(keep);
    el[p] = lp;
  }
  perm.clear();
  perm.reserve(n);
  for (int32_t p : seq)
    for (int32_t v = p; v >= 0; v = member_next[v]) perm.push_back(v);
}

// ------------------------------------------------------------------ dense update of a front
// C[i, j] -= sum_p L[i, p] * conj(W[j, p]) for j0 <= j < j1, i >= j (lower triangle; the strict upper part of the
// column blocks it touches is scratch).  L: columns of the front (leading dimension f), W: panel copy (ld f).
template <int TI>
static inline __attribute__((always_inline)) void update_block_d(double *A, const double *L, const double *W, int64_t f,
                                                                 int64_t nk, int64_t j0, int64_t j1) {
  int64_t j = j0;
  for (; j + 4 <= j1; j += 4) {
    int64_t i = j;
    for (; i + TI <= f; i += TI) {
      double acc[4][TI];
      for (int c = 0; c < 4; ++c)
        for (int r = 0; r < TI; ++r) acc[c][r] = 0.0;
      for (int64_t p = 0; p < nk; ++p) {
        const double *l = L + i + p * f;
        const double w0 = W[j + p * f], w1 = W[j + 1 + p * f], w2 = W[j + 2 + p * f], w3 = W[j + 3 + p * f];
        for (int r = 0; r < TI; ++r) {
          const double lv = l[r];
          acc[0][r] += lv * w0; acc[1][r] += lv * w1; acc[2][r] += lv * w2; acc[3][r] += lv * w3;
        }
      }
      for (int c = 0; c < 4; ++c) {
        double *dst = A + i + (j + c) * f;
        for (int r = 0; r < TI; ++r) dst[r] -= acc[c][r];
      }
    }
    for (; i < f; ++i) {
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      for (int64_t p = 0; p < nk; ++p) {
        const double lv = L[i + p * f];
        a0 += lv * W[j + p * f]; a1 += lv * W[j + 1 + p * f]; a2 += lv * W[j + 2 + p * f]; a3 += lv * W[j + 3 + p * f];
      }
      A[i + j * f] -= a0; A[i + (j + 1) * f] -= a1; A[i + (j + 2) * f] -= a2; A[i + (j + 3) * f] -= a3;
    }
  }
  for (; j < j1; ++j)
    for (int64_t p = 0; p < nk; ++p) {
      const double w = W[j + p * f];
      if (w == 0.0) continue;
      const double *l = L + p * f;
      double *dst = A + j * f;
      for (int64_t i = j; i < f; ++i) dst[i] -= l[i] * w;
    }
}
__attribute__((target("avx512f"))) static void update_block_d_avx512(double *A, const double *L, const double *W, int64_t f,
                                                                      int64_t nk, int64_t j0, int64_t j1) {
  update_block_d<16>(A, L, W, f, nk, j0, j1);
}
__attribute__((target("avx2,fma"))) static void update_block_d_avx2(double *A, const double *L, const double *W, int64_t f,
                                                                    int64_t nk, int64_t j0, int64_t j1) {
  update_block_d<8>(A, L, W, f, nk, j0, j1);
}
static void update_block_d_base(double *A, const double *L, const double *W, int64_t f, int64_t nk, int64_t j0, int64_t j1) {
  update_block_d<4>(A, L, W, f, nk, j0, j1);
}

// complex: two columns at a time, real and imaginary parts kept apart in the accumulators
template <int TI>
static inline __attribute__((always_inline)) void update_block_z(zd *Az, const zd *Lz, const zd *Wz, int64_t f, int64_t nk,
                                                                 int64_t j0, int64_t j1) {
  double *A = reinterpret_cast<double *>(Az);
  const double *L = reinterpret_cast<const double *>(Lz);
  const double *W = reinterpret_cast<const double *>(Wz);
  int64_t j = j0;
  for (; j + 2 <= j1; j += 2) {
    int64_t i = j;
    for (; i + TI <= f; i += TI) {
      double ar[2][TI], ai[2][TI];
      for (int c = 0; c < 2; ++c)
        for (int r = 0; r < TI; ++r) { ar[c][r] = 0.0; ai[c][r] = 0.0; }
      for (int64_t p = 0; p < nk; ++p) {
        const double *l = L + 2 * (i + p * f);
        // conj(w) = a - i b:  l * conj(w) = (lr a + li b) + i (li a - lr b)
        const double a0 = W[2 * (j + p * f)], b0 = W[2 * (j + p * f) + 1];
        const double a1 = W[2 * (j + 1 + p * f)], b1 = W[2 * (j + 1 + p * f) + 1];
        for (int r = 0; r < TI; ++r) {
          const double lr = l[2 * r], li = l[2 * r + 1];
          ar[0][r] += lr * a0 + li * b0; ai[0][r] += li * a0 - lr * b0;
          ar[1][r] += lr * a1 + li * b1; ai[1][r] += li * a1 - lr * b1;
        }
      }
      for (int c = 0; c < 2; ++c) {
        double *dst = A + 2 * (i + (j + c) * f);
        for (int r = 0; r < TI; ++r) { dst[2 * r] -= ar[c][r]; dst[2 * r + 1] -= ai[c][r]; }
      }
    }
    for (; i < f; ++i)
      for (int c = 0; c < 2; ++c) {
        zd s = 0;
        for (int64_t p = 0; p < nk; ++p) s += Lz[i + p * f] * std::conj(Wz[j + c + p * f]);
        Az[i + (j + c) * f] -= s;
      }
  }
  for (; j < j1; ++j)
    for (int64_t p = 0; p < nk; ++p) {
      const zd w = std::conj(Wz[j + p * f]);
      if (w == zd(0)) continue;
      const zd *l = Lz + p * f;
      zd *dst = Az + j * f;
      for (int64_t i = j; i < f; ++i) dst[i] -= l[i] * w;
    }
}
__attribute__((target("avx512f"))) static void update_block_z_avx512(zd *A, const zd *L, const zd *W, int64_t f, int64_t nk,
                                                                      int64_t j0, int64_t j1) {
  update_block_z<8>(A, L, W, f, nk, j0, j1);
}
__attribute__((target("avx2,fma"))) static void update_block_z_avx2(zd *A, const zd *L, const zd *W, int64_t f, int64_t nk,
                                                                    int64_t j0, int64_t j1) {
  update_block_z<4>(A, L, W, f, nk, j0, j1);
}
static void update_block_z_base(zd *A, const zd *L, const zd *W, int64_t f, int64_t nk, int64_t j0, int64_t j1) {
  update_block_z<2>(A, L, W, f, nk, j0, j1);
}

static int cpu_level() {
  static int level = -1;
  if (level < 0) {
    __builtin_cpu_init();
    const int cap = env_int("RLH_LDLT_ISA", 2);
    level = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
    if (level > cap) level = cap;
  }
  return level;
}
static void update_block(double *A, const double *L, const double *W, int64_t f, int64_t nk, int64_t j0, int64_t j1) {
  switch (cpu_level()) {
    case 2: update_block_d_avx512(A, L, W, f, nk, j0, j1); break;
    case 1: update_block_d_avx2(A, L, W, f, nk, j0, j1); break;
    default: update_block_d_base(A, L, W, f, nk, j0, j1);
  }
}
static void update_block(zd *A, const zd *L, const zd *W, int64_t f, int64_t nk, int64_t j0, int64_t j1) {
  switch (cpu_level()) {
    case 2: update_block_z_avx512(A, L, W, f, nk, j0, j1); break;
    case 1: update_block_z_avx2(A, L, W, f, nk, j0, j1); break;
    default: update_block_z_base(A, L, W, f, nk, j0, j1);
  }
}

// trailing update after a panel: columns [k1, f) of the front with the pivots [kp, k1)
template <typename T> static void trailing_update(T *A, const T *W, int64_t f, int64_t kp, int64_t k1) {
  const int64_t nk = k1 - kp, cols = f - k1;
  if (nk <= 0 || cols <= 0) return;
  const T *L = A + kp * f;
  const int64_t chunk = 32;
  const int64_t nchunks = (cols + chunk - 1) / chunk;
  const double work = (double)cols * (double)cols * (double)nk;
  if (work < 4e7 || nchunks < 4) {
    update_block(A, L, W, f, nk, k1, f);
    return;
  }
  std::atomic<int64_t> next(0);
  host_parallel((int)std::min<int64_t>(nchunks / 2, 16), [&](int, int) {
    for (;;) {
      const int64_t c = next.fetch_add(1);
      if (c >= nchunks) break;
      update_block(A, L, W, f, nk, k1 + c * chunk, std::min(f, k1 + (c + 1) * chunk));
    }
  });
}

// ------------------------------------------------------------------ one front
struct PivotStats {
  int64_t two_by_two = 0, delayed = 0, perturbed = 0, negative = 0, positive = 0, forced = 0;
  double t_update = 0, t_assemble = 0, t_pivot = 0, t_store = 0;    // (RLH_LDLT_VERBOSE: where the fronts' time goes)
};
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename T> struct FrontKernel {
  T *A;                 // f x f, column-major, lower triangle live
  T *W;                 // f x nb panel copy (updated columns before scaling)
  T *c1, *c2;           // candidate columns
  int32_t *idx;         // variable of every front position
  int64_t f, nF;
  int64_t k, kp;        // eliminated so far; start of the open panel
  const double *colmax; // largest entry of A in the column of every variable: what "numerically zero" is measured against
  double prel;
  int64_t *pivots;      // pivots accepted so far (all fronts), and per variable: when it may be examined again, when it
  int64_t *retry_at, *tried_at;   // was last examined, how often it has been refused
  int8_t *refused;
  static constexpr int64_t nb = 64;
  double floor_of(int64_t t) const { return prel * colmax[idx[t]]; }

  // symmetric column of position t over positions [k, f), brought up to date with the open panel
  void getcol(int64_t t, T *c) const {
    for (int64_t i = k; i < t; ++i) c[i] = cj(A[t + i * f]);
    c[t] = T(re(A[t + t * f]));
    for (int64_t i = t + 1; i < f; ++i) c[i] = A[i + t * f];
    for (int64_t p = kp; p < k; ++p) {
      const T w = cj(W[t + (p - kp) * f]);
      if (w == T(0)) continue;
      const T *l = A + p * f;
      for (int64_t i = k; i < f; ++i) c[i] -= l[i] * w;
    }
    c[t] = T(re(c[t]));
  }
  // positions p < q change places (rows and columns of the trailing matrix, rows of L and of the panel copy)
  void swap(int64_t p, int64_t q) {
    if (p == q) return;
    if (p > q) std::swap(p, q);
    for (int64_t j = 0; j < p; ++j) std::swap(A[p + j * f], A[q + j * f]);
    std::swap(A[p + p * f], A[q + q * f]);
    for (int64_t i = p + 1; i < q; ++i) {
      const T tmp = A[i + p * f];
      A[i + p * f] = cj(A[q + i * f]);
      A[q + i * f] = cj(tmp);
    }
    A[q + p * f] = cj(A[q + p * f]);
    for (int64_t i = q + 1; i < f; ++i) std::swap(A[i + p * f], A[i + q * f]);
    for (int64_t c = 0; c < k - kp; ++c) std::swap(W[p + c * f], W[q + c * f]);
    std::swap(idx[p], idx[q]);
  }
  double *t_update = nullptr;
  void close_panel() {
    const double t0 = t_update ? now_s() : 0.0;
    trailing_update(A, W, f, kp, k);
    if (t_update) *t_update += now_s() - t0;
    kp = k;
  }
  void pivot1(const T *c, PivotStats &st) {
    const double d = re(c[k]);
    T *w = W + (k - kp) * f;
    T *a = A + k * f;
    const double inv = 1.0 / d;
    for (int64_t i = k; i < f; ++i) { w[i] = c[i]; a[i] = c[i] * inv; }
    a[k] = T(d);
    (d < 0 ? st.negative : st.positive) += 1;
    k += 1;
  }
  void pivot2(const T *ca, const T *cb, PivotStats &st) {
    const double d11 = re(ca[k]), d22 = re(cb[k + 1]);
    const T d21 = ca[k + 1];
    const double det = d11 * d22 - mag(d21) * mag(d21);
    const double inv = 1.0 / det;
    T *w0 = W + (k - kp) * f, *w1 = w0 + f;
    T *a0 = A + k * f, *a1 = a0 + f;
    for (int64_t i = k; i < f; ++i) {
      w0[i] = ca[i]; w1[i] = cb[i];
      a0[i] = (ca[i] * d22 - cb[i] * d21) * inv;
      a1[i] = (cb[i] * d11 - ca[i] * cj(d21)) * inv;
    }
    a0[k] = T(d11); a0[k + 1] = d21; a1[k + 1] = T(d22);
    if (det < 0) { st.negative += 1; st.positive += 1; }
    else (d11 + d22 < 0 ? st.negative : st.positive) += 2;
    st.two_by_two += 1;
    k += 2;
  }

  // returns the number of pivots; kind[0 .. npiv) = 0 / 1 / 2 as in rlh_ldlt::blk
  int64_t run(bool root, double u, std::vector<int8_t> &kind, PivotStats &st) {
    k = kp = 0;
    kind.clear();
    const double alpha = 0.6403882032022076;            // (1 + sqrt(17)) / 8
    while (k < nF) {
      if (k - kp >= nb - 1) close_panel();
      const int64_t end = nF;
      int npv = 0;
      // a variable that was refused is not examined again at once (its column costs a pass over the front): only after
      // 2, 4, ... 64 further pivots -- or when nothing else is left (second pass), so that "no pivot" is never declared
      // on the strength of a skipped candidate
      for (int pass = 0; pass < 2 && !npv; ++pass)
      for (int64_t t = k; t < end && !npv; ++t) {
        const int32_t var = idx[t];
        if (pass == 0 ? retry_at[var] > *pivots : (retry_at[var] <= *pivots || tried_at[var] == *pivots)) continue;
        tried_at[var] = *pivots;
        if (refused[var] < 6) ++refused[var];
        retry_at[var] = *pivots + ((int64_t)1 << refused[var]);          // (stands if this candidate is refused)
        getcol(t, c1);
        const double dtt = mag(c1[t]);
        double gam = 0, lam = 0;
        int64_t r = -1;
        for (int64_t i = k; i < end; ++i)
          if (i != t) { const double v = mag(c1[i]); if (v > lam) { lam = v; r = i; } }
        gam = lam;
        for (int64_t i = end; i < f; ++i) gam = std::max(gam, mag(c1[i]));
        if (dtt > floor_of(t) && dtt >= u * gam) {
          swap(k, t); std::swap(c1[k], c1[t]);
          pivot1(c1, st); npv = 1;
          break;
        }
        if (r < 0 || lam <= 0.0) continue;
        getcol(r, c2);
        const double drr = mag(c2[r]);
        double gr_all = 0, gt2 = 0, gr2 = 0;
        for (int64_t i = k; i < f; ++i) {
          if (i != r) gr_all = std::max(gr_all, mag(c2[i]));
          if (i != r && i != t) { gt2 = std::max(gt2, mag(c1[i])); gr2 = std::max(gr2, mag(c2[i])); }
        }
        if (drr > floor_of(r) && drr >= u * gr_all) {
          swap(k, r); std::swap(c2[k], c2[r]);
          pivot1(c2, st); npv = 1;
          break;
        }
        const double d11 = re(c1[t]), d22 = re(c2[r]);
        const double det = d11 * d22 - lam * lam, adet = fabs(det);
        if (adet > prel * (fabs(d11 * d22) + lam * lam) && (fabs(d22) * gt2 + lam * gr2) * u <= adet && (lam * gt2 + fabs(d11) * gr2) * u <= adet) {
          int64_t rr = r;
          swap(k, t); std::swap(c1[k], c1[t]); std::swap(c2[k], c2[t]);
          if (rr == k) rr = t;
          swap(k + 1, rr); std::swap(c1[k + 1], c1[rr]); std::swap(c2[k + 1], c2[rr]);
          pivot2(c1, c2, st); npv = 2;
          break;
        }
      }
      if (!npv) {
        if (!root) break;                                // the rest is delayed to the parent
        // root: every variable is fully summed -- Bunch-Kaufman on the first one left always ends in a pivot
        st.forced += 1;
        const int64_t t = k;
        getcol(t, c1);
        double lam = 0;
        int64_t r = -1;
        for (int64_t i = k; i < f; ++i)
          if (i != t) { const double v = mag(c1[i]); if (v > lam) { lam = v; r = i; } }
        const double dtt = mag(c1[t]);
        const double perturb = std::max(floor_of(t), 1e-300);
        if (lam <= perturb && dtt <= perturb) {          // numerically zero column: perturbed pivot, counted
          c1[t] = T(re(c1[t]) < 0 ? -perturb : perturb);
          st.perturbed += 1;
          pivot1(c1, st); npv = 1;
        } else if (dtt >= alpha * lam) {
          pivot1(c1, st); npv = 1;
        } else {
          getcol(r, c2);
          double sig = 0;
          for (int64_t i = k; i < f; ++i) if (i != r) sig = std::max(sig, mag(c2[i]));
          if (dtt * sig >= alpha * lam * lam) {
            pivot1(c1, st); npv = 1;
          } else if (mag(c2[r]) >= alpha * sig) {
            swap(k, r); std::swap(c2[k], c2[r]);
            pivot1(c2, st); npv = 1;
          } else {
            swap(k + 1, r); std::swap(c1[k + 1], c1[r]); std::swap(c2[k + 1], c2[r]);
            pivot2(c1, c2, st); npv = 2;
          }
        }
      }
      if (npv == 1) kind.push_back(0);
      else { kind.push_back(1); kind.push_back(2); }
      *pivots += npv;
    }
    close_panel();
    st.delayed += nF - k;
    return k;
  }
};

// ------------------------------------------------------------------ the factorisation
// The contribution blocks live on a stack (a parent takes the blocks of its children, which are on top, and leaves its own):
// memory for them is bumped off a list of chunks that are kept until the factorisation ends -- the C library would map and
// unmap every block of more than 128 KB, i.e. page-fault it in again each time.
struct BlockStack {
  std::vector<std::unique_ptr<char[]>> chunks;
  std::vector<size_t> cap;
  size_t cur = 0, off = 0;
  struct Mark { size_t cur, off; };
  Mark mark() const { return Mark{cur, off}; }
  void release(Mark m) { cur = m.cur; off = m.off; }
  void *take(size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    while (cur < chunks.size() && off + bytes > cap[cur]) { ++cur; off = 0; }
    if (cur == chunks.size()) {
      const size_t c = std::max<size_t>(bytes, (size_t)64 << 20);
      chunks.emplace_back(new char[c]);
      cap.push_back(c);
      off = 0;
    }
    void *p = chunks[cur].get() + off;
    off += bytes;
    return p;
  }
};

template <typename T> struct Contribution {
  int32_t parent;                  // supernode that takes it
  int64_t ndelayed;                // leading variables that are still to be eliminated
  std::vector<int32_t> idx;
  T *a;                            // size x size, lower triangle live (BlockStack memory)
  BlockStack::Mark mark;           // the stack before this block was taken
};

template <typename T>
static int ldlt_factor(int64_t n64, const int64_t *indptr, const int32_t *indices, const T *values, const int64_t *user_perm,
                       double u, double perturb_rel, rlh_ldlt *out) {
  const int32_t n = (int32_t)n64;
  out->n = n;
  for (int i = 0; i < RLH_LDLT_INFO; ++i) out->info[i] = 0;
  out->lptr.assign((size_t)n + 1, 0);
  out->order.resize(n);
  out->blk.assign(n, 0);
  out->d.assign((size_t)n * sizeof(T), 0);
  out->e.assign((size_t)n * sizeof(T), 0);
  if (n == 0) return 0;

  // ---- pattern of A + A^T without the diagonal (entries below the diagonal of the input are ignored)
  std::vector<int64_t> ap((size_t)n + 1, 0);
  double amax = 0;
  for (int32_t r = 0; r < n; ++r)
    for (int64_t e = indptr[r]; e < indptr[r + 1]; ++e) {
      const int32_t c = indices[e];
      if (c < r) continue;
      const double v = mag(values[e]);
      if (!(v <= 1.79e308)) amax = v;                  // (a NaN would lose every comparison of a running maximum)
      else if (std::isfinite(amax)) amax = std::max(amax, v);
      if (c > r) { ++ap[r + 1]; ++ap[c + 1]; }
    }
  for (int32_t i = 0; i < n; ++i) ap[i + 1] += ap[i];
  std::vector<int32_t> ai((size_t)ap[n]);
  {
    std::vector<int64_t> fill(ap.begin(), ap.end() - 1);
    for (int32_t r = 0; r < n; ++r)
      for (int64_t e = indptr[r]; e < indptr[r + 1]; ++e) {
        const int32_t c = indices[e];
        if (c > r) { ai[fill[r]++] = c; ai[fill[c]++] = r; }
      }
    // duplicates in the input would only repeat a neighbour: harmless for the ordering, summed at assembly
  }
  if (!std::isfinite(amax)) { set_error("rlh_ldlt_factor: the matrix has a non-finite entry"); return 1; }

  const bool verbose = env_int("RLH_LDLT_VERBOSE", 0) != 0;
  auto clock0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!verbose) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "rlh_ldlt_factor: %-22s %8.3f s\n", what, std::chrono::duration<double>(now - clock0).count());
    clock0 = now;
  };
  lap("pattern");
  // ---- ordering
  std::vector<int32_t> perm;                 // perm[new] = old
  if (user_perm) {
    perm.resize(n);
    std::vector<char> seen(n, 0);
    for (int32_t i = 0; i < n; ++i) {
      const int64_t v = user_perm[i];
      if (v < 0 || v >= n || seen[v]) { set_error("rlh_ldlt_factor: perm is not a permutation"); return 1; }
      seen[v] = 1;
      perm[i] = (int32_t)v;
    }
  } else {
    minimum_degree(n, ap, ai, perm);
  }
  std::vector<int32_t> iperm(n);
  for (int32_t i = 0; i < n; ++i) iperm[perm[i]] = i;
  lap("ordering");

  // ---- elimination tree of the permuted matrix, then postorder
  std::vector<int32_t> parent(n, -1);
  auto etree = [&]() {
    std::vector<int32_t> anc(n, -1);
    for (int32_t k = 0; k < n; ++k) {
      parent[k] = -1;
      const int32_t old = perm[k];
      for (int64_t e = ap[old]; e < ap[old + 1]; ++e) {
        int32_t i = iperm[ai[e]];
        while (i >= 0 && i < k) {
          const int32_t next = anc[i];
          anc[i] = k;
          if (next < 0) parent[i] = k;
          i = next;
        }
      }
    }
  };
  etree();
  {
    std::vector<int32_t> head(n, -1), next(n, -1), post(n), stack;
    for (int32_t j = n - 1; j >= 0; --j)
      if (parent[j] >= 0) { next[j] = head[parent[j]]; head[parent[j]] = j; }
    int32_t count = 0;
    for (int32_t root = 0; root < n; ++root) {
      if (parent[root] >= 0) continue;
      stack.push_back(root);
      while (!stack.empty()) {
        const int32_t j = stack.back();
        const int32_t c = head[j];
        if (c >= 0) { head[j] = next[c]; stack.push_back(c); }
        else { post[j] = count++; stack.pop_back(); }
      }
    }
    std::vector<int32_t> np(n);
    for (int32_t j = 0; j < n; ++j) np[post[j]] = perm[j];
    perm.swap(np);
    for (int32_t i = 0; i < n; ++i) iperm[perm[i]] = i;
    etree();
  }

  // ---- column counts of L (row subtrees)
  std::vector<int64_t> cnt(n, 1);
  {
    std::vector<int32_t> flag(n, -1);
    for (int32_t i = 0; i < n; ++i) {
      flag[i] = i;
      const int32_t old = perm[i];
      for (int64_t e = ap[old]; e < ap[old + 1]; ++e) {
        int32_t j = iperm[ai[e]];
        if (j >= i) continue;
        while (flag[j] != i) { flag[j] = i; ++cnt[j]; j = parent[j]; }
      }
    }
  }

  // ---- supernodes: fundamental, then relaxed amalgamation of a last child into its parent
  std::vector<int32_t> sn_first;             // first column of every supernode (+ n at the end)
  {
    struct Sn { int32_t first, last; int64_t nnz, zeros; };
    std::vector<Sn> st;
    const bool relax = env_int("RLH_LDLT_RELAX", 1) != 0;
    int32_t j = 0;
    while (j < n) {
      int32_t l = j;
      int64_t nnz = cnt[j];
      while (l + 1 < n && parent[l] == l + 1 && cnt[l] == cnt[l + 1] + 1) { ++l; nnz += cnt[l]; }
      Sn cur = {j, l, nnz, 0};
      while (relax && !st.empty()) {
        const Sn &top = st.back();
        const int32_t pt = parent[top.last];
        if (pt < cur.first || pt > cur.last) break;
        // columns of top get the structure of the merged front
        const int64_t below = cnt[cur.last] - 1;                 // rows below the merged pivot block
        int64_t merged_nnz = 0;
        const int64_t nc = cur.last - top.first + 1;
        merged_nnz = nc * (nc + 1) / 2 + nc * below;
        const int64_t zeros = merged_nnz - (top.nnz - top.zeros) - (cur.nnz - cur.zeros);
        const double z = (double)zeros / (double)merged_nnz;
        const bool ok = nc <= 4 || (nc <= 16 && z < 0.8) || (nc <= 48 && z < 0.1) || z < 0.05;
        if (!ok) break;
        cur.first = top.first;
        cur.nnz = merged_nnz;
        cur.zeros = zeros;
        st.pop_back();
      }
      st.push_back(cur);
      j = l + 1;
    }
    for (const Sn &s : st) sn_first.push_back(s.first);
    sn_first.push_back(n);
  }
  const int32_t nsn = (int32_t)sn_first.size() - 1;
  std::vector<int32_t> sn_of(n);
  for (int32_t s = 0; s < nsn; ++s)
    for (int32_t j = sn_first[s]; j < sn_first[s + 1]; ++j) sn_of[j] = s;

  // ---- permuted lower triangle by columns: column j holds rows i >= j
  std::vector<int64_t> cp((size_t)n + 1, 0);
  for (int32_t r = 0; r < n; ++r)
    for (int64_t e = indptr[r]; e < indptr[r + 1]; ++e) {
      const int32_t c = indices[e];
      if (c < r) continue;
      ++cp[std::min(iperm[r], iperm[c]) + 1];
    }
  for (int32_t i = 0; i < n; ++i) cp[i + 1] += cp[i];
  std::vector<int32_t> ci((size_t)cp[n]);
  std::vector<T> cv((size_t)cp[n]);
  {
    std::vector<int64_t> fill(cp.begin(), cp.end() - 1);
    for (int32_t r = 0; r < n; ++r)
      for (int64_t e = indptr[r]; e < indptr[r + 1]; ++e) {
        const int32_t c = indices[e];
        if (c < r) continue;
        const int32_t a = iperm[r], b = iperm[c];      // permuted entry (a, b) = values[e]
        if (a >= b) { const int64_t q = fill[b]++; ci[q] = a; cv[q] = (a == b) ? T(re(values[e])) : values[e]; }
        else { const int64_t q = fill[a]++; ci[q] = b; cv[q] = cj(values[e]); }
      }
  }
  std::vector<int64_t>().swap(ap);
  std::vector<int32_t>().swap(ai);
  std::vector<double> colmax(n, 0.0);
  for (int32_t j = 0; j < n; ++j)
    for (int64_t e = cp[j]; e < cp[j + 1]; ++e) {
      const double v = mag(cv[e]);
      colmax[j] = std::max(colmax[j], v);
      colmax[ci[e]] = std::max(colmax[ci[e]], v);
    }

  lap("tree, counts, supernodes");
  // ---- numerical factorisation
  std::vector<Contribution<T>> stack;
  BlockStack blocks;
  std::vector<int64_t> colptr;               // L by pivot columns: rows as variables (postordered numbering)
  std::vector<int32_t> lrow;
  std::vector<T> lval;
  std::vector<int32_t> seq;                  // variables in pivot order
  std::vector<T> dd((size_t)n), de((size_t)n, T(0));
  std::vector<int8_t> &blk = out->blk;
  {
    int64_t est = 0;
    for (int32_t j = 0; j < n; ++j) est += cnt[j] - 1;
    lrow.reserve((size_t)(est + est / 8));
    lval.reserve((size_t)(est + est / 8));
  }
  colptr.reserve((size_t)n + 1);
  colptr.push_back(0);
  seq.reserve(n);
  std::vector<int32_t> loc(n, -1), idx, rows;
  std::vector<T> front, W, c1, c2;
  std::vector<int8_t> kind;
  PivotStats st;
  int64_t pivots_done = 0;
  std::vector<int64_t> retry_at((size_t)n, 0), tried_at((size_t)n, -1);
  std::vector<int8_t> refused((size_t)n, 0);
  int64_t max_front = 0;
  double flops = 0;
  for (int32_t s = 0; s < nsn; ++s) {
    const int32_t c0 = sn_first[s], c1col = sn_first[s + 1];
    const int32_t last = c1col - 1;
    const int32_t ps = parent[last] >= 0 ? sn_of[parent[last]] : -1;
    size_t first_child = stack.size();
    while (first_child > 0 && stack[first_child - 1].parent == s) --first_child;
    // variables: delayed ones of the children, the supernode's own columns, then the rows below
    idx.clear();
    for (size_t c = first_child; c < stack.size(); ++c)
      for (int64_t q = 0; q < stack[c].ndelayed; ++q) idx.push_back(stack[c].idx[q]);
    for (int32_t j = c0; j < c1col; ++j) idx.push_back(j);
    const int64_t nF = (int64_t)idx.size();
    for (int64_t q = 0; q < nF; ++q) loc[idx[q]] = (int32_t)q;
    rows.clear();
    for (int32_t j = c0; j < c1col; ++j)
      for (int64_t e = cp[j]; e < cp[j + 1]; ++e) {
        const int32_t i = ci[e];
        if (loc[i] < 0) { loc[i] = 0; rows.push_back(i); }
      }
    for (size_t c = first_child; c < stack.size(); ++c)
      for (size_t q = (size_t)stack[c].ndelayed; q < stack[c].idx.size(); ++q) {
        const int32_t i = stack[c].idx[q];
        if (loc[i] < 0) { loc[i] = 0; rows.push_back(i); }
      }
    std::sort(rows.begin(), rows.end());
    for (int32_t i : rows) { loc[i] = (int32_t)idx.size(); idx.push_back(i); }
    const int64_t f = (int64_t)idx.size();
    max_front = std::max(max_front, f);
    const double t_a = verbose ? now_s() : 0.0;
    if ((int64_t)front.size() < f * f) front.resize((size_t)(f * f));
    T *A = front.data();
    for (int64_t q = 0; q < f; ++q) memset((void *)(A + q * f + q), 0, (size_t)(f - q) * sizeof(T));   // (the lower triangle is what is used)
    for (int32_t j = c0; j < c1col; ++j) {
      const int64_t lj = loc[j];
      for (int64_t e = cp[j]; e < cp[j + 1]; ++e) A[loc[ci[e]] + lj * f] += cv[e];
    }
    for (size_t c = first_child; c < stack.size(); ++c) {
      const Contribution<T> &cb = stack[c];
      const int64_t m = (int64_t)cb.idx.size();
      for (int64_t q = 0; q < m; ++q) {
        const int64_t lq = loc[cb.idx[q]];
        const T *src = cb.a + q * m;
        for (int64_t p = q; p < m; ++p) {
          const int64_t lp = loc[cb.idx[p]];
          if (lp >= lq) A[lp + lq * f] += src[p];
          else A[lq + lp * f] += cj(src[p]);
        }
      }
    }
    if (first_child < stack.size()) blocks.release(stack[first_child].mark);
    stack.resize(first_child);
    const double t_b = verbose ? now_s() : 0.0;
    st.t_assemble += t_b - t_a;
    // pivots
    const int64_t nbw = FrontKernel<T>::nb;
    if ((int64_t)W.size() < f * nbw) W.resize((size_t)(f * nbw));
    if ((int64_t)c1.size() < f) { c1.resize((size_t)f); c2.resize((size_t)f); }
    FrontKernel<T> fk;
    fk.A = A; fk.W = W.data(); fk.c1 = c1.data(); fk.c2 = c2.data(); fk.idx = idx.data(); fk.f = f; fk.nF = nF;
    fk.colmax = colmax.data(); fk.prel = perturb_rel;
    fk.pivots = &pivots_done; fk.retry_at = retry_at.data(); fk.tried_at = tried_at.data(); fk.refused = refused.data();
    if (verbose) fk.t_update = &st.t_update;
    const int64_t npiv = fk.run(ps < 0, u, kind, st);
    const double t_c = verbose ? now_s() : 0.0;
    st.t_pivot += t_c - t_b;
    flops += (double)npiv * (double)f * (double)f;
    for (int64_t k = 0; k < npiv; ++k) {
      const int64_t pos = (int64_t)seq.size();
      seq.push_back(idx[k]);
      blk[pos] = kind[k];
      dd[pos] = A[k + k * f];
      int64_t from = k + 1;
      if (kind[k] == 1) { de[pos] = A[k + 1 + k * f]; from = k + 2; }
      const T *col = A + k * f;
      for (int64_t i = from; i < f; ++i)
        if (col[i] != T(0)) { lrow.push_back(idx[i]); lval.push_back(col[i]); }
      colptr.push_back((int64_t)lrow.size());
    }
    for (int64_t q = 0; q < f; ++q) loc[idx[q]] = -1;
    // what is left goes to the parent
    const int64_t m = f - npiv;
    if (ps >= 0 && m > 0) {
      stack.emplace_back();
      Contribution<T> &cb = stack.back();
      cb.parent = ps;
      cb.ndelayed = nF - npiv;
      cb.idx.assign(idx.begin() + npiv, idx.end());
      cb.mark = blocks.mark();
      cb.a = static_cast<T *>(blocks.take((size_t)(m * m) * sizeof(T)));
      for (int64_t q = 0; q < m; ++q) {
        const T *src = A + (npiv + q) * f + npiv;
        T *dst = cb.a + q * m;
        for (int64_t p = q; p < m; ++p) dst[p] = src[p];
      }
    }
    if (verbose) st.t_store += now_s() - t_c;
  }
  if (verbose)
    fprintf(stderr, "rlh_ldlt_factor:   fronts: assembly %.3f s, pivots + panels %.3f s (of which trailing updates %.3f s), factor columns + contribution blocks %.3f s\n",
            st.t_assemble, st.t_pivot, st.t_update, st.t_store);
  if ((int64_t)seq.size() != n) { set_error("rlh_ldlt_factor: internal error, %lld of %d pivots", (long long)seq.size(), n); return 1; }
  for (const T &v : lval)
    if (!std::isfinite(re(v))) { set_error("rlh_ldlt_factor: the factorisation broke down (non-finite factor entry)"); return 1; }

  lap("fronts");
  // ---- L by rows in pivot order
  std::vector<int32_t> pos(n);
  for (int32_t k = 0; k < n; ++k) pos[seq[k]] = k;
  const int64_t nnz = (int64_t)lrow.size();
  std::vector<int64_t> &lptr = out->lptr;
  out->lidx.resize((size_t)nnz);
  out->lval.resize((size_t)nnz * sizeof(T));
  T *ov = reinterpret_cast<T *>(out->lval.data());
  {
    // columns -> rows on the host threads: thread t owns a range of pivot columns (equal shares of the entries), counts
    // its entries per row, the offsets of a row are the prefix over the threads in order -- every row's entries come out in
    // ascending column order whatever the thread count (as many threads as 256 MB of counters allow)
    int nt = nnz > 4000000 ? (int)std::max<int64_t>(1, std::min<int64_t>(std::min(16, host_threads()), ((int64_t)64 << 20) / ((int64_t)n + 1))) : 1;
    std::vector<int32_t> cut((size_t)nt + 1, n);
    cut[0] = 0;
    for (int t = 1; t < nt; ++t)
      cut[t] = (int32_t)(std::lower_bound(colptr.begin(), colptr.end(), nnz * t / nt) - colptr.begin());
    for (int t = 1; t <= nt; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    cut[nt] = n;
    std::vector<std::vector<int32_t>> cnt((size_t)nt);
    host_parallel(nt, [&](int t, int threads) {
      for (int q = t; q < nt; q += threads) {
        cnt[q].assign((size_t)n, 0);
        for (int64_t e = colptr[cut[q]]; e < colptr[cut[q + 1]]; ++e) ++cnt[q][pos[lrow[e]]];
      }
    });
    for (int32_t i = 0; i < n; ++i) {
      int64_t tot = 0;
      for (int t = 0; t < nt; ++t) tot += cnt[t][i];
      lptr[i + 1] = lptr[i] + tot;
    }
    host_parallel(nt, [&](int t, int threads) {
      for (int q = t; q < nt; q += threads) {
        std::vector<int64_t> fill((size_t)n);
        for (int32_t i = 0; i < n; ++i) {
          int64_t off = lptr[i];
          for (int u = 0; u < q; ++u) off += cnt[u][i];
          fill[i] = off;
        }
        for (int32_t k = cut[q]; k < cut[q + 1]; ++k)
          for (int64_t e = colptr[k]; e < colptr[k + 1]; ++e) {
            const int64_t w = fill[pos[lrow[e]]]++;
            out->lidx[w] = k;
            ov[w] = lval[e];
          }
      }
    });
  }
  {
    // L^H by rows = L by columns, which is how the factorisation produced it: positions instead of variables, conjugated,
    // every row sorted (within a front the pivot order is not the variables' order)
    out->uptr.assign(colptr.begin(), colptr.end());
    out->uidx.resize((size_t)nnz);
    out->uval.resize((size_t)nnz * sizeof(T));
    T *uv = reinterpret_cast<T *>(out->uval.data());
    std::atomic<int32_t> next_col(0);
    host_parallel(nnz > 4000000 ? 16 : 1, [&](int, int) {
      std::vector<std::pair<int32_t, T>> row;
      for (;;) {
        const int32_t k0 = next_col.fetch_add(256);
        if (k0 >= n) break;
        for (int32_t k = k0; k < std::min(n, k0 + 256); ++k) {
          const int64_t e0 = colptr[k], e1 = colptr[k + 1];
          bool sorted = true;
          for (int64_t e = e0; e < e1; ++e) {
            out->uidx[(size_t)e] = pos[lrow[e]];
            uv[e] = cj(lval[e]);
            sorted = sorted && (e == e0 || out->uidx[(size_t)e - 1] < out->uidx[(size_t)e]);
          }
          if (sorted) continue;
          row.clear();
          for (int64_t e = e0; e < e1; ++e) row.emplace_back(out->uidx[(size_t)e], uv[e]);
          std::sort(row.begin(), row.end(), [](const std::pair<int32_t, T> &a, const std::pair<int32_t, T> &b) { return a.first < b.first; });
          for (int64_t e = e0; e < e1; ++e) { out->uidx[(size_t)e] = row[(size_t)(e - e0)].first; uv[e] = row[(size_t)(e - e0)].second; }
        }
      }
    });
  }
  memcpy(out->d.data(), dd.data(), (size_t)n * sizeof(T));
  memcpy(out->e.data(), de.data(), (size_t)n * sizeof(T));
  for (int32_t k = 0; k < n; ++k) out->order[k] = perm[seq[k]];
  lap("factor by rows");
  out->info[0] = nnz;
  out->info[1] = st.negative;
  out->info[2] = st.positive;
  out->info[3] = st.perturbed;
  out->info[4] = st.two_by_two;
  out->info[5] = st.delayed;
  out->info[6] = max_front;
  out->info[7] = nsn;
  out->info[8] = (int64_t)std::min(flops, 9e18);
  out->info[9] = st.forced;
  return 0;
}

}  // namespace
}  // namespace rlh

using namespace rlh;

extern "C" {

int rlh_ldlt_factor(rlh_ldlt_t *out, int dtype, int64_t n, const int64_t *indptr, const int32_t *indices, const void *values,
                    const int64_t *perm, double pivot_threshold, double perturb) {
  RLH_REQUIRE(out != nullptr, "rlh_ldlt_factor: null handle pointer");
  *out = nullptr;
  RLH_REQUIRE(dtype == RLH_D || dtype == RLH_Z, "rlh_ldlt_factor: the factorisation runs in double / complex double");
  RLH_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && indptr && (indptr[n] == 0 || (indices && values)), "rlh_ldlt_factor: bad matrix");
  RLH_REQUIRE(pivot_threshold >= 0.0 && pivot_threshold <= 0.5, "rlh_ldlt_factor: the pivot threshold must lie in [0, 0.5]");
  RLH_REQUIRE(perturb >= 0.0, "rlh_ldlt_factor: the perturbation must not be negative");
  for (int64_t i = 0; i < n; ++i) {
    RLH_REQUIRE(indptr[i + 1] >= indptr[i], "rlh_ldlt_factor: indptr decreases at row %lld", (long long)i);
    for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e)
      RLH_REQUIRE(indices[e] >= 0 && indices[e] < n, "rlh_ldlt_factor: column index out of range in row %lld", (long long)i);
  }
  rlh_ldlt *f = new rlh_ldlt();
  f->dtype = dtype;
  int rc = 1;
  try {
    rc = dtype == RLH_D ? ldlt_factor<double>(n, indptr, indices, (const double *)values, perm, pivot_threshold, perturb, f)
                        : ldlt_factor<zd>(n, indptr, indices, (const zd *)values, perm, pivot_threshold, perturb, f);
  } catch (const std::bad_alloc &) {
    set_error("rlh_ldlt_factor: out of host memory");
    rc = 1;
  }
  if (rc) { delete f; return rc; }
  *out = f;
  return 0;
}

int rlh_ldlt_info(rlh_ldlt_t f, int64_t *info) {
  RLH_REQUIRE(f != nullptr && info != nullptr, "rlh_ldlt_info: null argument");
  for (int i = 0; i < RLH_LDLT_INFO; ++i) info[i] = f->info[i];
  return 0;
}

int rlh_ldlt_get(rlh_ldlt_t f, int64_t *indptr, int32_t *indices, void *values, void *diag, void *subdiag, int8_t *block,
                 int64_t *order) {
  RLH_REQUIRE(f != nullptr, "rlh_ldlt_get: null handle");
  if (indptr) memcpy(indptr, f->lptr.data(), f->lptr.size() * sizeof(int64_t));
  if (indices && !f->lidx.empty()) memcpy(indices, f->lidx.data(), f->lidx.size() * sizeof(int32_t));
  if (values && !f->lval.empty()) memcpy(values, f->lval.data(), f->lval.size());
  if (diag && !f->d.empty()) memcpy(diag, f->d.data(), f->d.size());
  if (subdiag && !f->e.empty()) memcpy(subdiag, f->e.data(), f->e.size());
  if (block && !f->blk.empty()) memcpy(block, f->blk.data(), f->blk.size());
  if (order && !f->order.empty()) memcpy(order, f->order.data(), f->order.size() * sizeof(int64_t));
  return 0;
}

int rlh_ldlt_get_transposed(rlh_ldlt_t f, int64_t *indptr, int32_t *indices, void *values) {
  RLH_REQUIRE(f != nullptr, "rlh_ldlt_get_transposed: null handle");
  if (indptr) {
    if (f->uptr.empty()) memset(indptr, 0, ((size_t)f->n + 1) * sizeof(int64_t));
    else memcpy(indptr, f->uptr.data(), f->uptr.size() * sizeof(int64_t));
  }
  if (indices && !f->uidx.empty()) memcpy(indices, f->uidx.data(), f->uidx.size() * sizeof(int32_t));
  if (values && !f->uval.empty()) memcpy(values, f->uval.data(), f->uval.size());
  return 0;
}

int rlh_ldlt_destroy(rlh_ldlt_t f) {
  delete f;
  return 0;
}

}  // extern "C"

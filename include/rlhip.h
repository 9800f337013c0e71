/*
 * rlhip.h -- C ABI of librlhip.so, the MI355X (gfx950) abstract-vectors backend
 * for the RALEIGH block-JCG eigensolver.
 *
 * This is the drop-in boundary: the entry points a ctypes binding of the
 * reference would call in place of libcudart / libcublas / libcusolver
 * (raleigh/algebra/cuda_wrap.py, cublas_wrap.py) and libmkl_rt
 * (raleigh/algebra/mkl_wrap.py).  Every entry point cites the reference
 * interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *  - A block of vectors is a column-major n x m matrix in device memory with
 *    leading dimension ld (in ELEMENTS, ld >= n): vector j starts at
 *    base + j*ld.  This is the reference's (nvec, dim) C-ordered array
 *    (dense_ndarray.py:52-83, dense_cublas.py:355-428) with ld = dim; a
 *    `select(nv, first)` window is the pointer base + first*ld.
 *  - All sizes are int64_t (the reference passes c_int and overflows at 2^31).
 *  - dtype: RLH_S float, RLH_D double, RLH_C complex64, RLH_Z complex128
 *    (interleaved re,im).
 *  - Small coefficient arrays (q, s, ind) are HOST pointers, borrowed for the
 *    call only; q is addressed through element strides so C- and F-ordered
 *    numpy arrays need no copy (dense_cublas.py:283-294).
 *  - Results that feed host math are written to HOST pointers; such calls
 *    synchronise the stream before returning.  Device-only calls are
 *    asynchronous on the library stream (rlh_set_stream).
 *  - Every function returns 0 on success, non-zero on failure;
 *    rlh_last_error() gives the message (the reference raises
 *    RuntimeError('cuda error %d'), dense_cublas.py:779-781).
 *  - One calling thread per process; one device per process (one process per
 *    GPU, RCCL reductions are driven by the caller on the same stream).
 */
#ifndef RLHIP_H
#define RLHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLH_VERSION 100

enum { RLH_S = 0, RLH_D = 1, RLH_C = 2, RLH_Z = 3 };

/* ---- context (cuda_wrap.py:139-162; cublas_wrap.py Cublas.__init__) ---- */
int rlh_version(void);
const char *rlh_last_error(void);
int rlh_device_count(int *count);
/* Binds the process to `device`, creates the stream, pinned staging ring and
 * reduction workspace.  Idempotent for the same device. */
int rlh_init(int device);
int rlh_finalize(void);
/* Use an externally owned hipStream_t (e.g. torch's current stream) so that
 * RCCL collectives issued by the caller are ordered with the kernels;
 * NULL restores the library's own stream. */
int rlh_set_stream(void *hip_stream);
int rlh_sync(void);                          /* cuda_wrap.py synchronize */
int rlh_mem_info(int64_t *free_bytes, int64_t *total_bytes);

/* ---- device memory (cuda_wrap.py malloc/free/memset/memcpy/memcpy2D;
 *      dense_cublas.py:801-811 _Data) ---- */
int rlh_malloc(void **dptr, int64_t bytes);
int rlh_free(void *dptr);
int rlh_memset(void *dptr, int value, int64_t bytes);            /* async */
int rlh_h2d(void *dptr, const void *hptr, int64_t bytes);        /* sync  */
int rlh_d2h(void *hptr, const void *dptr, int64_t bytes);        /* sync  */
int rlh_d2d(void *dst, const void *src, int64_t bytes);          /* async */
/* small device result -> host through the library's pinned buffer (one stream synchronisation);
 * used after an RCCL all-reduce of a Gram / dots result */
int rlh_fetch(void *hptr, const void *dptr, int64_t bytes);      /* sync  */
/* rows x width_bytes strided copy, kind: 0 h2d, 1 d2h, 2 d2d
 * (dense_cublas.py:61-68 append(axis=1); padded-ld upload/download). */
int rlh_copy2d(void *dst, int64_t dpitch, const void *src, int64_t spitch,
               int64_t width_bytes, int64_t rows, int kind);

/* ---- K1: Gram / dot (dense_numpy.py:78-82; dense_cublas.py:245-269) ----
 * out[i*mx + j] = sum_r conj(Y[r,i]) * X[r,j], i < my, j < mx
 * (shape (my, mx), C order == `X.dot(Y)` of the reference).
 * d_out: device buffer of my*mx elements or NULL (internal buffer);
 * h_out: host buffer or NULL.  With h_out the call synchronises; with d_out
 * only, it is asynchronous (the caller all-reduces d_out with RCCL, then
 * rlh_d2h).  X == Y with equal shapes is detected and read once. */
int rlh_gram(int dtype, int64_t n, int64_t mx, const void *X, int64_t ldx,
             int64_t my, const void *Y, int64_t ldy, void *d_out, void *h_out);

/* Gram of two CONCATENATED windows in one pass (the solver's back-to-back Gram
 * pairs share an operand -- solver.py:854-861 XAX + XBX, :1321-1339 ZAY + ZBY, :1376-1381
 * XBY + YBY, :1444-1447 XAY + YAY): out = [Y_0 | ... | Y_{ny-1}]^H [X_0 | ... | X_{nx-1}], shape
 * (sum my, sum mx), C order; up to 4 blocks per window, every block is read once.  X, ldx, mx,
 * Y, ldy, my: HOST arrays of nx / ny device pointers, leading dimensions and column counts.
 * d_out / h_out as rlh_gram. */
int rlh_gram_multi(int dtype, int64_t n, int nx, const void *const *X, const int64_t *ldx,
                   const int64_t *mx, int ny, const void *const *Y, const int64_t *ldy,
                   const int64_t *my, void *d_out, void *h_out);

/* ---- K2: column-wise dots (dense_numpy.py:68-76; dense_cublas.py:233-243)
 * out[i] = sum_r conj(Y[r,i]) * X[r,i], i < m. */
int rlh_dots(int dtype, int64_t n, int64_t m, const void *X, int64_t ldx,
             const void *Y, int64_t ldy, void *d_out, void *h_out);

/* ---- K2t: transposed dots (dense_numpy.py:55-66; dense_cublas.py:175-221)
 * d_out[r] = sum_i conj(Y[r,i]) * X[r,i], r < n; device output of n elements. */
int rlh_dots_transp(int dtype, int64_t n, int64_t m, const void *X, int64_t ldx,
                    const void *Y, int64_t ldy, void *d_out);
/* Largest modulus of the real / imaginary parts of the entries of a block (AMatrix.scale():
 * the reference scans the host array, raleigh/algebra/dense_matrix.py:44-49); *h_out is a host
 * double; the call synchronises. */
int rlh_absmax(int dtype, int64_t n, int64_t m, const void *X, int64_t ldx, double *h_out);

/* ---- K3/K4: block update (dense_numpy.py:84-105; dense_cublas.py:271-342)
 * Out[:,j] = beta*Out[:,j] + alpha * sum_{i<k} q[i,j] * X[:,i], j < m.
 * q: HOST, element (i,j) at q[i*q_rs + j*q_cs]; alpha: HOST, 2 doubles (re,im);
 * beta is 0 (multiply) or 1 (add).  Out must not overlap X. */
int rlh_block_update(int dtype, int64_t n, int64_t k, const void *X, int64_t ldx,
                     int64_t m, void *Out, int64_t ldo, const void *q,
                     int64_t q_rs, int64_t q_cs, const double *alpha, int beta);

/* ---- fused forms used by this repository's driver; the reference issues the
 *      same arithmetic as multiply + add (solver.py:1609-1656) and copy + add (solver.py:942-952).
 * rlh_block_update2: Out[:,j] = beta*Out[:,j] + alpha*(sum_i q1[i,j] X1[:,i] + sum_i q2[i,j] X2[:,i])
 * in one pass over X1 and X2.
 * rlh_lincomb_cols: Out[:,i] = a[i]*A[:,i] + b[i]*B[:,i]; a, b HOST arrays of the vectors' dtype;
 * Out may alias A or B. */
int rlh_block_update2(int dtype, int64_t n, int64_t k1, const void *X1, int64_t ldx1,
                      const void *q1, int64_t q1_rs, int64_t q1_cs, int64_t k2,
                      const void *X2, int64_t ldx2, const void *q2, int64_t q2_rs,
                      int64_t q2_cs, int64_t m, void *Out, int64_t ldo,
                      const double *alpha, int beta);
/* Two result blocks from ONE pass over the two sources (the Rayleigh-Ritz update forms the new
 * X and the new search directions Z from the same X, Y -- solver.py:1609-1656 issues a
 * multiply + add pair for each):  [OutA | OutB] = X1 * q1 + X2 * q2,  q1: k1 x (ma + mb),
 * q2: k2 x (ma + mb) host matrices (element strides as above), the first ma result columns go
 * to OutA, the other mb to OutB. */
int rlh_block_update2x2(int dtype, int64_t n, int64_t k1, const void *X1, int64_t ldx1,
                        const void *q1, int64_t q1_rs, int64_t q1_cs, int64_t k2, const void *X2,
                        int64_t ldx2, const void *q2, int64_t q2_rs, int64_t q2_cs, int64_t ma,
                        void *OutA, int64_t ldoa, int64_t mb, void *OutB, int64_t ldob);
int rlh_lincomb_cols(int dtype, int64_t n, int64_t m, const void *a, const void *A,
                     int64_t lda, const void *b, const void *B, int64_t ldb, void *Out,
                     int64_t ldo);

/* ---- K5: Y += alpha * X on an n x m window (dense_cublas.py:311-316) ---- */
int rlh_axpy(int dtype, int64_t n, int64_t m, const double *alpha,
             const void *X, int64_t ldx, void *Y, int64_t ldy);
/* ---- K6: Y[:,i] += s[i] * X[:,i]; s HOST, vectors' dtype
 *      (dense_cublas.py:343-350) ---- */
int rlh_axpy_cols(int dtype, int64_t n, int64_t m, const void *s,
                  const void *X, int64_t ldx, void *Y, int64_t ldy);
/* ---- K7: window copy / column gather (dense_cublas.py:133-153)
 * rlh_copy: Y[:, j] = X[:, j], j < m.
 * rlh_copy_cols: Y[:, k] = Xall[:, ind[k]], k < m; ind HOST int64, absolute
 * indices into the storage Xall points at. */
int rlh_copy(int dtype, int64_t n, int64_t m, const void *X, int64_t ldx,
             void *Y, int64_t ldy);
int rlh_copy_cols(int dtype, int64_t n, int64_t m, const int64_t *ind,
                  const void *Xall, int64_t ldx, void *Y, int64_t ldy);
/* ---- K8: column scale (dense_numpy.py:44-52; dense_cublas.py:155-172)
 * s HOST doubles (re,im pairs when the dtype is complex).
 * mode 1: X[:,i] *= s[i]; mode 0: X[:,i] /= s[i] unless s[i] == 0. */
int rlh_scale_cols(int dtype, int64_t n, int64_t m, const double *s, int mode,
                   void *X, int64_t ldx);
/* precision conversion Y = (dst type) X between s<->d or c<->z blocks (mixed-precision
 * preconditioning; not in the reference) */
int rlh_convert(int src_dtype, int dst_dtype, int64_t n, int64_t m, const void *X,
                int64_t ldx, void *Y, int64_t ldy);
/* Vectors.fill_random for large blocks (dense_cublas.py:119-131 draws numpy.random.rand on the
 * host and uploads: 0.6 s for 10^7 x 20): X[i, j] = uniform in [-1, 1), a pure function of
 * (seed, col0 + j, row0 + i) -- splitmix64 of the counter, restated in oracle/ops.py
 * uniform_block -- so row shards generate their rows of one global block; imaginary parts 0. */
int rlh_fill_random(int dtype, int64_t n, int64_t m, void *X, int64_t ldx, uint64_t seed,
                    int64_t row0, int64_t col0);
/* in-place complex conjugate (dense_cublas.py:503-511); no-op for real */
int rlh_conj(int dtype, int64_t n, int64_t m, void *X, int64_t ldx);

/* ---- K13: sparse symmetric/Hermitian operator
 *      (sparse_mkl.py:16-48; mkl_wrap.py:204-276 mkl_?csrmm 'SUNF'/'HUNF')
 * The caller passes the FULL matrix (both triangles) as 0-based CSR in host
 * memory; rows [row0, row0+n_rows) of a matrix with n_cols columns (row-shard
 * of the operator).  The library converts to one of three device layouts
 * (rlh_csr_layout).
 * Y[:, j] = A * X[:, j]; X has n_cols rows, Y has n_rows rows. */
typedef struct rlh_csr *rlh_csr_t;
int rlh_csr_create(rlh_csr_t *h, int dtype, int64_t n_rows, int64_t n_cols,
                   const int64_t *indptr, const int32_t *indices,
                   const void *values);
/* The Hermitian operator defined by the UPPER triangle of a square 0-based CSR matrix with sorted rows, as the
 * reference hands it to mkl_?csrmm with the 'SUNF' / 'HUNF' descriptor (mkl_wrap.py:211-276: A = U + U^H - diag(U));
 * entries below the diagonal, if stored, are ignored.  Single-GPU operator (n_own = n). */
int rlh_csr_create_upper(rlh_csr_t *h, int dtype, int64_t n, const int64_t *indptr,
                         const int32_t *indices, const void *values);
int rlh_csr_destroy(rlh_csr_t h);
int rlh_csr_info(rlh_csr_t h, int64_t *n_rows, int64_t *n_cols, int64_t *nnz,
                 int64_t *device_bytes);
/* Device layout the library chose for the handle (diagnostic): *layout = 0 sliced ELL
 * (one gather per stored entry and vector: no column locality), 1 windowed ELL (rows of at most
 * 8 entries of a real type: the column windows of every 1024-row block are staged through the
 * LDS, the row's entries stay in registers), 2 interleaved windowed layout (any row length, any
 * type: 256-row blocks, LDS image [column][vector], entries streamed in chunks of 8);
 * *stored = entry slots incl. padding; *staged_per_slot = staged vector elements per entry
 * slot (windowed-layout analysis; 0 if it was not run).  RLH_SPMM_FORMAT=sell|well|wide in the
 * environment of rlh_csr_create overrides the choice. */
int rlh_csr_layout(rlh_csr_t h, int *layout, int64_t *stored, double *staged_per_slot);
/* Stacked row blocks of the windowed layout (diagnostic): *stacks = workgroup-sized stacks of two 1024-row blocks whose
 * column windows overlap (0: the layout was not built: no windowed layout, or stacking stages less than 10 % fewer
 * elements); *staged_per_row / *staged_per_row_stacked = vector elements staged per row and vector without / with the
 * stacks (3.42 / 2.42 for the 7-point stencil on 215^3).  rlh_spmm on the whole operator uses the stacks when they exist;
 * RLH_SPMM_STACK=0 in the environment (create or call time) turns them off, =2 at create time builds them regardless. */
int rlh_csr_stacks(rlh_csr_t h, int64_t *stacks, double *staged_per_row, double *staged_per_row_stacked);
/* Whether rlh_spmm_cheb_bf16_part would take this operator: *ok = 1 if it is a float32 operator in the 1024-row windowed
 * layout whose staging groups all lie inside the column range and -- with a halo block (n_own < the column count: a row
 * shard) of leading dimension ldh -- n_own and ldh are multiples of 8 and the groups start on multiples of 8 columns, so
 * that the 2-byte staging stays on 16-byte pieces.  The layout conditions depend on the SHARD, so ranks of one row-sharded
 * operator can differ: callers agree on the answer (all-reduce MIN) BEFORE the first halo exchange of a bfloat16 step,
 * never by catching the launch's error afterwards. */
int rlh_csr_bf16_ready(rlh_csr_t h, int64_t n_own, int64_t ldh, int *ok);
/* Columns [0, n_own) are read from X, columns [n_own, n_cols) from the halo block
 * H (row c - n_own), which holds the off-shard rows received from other ranks;
 * single GPU: n_own = n_cols, H = NULL. */
int rlh_spmm(rlh_csr_t h, int64_t m, const void *X, int64_t ldx, int64_t n_own,
             const void *H, int64_t ldh, void *Y, int64_t ldy);
/* One step of the Chebyshev semi-iteration (three-term form) fused with the operator
 * application (device polynomial preconditioner): per row and vector
 *   P = cy*Y + cp*P + cb*(B - A*Y)
 * in one pass: Y = y_k (gathered, read only), P = y_{k-1} on entry and y_{k+1} on return
 * (updated in place), B the right-hand side.  A square in the own rows; P distinct from Y, B. */
int rlh_spmm_cheb(rlh_csr_t h, int64_t m, const void *Y, int64_t ldy, int64_t n_own,
                  const void *H, int64_t ldh, void *P, int64_t ldp, const void *B, int64_t ldb,
                  double cy, double cp, double cb);
/* The same two operations on a part of the rows, so that a row shard can overlap its halo
 * exchange with the rows that do not need it: part 1 = the rows whose entries all lie in columns
 * < n_own (H is not read and may still be in flight), part 2 = the other rows, part 0 = all.
 * Parts 1 and 2 together write every row exactly once.  (Granularity: the layout's 1024-row blocks;
 * with the sliced layout part 1 is empty and part 2 is everything.) */
int rlh_spmm_part(rlh_csr_t h, int part, int64_t m, const void *X, int64_t ldx, int64_t n_own,
                  const void *H, int64_t ldh, void *Y, int64_t ldy);
int rlh_spmm_cheb_part(rlh_csr_t h, int part, int64_t m, const void *Y, int64_t ldy, int64_t n_own,
                       const void *H, int64_t ldh, void *P, int64_t ldp, const void *B, int64_t ldb,
                       double cy, double cp, double cb);
/* bfloat16 storage for the device polynomial preconditioner (not in the reference).  A bf16 block
 * is a column-major array of 16-bit words, leading dimension in elements (a multiple of 8), base
 * 16-byte aligned.  pack: Y16 = bf16(scale * X) (round to nearest even) from a float32 / float64
 * block; unpack: back to float32 / float64.  rlh_spmm_cheb_bf16 is rlh_spmm_cheb on three bf16
 * blocks with float32 arithmetic against a square float32 operator in the windowed layout
 * (returns an error otherwise, the caller then stays in float32). */
int rlh_bf16_pack(int src_dtype, int64_t n, int64_t m, const void *X, int64_t ldx, double scale,
                  void *Y16, int64_t ldy);
int rlh_bf16_unpack(int dst_dtype, int64_t n, int64_t m, const void *X16, int64_t ldx, void *Y,
                    int64_t ldy);
int rlh_spmm_cheb_bf16(rlh_csr_t h, int64_t m, const void *Y16, int64_t ldy, void *P16, int64_t ldp,
                       const void *B16, int64_t ldb, double cy, double cp, double cb);
/* The bfloat16 step on a row shard (n_own / H16 / part as in rlh_spmm_part; n_own and ldh multiples
 * of 8), and the row packing of its halo exchange. */
int rlh_spmm_cheb_bf16_part(rlh_csr_t h, int part, int64_t m, const void *Y16, int64_t ldy,
                            int64_t n_own, const void *H16, int64_t ldh, void *P16, int64_t ldp,
                            const void *B16, int64_t ldb, double cy, double cp, double cb);
int rlh_gather_rows_bf16(int64_t nidx, const int64_t *d_idx, int64_t m, const void *X16,
                         int64_t ldx, void *Out16, int64_t ldo);
/* Packs rows for the halo exchange: Out[i, j] = X[idx[i], j], i < nidx, j < m;
 * idx: DEVICE int64 (built once per operator). */
int rlh_gather_rows(int dtype, int64_t nidx, const int64_t *d_idx, int64_t m,
                    const void *X, int64_t ldx, void *Out, int64_t ldo);

/* ---- incomplete LU preconditioner on the device
 *      (sparse_mkl.py:122-140 IncompleteLU -> mkl_wrap.py:279-347: mkl dcsrilut once, then two
 *      mkl_dcsrtrsv per VECTOR on the host)
 * rlh_ilut_factor: dual-threshold ILUT(p = maxfil, tau = tol) of the FULL matrix given as 0-based
 * CSR in HOST memory (dtype RLH_D or RLH_Z); runs on the host (no GPU needed): entries below
 * tol * ||row||_2 are dropped, at most maxfil entries are kept in the L part and in the U part of
 * every row (mkl_wrap.py:305-331: tol, max_fill_rel * nnz / n).  The factors are read back as CSR
 * with rlh_factors_get: which = 0 -> L strictly lower (unit diagonal implied), 1 -> U upper
 * including the diagonal (diagonal first in every row, then ascending columns). */
typedef struct rlh_factors *rlh_factors_t;
int rlh_ilut_factor(rlh_factors_t *f, int dtype, int64_t n, const int64_t *indptr,
                    const int32_t *indices, const void *values, double tol, int64_t maxfil);
int rlh_factors_nnz(rlh_factors_t f, int64_t *nnz_l, int64_t *nnz_u);
int rlh_factors_get(rlh_factors_t f, int which, int64_t *indptr, int32_t *indices, void *values);
int rlh_factors_destroy(rlh_factors_t f);
/* Sparse triangular operator for blocks of vectors (mkl_wrap.py:333-347 mkl_dcsrtrsv 'L','N','U'
 * and 'U','N','N').  The triangular factor is given as 0-based CSR in HOST memory (the diagonal
 * stored unless unit_diag); rows are grouped into dependency levels at creation.
 * rlh_sptrsv_solve_chain: X = op[nops-1]^-1 ... op[0]^-1 B for column-major n x m blocks (B may be
 * X); d_perm_in / d_perm_out: DEVICE int64 arrays or NULL -- row r of the internal scratch is row
 * d_perm_in[r] of B, and is written to row d_perm_out[r] of X (row / column permutations of a
 * factorisation P_r A P_c = L U).  Asynchronous on the library stream. */
typedef struct rlh_sptrsv *rlh_sptrsv_t;
int rlh_sptrsv_create(rlh_sptrsv_t *t, int dtype, int64_t n, const int64_t *indptr,
                      const int32_t *indices, const void *values, int lower, int unit_diag);
int rlh_sptrsv_info(rlh_sptrsv_t t, int64_t *nnz, int64_t *levels, int64_t *device_bytes);
int rlh_sptrsv_solve_chain(int nops, const rlh_sptrsv_t *ops, const int64_t *d_perm_in,
                           const int64_t *d_perm_out, int64_t m, const void *B, int64_t ldb,
                           void *X, int64_t ldx);
int rlh_sptrsv_destroy(rlh_sptrsv_t t);

/* ---- symmetric indefinite factorisation for the direct shift-invert operator
 *      (sparse_mkl.py:51-119 SparseSymmetricSolver -> mkl_wrap.py:354-489 class ParDiSo: PARDISO
 *      mtype -2 / -4 (2 / 4 when pos_def), phases 11 / 22 / 33, inertia from iparm[21], iparm[22])
 * rlh_ldlt_factor: P A P^T = L D L^H of a real symmetric / Hermitian matrix given by its UPPER
 * triangle (entries below the diagonal are ignored) as 0-based CSR in HOST memory (dtype RLH_D or
 * RLH_Z); runs on the host (no GPU needed).  perm: NULL -> own minimum-degree ordering, else
 * perm[new] = old.  D has 1 x 1 and 2 x 2 blocks chosen by threshold partial pivoting among the
 * fully summed variables of a front (pivot_threshold u in [0, 0.5]: 0.01 is the usual choice, 0
 * = no pivoting, for positive definite matrices); variables with no acceptable pivot are delayed
 * to the parent front.  perturb: a pivot below perturb * (largest entry of A in its column) is
 * taken for zero: such a pivot is replaced and counted (info[3]) -- the matrix is numerically
 * singular.
 * rlh_ldlt_info: info[0] entries of L, [1] negative and [2] positive eigenvalues of D (the
 * inertia of A), [3] perturbed pivots, [4] 2 x 2 pivots, [5] delayed pivots, [6] largest front,
 * [7] supernodes, [8] multiply-adds (estimate), [9] pivots forced at a root.
 * rlh_ldlt_get: L strictly lower (unit diagonal implied) as CSR in PIVOT order; diag[k] = D[k, k];
 * subdiag[k] = D[k + 1, k] where block[k] == 1 (first row of a 2 x 2 pivot; block 2 = second row,
 * 0 = 1 x 1 pivot); order[k] = the row of A eliminated k-th, so that with c[k] = b[order[k]]
 * the solution of A x = b is x[order[k]] = (L^-H D^-1 L^-1 c)[k].  Any pointer may be NULL. */
#define RLH_LDLT_INFO 10
typedef struct rlh_ldlt *rlh_ldlt_t;
int rlh_ldlt_factor(rlh_ldlt_t *f, int dtype, int64_t n, const int64_t *indptr,
                    const int32_t *indices, const void *values, const int64_t *perm,
                    double pivot_threshold, double perturb);
int rlh_ldlt_info(rlh_ldlt_t f, int64_t *info);
int rlh_ldlt_get(rlh_ldlt_t f, int64_t *indptr, int32_t *indices, void *values, void *diag,
                 void *subdiag, int8_t *block, int64_t *order);
/* L^H (strictly upper, unit diagonal implied, entries conjugated, columns ascending) as CSR in pivot order: the operator
 * of the backward solve (PARDISO phase 333), which the factorisation holds anyway (it produces L by columns). */
int rlh_ldlt_get_transposed(rlh_ldlt_t f, int64_t *indptr, int32_t *indices, void *values);
int rlh_ldlt_destroy(rlh_ldlt_t f);
/* X <- D^-1 X for the block diagonal D of such a factorisation, on a column-major n x m block in
 * DEVICE memory (PARDISO phase 332): d_coef holds two entries per row (DEVICE, the block's dtype):
 * X'[i] = coef[2i] X[i] + coef[2i+1] X[i + d_shift[i]], d_shift[i] in {-1, 0, +1} (DEVICE int32). */
int rlh_bdiag_solve(int dtype, int64_t n, const void *d_coef, const int32_t *d_shift, int64_t m,
                    void *X, int64_t ldx);

/* ---- sum of a SMALL host array over the processes of one node (shared memory; host only, no GPU needed)
 *      The reductions of the hot path end on the host (solver.py:1117-1187 reads the Gram matrices as NumPy arrays):
 *      on a row-sharded run each rank fetches its partial result and the ranks add them up here -- a flag-per-rank
 *      hand-off in a POSIX shared-memory segment, the slots summed in rank order (the same bits on every rank) --
 *      instead of one RCCL launch + copy per reduction.  name: "/..." chosen by rank 0 and told to the others;
 *      rank 0 creates the segment, the others wait for it; slot_bytes = the largest array ever reduced; after all
 *      ranks hold a handle rank 0 may remove the name (rlh_shm_unlink: the mappings live on).  dtype: RLH_S or RLH_D
 *      (complex data as pairs).  All ranks must make the same sequence of calls.  A rank that waits longer than
 *      RLH_SHM_TIMEOUT seconds (300) for another gets an error. */
typedef struct rlh_shm *rlh_shm_t;
int rlh_shm_create(rlh_shm_t *s, const char *name, int rank, int nranks, int64_t slot_bytes);
int rlh_shm_unlink(const char *name);
int rlh_shm_allreduce(rlh_shm_t s, int dtype, int64_t count, void *inout);
int rlh_shm_destroy(rlh_shm_t s);

/* ---- K12: dense operator (dense_numpy.py:153-175; dense_cublas.py:732-776)
 * A: DEVICE, M x N, row-major (order 0, numpy C_CONTIGUOUS, lda >= N) or
 * column-major (order 1, F_CONTIGUOUS, lda >= M).
 * transp 0: Y[:, j] = A   * X[:, j]   (X: N rows, Y: M rows)
 * transp 1: Y[:, j] = A^H * X[:, j]   (X: M rows, Y: N rows) */
int rlh_dense_apply(int dtype, int64_t M, int64_t N, const void *A, int64_t lda,
                    int order, int transp, int64_t m, const void *X,
                    int64_t ldx, void *Y, int64_t ldy);

/* The same product with a rank-one correction folded into its epilogue (no further pass over Y):
 *   Y[:, j] = Op(A) X[:, j] - c[j] * u,   u, c DEVICE arrays (u of the output dimension, NULL = a
 * vector of ones; c of m coefficients; both NULL = rlh_dense_apply).  This is how the mean shift
 * of the PCA operator A_s = A - e a^T is applied (raleigh/interfaces/partial_svd.py:258-291 removes
 * it from the intermediate block with two dot + add passes per product): c is produced on the
 * device by rlh_gram / rlh_dots with d_out, so one operator application is two GEMMs and two small
 * reductions with no host synchronisation. */
int rlh_dense_apply_r1(int dtype, int64_t M, int64_t N, const void *A, int64_t lda, int order,
                       int transp, int64_t m, const void *X, int64_t ldx, void *Y, int64_t ldy,
                       const void *d_u, const void *d_c);

/* ---- profiling aid: HIP-event time of the last `count` kernels ---- */
int rlh_timer_start(void);
int rlh_timer_stop(float *milliseconds);

#ifdef __cplusplus
}
#endif
#endif /* RLHIP_H */

/*
 * sparse_linear_hip.h — C ABI of the MI355X (gfx950) backend for the
 * SpMV / SpGEMM / sparse-add / transpose / compress hot path under
 * ttuegel/sparse-linear's Data.Matrix.Sparse.
 *
 * There is no existing FFI for these operations in the reference (they are
 * pure Haskell); the reference's designated seam is
 *   withConstMatrix :: Matrix v a -> (CInt -> CInt -> Ptr CInt -> Ptr CInt -> Ptr a -> IO b) -> IO b
 *       (sparse-linear/src/Data/Matrix/Sparse/Foreign.hs:24-41)   inputs
 *   fromForeign :: Bool -> CInt -> CInt -> Ptr CInt -> Ptr CInt -> Ptr a -> IO (Matrix v a)
 *       (Foreign.hs:43-88)                                           outputs
 * so every matrix crosses this ABI as that 5-tuple
 *   (int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax):
 * CSC, 0-based, int32, Ap[ncols] = nnz, row indices strictly ascending inside
 * a column, borrowed only for the duration of the call.  The LU / solve step
 * keeps its existing link-time ABI: see umfpack_hip.h.
 *
 * Conventions (those of the reference's UMFPACK binding,
 * suitesparse/src/Numeric/LinearAlgebra/Umfpack.hs:60-102):
 *   - every function returns an int status: 0 OK, < 0 fatal (the Haskell
 *     wrapper throws), > 0 warning;
 *   - handles are allocated by the callee and written through a void**; the
 *     free function takes that void**, releases everything and nulls it, and
 *     may be called from any thread (GHC finalizer thread);
 *   - output matrices of unknown size are returned as malloc()'d arrays so
 *     that `fromForeign False` can adopt them (it frees with C free(),
 *     Foreign.hs:54-55); spl_free is free().
 *   - all entry points are thread-safe; a handle may be used from several
 *     threads for read-only operations (spmv) concurrently.
 *
 * No torch / HIP types appear in any signature; `stream` arguments are an
 * opaque hipStream_t passed as void* (NULL = the default stream).
 */
#ifndef SPARSE_LINEAR_HIP_H
#define SPARSE_LINEAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (UMFPACK's numbering where one exists) ------------------- */
#define SPL_OK 0
#define SPL_WARNING_singular_matrix 1
#define SPL_ERROR_out_of_memory (-1)
#define SPL_ERROR_invalid_handle (-3)
#define SPL_ERROR_argument_missing (-5)
#define SPL_ERROR_n_nonpositive (-6)
#define SPL_ERROR_invalid_matrix (-8)   /* pointers not monotone / index out of range */
#define SPL_ERROR_dimension_mismatch (-20) /* the reference's errorWithStackTrace sites */
#define SPL_ERROR_index_out_of_bounds (-21) /* compress: Sparse.hs:196-212 */
#define SPL_ERROR_index_overflow (-22)  /* result does not fit int32 at the seam */
#define SPL_ERROR_device (-30)          /* HIP runtime failure, no GPU, wrong arch */
#define SPL_ERROR_internal (-911)

/* human-readable name of a status code (static string) */
const char *spl_status_string(int status);
/* number of visible HIP devices (0 if none); never fails */
int spl_device_count(void);
/* last HIP error text seen by this thread ("" if none) */
const char *spl_last_error(void);
void spl_free(void *p);
/* Device blocks of 1 GiB and more (LU factors, fronts, SpGEMM work space) are kept by the library
 * when their owner is freed and reused by the next request they fit, because on this platform a
 * fresh hipMalloc of memory released a moment before waits seconds for the driver's wipe
 * (csrc/device_pool.hip).  They are given back automatically when an allocation of this library
 * fails; this gives them back now (e.g. before another library needs the memory) and returns the
 * number of bytes released.  SPL_CACHE_DEVICE_MEMORY=0 in the environment: never keep any. */
unsigned long long spl_release_cached_memory(void);
/* Seconds this process has spent inside hipMalloc on behalf of the library so far (requests the kept blocks could
 * not serve).  A request that reaches into memory the driver is still wiping — released by this or an earlier
 * process a few seconds before — waits there, and nothing the process has queued on the device runs meanwhile
 * (tools/probe/malloc_overlap_probe.hip): the difference around a call tells how much of it was that wait
 * (benchmarks: the first factorisation of a large matrix against the steady state). */
double spl_device_alloc_seconds(void);

/* ---- one-shot operations on borrowed host CSC 5-tuples ---------------------- */

/* axpy_ (Sparse.hs:433-453):  y <- A x + y.  xlen must equal ncols and ylen
 * nrows, else SPL_ERROR_dimension_mismatch (the reference's two guards). */
int spl_gaxpy(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax,
              int xlen, const double *x, int ylen, double *y);

/* mulV (Sparse.hs:464-471):  y = A x  (y need not be initialised). */
int spl_mulv(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax,
             int xlen, const double *x, double *y);

/* y <- A^T x + y : a pure gather on the CSC arrays. */
int spl_gaxpy_t(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax,
                int xlen, const double *x, int ylen, double *y);

/* mulM (Sparse.hs:473-498): C = A B with B dense brows x bcols, row-major
 * (hmatrix's default order); C is nrows x bcols row-major. */
int spl_mulm(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax,
             int brows, int bcols, const double *B, double *C);

/* mm / (*) (Sparse.hs:691-702): C = A B.  Union pattern (cancellation keeps a
 * stored zero), row indices ascending, Cp = exclusive prefix sum.  Outputs are
 * malloc()'d.  SPL_ERROR_index_overflow if nnz(C) >= 2^31 (use the handle API). */
int spl_spgemm(int nrowsA, int ncolsA, const int *Ap, const int *Ai, const double *Ax,
               int nrowsB, int ncolsB, const int *Bp, const int *Bi, const double *Bx,
               int *nrowsC, int *ncolsC, int **Cp, int **Ci, double **Cx);
/* mm on `Matrix U.Vector (Complex Double)` (Sparse.hs:691-702 under the SPECIALIZE of :456-457): values as packed
 * (re, im) pairs; the pattern is that of the real product of the two patterns, every value the sum over ascending k
 * of A[i,k] * B[k,j] in Data.Complex's arithmetic, started from 0 — bit-identical to the Haskell code.  *Cz is
 * malloc()'d with 2 * nnz doubles. */
int spl_spgemm_z(int nrowsA, int ncolsA, const int *Ap, const int *Ai, const double *Az, int nrowsB, int ncolsB,
                 const int *Bp, const int *Bi, const double *Bz, int *nrowsC, int *ncolsC, int **Cp, int **Ci,
                 double **Cz);

/* lin (Sparse.hs:426-431):  C = alpha A + beta B, union pattern. malloc()'d outputs. */
int spl_lin(double alpha, int nrowsA, int ncolsA, const int *Ap, const int *Ai, const double *Ax,
            double beta, int nrowsB, int ncolsB, const int *Bp, const int *Bi, const double *Bx,
            int *nrowsC, int *ncolsC, int **Cp, int **Ci, double **Cx);
/* The same on `Matrix U.Vector (Complex Double)` (the second SPECIALIZE instance, Sparse.hs:456-457; what
 * Feast.hs:216 calls with a complex contour point: `lin (-1) matA _ze matB`): values and the two scalars are
 * (re, im) pairs, products and sums evaluated in Data.Complex's order, so the result is bit-identical to the
 * Haskell code.  *Cz is malloc()'d with 2 * nnz doubles. */
int spl_lin_z(const double alpha[2], int nrowsA, int ncolsA, const int *Ap, const int *Ai, const double *Az,
              const double beta[2], int nrowsB, int ncolsB, const int *Bp, const int *Bi, const double *Bz,
              int *nrowsC, int *ncolsC, int **Cp, int **Ci, double **Cz);

/* transpose (Sparse.hs:301-329): CSC(A) -> CSC(A^T) == CSR(A).  Caller
 * allocates Tp[nrows+1], Ti[nnz], Tx[nnz]. */
int spl_transpose(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax,
                  int *Tp, int *Ti, double *Tx);

/* C = A (x) B, the Kronecker product (`kronecker`, Sparse.hs:597-634): column ja*ncolsB + jb of C
 * holds the rows ia*nrowsB + ib (ascending) with values b*a.  The reference assembles its model
 * problems this way (`kronecker (ident n) T + kronecker T (ident n)`).  Outputs malloc()'d as for
 * spl_spgemm; SPL_ERROR_index_overflow when a dimension or nnz(A)*nnz(B) does not fit int32. */
int spl_kronecker(int nrowsA, int ncolsA, const int *Ap, const int *Ai, const double *Ax, int nrowsB,
                  int ncolsB, const int *Bp, const int *Bi, const double *Bx, int *nrowsC, int *ncolsC,
                  int **Cp, int **Ci, double **Cx);

/* hcat / vcat / fromBlocks / fromBlocksDiag (Sparse.hs:500-595) in one call: nblocks CSC blocks (block b is
 * nrows[b] x ncols[b] with arrays Ap[b], Ai[b], Ax[b]) are placed at (row_off[b], col_off[b]) of an
 * nrowsC x ncolsC result.  Column c of the result is the concatenation, in list order, of the columns of the
 * blocks that cover it, rows shifted by the block's row offset (vcat's copyWithOffset, Sparse.hs:551-559):
 * blocks sharing columns must be listed by ascending row offset, as vcat stacks them.  hcat: row_off = 0,
 * col_off = running widths; vcat: col_off = 0, row_off = running heights; fromBlocks: both.  value_width = 1
 * (double) or 2 (packed Complex Double: the entries are moved, never combined).  Outputs malloc()'d as for
 * spl_spgemm; SPL_ERROR_dimension_mismatch if a block leaves the result. */
int spl_assemble_blocks(int nblocks, const int *nrows, const int *ncols, const int *const *Ap, const int *const *Ai,
                        const double *const *Ax, int value_width, const int *row_off, const int *col_off, int nrowsC,
                        int ncolsC, int **Cp, int **Ci, double **Cx);

/* d[c] = A[c,c], or 0 where no entry is stored, c < min(nrows, ncols) (`takeDiag`, Sparse.hs:636-648) */
int spl_take_diag(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, double *d);

/* compress / fromTriples (Sparse.hs:184-255): COO -> CSC, duplicates summed,
 * explicit zeros kept.  Ap[ncols+1] caller-allocated; *Ai,*Ax malloc()'d with
 * Ap[ncols] entries.  On SPL_ERROR_index_out_of_bounds *bad is the first
 * offending position (rows checked first, then columns). */
int spl_compress(int nrows, int ncols, int64_t nnz, const int *rows, const int *cols,
                 const double *vals, int *Ap, int **Ai, double **Ax, int64_t *bad);

/* ---- device-resident matrix handles ------------------------------------------ */

/* Upload the CSC 5-tuple once and build the row-major (CSR) image the SpMV
 * kernels read.  The inputs may be freed as soon as the call returns. */
int spl_matrix_create(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax,
                      void **H);
/* The same for Complex Double: Az holds packed (re, im) pairs, 2 * Ap[ncols] doubles (the form the reference
 * passes for its complex instance, Umfpack/Internal.hs:124-132).  On such a handle spl_matrix_mulv / _gaxpy /
 * _spmv_dev take packed complex vectors (2 * ncols and 2 * nrows doubles; xlen, ylen still count entries) and
 * compute  y <- a * x + y  per stored entry in ascending column order with Data.Complex's arithmetic, every
 * real operation separately rounded (csrc/spmv_z.hip: 20 bytes per stored entry instead of the 48 of the real
 * 2n x 2n embedding).  Other handle operations (spgemm, export, spmm, images) are for real handles. */
int spl_matrix_create_z(int nrows, int ncols, const int *Ap, const int *Ai, const double *Az, void **H);
/* 1 for a handle made by spl_matrix_create_z, 0 for a real one */
int spl_matrix_is_complex(void *H);
/* Same, but keep only the rows of the part-th of nparts nnz-balanced
 * contiguous row blocks (1-D row partition). */
int spl_matrix_create_rowblock(int nrows, int ncols, const int *Ap, const int *Ai,
                               const double *Ax, int part, int nparts, void **H);
/* Build from CSR arrays directly (== the CSC arrays of A^T). Rows [row0,row0+nrows_local)
 * of a matrix with nrows_global rows. */
int spl_matrix_create_csr(int64_t nrows_global, int64_t ncols, int64_t row0, int64_t nrows_local,
                          const int *rowptr, const int *colidx, const double *val, void **H);
/* Synthetic workloads generated on the device (include/spl_synth.h):
 * kind 0 random(n,K), 1 banded(n), 2 poisson2d(m) [n=m*m], 3 poisson3d(m) [n=m^3].
 * Generates rows [row0,row1) only. */
int spl_matrix_create_synthetic(int kind, int64_t n_or_m, int K, uint64_t seed, int64_t row0,
                                int64_t row1, void **H);
/* R-MAT graph of 2^scale vertices and edge_factor * 2^scale edges with quadrant
 * probabilities (a, b, c, 1-a-b-c); duplicate edges summed (compress semantics). */
int spl_matrix_create_rmat(int scale, int edge_factor, double a, double b, double c, uint64_t seed,
                           void **H);
/* C = A * B on device-resident operands (mm, Sparse.hs:691-702); C gets 64-bit row
 * pointers, so nnz(C) >= 2^31 is fine.  *products (may be NULL) receives the number of
 * intermediate products. */
int spl_matrix_spgemm(void *HA, void *HB, void **HC, int64_t *products);
void spl_matrix_free(void **H);

/* info[0..7] = nrows_global, ncols, row0, nrows_local, nnz_local, device,
 * rows per panel of the column-blocked image (0 = CSR-stream kernel in use, -64 = sliced-ELL
 * image in use), its cols_log2 */
int spl_matrix_info(void *H, int64_t info[8]);
/* copy the device CSR image back: rowptr[nrows_local+1] (relative to the block,
 * rowptr[0]=0), colidx[nnz_local], val[nnz_local] (2 nnz_local doubles, packed (re, im), for a complex handle) */
int spl_matrix_export_csr(void *H, int64_t *rowptr, int *colidx, double *val);
/* the same for rows [row0, row1) of the block only (a window of a result too large to copy whole):
 * rowptr[row1-row0+1] keeps the block's offsets (rowptr[0] = first entry of row0, not 0); colidx / val
 * receive the rowptr[row1-row0] - rowptr[0] entries of those rows and must hold `capacity` entries
 * (SPL_ERROR_argument_missing if they do not: call once with capacity 0 to learn the count from rowptr) */
int spl_matrix_export_csr_rows(void *H, int64_t row0, int64_t row1, int64_t *rowptr, int64_t capacity, int *colidx,
                               double *val);

/* ---- device-resident forms of lin / transpose / compress: handle in, handle out, nothing crosses PCIe ----
 * spl_matrix_lin: C = alpha A + beta B (Sparse.hs:401-431) for two handles of the same shape, row block and
 *   scalar kind; alpha / beta are (re, im) pairs, the imaginary parts must be 0 for real handles (complex scalars
 *   on real matrices: spl_matrix_to_complex first, as the reference's `cmap (:+ 0)` does, Feast.hs:214).
 * spl_matrix_to_complex: the Complex Double handle (x :+ 0) of a real one.
 * spl_matrix_transpose: handle of A^T (Sparse.hs:301-329); whole real matrices.
 * spl_matrix_compress_dev: COO triples in DEVICE memory -> handle (compress / fromTriples, Sparse.hs:184-280):
 *   bounds checked rows first, then columns (*bad = first offending position, may be NULL), duplicates summed. */
int spl_matrix_lin(void *HA, const double alpha[2], void *HB, const double beta[2], void **HC);
int spl_matrix_to_complex(void *H, void **HZ);
int spl_matrix_transpose(void *H, void **HT);
int spl_matrix_compress_dev(int nrows, int ncols, int64_t ntriples, const int *d_rows, const int *d_cols,
                            const double *d_vals, void **H, int64_t *bad);
/* transpose the block on the device (Sparse.hs:301-329) and copy out its
 * column-major image: colptr[ncols+1], rowidx[nnz_local] (LOCAL row ids, ascending
 * inside a column), val[nnz_local] — i.e. the reference's own CSC Matrix fields */
int spl_matrix_export_csc(void *H, int64_t *colptr, int *rowidx, double *val);

/* y = A x  /  y <- A x + y  with HOST vectors (upload, run, download) */
int spl_matrix_mulv(void *H, int xlen, const double *x, double *y);
int spl_matrix_gaxpy(void *H, int xlen, const double *x, int ylen, double *y);

/* Device-resident SpMV: d_x (ncols doubles) and d_y (nrows_local doubles) are
 * DEVICE pointers; the kernel is enqueued on `stream` and the call returns
 * without synchronising.  accumulate != 0: y <- A x + y. */
int spl_matrix_spmv_dev(void *H, const double *d_x, double *d_y, int accumulate, void *stream);
/* Device-resident sparse x dense (mulM): d_B is ncols x k, d_C is nrows_local x k, both
 * row-major DEVICE arrays; A is read once for all k columns.  accumulate != 0: C <- A B + C. */
int spl_matrix_spmm_dev(void *H, const double *d_B, double *d_C, int k, int accumulate, void *stream);
/* select a kernel variant for spl_matrix_spmv_dev (tuning / ablation only): 0 = default
 * (whatever spl_matrix_optimize chose), 1-6 CSR-stream shapes, 7 sub-wavefront kernel, 8
 * column-blocked image, 9-11 gather cache policies, 15 sliced-ELL image, 16 column-sorted panel
 * image (order-free sums, see spl_matrix_set_spmv_order); 12-14 are timing-only
 * ablations that do not compute A x and are refused unless SPL_ALLOW_ABLATION=1.
 * Returns SPL_ERROR_argument_missing for an unknown or refused variant. */
int spl_matrix_set_variant(void *H, int variant);

/* Analyse the matrix once (like umfpack_*_symbolic) and build the image variant 0 then uses:
 * the column-blocked image when columns have no locality and x exceeds the L2s
 * (csrc/spmv_blocked.hip), the sliced-ELL image for regular rows with locality
 * (csrc/spmv_sell.hip), or nothing (CSR-stream kernel).  Results are bit-identical either way.
 * After spl_matrix_set_spmv_order(H, SPL_ORDER_FREE) the column-sorted panel image (csrc/spmv_panel.hip) takes
 * the place of the column-blocked one where it pays; its two launch parameters are then timed against their
 * neighbours on a scratch vector (about a hundred launches, once) and the fastest pair kept. */
int spl_matrix_optimize(void *H);
/* build that image with an explicit shape (tuning / ablation): panels of rows_per_panel rows,
 * column blocks of 2^cols_log2 columns (rows_per_panel << cols_log2 must fit 31 bits);
 * 0,0 = choose.  unroll: 0 default, {4,8,10,12} 64-entry chunks per register set of the
 * lockstep kernel; negative {-1,-2,-4,-8} selects the free-running baseline kernel. */
int spl_matrix_build_blocked(void *H, int rows_per_panel, int cols_log2, int unroll);

/* Order of the floating-point sums of spl_matrix_spmv_dev / mulv / gaxpy on this handle.
 * The CONTRACT of both modes is north_star's: every value within 1e-10 relative of the reference's.
 * SPL_ORDER_REFERENCE (default): every y[r] receives a*x + y in ascending column order, each
 * multiply and add separately rounded — the evaluation order of axpy_ (Sparse.hs:447-451).  On gfx950
 * this reproduces the reference's bits, run after run (what the parity tests check), with two caveats that
 * keep bit-identity an observation rather than a promise: the column-blocked kernel relies on same-address
 * lanes of one LDS atomic being applied in lane order (pinned by tests/test_gpu_spmv_blocked.py, not by an
 * ISA document), and the CSR-stream kernel sums a row longer than one 512-entry chunk with a wavefront tree.
 * SPL_ORDER_FREE: the products of a row may be added in any order (still separately rounded
 * multiplies and adds): results agree with the reference to rounding level (1e-10 relative is
 * north_star's contract; observed ~1e-16) but need not be bit-identical, nor identical from run to
 * run.  Lets spl_matrix_optimize use the column-sorted panel image (csrc/spmv_panel.hip), which
 * sends about a quarter fewer requests to the L2 on matrices without column locality. */
#define SPL_ORDER_REFERENCE 0
#define SPL_ORDER_FREE 1
int spl_matrix_set_spmv_order(void *H, int order);
/* CUs to leave free for a kernel that runs beside the SpMV (the collective of a multi-GPU step): the images
 * spl_matrix_optimize / _build_* lay out afterwards have one panel (group) per remaining CU when the row block
 * takes one generation, so the persistent grid is that much smaller.  Call before spl_matrix_optimize.
 * Default 0, or the environment variable SPL_SPMV_RESERVED_CUS. */
int spl_matrix_set_reserved_cus(void *H, int reserved);
/* Diagnostics only: `blocks` workgroups of `threads` threads copy d_buf (count doubles) onto itself for
 * `milliseconds` (at most 2000) on `stream` — a stand-in for a collective's channel kernels when measuring what
 * reserved CUs are worth on one GPU (tools/bench_reserved_cus.py). */
int spl_debug_occupy(int blocks, int threads, double milliseconds, double *d_buf, size_t count, void *stream);
/* diagnostics / tests: the library's own radix sort (csrc/radix_sort.hip; the nested dissection orders its level
 * structures with it) on d_keys[0 .. n) in device memory, in place, ascending by the low nbits bits of the keys */
int spl_debug_sort_u64(unsigned long long *d_keys, long long n, int nbits, void *stream);
/* build the column-sorted panel image with an explicit shape (tuning / ablation): panels of
 * rows_per_panel rows (<= 20479: one workgroup's LDS), index blocks of 2^cols_log2 columns
 * (<= 17); 0,0 = choose.  form: 0 default; 1 / 2 = one 64-entry chunk per load instruction with 1 / 2
 * index blocks per barrier phase; 4 / 5 = paired storage (two chunks per 8-byte key / 16-byte value
 * load) with 1 / 2 index blocks per phase.  unroll: 0 default, else chunks ({4,6,8,10,12}) or pairs
 * ({2..6}) per wavefront and register set.  6 / 7 = ring form on the paired storage (a few wavefronts of
 * the workgroup only stream the image and hand it to the others, which only gather and fold, through
 * 1.5 KiB slots in LDS; 1 / 2 index blocks per phase; unroll = units in flight per loader, 0 = 6;
 * rows_per_panel then has to leave room for the slots: 19 700 with the default 4 loaders).  Used by
 * variant 16, and by variant 0 once spl_matrix_set_spmv_order(H, SPL_ORDER_FREE) was called. */
int spl_matrix_build_panel(void *H, int rows_per_panel, int cols_log2, int unroll, int form);
/* Diagnostics: synchronises the device and returns 1 when a bounded wait of the ring form's hand-over gave
 * up during the last panel SpMV of this handle (its result is then invalid), 0 otherwise, < 0 on error. */
int spl_matrix_panel_errors(void *H);
/* the kernel spl_matrix_spmv_dev launches for this handle now: 0 CSR-stream, 8 column-blocked
 * lockstep, 15 sliced ELL, 16 column-sorted panels (other values: the forced ablation variant) */
int spl_matrix_spmv_kernel(void *H);

/* ---- one-sided exchange of y for the row-partitioned SpMV (one process per GPU; csrc/peer.hip) ----
 * Every rank pushes its block of y straight into every peer's copy with device-to-device copies on one
 * stream per peer (no collective kernel, no CU taken from the SpMV), then a 4-byte step flag; a one-thread
 * kernel on the compute stream waits for the N - 1 flags.  Peers' buffers are reached through IPC memory
 * handles that the caller passes between the processes (3 x 64 bytes per rank).
 * create: y is cut into chunks * world pieces, piece q = c * world + p = y[bounds[q], bounds[q+1]) belongs to
 *   rank p (chunks = 1: one block per rank); handles_out = this rank's 192 bytes.
 * connect: all_handles = world x 192 bytes in rank order.
 * A step = one push per chunk + finish.  push: the rank's piece of chunk c (produced on `stream`) goes to every
 *   rank (the copies run on the per-peer streams, i.e. under whatever `stream` does next — the kernel of the
 *   next chunk).  finish: after the work enqueued here *y_full (n doubles, device, valid until the step after
 *   next) is the whole y.  No host synchronisation.
 * failed: 1 if a wait gave up after ~2 s (a peer did not deliver). */
int spl_peer_exchange_create(int rank, int world, int chunks, int64_t n, const int64_t *bounds,
                             unsigned char *handles_out, void **X);
int spl_peer_exchange_connect(void *X, const unsigned char *all_handles);
int spl_peer_exchange_push(void *X, int chunk, const double *d_piece, void *stream);
int spl_peer_exchange_finish(void *X, void *stream, double **y_full);
int spl_peer_exchange_failed(void *X);
/* 1 when the step flags live in fine-grained device memory (polled by a kernel while a peer's copy engine
 * writes them: csrc/peer.hip), 0 when the runtime refused it and plain device memory is used */
int spl_peer_exchange_flags_finegrained(void *X);
void spl_peer_exchange_free(void **X);

/* fill a device vector with the synthetic entries j in [j0,j1) */
int spl_vector_synthetic_dev(uint64_t seed, int64_t j0, int64_t j1, double *d_x, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPARSE_LINEAR_HIP_H */

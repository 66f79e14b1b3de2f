/*
 * umfpack_hip.h — the link-time ABI of the LU / solve step, served by the MI355X backend.
 *
 * These are exactly the real-valued symbols that the reference imports from SuiteSparse
 * UMFPACK (suitesparse/src/Numeric/LinearAlgebra/Umfpack/Internal.hs:137-148):
 *     umfpack_di_symbolic   (called by `analyze`,      Umfpack.hs:64)
 *     umfpack_di_numeric    (called by `factor`,       Umfpack.hs:78)
 *     umfpack_di_solve      (called by `linearSolve_`, Umfpack.hs:99)
 *     umfpack_di_free_symbolic / umfpack_di_free_numeric   (ForeignPtr finalizers, :65,79)
 *     umfpack_di_report_status                              (:66,80,100)
 * with the argument lists of the Haskell type synonyms UmfpackSymbolic / UmfpackNumeric /
 * UmfpackSolve / UmfpackReport (Internal.hs:26-59).  Linking the reference's `suitesparse`
 * package against libsparse_linear_hip.so instead of libumfpack routes `analyze`, `factor`,
 * `linearSolve_`, `linearSolve` and `(<\>)` to the GPU with no Haskell change.
 *
 * Conventions honoured: CSC, 0-based, int32, borrowed for the call only;
 * Control and Info may be NULL (the reference always passes NULL => UMFPACK defaults, which
 * include up to 2 steps of iterative refinement in solve); Symbolic/Numeric are opaque,
 * callee-allocated, written through void**; the free functions take that void**, are
 * idempotent and may run on any thread; status: 0 OK, 1 singular-matrix warning, < 0 error;
 * only sys = 0 (A x = b) and sys = 1 (A^T x = b; UMFPACK_At) are ever passed (Umfpack.hs:95-97).
 *
 * Algorithm (HISTORY.md §4.5 for how it got there).  symbolic (host): a reverse-Cuthill-McKee band ordering and, from 1024
 * unknowns on, a nested-dissection ordering with its frontal tree; the one a measured time model
 * predicts to factor faster is kept (the tree from about 2000 unknowns on, narrow bands included:
 * the band factorisation is a chain of n / 64 block steps however narrow the band).  numeric (GPU): LU of the
 * permuted matrix WITHOUT row interchanges, blocked on the fp64 matrix cores — in band storage,
 * or multifrontal on the tree (dense frontal matrices, Schur complements passed to the parent) —
 * when the matrix is diagonally dominant by columns (provably safe) and, as a speculation, for
 * every other matrix too — on the tree then with threshold partial pivoting (0.1, after UMFPACK's row scaling)
 * among the fully summed rows of every 64 x 64 pivot block, the interchange folded into the block's stored
 * inverse (round 3; a symmetric matrix is tried as L D L^T first); LAPACK-style band LU with partial pivoting
 * otherwise (a zero pivot, SPL_LU_FORCE_PIVOT=1, or a failed speculation).  solve (GPU): blocked substitution through the
 * band or up and down the tree, and SpMV-based iterative refinement with UMFPACK's stopping rules.
 * A speculation is checked by every solve: unless the refined solution is backward stable to
 * rounding level (componentwise backward error <= 1e-13), the object is refactored and the system
 * solved again, so the caller sees pivoted-LU accuracy either way.  The refactorisation has two
 * stages.  First STATIC PIVOTING (csrc/static_pivot.hpp), which stays on the tree: a
 * maximum-product transversal with its scalings turns A into B = Dr P A Dc with |b_jj| = 1 >=
 * |b_ij| (the pivots are chosen before the factorisation instead of during it — what UMFPACK's
 * threshold pivoting with delayed pivots would do inside the fronts cannot be done on fronts whose
 * sizes are planned ahead), B is ordered, factored without interchanges and checked like any
 * speculation.  Only if that fails too: the band factorisation with partial pivoting (when its
 * band does not fit the HBM — a large mesh — solve returns UMFPACK_ERROR_out_of_memory and the
 * object keeps its previous factors).  SPL_LU_STATIC_PIVOT=0 skips the first stage.  SPL_LU_METHOD=band|mf forces the ordering.
 * RECTANGULAR matrices (n_row != n_col) are analysed and "factored" as far as the reference's binding can observe:
 * symbolic records shape and pattern, numeric checks the pattern and returns UMFPACK_OK when min(n_row, n_col) non-zero
 * pivots exist and UMFPACK_WARNING_singular_matrix otherwise, without keeping factors: the structural rank over the
 * non-zero entries decides for large matrices, and a matrix with rows x columns x min(rows, columns) <= 4e8 that is
 * structurally regular is eliminated on the host (row pivoting, a pivot counts unless exactly zero, as in UMFPACK), so
 * the 3 x 2 matrix of ones is reported singular as UMFPACK reports it.  KNOWN DIVERGENCE: a large rectangular matrix
 * that is structurally regular but numerically rank deficient gets UMFPACK_OK here.  Every solve returns
 * UMFPACK_ERROR_invalid_system, as UMFPACK's does ("the matrix is not square"; the reference's linearSolve_ assumes
 * square, Umfpack.hs:93).
 * The complex (`zi`) entry points (Internal.hs:69-115): a Numeric object holds the real 2n x 2n
 * embedding with interleaved unknowns (csrc/umfpack_zi.hip) — packed complex arrays (imaginary
 * pointer NULL, the only form the reference uses, Internal.hs:124-132) ARE the real arrays of the
 * embedded system; sys = 1 is the conjugate transpose, as in UMFPACK — for residuals, refinement
 * and every fallback above.  The factors themselves are NATIVE COMPLEX fronts (two planes per
 * front, complex arithmetic on the fp64 matrix cores, L D L^T when A == A^T; csrc/multifrontal.hip,
 * round 3) whenever the analysis chose the multifrontal tree and that tree has work to halve
 * (1e12 flops; SPL_ZI_NATIVE=1|0 forces); the band path and small trees factor the embedding.
 */
#ifndef UMFPACK_HIP_H
#define UMFPACK_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define UMFPACK_OK 0
#define UMFPACK_WARNING_singular_matrix 1
#define UMFPACK_ERROR_out_of_memory (-1)
#define UMFPACK_ERROR_invalid_Numeric_object (-3)
#define UMFPACK_ERROR_invalid_Symbolic_object (-4)
#define UMFPACK_ERROR_argument_missing (-5)
#define UMFPACK_ERROR_n_nonpositive (-6)
#define UMFPACK_ERROR_invalid_matrix (-8)
#define UMFPACK_ERROR_different_pattern (-11)
#define UMFPACK_ERROR_invalid_system (-13)
#define UMFPACK_ERROR_internal_error (-911)

#define UMFPACK_A 0   /* A x = b   */
#define UMFPACK_At 1  /* A' x = b  */

int umfpack_di_symbolic(int n_row, int n_col, const int Ap[], const int Ai[], const double Ax[],
                        void **Symbolic, const double Control[], double Info[]);

int umfpack_di_numeric(const int Ap[], const int Ai[], const double Ax[], void *Symbolic,
                       void **Numeric, const double Control[], double Info[]);

int umfpack_di_solve(int sys, const int Ap[], const int Ai[], const double Ax[], double X[],
                     const double B[], void *Numeric, const double Control[], double Info[]);

void umfpack_di_free_symbolic(void **Symbolic);
void umfpack_di_free_numeric(void **Numeric);
void umfpack_di_report_status(const double Control[], int status);

/* ---- complex: Ax/Az, Xx/Xz, Bx/Bz are split real/imaginary arrays, or, when the imaginary
 * pointer is NULL, packed interleaved (re, im) pairs in the real pointer ---------------------- */
int umfpack_zi_symbolic(int n_row, int n_col, const int Ap[], const int Ai[], const double Ax[],
                        const double Az[], void **Symbolic, const double Control[], double Info[]);
int umfpack_zi_numeric(const int Ap[], const int Ai[], const double Ax[], const double Az[],
                       void *Symbolic, void **Numeric, const double Control[], double Info[]);
int umfpack_zi_solve(int sys, const int Ap[], const int Ai[], const double Ax[], const double Az[],
                     double Xx[], double Xz[], const double Bx[], const double Bz[], void *Numeric,
                     const double Control[], double Info[]);
void umfpack_zi_free_symbolic(void **Symbolic);
void umfpack_zi_free_numeric(void **Numeric);
void umfpack_zi_report_status(const double Control[], int status);

/* ---- batched linearSolve ------------------------------------------------------------------
 * The reference's `linearSolve` (Umfpack.hs:103-108) maps `linearSolve_` over a list of
 * right-hand sides, one umfpack_*_solve call each; FEAST solves a whole subspace that way
 * (Feast.hs:197-201).  These entry points take all nrhs right-hand sides through the factors
 * together (X and B are n x nrhs, column-major; complex ones packed when Xz = Bz = NULL): the
 * blocked solves are latency-bound, so nrhs columns cost about as much as one.  Each column gets
 * the same refinement as umfpack_*_solve; status as umfpack_*_solve. */
int spl_umfpack_di_solve_many(int sys, const int Ap[], const int Ai[], const double Ax[], int nrhs, double X[],
                              const double B[], void *Numeric);
int spl_umfpack_zi_solve_many(int sys, const int Ap[], const int Ai[], const double Ax[], const double Az[],
                              int nrhs, double Xx[], double Xz[], const double Bx[], const double Bz[],
                              void *Numeric);

/* The same with the right-hand sides and the solutions in DEVICE memory (n x nrhs doubles, column-major;
 * complex ones packed (re, im) pairs: 2 n x nrhs doubles), for callers that keep a whole subspace in HBM
 * between the solves, the SpMVs and the dense products of an iteration (the FEAST-style loop,
 * Feast.hs:197-233).  The call returns when the solutions are written.  Streams: numeric and solve entry points
 * work on the calling thread's default stream (hipStreamPerThread) — ordered with the legacy default stream, where a
 * caller's own kernels usually produce d_B, exactly as the legacy stream itself is; a caller that fills d_B on a
 * non-blocking stream of its own synchronises that stream first.  Calls from different host threads (the contour
 * points of a FEAST iteration) overlap on the device.  Ap/Ai/Ax stay HOST arrays — the
 * arguments umfpack_*_solve ignores on the way to the factors are what a failed speculation is refactored
 * from — and may be NULL, which rules that refactoring out (the band fallback remains). */
int spl_umfpack_di_solve_many_dev(int sys, const int Ap[], const int Ai[], const double Ax[], int nrhs, double *d_X,
                                  const double *d_B, void *Numeric);
int spl_umfpack_zi_solve_many_dev(int sys, const int Ap[], const int Ai[], const double Ax[], int nrhs, double *d_X,
                                  const double *d_B, void *Numeric);

/* dimension of the system a Numeric object factors (0 if invalid); helper of the zi wrappers */
int spl_umfpack_dimension(void *Numeric);

/* factorisation a Numeric object holds now: 0 band, partial pivoting; 1 band, no interchanges
 * (diagonally dominant matrix); 2 the same as a speculation; 3 multifrontal, no interchanges
 * (dominant); 4 the same as a speculation; 5 multifrontal with static pivoting (the
 * maximum-product transversal on the diagonal, scaled; checked by every solve); -1 if invalid */
int spl_umfpack_path(void *Numeric);

/* figures of the factorisation a Numeric object holds now (diagnostics and benchmarks; UMFPACK
 * reports the like through Info[], which the reference never reads: Umfpack/Internal.hs passes
 * nullPtr).  out[0] path as above, out[1] n, out[2] kl and out[3] ku of the reordered matrix
 * (band paths), out[4] bytes of device memory the factors occupy, out[5] flops of the numeric
 * factorisation (band: 2 n kl ku; multifrontal: summed over the fronts), out[6] number of fronts
 * (0 on the band paths), out[7] flags: bit 0 the fronts are native complex ones (`zi` objects; out[5]
 * then counts 4 real flops per complex multiply-add pair), bit 1 threshold pivoting inside the diagonal
 * blocks of the fronts was on (matrices that are not diagonally dominant).  Returns 0, or -1 if the object
 * is invalid. */
int spl_umfpack_stats(void *Numeric, double out[8]);

/* the most recent solve call that finished on a Numeric object, and the size of a walk over its factors (the
 * measurement of the triangular solves: SURVEY.md 8d "SpTRSV").  UMFPACK reports the first four through
 * Info[UMFPACK_IR_TAKEN], [UMFPACK_IR_ATTEMPTED], [UMFPACK_OMEGA1] (umfpack_di_solve here fills them too when Info is
 * not NULL; the reference passes nullPtr, Umfpack.hs:99).  out[0] walks over the factors (forward + backward
 * substitution: the first solve plus one per refinement step attempted), out[1] refinement steps kept, out[2]
 * attempted, out[3] largest componentwise backward error max_i |r_i| / (|A||x| + |b|)_i among the delivered columns,
 * out[4] bytes ONE walk reads: 8 per stored entry of L and U (dense panels, no index arrays) + 16 n for the vectors.
 * out[5 .. 7]: the chain matrices a multifrontal object builds at its first solve of A x = b (and, for unsymmetric
 * factorisations, a second set at its first solve of A^T x = b)
 * (explicit inverses of 512 x 512 blocks of the large fronts' pivot blocks, so that a step of the triangular passes is one
 * matrix-vector product): bytes of device memory, milliseconds their construction took, pivots per block (0: none).
 * Returns 0, or -1 if the object is invalid. */
int spl_umfpack_solve_report(void *Numeric, double out[8]);

#ifdef __cplusplus
}
#endif
#endif /* UMFPACK_HIP_H */

/*
 * sparse_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Serial CPU restatement of the hot path of ttuegel/sparse-linear
 * (Data.Matrix.Sparse + Data.Vector.Sparse.ScatterGather), written from the
 * reference's Haskell sources as a specification.  Each function cites the
 * reference file:line it follows (paths relative to /root/reference).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (sparse-linear_amd/) never does.
 *
 * PINNING.  The reference is Haskell and no GHC exists in this pipeline, so the
 * oracle is never checked against a run of the reference.  It is pinned by the
 * reference's own test properties and fixtures (sparse-linear/tests/Sparse.hs,
 * tests/Test/LinearAlgebra.hs, suitesparse/tests/test-umfpack.hs,
 * feast/tests/test-feast.hs), re-expressed in tests/test_oracle_*.py, plus
 * closed-form known answers and scipy as an independent third party.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off reproduces GHC's separately rounded `a * x + y`
 * (mulsd + addsd; the cabal file passes only -msse2).
 *
 * Index type is int64_t = Haskell Int (Sparse.hs:67-76).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "../include/spl_synth.h"

typedef int64_t Int;

#define ORC_OK 0
#define ORC_ERR_DIM (-1)      /* dimension mismatch (errorWithStackTrace sites) */
#define ORC_ERR_BOUNDS (-2)   /* index out of bounds in compress */
#define ORC_ERR_ALLOC (-3)
#define ORC_ERR_FORMAT (-4)

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------
 * computePtrs — Sparse.hs:282-291: count occurrences, exclusive prefix sum.
 * ptrs has n+1 entries, ptrs[n] = nnz.
 * ---------------------------------------------------------------------- */
void orc_compute_ptrs(Int n, Int nnz, const Int *indices, Int *ptrs) {
  for (Int i = 0; i <= n; ++i) ptrs[i] = 0;
  for (Int k = 0; k < nnz; ++k) ptrs[indices[k] + 1] += 1;
  for (Int i = 0; i < n; ++i) ptrs[i + 1] += ptrs[i];
}

/* stable sort of (key, value) pairs by key: insertion sort for short runs,
 * bottom-up merge sort otherwise.  The reference uses an UNSTABLE introsort
 * (Sparse.hs:219,262), so the order of equal keys is implementation-defined
 * there; stability here fixes one legal order. */
static void sort_pairs(Int len, Int *ix, double *x) {
  if (len < 2) return;
  if (len <= 32) {
    for (Int i = 1; i < len; ++i) {
      Int ki = ix[i];
      double vi = x[i];
      Int j = i - 1;
      while (j >= 0 && ix[j] > ki) {
        ix[j + 1] = ix[j];
        x[j + 1] = x[j];
        --j;
      }
      ix[j + 1] = ki;
      x[j + 1] = vi;
    }
    return;
  }
  Int *tix = (Int *)malloc((size_t)len * sizeof(Int));
  double *tx = (double *)malloc((size_t)len * sizeof(double));
  Int *sa = ix, *da = tix;
  double *sx = x, *dx = tx;
  for (Int w = 1; w < len; w *= 2) {
    for (Int lo = 0; lo < len; lo += 2 * w) {
      Int mid = lo + w < len ? lo + w : len;
      Int hi = lo + 2 * w < len ? lo + 2 * w : len;
      Int i = lo, j = mid, k = lo;
      while (i < mid && j < hi) {
        if (sa[j] < sa[i]) { da[k] = sa[j]; dx[k++] = sx[j++]; }
        else { da[k] = sa[i]; dx[k++] = sx[i++]; }
      }
      while (i < mid) { da[k] = sa[i]; dx[k++] = sx[i++]; }
      while (j < hi) { da[k] = sa[j]; dx[k++] = sx[j++]; }
    }
    Int *t = sa; sa = da; da = t;
    double *u = sx; sx = dx; dx = u;
  }
  if (sa != ix) {
    memcpy(ix, sa, (size_t)len * sizeof(Int));
    memcpy(x, sx, (size_t)len * sizeof(double));
  }
  free(tix);
  free(tx);
}

/* ------------------------------------------------------------------------
 * dedupInPlace — Sparse.hs:257-280.
 * Sort by index; walk with a write cursor w and read cursor r: on equal index
 * add xs[r] into xs[w] and overwrite ixs[r] with the sentinel `idim`; sort
 * again so the sentinels sink to the end.  Returns the number deleted.
 * ---------------------------------------------------------------------- */
Int orc_dedup_in_place(Int idim, Int len, Int *ixs, double *xs) {
  sort_pairs(len, ixs, xs);
  Int w = 0, del = 0;
  for (Int r = 1; r < len; ++r) {
    if (ixs[r] == ixs[w]) {
      ixs[r] = idim;
      xs[w] = xs[w] + xs[r];
      ++del;
    } else {
      w = r;
    }
  }
  sort_pairs(len, ixs, xs);
  return del;
}

/* ------------------------------------------------------------------------
 * compress — Sparse.hs:184-255.  COO -> CSC, duplicates summed, explicit
 * zeros kept.  Outputs: ptrs[ncols+1], and the first ptrs[ncols] entries of
 * idx_out / val_out (caller allocates nnz entries each).
 * Returns ORC_OK, or ORC_ERR_BOUNDS with *bad = offending position
 * (row check first, then column check, as :196-212).
 * ---------------------------------------------------------------------- */
int orc_compress(Int nrows, Int ncols, Int nnz, const Int *rows, const Int *cols,
                 const double *vals, Int *ptrs, Int *idx_out, double *val_out, Int *bad) {
  for (Int k = 0; k < nnz; ++k)
    if (!(rows[k] >= 0 && rows[k] < nrows)) { if (bad) *bad = k; return ORC_ERR_BOUNDS; }
  for (Int k = 0; k < nnz; ++k)
    if (!(cols[k] >= 0 && cols[k] < ncols)) { if (bad) *bad = k; return ORC_ERR_BOUNDS; }

  /* sort entries by column (:219) — stable counting sort */
  Int *start = (Int *)malloc((size_t)(ncols + 1) * sizeof(Int));
  Int *cursor = (Int *)malloc((size_t)(ncols + 1) * sizeof(Int));
  if (!start || !cursor) { free(start); free(cursor); return ORC_ERR_ALLOC; }
  orc_compute_ptrs(ncols, nnz, cols, start); /* ptrs = computePtrs ncols _cols (:253) */
  memcpy(cursor, start, (size_t)(ncols + 1) * sizeof(Int));
  for (Int k = 0; k < nnz; ++k) {
    Int p = cursor[cols[k]]++;
    idx_out[p] = rows[k];
    val_out[p] = vals[k];
  }
  /* per-column dedup (:224-225), then shift columns down (:227-244) */
  Int shift = 0;
  ptrs[0] = 0;
  for (Int m = 0; m < ncols; ++m) {
    Int s = start[m], len = start[m + 1] - start[m];
    Int del = orc_dedup_in_place(nrows, len, idx_out + s, val_out + s);
    if (shift > 0) {
      memmove(idx_out + s - shift, idx_out + s, (size_t)len * sizeof(Int));
      memmove(val_out + s - shift, val_out + s, (size_t)len * sizeof(double));
    }
    shift += del;
    ptrs[m + 1] = start[m + 1] - shift;
  }
  free(start);
  free(cursor);
  return ORC_OK;
}

/* ------------------------------------------------------------------------
 * transpose — Sparse.hs:301-329 (+ preincrement, Data/Vector/Util.hs:52-58).
 * Counting-sort transpose; within each new column the old column index
 * ascends because old columns are visited in order.
 * In: CSC (nrows x ncols).  Out: ptrsT[nrows+1], idxT[nnz], valT[nnz].
 * ---------------------------------------------------------------------- */
void orc_transpose(Int nrows, Int ncols, const Int *ptrs, const Int *idx, const double *val,
                   Int *ptrsT, Int *idxT, double *valT) {
  Int nz = ptrs[ncols];
  orc_compute_ptrs(nrows, nz, idx, ptrsT);
  Int *count = (Int *)malloc((size_t)(nrows > 0 ? nrows : 1) * sizeof(Int));
  memcpy(count, ptrsT, (size_t)nrows * sizeof(Int));
  for (Int m = 0; m < ncols; ++m)
    for (Int k = ptrs[m]; k < ptrs[m + 1]; ++k) {
      Int ix = count[idx[k]]++;
      idxT[ix] = m;
      valT[ix] = val[k];
    }
  free(count);
}

/* ------------------------------------------------------------------------
 * axpy_ — Sparse.hs:433-453.  y <- A x + y, column-major scatter, each update
 * `a * x + y` separately rounded, columns ascending.
 * ---------------------------------------------------------------------- */
int orc_axpy_(Int nrows, Int ncols, const Int *ptrs, const Int *idx, const double *val,
              Int xlen, const double *x, Int ylen, double *y) {
  if (xlen != ncols) return ORC_ERR_DIM; /* :438-441 */
  if (ylen != nrows) return ORC_ERR_DIM; /* :442-445 */
  for (Int c = 0; c < ncols; ++c) {
    for (Int k = ptrs[c]; k < ptrs[c + 1]; ++k) {
      double xv = x[c];
      Int r = idx[k];
      y[r] = val[k] * xv + y[r];
    }
  }
  return ORC_OK;
}

/* axpy_ on Complex Double — the reference's second SPECIALIZE instance (Sparse.hs:456-457, 465-466).
 * Values, x and y are packed (re, im) pairs.  `a * x + y` with base's Data.Complex instance
 *   (x :+ y) * (x' :+ y') = (x*x' - y*y') :+ (x*y' + y*x')      (+) componentwise
 * every real operation separately rounded (this file is built with -ffp-contract=off). */
int orc_axpy_z(Int nrows, Int ncols, const Int *ptrs, const Int *idx, const double *val,
               Int xlen, const double *x, Int ylen, double *y) {
  if (xlen != ncols) return ORC_ERR_DIM;
  if (ylen != nrows) return ORC_ERR_DIM;
  for (Int c = 0; c < ncols; ++c) {
    for (Int k = ptrs[c]; k < ptrs[c + 1]; ++k) {
      const double xr = x[2 * c], xi = x[2 * c + 1];
      const double ar = val[2 * k], ai = val[2 * k + 1];
      const Int r = idx[k];
      const double pr = ar * xr - ai * xi;
      const double pi = ar * xi + ai * xr;
      y[2 * r] = pr + y[2 * r];
      y[2 * r + 1] = pi + y[2 * r + 1];
    }
  }
  return ORC_OK;
}

/* mulV — Sparse.hs:464-471: thaw (copy) x, zero y, axpy_, freeze (copy) y. */
int orc_mulv(Int nrows, Int ncols, const Int *ptrs, const Int *idx, const double *val,
             Int xlen, const double *x, double *y_out) {
  if (xlen != ncols) return ORC_ERR_DIM;
  double *xc = (double *)malloc((size_t)(ncols > 0 ? ncols : 1) * sizeof(double));
  double *yc = (double *)malloc((size_t)(nrows > 0 ? nrows : 1) * sizeof(double));
  if (!xc || !yc) { free(xc); free(yc); return ORC_ERR_ALLOC; }
  memcpy(xc, x, (size_t)ncols * sizeof(double));          /* G.thaw _x  :468 */
  for (Int i = 0; i < nrows; ++i) yc[i] = 0.0;             /* replicate 0 :469 */
  int st = orc_axpy_(nrows, ncols, ptrs, idx, val, ncols, xc, nrows, yc);
  memcpy(y_out, yc, (size_t)nrows * sizeof(double));      /* G.freeze y :471 */
  free(xc);
  free(yc);
  return st;
}

/* axpy — Sparse.hs:455-462: thaw both, axpy_, freeze y. */
int orc_axpy(Int nrows, Int ncols, const Int *ptrs, const Int *idx, const double *val,
             Int xlen, const double *x, Int ylen, const double *y, double *y_out) {
  if (xlen != ncols || ylen != nrows) return ORC_ERR_DIM;
  memcpy(y_out, y, (size_t)nrows * sizeof(double));
  return orc_axpy_(nrows, ncols, ptrs, idx, val, xlen, x, ylen, y_out);
}

/* mulM — Sparse.hs:473-498 (unexported): C = A B, B dense; one axpy_ per
 * column of B.  B and C are given/returned ROW-major (hmatrix default), the
 * reference transposes to column-major slices internally (:480-490). */
int orc_mulm(Int nrows, Int ncols, const Int *ptrs, const Int *idx, const double *val,
             Int brows, Int bcols, const double *B, double *C) {
  if (ncols != brows) return ORC_ERR_DIM;
  double *xc = (double *)malloc((size_t)(brows > 0 ? brows : 1) * sizeof(double));
  double *yc = (double *)malloc((size_t)(nrows > 0 ? nrows : 1) * sizeof(double));
  for (Int j = 0; j < bcols; ++j) {
    for (Int i = 0; i < brows; ++i) xc[i] = B[i * bcols + j];
    for (Int i = 0; i < nrows; ++i) yc[i] = 0.0;
    orc_axpy_(nrows, ncols, ptrs, idx, val, brows, xc, nrows, yc);
    for (Int i = 0; i < nrows; ++i) C[i * bcols + j] = yc[i];
  }
  free(xc);
  free(yc);
  return ORC_OK;
}

/* ------------------------------------------------------------------------
 * The SG monad — Data/Vector/Sparse/ScatterGather.hs:29-147.
 * Dense sparse accumulator: Bool pattern[len] + double values[len].
 * ---------------------------------------------------------------------- */
typedef struct {
  Int len;
  unsigned char *pattern;
  double *values;
} orc_sg;

static int sg_run(orc_sg *sg, Int len) { /* run :37-43 */
  sg->len = len;
  sg->pattern = (unsigned char *)malloc((size_t)(len > 0 ? len : 1));
  sg->values = (double *)malloc((size_t)(len > 0 ? len : 1) * sizeof(double));
  return (sg->pattern && sg->values) ? ORC_OK : ORC_ERR_ALLOC;
}
static void sg_done(orc_sg *sg) { free(sg->pattern); free(sg->values); }
static void sg_reset(orc_sg *sg, double a0) { /* reset :47-53 */
  memset(sg->pattern, 0, (size_t)sg->len);
  for (Int i = 0; i < sg->len; ++i) sg->values[i] = a0;
}
static void sg_scatter_indices(orc_sg *sg, Int n, const Int *ix) { /* :77-81 */
  for (Int k = 0; k < n; ++k) sg->pattern[ix[k]] = 1;
}
static Int sg_count(const orc_sg *sg) { /* count :107-114 */
  Int n = 0;
  for (Int i = 0; i < sg->len; ++i) n += sg->pattern[i] ? 1 : 0;
  return n;
}
static void sg_gather(const orc_sg *sg, Int *ix_out, double *x_out) { /* :116-147 */
  Int k = 0;
  for (Int i = 0; i < sg->len; ++i)
    if (sg->pattern[i]) ix_out[k++] = i;
  k = 0;
  for (Int i = 0; i < sg->len; ++i)
    if (sg->pattern[i]) x_out[k++] = sg->values[i];
}

/* growable column store used by unsafeFromColumns (Sparse.hs:381-399) */
typedef struct {
  Int cap, nnz;
  Int *idx;
  double *val;
} colstore;
static int cs_reserve(colstore *cs, Int extra) {
  if (cs->nnz + extra <= cs->cap) return ORC_OK;
  Int ncap = cs->cap ? cs->cap : 1024;
  while (ncap < cs->nnz + extra) ncap *= 2;
  Int *ni = (Int *)realloc(cs->idx, (size_t)ncap * sizeof(Int));
  if (!ni) return ORC_ERR_ALLOC;
  cs->idx = ni;
  double *nv = (double *)realloc(cs->val, (size_t)ncap * sizeof(double));
  if (!nv) return ORC_ERR_ALLOC;
  cs->val = nv;
  cs->cap = ncap;
  return ORC_OK;
}

/* ------------------------------------------------------------------------
 * mm — Sparse.hs:691-702.  C = A B, column-wise Gustavson with the dense SPA.
 * For every column j of B: reset 0; for each (k, b) of B[:,j] ascending k:
 * scatter A[:,k] with  w[i] = w[i] + a*b  (pattern set regardless of value);
 * gather.  `literal` != 0: full O(nrows) reset + 3 gather sweeps exactly as
 * the reference (ScatterGather.hs:47-53,97-147).  `literal` == 0: identical
 * results via a touched list (reset/gather only the touched entries, sorted) —
 * the reference algorithm is O(nrows*ncols) and infeasible beyond n ~ 1e5.
 * Outputs malloc'd: *Cp[ncolsB+1], *Ci[nnz], *Cx[nnz].
 * ---------------------------------------------------------------------- */
static int cmp_int(const void *a, const void *b) {
  Int x = *(const Int *)a, y = *(const Int *)b;
  return (x > y) - (x < y);
}

int orc_mm(Int nrowsA, Int ncolsA, const Int *Ap, const Int *Ai, const double *Ax,
           Int nrowsB, Int ncolsB, const Int *Bp, const Int *Bi, const double *Bx,
           int literal, Int **Cp_out, Int **Ci_out, double **Cx_out) {
  if (ncolsA != nrowsB) return ORC_ERR_DIM; /* :694 */
  orc_sg sg;
  if (sg_run(&sg, nrowsA) != ORC_OK) return ORC_ERR_ALLOC;
  colstore cs = {0, 0, NULL, NULL};
  Int *Cp = (Int *)malloc((size_t)(ncolsB + 1) * sizeof(Int));
  Int *touched = (Int *)malloc((size_t)(nrowsA > 0 ? nrowsA : 1) * sizeof(Int));
  if (!Cp || !touched) return ORC_ERR_ALLOC;
  Cp[0] = 0;
  if (!literal) sg_reset(&sg, 0.0);
  for (Int j = 0; j < ncolsB; ++j) {
    Int ntouched = 0;
    if (literal) sg_reset(&sg, 0.0); /* :697 */
    for (Int q = Bp[j]; q < Bp[j + 1]; ++q) { /* iforM_ colB :698 */
      Int k = Bi[q];
      double b = Bx[q];
      Int s = Ap[k], e = Ap[k + 1];
      if (literal) {
        sg_scatter_indices(&sg, e - s, Ai + s);
      } else {
        for (Int p = s; p < e; ++p)
          if (!sg.pattern[Ai[p]]) { sg.pattern[Ai[p]] = 1; touched[ntouched++] = Ai[p]; }
      }
      for (Int p = s; p < e; ++p) /* unsafeScatterValues, add = \c a -> c + a*b :699 */
        sg.values[Ai[p]] = sg.values[Ai[p]] + Ax[p] * b;
    }
    Int pop = literal ? sg_count(&sg) : ntouched;
    if (cs_reserve(&cs, pop) != ORC_OK) return ORC_ERR_ALLOC;
    if (literal) {
      sg_gather(&sg, cs.idx + cs.nnz, cs.val + cs.nnz);
    } else {
      qsort(touched, (size_t)ntouched, sizeof(Int), cmp_int);
      for (Int t = 0; t < ntouched; ++t) {
        Int i = touched[t];
        cs.idx[cs.nnz + t] = i;
        cs.val[cs.nnz + t] = sg.values[i];
        sg.pattern[i] = 0;
        sg.values[i] = 0.0;
      }
    }
    cs.nnz += pop;
    Cp[j + 1] = cs.nnz; /* pointers = scanl' (+) 0 nonZeros :393 */
  }
  sg_done(&sg);
  free(touched);
  *Cp_out = Cp;
  *Ci_out = cs.idx ? cs.idx : (Int *)malloc(sizeof(Int));
  *Cx_out = cs.val ? cs.val : (double *)malloc(sizeof(double));
  return ORC_OK;
}

/* mm on Complex Double (Sparse.hs:691-702 at the SPECIALIZE instance of :456-457).  The same loop as orc_mm
 * (touched-list form: identical results to the literal O(nrows) reset / gather sweeps, see there) with a complex
 * accumulator per row: w[i] = w[i] + a * b, Data.Complex's product (x*x' - y*y') :+ (x*y' + y*x'), componentwise
 * sum.  Values are packed (re, im) pairs. */
int orc_mm_z(Int nrowsA, Int ncolsA, const Int *Ap, const Int *Ai, const double *Ax,
             Int nrowsB, Int ncolsB, const Int *Bp, const Int *Bi, const double *Bx,
             Int **Cp_out, Int **Ci_out, double **Cx_out) {
  if (ncolsA != nrowsB) return ORC_ERR_DIM; /* :694 */
  const size_t nr = (size_t)(nrowsA > 0 ? nrowsA : 1);
  unsigned char *pattern = (unsigned char *)calloc(nr, 1);
  double *wr = (double *)calloc(nr, sizeof(double)), *wi = (double *)calloc(nr, sizeof(double));
  Int *touched = (Int *)malloc(nr * sizeof(Int));
  Int *Cp = (Int *)malloc((size_t)(ncolsB + 1) * sizeof(Int));
  Int cap = 16, nnz = 0;
  Int *idx = (Int *)malloc((size_t)cap * sizeof(Int));
  double *val = (double *)malloc((size_t)cap * 2 * sizeof(double));
  if (!pattern || !wr || !wi || !touched || !Cp || !idx || !val) return ORC_ERR_ALLOC;
  Cp[0] = 0;
  for (Int j = 0; j < ncolsB; ++j) {
    Int ntouched = 0; /* SG.reset 0 :697 */
    for (Int q = Bp[j]; q < Bp[j + 1]; ++q) { /* iforM_ colB :698 */
      const Int k = Bi[q];
      const double br = Bx[2 * q], bi = Bx[2 * q + 1];
      for (Int p = Ap[k]; p < Ap[k + 1]; ++p) { /* add = \c a -> c + a*b :699 */
        const Int i = Ai[p];
        if (!pattern[i]) { pattern[i] = 1; touched[ntouched++] = i; }
        const double ar = Ax[2 * p], ai = Ax[2 * p + 1];
        const double pr = ar * br - ai * bi;
        const double pi = ar * bi + ai * br;
        wr[i] = wr[i] + pr;
        wi[i] = wi[i] + pi;
      }
    }
    if (nnz + ntouched > cap) {
      while (nnz + ntouched > cap) cap *= 2;
      idx = (Int *)realloc(idx, (size_t)cap * sizeof(Int));
      val = (double *)realloc(val, (size_t)cap * 2 * sizeof(double));
      if (!idx || !val) return ORC_ERR_ALLOC;
    }
    qsort(touched, (size_t)ntouched, sizeof(Int), cmp_int);
    for (Int t = 0; t < ntouched; ++t) {
      const Int i = touched[t];
      idx[nnz + t] = i;
      val[2 * (nnz + t)] = wr[i];
      val[2 * (nnz + t) + 1] = wi[i];
      pattern[i] = 0;
      wr[i] = 0.0;
      wi[i] = 0.0;
    }
    nnz += ntouched;
    Cp[j + 1] = nnz;
  }
  free(pattern); free(wr); free(wi); free(touched);
  *Cp_out = Cp;
  *Ci_out = idx;
  *Cx_out = val;
  return ORC_OK;
}

/* ------------------------------------------------------------------------
 * lin — Sparse.hs:426-431 on top of glin :401-424:
 *   glin 0 (\r a -> r + alpha*a) A (\r b -> r + beta*b) B
 * Per column: if A's column is empty -> cmap (fB 0) colB; if B's is empty ->
 * cmap (fA 0) colA (:413-414); else reset 0, scatter A, scatter B, gather.
 * Union pattern, cancellation kept.  Outputs malloc'd.
 * ---------------------------------------------------------------------- */
int orc_lin(double alpha, Int nrowsA, Int ncolsA, const Int *Ap, const Int *Ai, const double *Ax,
            double beta, Int nrowsB, Int ncolsB, const Int *Bp, const Int *Bi, const double *Bx,
            Int **Cp_out, Int **Ci_out, double **Cx_out) {
  if (nrowsA != nrowsB) return ORC_ERR_DIM; /* :408 */
  if (ncolsA != ncolsB) return ORC_ERR_DIM; /* :409 */
  orc_sg sg;
  if (sg_run(&sg, nrowsA) != ORC_OK) return ORC_ERR_ALLOC;
  colstore cs = {0, 0, NULL, NULL};
  Int *Cp = (Int *)malloc((size_t)(ncolsA + 1) * sizeof(Int));
  Cp[0] = 0;
  for (Int j = 0; j < ncolsA; ++j) {
    Int na = Ap[j + 1] - Ap[j], nb = Bp[j + 1] - Bp[j];
    if (cs_reserve(&cs, na + nb) != ORC_OK) return ORC_ERR_ALLOC;
    Int *ci = cs.idx + cs.nnz;
    double *cx = cs.val + cs.nnz;
    Int pop;
    if (na == 0) {
      for (Int t = 0; t < nb; ++t) { ci[t] = Bi[Bp[j] + t]; cx[t] = 0.0 + beta * Bx[Bp[j] + t]; }
      pop = nb;
    } else if (nb == 0) {
      for (Int t = 0; t < na; ++t) { ci[t] = Ai[Ap[j] + t]; cx[t] = 0.0 + alpha * Ax[Ap[j] + t]; }
      pop = na;
    } else {
      sg_reset(&sg, 0.0);
      sg_scatter_indices(&sg, na, Ai + Ap[j]);
      for (Int p = Ap[j]; p < Ap[j + 1]; ++p) sg.values[Ai[p]] = sg.values[Ai[p]] + alpha * Ax[p];
      sg_scatter_indices(&sg, nb, Bi + Bp[j]);
      for (Int p = Bp[j]; p < Bp[j + 1]; ++p) sg.values[Bi[p]] = sg.values[Bi[p]] + beta * Bx[p];
      pop = sg_count(&sg);
      sg_gather(&sg, ci, cx);
    }
    cs.nnz += pop;
    Cp[j + 1] = cs.nnz;
  }
  sg_done(&sg);
  *Cp_out = Cp;
  *Ci_out = cs.idx ? cs.idx : (Int *)malloc(sizeof(Int));
  *Cx_out = cs.val ? cs.val : (double *)malloc(sizeof(double));
  return ORC_OK;
}

/* lin on Complex Double (Sparse.hs:426-431 over glin :401-424 at a = Complex Double; the call of
 * feast/src/Numeric/LinearAlgebra/Feast.hs:216, `lin (-1) matA _ze matB`).  Values and the scalars are packed
 * (re, im) pairs; `r + alpha * a` with Data.Complex's instance: the product (x*x' - y*y') :+ (x*y' + y*x'),
 * the sum componentwise.  The same three cases as orc_lin; the scatter-gather workspace holds a complex number
 * per row (two planes here). */
int orc_lin_z(const double *alpha, Int nrowsA, Int ncolsA, const Int *Ap, const Int *Ai, const double *Ax,
              const double *beta, Int nrowsB, Int ncolsB, const Int *Bp, const Int *Bi, const double *Bx,
              Int **Cp_out, Int **Ci_out, double **Cx_out) {
  if (nrowsA != nrowsB) return ORC_ERR_DIM; /* :408 */
  if (ncolsA != ncolsB) return ORC_ERR_DIM; /* :409 */
  orc_sg sg;
  if (sg_run(&sg, nrowsA) != ORC_OK) return ORC_ERR_ALLOC;
  double *im = (double *)malloc((size_t)(nrowsA > 0 ? nrowsA : 1) * sizeof(double));
  Int cap = 16, nnz = 0;
  Int *idx = (Int *)malloc((size_t)cap * sizeof(Int));
  double *val = (double *)malloc((size_t)cap * 2 * sizeof(double));
  Int *Cp = (Int *)malloc((size_t)(ncolsA + 1) * sizeof(Int));
  if (!im || !idx || !val || !Cp) return ORC_ERR_ALLOC;
  Cp[0] = 0;
  const double ar = alpha[0], ai = alpha[1], br = beta[0], bi = beta[1];
  for (Int j = 0; j < ncolsA; ++j) {
    const Int na = Ap[j + 1] - Ap[j], nb = Bp[j + 1] - Bp[j];
    if (nnz + na + nb > cap) {
      while (nnz + na + nb > cap) cap *= 2;
      idx = (Int *)realloc(idx, (size_t)cap * sizeof(Int));
      val = (double *)realloc(val, (size_t)cap * 2 * sizeof(double));
      if (!idx || !val) return ORC_ERR_ALLOC;
    }
    Int *ci = idx + nnz;
    double *cx = val + 2 * nnz;
    Int pop;
    if (na == 0) { /* S.cmap (fB 0) colB */
      for (Int t = 0; t < nb; ++t) {
        const double xr = Bx[2 * (Bp[j] + t)], xi = Bx[2 * (Bp[j] + t) + 1];
        ci[t] = Bi[Bp[j] + t];
        cx[2 * t] = 0.0 + (br * xr - bi * xi);
        cx[2 * t + 1] = 0.0 + (br * xi + bi * xr);
      }
      pop = nb;
    } else if (nb == 0) { /* S.cmap (fA 0) colA */
      for (Int t = 0; t < na; ++t) {
        const double xr = Ax[2 * (Ap[j] + t)], xi = Ax[2 * (Ap[j] + t) + 1];
        ci[t] = Ai[Ap[j] + t];
        cx[2 * t] = 0.0 + (ar * xr - ai * xi);
        cx[2 * t + 1] = 0.0 + (ar * xi + ai * xr);
      }
      pop = na;
    } else {
      sg_reset(&sg, 0.0);
      for (Int i = 0; i < nrowsA; ++i) im[i] = 0.0;
      sg_scatter_indices(&sg, na, Ai + Ap[j]);
      for (Int p = Ap[j]; p < Ap[j + 1]; ++p) {
        const double xr = Ax[2 * p], xi = Ax[2 * p + 1];
        sg.values[Ai[p]] = sg.values[Ai[p]] + (ar * xr - ai * xi);
        im[Ai[p]] = im[Ai[p]] + (ar * xi + ai * xr);
      }
      sg_scatter_indices(&sg, nb, Bi + Bp[j]);
      for (Int p = Bp[j]; p < Bp[j + 1]; ++p) {
        const double xr = Bx[2 * p], xi = Bx[2 * p + 1];
        sg.values[Bi[p]] = sg.values[Bi[p]] + (br * xr - bi * xi);
        im[Bi[p]] = im[Bi[p]] + (br * xi + bi * xr);
      }
      pop = 0;
      for (Int i = 0; i < nrowsA; ++i)
        if (sg.pattern[i]) {
          ci[pop] = i;
          cx[2 * pop] = sg.values[i];
          cx[2 * pop + 1] = im[i];
          ++pop;
        }
    }
    nnz += pop;
    Cp[j + 1] = nnz;
  }
  sg_done(&sg);
  free(im);
  *Cp_out = Cp;
  *Ci_out = idx;
  *Cx_out = val;
  return ORC_OK;
}

/* ------------------------------------------------------------------------
 * checkMatrix — sparse-linear/tests/Test/LinearAlgebra.hs:40-67.
 * Returns 0 if all format invariants hold, else the 1-based number of the
 * first failing clause.
 * ---------------------------------------------------------------------- */
int orc_check_matrix(Int nrows, Int ncols, Int nptrs, const Int *ptrs, Int nidx, const Int *idx,
                     Int nval) {
  if (nptrs != ncols + 1) return 2;
  for (Int c = 0; c < ncols; ++c)
    if (ptrs[c] > ptrs[c + 1]) return 1;
  if (nval != ptrs[ncols]) return 3;
  if (nidx != ptrs[ncols]) return 4;
  for (Int c = 0; c < ncols; ++c)
    for (Int k = ptrs[c] + 1; k < ptrs[c + 1]; ++k)
      if (!(idx[k - 1] < idx[k])) return 5;
  for (Int k = 0; k < nidx; ++k) {
    if (idx[k] < 0) return 6;
    if (idx[k] >= nrows) return 7;
  }
  return 0;
}

/* ------------------------------------------------------------------------
 * fromForeign — Data/Matrix/Sparse/Foreign.hs:43-88: widen int32 -> Int and
 * dedupInPlace every column WITHOUT compaction (the deletion count is
 * ignored, :74-78): duplicate-free input round-trips unchanged.
 * withConstMatrix — Foreign.hs:24-41: narrow Int -> int32.
 * ---------------------------------------------------------------------- */
void orc_with_const_matrix(Int ncols, const Int *ptrs, const Int *idx, int32_t *Ap, int32_t *Ai) {
  Int nz = ptrs[ncols];
  for (Int c = 0; c <= ncols; ++c) Ap[c] = (int32_t)ptrs[c];
  for (Int k = 0; k < nz; ++k) Ai[k] = (int32_t)idx[k];
}
void orc_from_foreign(int32_t nrows, int32_t ncols, const int32_t *Ap, const int32_t *Ai,
                      const double *Ax, Int *ptrs, Int *idx, double *val) {
  for (Int c = 0; c <= ncols; ++c) ptrs[c] = Ap[c];
  Int nz = ptrs[ncols];
  for (Int k = 0; k < nz; ++k) { idx[k] = Ai[k]; val[k] = Ax[k]; }
  for (Int m = 0; m < ncols; ++m)
    orc_dedup_in_place(nrows, ptrs[m + 1] - ptrs[m], idx + ptrs[m], val + ptrs[m]);
}

/* ------------------------------------------------------------------------
 * CSR row-gather SpMV in the reference's evaluation order:
 * y[r] = fold (\acc (c,a) -> a*x[c] + acc) y0[r] over row r, ascending c.
 * Bit-identical to orc_axpy_ on the transposed arrays; int32 indices as at
 * the FFI seam.  Used as the checker for the HIP CSR kernels.
 * ---------------------------------------------------------------------- */
void orc_csr_gaxpy32(int64_t nrows, const int32_t *rowptr, const int32_t *colidx,
                     const double *val, const double *x, double *y) {
  for (int64_t r = 0; r < nrows; ++r) {
    double acc = y[r];
    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) acc = val[k] * x[colidx[k]] + acc;
    y[r] = acc;
  }
}

/* ------------------------------------------------------------------------
 * Synthetic workloads, restated on the host from
 * include/spl_synth.h.  All emit CSR with int32 indices, rows [row0,row1).
 * Two-call protocol: pass colidx = NULL to get row pointers (relative to
 * row0, rowptr[0] = 0) and the nnz; then call again with buffers.
 * ---------------------------------------------------------------------- */
static int gen_random_row(uint64_t seed, uint64_t n, int K, uint64_t r, int32_t *c_out,
                          double *v_out) {
  uint64_t c[SPL_MAX_DRAWS];
  double v[SPL_MAX_DRAWS];
  for (int k = 0; k < K; ++k) {
    uint64_t ck = spl_random_col(seed, r, (uint64_t)k, n);
    double vk = spl_random_val(seed, r, (uint64_t)k);
    int j = k - 1; /* stable insertion by column: equal columns keep draw order */
    while (j >= 0 && c[j] > ck) { c[j + 1] = c[j]; v[j + 1] = v[j]; --j; }
    c[j + 1] = ck;
    v[j + 1] = vk;
  }
  int m = 0;
  for (int k = 0; k < K; ++k) {
    if (m > 0 && c[k] == (uint64_t)c_out[m - 1]) v_out[m - 1] = v_out[m - 1] + v[k];
    else { c_out[m] = (int32_t)c[k]; v_out[m] = v[k]; ++m; }
  }
  return m;
}

int64_t orc_gen_random_csr(uint64_t seed, int64_t n, int K, int64_t row0, int64_t row1,
                           int64_t *rowptr, int32_t *colidx, double *val) {
  int32_t c[SPL_MAX_DRAWS];
  double v[SPL_MAX_DRAWS];
  int64_t nnz = 0;
  rowptr[0] = 0;
  for (int64_t r = row0; r < row1; ++r) {
    int m = gen_random_row(seed, (uint64_t)n, K, (uint64_t)r, c, v);
    if (colidx) {
      memcpy(colidx + nnz, c, (size_t)m * sizeof(int32_t));
      memcpy(val + nnz, v, (size_t)m * sizeof(double));
    }
    nnz += m;
    rowptr[r - row0 + 1] = nnz;
  }
  return nnz;
}

int64_t orc_gen_banded_csr(uint64_t seed, int64_t n, int64_t row0, int64_t row1, int64_t *rowptr,
                           int32_t *colidx, double *val) {
  int64_t nnz = 0;
  rowptr[0] = 0;
  for (int64_t r = row0; r < row1; ++r) {
    for (int d = 0; d < SPL_BAND_DIAGS; ++d) {
      int64_t c = r + spl_band_offset(d);
      if (c < 0 || c >= n) continue;
      if (colidx) {
        colidx[nnz] = (int32_t)c;
        val[nnz] = spl_random_val(seed, (uint64_t)r, (uint64_t)d);
      }
      ++nnz;
    }
    rowptr[r - row0 + 1] = nnz;
  }
  return nnz;
}

/* 2-D 5-point Poisson on an m x m grid (N = m*m): diag 4, neighbours -1. */
int64_t orc_gen_poisson2d_csr(int64_t m, int64_t *rowptr, int32_t *colidx, double *val) {
  int64_t nnz = 0;
  rowptr[0] = 0;
  for (int64_t iy = 0; iy < m; ++iy)
    for (int64_t ix = 0; ix < m; ++ix) {
      int64_t r = iy * m + ix;
      int64_t cc[5] = {r - m, r - 1, r, r + 1, r + m};
      int ok[5] = {iy > 0, ix > 0, 1, ix < m - 1, iy < m - 1};
      for (int t = 0; t < 5; ++t)
        if (ok[t]) {
          if (colidx) { colidx[nnz] = (int32_t)cc[t]; val[nnz] = (t == 2) ? 4.0 : -1.0; }
          ++nnz;
        }
      rowptr[r + 1] = nnz;
    }
  return nnz;
}

/* 3-D 7-point Poisson on an m^3 grid: diag 6, neighbours -1. */
int64_t orc_gen_poisson3d_csr(int64_t m, int64_t *rowptr, int32_t *colidx, double *val) {
  int64_t nnz = 0;
  rowptr[0] = 0;
  for (int64_t iz = 0; iz < m; ++iz)
    for (int64_t iy = 0; iy < m; ++iy)
      for (int64_t ix = 0; ix < m; ++ix) {
        int64_t r = (iz * m + iy) * m + ix;
        int64_t cc[7] = {r - m * m, r - m, r - 1, r, r + 1, r + m, r + m * m};
        int ok[7] = {iz > 0, iy > 0, ix > 0, 1, ix < m - 1, iy < m - 1, iz < m - 1};
        for (int t = 0; t < 7; ++t)
          if (ok[t]) {
            if (colidx) { colidx[nnz] = (int32_t)cc[t]; val[nnz] = (t == 3) ? 6.0 : -1.0; }
            ++nnz;
          }
        rowptr[r + 1] = nnz;
      }
  return nnz;
}

/* R-MAT edge list (COO, duplicates present): rows/cols/vals of `nedges`
 * edges e in [e0, e0+nedges). */
void orc_gen_rmat_coo(uint64_t seed, int scale, uint32_t ta, uint32_t tb, uint32_t tc, int64_t e0,
                      int64_t nedges, int64_t *rows, int64_t *cols, double *vals) {
  for (int64_t e = 0; e < nedges; ++e) {
    uint64_t r, c;
    spl_rmat_edge(seed, (uint64_t)(e0 + e), scale, ta, tb, tc, &r, &c);
    rows[e] = (int64_t)r;
    cols[e] = (int64_t)c;
    vals[e] = spl_uniform_value(spl_hash(seed ^ SPL_VAL_SALT, (uint64_t)(e0 + e), 63));
  }
}

void orc_gen_vector(uint64_t seed, int64_t j0, int64_t j1, double *x) {
  for (int64_t j = j0; j < j1; ++j) x[j - j0] = spl_vector_entry(seed, (uint64_t)j);
}

/* closeness predicate of feast/tests/test-feast.hs:17-19:
 * x == y || |x-y| / |x+y| < tol.  Returns the number of failing elements. */
int64_t orc_count_not_close(int64_t n, const double *a, const double *b, double tol) {
  int64_t bad = 0;
  for (int64_t i = 0; i < n; ++i) {
    double x = a[i], y = b[i];
    if (x == y) continue;
    if (fabs(x - y) / fabs(x + y) < tol) continue;
    ++bad;
  }
  return bad;
}

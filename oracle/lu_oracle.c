/*
 * lu_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU checker for the solve step `mat <\> b` (suitesparse/src/Numeric/
 * LinearAlgebra/Umfpack.hs:38-102).  In the reference ALL of the arithmetic is
 * inside the third-party, un-vendored, un-pinned SuiteSparse UMFPACK
 * (suitesparse/suitesparse.cabal:30-32 `extra-libraries: openblas suitesparse`,
 * no version bound; no umfpack.h / libumfpack in this pipeline).  UMFPACK's
 * published algorithm is: fill-reducing column pre-ordering, unsymmetric
 * multifrontal LU with threshold partial pivoting, P/R/L/U/Q solve and (default
 * Control, as the reference always passes NULL, Umfpack.hs:64,78,99) up to two
 * steps of iterative refinement with the sparse backward error as the stopping
 * test.  Pivot order is an implementation detail, so parity for this step is
 * defined on the SOLUTION: `ident <\> v == v` exactly
 * (suitesparse/tests/test-umfpack.hs:16-19) and residual / manufactured-
 * solution checks elsewhere.  PARITY UNPINNED beyond that one reference test.
 *
 * This file: left-looking sparse LU with partial pivoting (natural column
 * order; dense work vector; O(n^2) scanning, fine at oracle sizes), forward /
 * back substitution for A x = b (sys 0) and A^T x = b (sys 1, the real case of
 * UMFPACK_At, Umfpack.hs:95-97), and `irsteps` refinement steps.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef int64_t Int;

typedef struct {
  Int n;
  Int *Lp, *Li; double *Lx;   /* unit lower, by columns; rows are ORIGINAL row ids */
  Int *Up, *Ui; double *Ux;   /* upper, by columns; row index = pivot step k; diagonal last */
  Int *prow;                  /* prow[k] = original row chosen as k-th pivot */
  Int *pinv;                  /* pinv[row] = k */
  int singular;
} orc_lu;

static void *xrealloc(void *p, size_t n) { return realloc(p, n ? n : 1); }

void orc_lu_free(orc_lu *f) {
  if (!f) return;
  free(f->Lp); free(f->Li); free(f->Lx);
  free(f->Up); free(f->Ui); free(f->Ux);
  free(f->prow); free(f->pinv);
  free(f);
}

orc_lu *orc_lu_factor(Int n, const Int *Ap, const Int *Ai, const double *Ax) {
  orc_lu *f = (orc_lu *)calloc(1, sizeof(orc_lu));
  f->n = n;
  f->Lp = (Int *)calloc((size_t)n + 1, sizeof(Int));
  f->Up = (Int *)calloc((size_t)n + 1, sizeof(Int));
  f->prow = (Int *)malloc((size_t)(n ? n : 1) * sizeof(Int));
  f->pinv = (Int *)malloc((size_t)(n ? n : 1) * sizeof(Int));
  for (Int i = 0; i < n; ++i) f->pinv[i] = -1;
  Int lcap = 0, ucap = 0, lnz = 0, unz = 0;
  double *w = (double *)calloc((size_t)(n ? n : 1), sizeof(double));
  unsigned char *mark = (unsigned char *)calloc((size_t)(n ? n : 1), 1);
  Int *list = (Int *)malloc((size_t)(n ? n : 1) * sizeof(Int));
  for (Int j = 0; j < n; ++j) {
    Int nlist = 0;
    for (Int p = Ap[j]; p < Ap[j + 1]; ++p) {
      Int i = Ai[p];
      if (!mark[i]) { mark[i] = 1; list[nlist++] = i; }
      w[i] += Ax[p];
    }
    /* apply previous columns of L in pivot order */
    if (unz + j + 1 > ucap) {
      ucap = 2 * (unz + j + 1);
      f->Ui = (Int *)xrealloc(f->Ui, (size_t)ucap * sizeof(Int));
      f->Ux = (double *)xrealloc(f->Ux, (size_t)ucap * sizeof(double));
    }
    for (Int k = 0; k < j; ++k) {
      Int pr = f->prow[k];
      if (!mark[pr]) continue;
      double ukj = w[pr];
      f->Ui[unz] = k;
      f->Ux[unz++] = ukj;
      if (ukj != 0.0)
        for (Int p = f->Lp[k]; p < f->Lp[k + 1]; ++p) {
          Int i = f->Li[p];
          if (!mark[i]) { mark[i] = 1; list[nlist++] = i; }
          w[i] -= f->Lx[p] * ukj;
        }
    }
    /* partial pivoting among not-yet-pivotal rows; ties -> smallest row id */
    Int piv = -1;
    double best = -1.0;
    for (Int t = 0; t < nlist; ++t) {
      Int i = list[t];
      if (f->pinv[i] >= 0) continue;
      double a = fabs(w[i]);
      if (a > best || (a == best && i < piv)) { best = a; piv = i; }
    }
    if (piv < 0) { /* structurally singular column: pick any free row */
      for (Int i = 0; i < n; ++i) if (f->pinv[i] < 0) { piv = i; break; }
      best = 0.0;
    }
    double d = w[piv];
    if (d == 0.0) f->singular = 1;
    f->prow[j] = piv;
    f->pinv[piv] = j;
    f->Ui[unz] = j;
    f->Ux[unz++] = d;
    f->Up[j + 1] = unz;
    if (lnz + nlist > lcap) {
      lcap = 2 * (lnz + nlist);
      f->Li = (Int *)xrealloc(f->Li, (size_t)lcap * sizeof(Int));
      f->Lx = (double *)xrealloc(f->Lx, (size_t)lcap * sizeof(double));
    }
    for (Int t = 0; t < nlist; ++t) {
      Int i = list[t];
      if (f->pinv[i] < 0) { f->Li[lnz] = i; f->Lx[lnz++] = w[i] / d; }
      w[i] = 0.0;
      mark[i] = 0;
    }
    f->Lp[j + 1] = lnz;
  }
  free(w); free(mark); free(list);
  return f;
}

int orc_lu_is_singular(const orc_lu *f) { return f->singular; }

/* solve with the factors only (no refinement): sys 0: A x = b; 1: A^T x = b */
static void lu_solve_raw(const orc_lu *f, int sys, const double *b, double *x, double *w) {
  Int n = f->n;
  if (sys == 0) {
    /* P A = L U.  w indexed by ORIGINAL row; forward: for k ascending */
    memcpy(w, b, (size_t)n * sizeof(double));
    for (Int k = 0; k < n; ++k) {
      double yk = w[f->prow[k]];
      for (Int p = f->Lp[k]; p < f->Lp[k + 1]; ++p) w[f->Li[p]] -= f->Lx[p] * yk;
      x[k] = yk;
    }
    /* back substitution, U by columns with diagonal last */
    for (Int j = n - 1; j >= 0; --j) {
      Int pd = f->Up[j + 1] - 1;
      x[j] = x[j] / f->Ux[pd];
      for (Int p = f->Up[j]; p < pd; ++p) x[f->Ui[p]] -= f->Ux[p] * x[j];
    }
  } else {
    /* A^T = U^T L^T P  =>  U^T z = b ; L^T (P x) = z */
    for (Int j = 0; j < n; ++j) {
      double s = b[j];
      Int pd = f->Up[j + 1] - 1;
      for (Int p = f->Up[j]; p < pd; ++p) s -= f->Ux[p] * w[f->Ui[p]];
      w[j] = s / f->Ux[pd];
    }
    for (Int k = n - 1; k >= 0; --k) {
      double s = w[k];
      for (Int p = f->Lp[k]; p < f->Lp[k + 1]; ++p) s -= f->Lx[p] * x[f->Li[p]];
      x[f->prow[k]] = s;
    }
  }
}

/* r = b - op(A) x ; op = A (sys 0) or A^T (sys 1); A is CSC */
static void residual(Int n, const Int *Ap, const Int *Ai, const double *Ax, int sys,
                     const double *x, const double *b, double *r) {
  memcpy(r, b, (size_t)n * sizeof(double));
  if (sys == 0) {
    for (Int j = 0; j < n; ++j)
      for (Int p = Ap[j]; p < Ap[j + 1]; ++p) r[Ai[p]] -= Ax[p] * x[j];
  } else {
    for (Int j = 0; j < n; ++j) {
      double s = 0.0;
      for (Int p = Ap[j]; p < Ap[j + 1]; ++p) s += Ax[p] * x[Ai[p]];
      r[j] -= s;
    }
  }
}

int orc_lu_solve(const orc_lu *f, const Int *Ap, const Int *Ai, const double *Ax, int sys,
                 const double *b, double *x, int irsteps) {
  Int n = f->n;
  double *w = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));
  double *r = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));
  double *d = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));
  lu_solve_raw(f, sys, b, x, w);
  for (int it = 0; it < irsteps && !f->singular; ++it) {
    residual(n, Ap, Ai, Ax, sys, x, b, r);
    double rmax = 0.0;
    for (Int i = 0; i < n; ++i) rmax = fmax(rmax, fabs(r[i]));
    if (rmax == 0.0) break;
    lu_solve_raw(f, sys, r, d, w);
    for (Int i = 0; i < n; ++i) x[i] += d[i];
  }
  free(w); free(r); free(d);
  return f->singular ? 1 : 0;
}

/* one-shot: mat <\> b  (Umfpack.hs:48-50) with UMFPACK's default irstep = 2 */
int orc_linear_solve(Int n, const Int *Ap, const Int *Ai, const double *Ax, int sys,
                     const double *b, double *x) {
  orc_lu *f = orc_lu_factor(n, Ap, Ai, Ax);
  int st = orc_lu_solve(f, Ap, Ai, Ax, sys, b, x, 2);
  orc_lu_free(f);
  return st;
}

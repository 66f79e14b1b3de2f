/*
 * cpu_fair.c — TEST INFRASTRUCTURE.  NOT the reference's algorithm: an OpenMP
 * CSR row-gather SpMV with int32 indices on all host cores, reported next to
 * the reference-faithful single-thread baseline so the GPU is also compared
 * against a fair CPU.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

int orc_omp_threads(void) { return omp_get_max_threads(); }

void orc_csr_spmv_omp(int64_t nrows, const int32_t *rowptr, const int32_t *colidx,
                      const double *val, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nrows; ++r) {
    double acc = 0.0;
    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) acc = val[k] * x[colidx[k]] + acc;
    y[r] = acc;
  }
}

/* The same product timed fairly on a multi-socket host: the caller's arrays were written by ONE thread
 * (their pages sit on one NUMA node), so the row blocks are first copied into arrays that every thread
 * first-touches for the rows it will own (static schedule, same partition as the product loop; x is
 * touched in equal slices, i.e. spread over the nodes).  Returns the best of `reps` runs in seconds,
 * y_out = the product; < 0 if memory is short. */
double orc_csr_spmv_omp_timed(int64_t nrows, int64_t ncols, const int32_t *rowptr, const int32_t *colidx,
                              const double *val, const double *x, double *y_out, int reps) {
  const int64_t nnz = rowptr[nrows];
  int32_t *rp = malloc((size_t)(nrows + 1) * sizeof(int32_t));
  int32_t *ci = malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t));
  double *v = malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
  double *xx = malloc((size_t)(ncols > 0 ? ncols : 1) * sizeof(double));
  double *y = malloc((size_t)(nrows > 0 ? nrows : 1) * sizeof(double));
  double best = -1.0;
  if (rp && ci && v && xx && y) {
#pragma omp parallel
    {
#pragma omp for schedule(static)
      for (int64_t r = 0; r < nrows; ++r) {  /* first touch by the owner of row r */
        const int32_t a = rowptr[r], b = rowptr[r + 1];
        rp[r] = a;
        memcpy(ci + a, colidx + a, (size_t)(b - a) * sizeof(int32_t));
        memcpy(v + a, val + a, (size_t)(b - a) * sizeof(double));
        y[r] = 0.0;
      }
#pragma omp for schedule(static)
      for (int64_t j = 0; j < ncols; ++j) xx[j] = x[j];
    }
    rp[nrows] = (int32_t)nnz;
    for (int it = 0; it < (reps > 0 ? reps : 1); ++it) {
      const double t0 = omp_get_wtime();
      orc_csr_spmv_omp(nrows, rp, ci, v, xx, y);
      const double t = omp_get_wtime() - t0;
      if (best < 0.0 || t < best) best = t;
    }
    memcpy(y_out, y, (size_t)nrows * sizeof(double));
  }
  free(rp); free(ci); free(v); free(xx); free(y);
  return best;
}

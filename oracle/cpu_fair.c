/*
 * cpu_fair.c — TEST INFRASTRUCTURE.  NOT the reference's algorithm: an OpenMP
 * CSR row-gather SpMV with int32 indices on all host cores, reported next to
 * the reference-faithful single-thread baseline so the GPU is also compared
 * against a fair CPU.
 */
#include <stdint.h>
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

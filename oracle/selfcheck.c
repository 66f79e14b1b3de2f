/*
 * selfcheck.c — TEST INFRASTRUCTURE.  A small C driver that exercises every oracle entry
 * point on fixed inputs; tests/test_oracle_sanitizers.py builds it together with the oracle
 * sources under -fsanitize=address,undefined and runs it, so that out-of-bounds accesses,
 * leaks and undefined behaviour in the checker itself are caught (the reference has no
 * sanitizer story at all; GPU sanitizers are not available on this pool).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t Int;
void orc_free(void *p);
int orc_compress(Int, Int, Int, const Int *, const Int *, const double *, Int *, Int *, double *, Int *);
void orc_transpose(Int, Int, const Int *, const Int *, const double *, Int *, Int *, double *);
int orc_mulv(Int, Int, const Int *, const Int *, const double *, Int, const double *, double *);
int orc_mm(Int, Int, const Int *, const Int *, const double *, Int, Int, const Int *, const Int *, const double *,
           int, Int **, Int **, double **);
int orc_lin(double, Int, Int, const Int *, const Int *, const double *, double, Int, Int, const Int *, const Int *,
            const double *, Int **, Int **, double **);
int orc_check_matrix(Int, Int, Int, const Int *, Int, const Int *, Int);
int64_t orc_gen_random_csr(uint64_t, int64_t, int, int64_t, int64_t, int64_t *, int32_t *, double *);
int64_t orc_gen_poisson3d_csr(int64_t, int64_t *, int32_t *, double *);
int orc_linear_solve(Int, const Int *, const Int *, const double *, int, const double *, double *);

#define CHECK(c) do { if (!(c)) { printf("selfcheck FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(void) {
  enum { NR = 37, NC = 29, K = 400 };
  Int rows[K], cols[K], ptr[NC + 1], idx[K], bad = -1;
  double vals[K], val[K];
  uint64_t s = 12345;
  for (int k = 0; k < K; ++k) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    rows[k] = (Int)((s >> 33) % NR);
    cols[k] = (Int)((s >> 13) % NC);
    vals[k] = (double)((int)((s >> 50) % 11) - 5);
  }
  CHECK(orc_compress(NR, NC, K, rows, cols, vals, ptr, idx, val, &bad) == 0);
  const Int nz = ptr[NC];
  CHECK(orc_check_matrix(NR, NC, NC + 1, ptr, nz, idx, nz) == 0);
  Int tptr[NR + 1], tidx[K];
  double tval[K];
  orc_transpose(NR, NC, ptr, idx, val, tptr, tidx, tval);
  CHECK(orc_check_matrix(NC, NR, NR + 1, tptr, nz, tidx, nz) == 0);
  double x[NC], y[NR];
  for (int j = 0; j < NC; ++j) x[j] = j + 1;
  CHECK(orc_mulv(NR, NC, ptr, idx, val, NC, x, y) == 0);
  CHECK(orc_mulv(NR, NC, ptr, idx, val, NC - 1, x, y) != 0);
  Int *cp, *ci;
  double *cx;
  for (int literal = 0; literal < 2; ++literal) {
    CHECK(orc_mm(NR, NC, ptr, idx, val, NC, NR, tptr, tidx, tval, literal, &cp, &ci, &cx) == 0);
    CHECK(orc_check_matrix(NR, NR, NR + 1, cp, cp[NR], ci, cp[NR]) == 0);
    orc_free(cp); orc_free(ci); orc_free(cx);
  }
  CHECK(orc_lin(2.0, NR, NC, ptr, idx, val, -1.0, NR, NC, ptr, idx, val, &cp, &ci, &cx) == 0);
  CHECK(cp[NC] == nz);
  orc_free(cp); orc_free(ci); orc_free(cx);
  rows[3] = NR;  /* out of bounds must be reported, not written */
  CHECK(orc_compress(NR, NC, K, rows, cols, vals, ptr, idx, val, &bad) != 0 && bad == 3);
  /* generators: two-call protocol */
  int64_t rp[101];
  int64_t n1 = orc_gen_random_csr(0x5EED, 100000, 20, 500, 600, rp, NULL, NULL);
  int32_t *c32 = malloc((size_t)n1 * sizeof(int32_t));
  double *v64 = malloc((size_t)n1 * sizeof(double));
  CHECK(orc_gen_random_csr(0x5EED, 100000, 20, 500, 600, rp, c32, v64) == n1 && rp[100] == n1);
  free(c32); free(v64);
  /* solve: 5^3 Poisson, b = A*1 */
  enum { M = 5, N3 = M * M * M };
  int64_t prp[N3 + 1];
  int64_t n3 = orc_gen_poisson3d_csr(M, prp, NULL, NULL);
  int32_t *pc = malloc((size_t)n3 * sizeof(int32_t));
  double *pv = malloc((size_t)n3 * sizeof(double));
  orc_gen_poisson3d_csr(M, prp, pc, pv);
  Int *pi = malloc((size_t)n3 * sizeof(Int));
  for (int64_t k = 0; k < n3; ++k) pi[k] = pc[k];
  double ones[N3], b[N3], sol[N3];
  for (int i = 0; i < N3; ++i) ones[i] = 1.0;
  CHECK(orc_mulv(N3, N3, prp, pi, pv, N3, ones, b) == 0);
  CHECK(orc_linear_solve(N3, prp, pi, pv, 0, b, sol) == 0);
  for (int i = 0; i < N3; ++i) CHECK(sol[i] > 1 - 1e-12 && sol[i] < 1 + 1e-12);
  CHECK(orc_linear_solve(N3, prp, pi, pv, 1, b, sol) == 0);
  free(pc); free(pv); free(pi);
  printf("selfcheck OK\n");
  return 0;
}

// CPU driver of csrc/static_pivot.hpp for tests/test_static_pivot.py: reads a CSC matrix
// (n nnz / Ap / Ai / Ax) from stdin, prints ok=<0|1> and, when ok, the matched row of every column,
// the scalings and the entries of B = Dr P A Dc.
#include <cstdio>
#include <vector>
#include <algorithm>
#include "static_pivot.hpp"

int main() {
  int n = 0;
  long long nnz = 0;
  if (scanf("%d %lld", &n, &nnz) != 2) return 2;
  std::vector<int> Ap((size_t)n + 1), Ai((size_t)nnz);
  std::vector<double> Ax((size_t)nnz);
  for (auto &v : Ap) if (scanf("%d", &v) != 1) return 2;
  for (auto &v : Ai) if (scanf("%d", &v) != 1) return 2;
  for (auto &v : Ax) if (scanf("%lf", &v) != 1) return 2;
  spl::sp::Transversal T;
  const bool ok = spl::sp::max_product_transversal(n, Ap.data(), Ai.data(), Ax.data(), T);
  printf("ok=%d\n", ok ? 1 : 0);
  if (!ok) return 0;
  printf("rows");
  for (int j = 0; j < n; ++j) printf(" %d", T.row_of_col[(size_t)j]);
  printf("\ndr");
  for (int i = 0; i < n; ++i) printf(" %.17g", T.dr[(size_t)i]);
  printf("\ndc");
  for (int j = 0; j < n; ++j) printf(" %.17g", T.dc[(size_t)j]);
  std::vector<int> Bp, Bi;
  std::vector<double> Bx;
  spl::sp::permuted_scaled_csc(n, Ap.data(), Ai.data(), Ax.data(), T, Bp, Bi, Bx);
  printf("\nbi");
  for (int v : Bi) printf(" %d", v);
  printf("\nbx");
  for (double v : Bx) printf(" %.17g", v);
  printf("\n");
  return 0;
}

// Host-only timing of the maximum-product transversal (csrc/static_pivot.hpp) on a matrix written by
// tools/transversal_gen.py (families perm2d / mesh3d of tools/fuzz_lu_scale.py):
//   python tools/transversal_gen.py mesh3d 100 /tmp/m.bin && g++ -O2 -std=c++17 -pthread -I sparse-linear_amd/csrc \
//     tools/transversal_bench.cpp -o /tmp/tb && SPL_SP_VERBOSE=1 SPL_SP_THREADS=8 /tmp/tb /tmp/m.bin
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "static_pivot.hpp"
int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb");
  int64_t hdr[2]; if (fread(hdr, 8, 2, f) != 2) return 1;
  int n = (int)hdr[0]; int64_t nnz = hdr[1];
  std::vector<int> Ap(n + 1), Ai(nnz); std::vector<double> Ax(nnz);
  if (fread(Ap.data(), 4, n + 1, f) != (size_t)n + 1 || fread(Ai.data(), 4, nnz, f) != (size_t)nnz || fread(Ax.data(), 8, nnz, f) != (size_t)nnz) return 1;
  fclose(f);
  spl::sp::Transversal T;
  auto t0 = std::chrono::steady_clock::now();
  bool ok = spl::sp::max_product_transversal(n, Ap.data(), Ai.data(), Ax.data(), T);
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  double logprod = 0;
  if (ok) for (int j = 0; j < n; ++j) { int i = T.row_of_col[j]; for (int p = Ap[j]; p < Ap[j+1]; ++p) if (Ai[p] == i) logprod += std::log(std::fabs(Ax[p])); }
  printf("n=%d ok=%d seconds=%.3f logprod=%.9f\n", n, (int)ok, dt, logprod);
  return 0;
}

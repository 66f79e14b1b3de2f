// Host-only timing of the nested-dissection analysis (csrc/mf_symbolic.hpp) on 2-D / 3-D grid patterns:
//   g++ -O2 -std=c++17 -pthread -I sparse-linear_amd/csrc tools/mf_bench.cpp -o /tmp/mf_bench && /tmp/mf_bench 2 3000
// prints the seconds of build_tree, the number of fronts and the flops of the factorisation it plans.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mf_symbolic.hpp"

int main(int argc, char **argv) {
  const int dim = argc > 1 ? atoi(argv[1]) : 3, m = argc > 2 ? atoi(argv[2]) : 100, reps = argc > 3 ? atoi(argv[3]) : 1;
  const long n = dim == 2 ? (long)m * m : (long)m * m * m;
  std::vector<int> Ap((size_t)n + 1, 0), Ai;
  Ai.reserve((size_t)n * (dim == 2 ? 5 : 7));
  auto id = [&](int x, int y, int z) { return (long)x + (long)m * ((long)y + (long)m * z); };
  const int mz = dim == 2 ? 1 : m;
  for (int z = 0; z < mz; ++z)
    for (int y = 0; y < m; ++y)
      for (int x = 0; x < m; ++x) {
        const long j = id(x, y, z);
        if (z > 0) Ai.push_back((int)id(x, y, z - 1));
        if (y > 0) Ai.push_back((int)id(x, y - 1, z));
        if (x > 0) Ai.push_back((int)id(x - 1, y, z));
        Ai.push_back((int)j);
        if (x + 1 < m) Ai.push_back((int)id(x + 1, y, z));
        if (y + 1 < m) Ai.push_back((int)id(x, y + 1, z));
        if (z + 1 < mz) Ai.push_back((int)id(x, y, z + 1));
        Ap[(size_t)j + 1] = (int)Ai.size();
      }
  for (int r = 0; r < reps; ++r) {
    spl::mf::Tree T;
    const auto t0 = std::chrono::steady_clock::now();
    spl::mf::build_tree((int)n, Ap.data(), Ai.data(), 256, T);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int maxfs = 0;
    for (int f = 0; f < T.nfronts; ++f) maxfs = maxfs > T.fs(f) ? maxfs : T.fs(f);
    // order-sensitive checksums of the permutation and of the boundary lists: two builds of the analysis agree on the tree
    unsigned long long hp = 1469598103934665603ull, hb = hp;
    for (int v : T.perm) hp = (hp ^ (unsigned)v) * 1099511628211ull;
    for (int v : T.bidx) hb = (hb ^ (unsigned)v) * 1099511628211ull;
    for (int f = 0; f < T.nfronts; ++f) hb = (hb ^ (unsigned)T.nb[(size_t)f]) * 1099511628211ull;
    printf("%d-D grid %d: n=%ld analyze %.3f s fronts=%d depth=%d maxfront=%d flops=%.4g panel_GB=%.2f perm=%016llx bnd=%016llx (%zu)\n",
           dim, m, n, dt, T.nfronts, T.maxdepth, maxfs, T.flops, (double)T.panel_elems * 8 / 1e9, hp, hb, T.bidx.size());
  }
  return 0;
}

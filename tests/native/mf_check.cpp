// CPU self-check of the nested-dissection ordering / frontal tree (sparse-linear_amd/csrc/mf_symbolic.hpp):
// reads "n nnz" then Ap (n+1) and Ai (nnz) from stdin, builds the tree and verifies its invariants.
// Prints one line of statistics; exit status 0 iff every invariant holds.
#include <cstdio>
#include <vector>

#include "mf_symbolic.hpp"

using namespace spl::mf;

int main(int argc, char **argv) {
  int leaf = argc > 1 ? atoi(argv[1]) : 256;
  int n = 0, nnz = 0;
  if (scanf("%d %d", &n, &nnz) != 2) return 2;
  std::vector<int> Ap((size_t)n + 1), Ai((size_t)nnz);
  for (int &v : Ap) if (scanf("%d", &v) != 1) return 2;
  for (int &v : Ai) if (scanf("%d", &v) != 1) return 2;
  const int mult = argc > 2 ? atoi(argv[2]) : 1;
  Tree T;
  build_tree(n, Ap.data(), Ai.data(), leaf, T, mult);
  if (mult > 1) {  // the invariants are checked against the expanded pattern: a dense mult x mult block per entry
    std::vector<int> Ep((size_t)n * mult + 1, 0), Ei;
    for (int j = 0; j < n; ++j)
      for (int h = 0; h < mult; ++h) {
        for (int q = Ap[(size_t)j]; q < Ap[(size_t)j + 1]; ++q)
          for (int a = 0; a < mult; ++a) Ei.push_back(Ai[(size_t)q] * mult + a);
        Ep[(size_t)j * mult + h + 1] = (int)Ei.size();
      }
    n *= mult;
    Ap.swap(Ep);
    Ai.swap(Ei);
  }
  long bad = T.n == n ? 0 : 1;
  // perm / inv are inverse permutations
  std::vector<char> hit((size_t)n, 0);
  for (int g = 0; g < n; ++g) {
    const int v = T.perm[(size_t)g];
    if (v < 0 || v >= n || hit[(size_t)v] || T.inv[(size_t)v] != g) ++bad; else hit[(size_t)v] = 1;
  }
  // fronts: post-order (children before parents), pivot ranges tile [0, n)
  int next = 0;
  for (int f = 0; f < T.nfronts; ++f) {
    if (T.p0[(size_t)f] != next) ++bad;
    next += T.np[(size_t)f];
    if (T.parent[(size_t)f] >= 0 && T.parent[(size_t)f] <= f) ++bad;
    for (int g = T.p0[(size_t)f]; g < T.p0[(size_t)f] + T.np[(size_t)f]; ++g)
      if (T.front_of[(size_t)g] != f) ++bad;
  }
  if (next != n) ++bad;
  // boundary lists: ascending, later than the pivots, owned by an ancestor, and passed on to the parent
  for (int f = 0; f < T.nfronts; ++f) {
    const int last = T.p0[(size_t)f] + T.np[(size_t)f], p = T.parent[(size_t)f];
    for (int64_t q = T.bptr[(size_t)f]; q < T.bptr[(size_t)f + 1]; ++q) {
      const int g = T.bidx[(size_t)q];
      if (g < last || (q > T.bptr[(size_t)f] && T.bidx[(size_t)q - 1] >= g)) ++bad;
      int a = T.front_of[(size_t)g], c = f;
      while (c >= 0 && c != a) c = T.parent[(size_t)c];
      if (c != a) ++bad;
      if (p < 0) { ++bad; continue; }
      const int plast = T.p0[(size_t)p] + T.np[(size_t)p];
      if (g >= plast) {  // not one of the parent's pivots: must be in the parent's boundary
        bool found = false;
        for (int64_t r = T.bptr[(size_t)p]; r < T.bptr[(size_t)p + 1] && !found; ++r) found = T.bidx[(size_t)r] == g;
        if (!found) ++bad;
      }
    }
  }
  // every entry of A + A^T couples an index with one inside the front of the earlier-eliminated one
  for (int j = 0; j < n; ++j)
    for (int q = Ap[(size_t)j]; q < Ap[(size_t)j + 1]; ++q) {
      int gi = T.inv[(size_t)Ai[(size_t)q]], gj = T.inv[(size_t)j];
      if (gi == gj) continue;
      const int lo = gi < gj ? gi : gj, hi = gi < gj ? gj : gi, f = T.front_of[(size_t)lo];
      if (hi < T.p0[(size_t)f] + T.np[(size_t)f]) continue;  // both pivots of the same front
      bool found = false;
      for (int64_t r = T.bptr[(size_t)f]; r < T.bptr[(size_t)f + 1] && !found; ++r) found = T.bidx[(size_t)r] == hi;
      if (!found) ++bad;
    }
  int maxfs = 0;
  for (int f = 0; f < T.nfronts; ++f) maxfs = maxfs > T.fs(f) ? maxfs : T.fs(f);
  unsigned long long h = 1469598103934665603ull;  // FNV-1a of the ordering and the tree
  for (int v : T.perm) { h ^= (unsigned)v; h *= 1099511628211ull; }
  for (int v : T.parent) { h ^= (unsigned)v; h *= 1099511628211ull; }
  printf("n=%d fronts=%d depth=%d maxfront=%d flops=%.4g bad=%ld hash=%016llx\n", n, T.nfronts, T.maxdepth, maxfs, T.flops,
         bad, h);
  return bad != 0;
}

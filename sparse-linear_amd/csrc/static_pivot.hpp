// static_pivot.hpp — host side of the "static pivoting" stage of the LU (plain C++17, no HIP).
//
// Why: the multifrontal factorisation (multifrontal.hip) eliminates without interchanges.  That is provably
// safe for column-dominant matrices and is kept as a checked speculation for every other matrix; when the
// check fails (a solve's backward error does not reach rounding level) the reference's solver — UMFPACK,
// suitesparse/src/Numeric/LinearAlgebra/Umfpack.hs:71-83 — would have pivoted.  Pivoting inside a front with
// pivots delayed to the parent changes the sizes of the fronts while they are being factored, which the
// flat, pre-planned layout of the GPU factorisation (panels, level regions, lockstep launches) cannot do.
// What can be done before the factorisation is what distributed-memory direct solvers do in the same
// situation (SuperLU_DIST's GESP, MUMPS' static pivoting): choose the pivots up front by a
// maximum-product transversal — a row permutation that puts on the diagonal entries whose product is as
// large as possible — together with the row and column scalings its dual variables give (Duff & Koster,
// "On algorithms for permuting large entries to the diagonal of a sparse matrix", SIAM J. Matrix Anal.
// Appl. 22 (2001); the algorithm behind HSL MC64 job 5, restated here from the paper):
//   B = Dr P A Dc   with   |b_jj| = 1,  |b_ij| <= 1.
// B is then ordered (nested dissection on its own pattern), factored without interchanges on the tree and
// every solve still checks its backward error against the ORIGINAL A (iterative refinement as before); the
// band factorisation with partial pivoting stays the last resort.
//
// Matching: cost c_ij = log(max_i |a_ij|) - log |a_ij| >= 0; minimum-cost perfect matching by successive
// shortest augmenting paths (Dijkstra on reduced costs, one search per column the greedy start left
// unmatched), dual variables u (rows), v (columns) with u_i + v_j <= c_ij, equality on matched entries;
// scalings dr_i = exp(u_i), dc_j = exp(v_j) / max_i |a_ij|.
#pragma once

#include <stdint.h>
#include <chrono>
#include <cmath>
#include <limits>
#include <queue>
#include <utility>
#include <vector>

namespace spl {
namespace sp {

struct Transversal {
  std::vector<int> row_of_col;  // row matched to column j (that entry becomes b_jj)
  std::vector<int> col_of_row;  // = new index of row i
  std::vector<double> dr, dc;   // row / column scalings
};

// false: structurally singular (no perfect matching over the non-zero entries), n == 0, or out of time
inline bool max_product_transversal(int n, const int *Ap, const int *Ai, const double *Ax, Transversal &T,
                                    double max_seconds = 1e30) {
  if (n <= 0) return false;
  const double inf = std::numeric_limits<double>::infinity();
  const int64_t nnz = Ap[n];
  std::vector<double> c((size_t)nnz), cmax((size_t)n, 0.0);
  for (int j = 0; j < n; ++j) {
    double m = 0.0;
    for (int p = Ap[j]; p < Ap[j + 1]; ++p) m = std::max(m, std::fabs(Ax[p]));
    if (!(m > 0.0) || !std::isfinite(m)) return false;  // an empty (or all-zero, or non-finite) column
    cmax[(size_t)j] = m;
    const double lm = std::log(m);
    for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
      const double a = std::fabs(Ax[p]);
      c[(size_t)p] = a > 0.0 ? lm - std::log(a) : inf;
    }
  }
  std::vector<double> u((size_t)n, inf), v((size_t)n, 0.0);
  for (int j = 0; j < n; ++j)
    for (int p = Ap[j]; p < Ap[j + 1]; ++p) u[(size_t)Ai[p]] = std::min(u[(size_t)Ai[p]], c[(size_t)p]);
  for (int i = 0; i < n; ++i)
    if (u[(size_t)i] == inf) return false;  // an empty row
  // column duals: v_j = min_i (c_ij - u_i), so that every column has a tight entry as well as every row
  for (int j = 0; j < n; ++j) {
    double m = inf;
    for (int p = Ap[j]; p < Ap[j + 1]; ++p) m = std::min(m, c[(size_t)p] - u[(size_t)Ai[p]]);
    v[(size_t)j] = m;
  }
  T.row_of_col.assign((size_t)n, -1);
  T.col_of_row.assign((size_t)n, -1);
  auto tight = [&](int p, int i, int j) { return c[(size_t)p] - u[(size_t)i] - v[(size_t)j] <= 1e-14; };
  // greedy start: tight entries of unmatched rows ...
  for (int j = 0; j < n; ++j)
    for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
      const int i = Ai[p];
      if (T.col_of_row[(size_t)i] < 0 && tight(p, i, j)) {
        T.col_of_row[(size_t)i] = j;
        T.row_of_col[(size_t)j] = i;
        break;
      }
    }
  // ... then augmenting paths of length two: a tight entry (i, j) whose row is taken by column j2, which has
  // another tight entry in a free row (each column is rescanned from where its last scan stopped)
  {
    std::vector<int> next((size_t)n);
    for (int j = 0; j < n; ++j) next[(size_t)j] = Ap[j];
    for (int j = 0; j < n; ++j) {
      if (T.row_of_col[(size_t)j] >= 0) continue;
      for (int p = Ap[j]; p < Ap[j + 1] && T.row_of_col[(size_t)j] < 0; ++p) {
        const int i = Ai[p];
        if (!tight(p, i, j)) continue;
        const int j2 = T.col_of_row[(size_t)i];
        if (j2 < 0) {  // freed meanwhile
          T.col_of_row[(size_t)i] = j;
          T.row_of_col[(size_t)j] = i;
          break;
        }
        for (int &q = next[(size_t)j2]; q < Ap[j2 + 1]; ++q) {
          const int i2 = Ai[q];
          if (T.col_of_row[(size_t)i2] < 0 && tight(q, i2, j2)) {
            T.col_of_row[(size_t)i2] = j2;
            T.row_of_col[(size_t)j2] = i2;
            T.col_of_row[(size_t)i] = j;
            T.row_of_col[(size_t)j] = i;
            ++q;
            break;
          }
        }
      }
    }
  }
  // shortest augmenting paths for the rest
  std::vector<double> d((size_t)n, inf);
  std::vector<int> pred((size_t)n, -1);     // column from which row i was reached
  std::vector<char> done((size_t)n, 0);
  std::vector<int> touched, settled;
  typedef std::pair<double, int> Item;
  std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
  // Time budget: the searches are sequential host work (about 45 rows settled per unknown on a 3-D mesh with a
  // useless diagonal: 3 s at 2e5 unknowns, 34 s at 1e6 on one core).  A matrix that needs more than
  // `max_seconds` is left to the caller's next fallback rather than holding a solve call for many minutes.
  const auto t_start = std::chrono::steady_clock::now();
  int64_t work = 0, next_check = 1 << 20;
  for (int j0 = 0; j0 < n; ++j0) {
    if (T.row_of_col[(size_t)j0] >= 0) continue;
    if (work > next_check) {
      next_check = work + (1 << 20);
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > max_seconds) return false;
    }
    touched.clear();
    settled.clear();
    while (!heap.empty()) heap.pop();
    int j = j0, end_row = -1;
    double lowest = 0.0;  // distance at which column j was reached
    double best_free = inf;  // shortest distance to a free row seen so far: nothing longer can be the answer
    for (;;) {
      for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
        const int i = Ai[p];
        if (done[(size_t)i] || c[(size_t)p] == inf) continue;
        const double dn = lowest + (c[(size_t)p] - u[(size_t)i] - v[(size_t)j]);
        if (dn >= best_free) continue;
        if (T.col_of_row[(size_t)i] < 0) best_free = dn;
        if (dn < d[(size_t)i]) {
          if (d[(size_t)i] == inf) touched.push_back(i);
          d[(size_t)i] = dn;
          pred[(size_t)i] = j;
          heap.push(Item(dn, i));
        }
      }
      int i = -1;
      while (!heap.empty()) {
        const Item it = heap.top();
        heap.pop();
        if (!done[(size_t)it.second] && it.first <= d[(size_t)it.second]) { i = it.second; break; }
      }
      if (i < 0) break;  // no augmenting path: structurally singular
      done[(size_t)i] = 1;
      settled.push_back(i);
      ++work;
      lowest = d[(size_t)i];
      if (T.col_of_row[(size_t)i] < 0) { end_row = i; break; }
      j = T.col_of_row[(size_t)i];
    }
    if (end_row < 0) return false;
    const double L = d[(size_t)end_row];
    // dual update: settled rows u_i += d_i - L; their matched columns (and j0) v_j += L - (distance of j)
    for (int i : settled) {
      const int jm = T.col_of_row[(size_t)i];
      if (jm >= 0) v[(size_t)jm] += L - d[(size_t)i];
      u[(size_t)i] += d[(size_t)i] - L;
    }
    v[(size_t)j0] += L;
    // augment along the predecessor columns
    for (int i = end_row;;) {
      const int jc = pred[(size_t)i];
      const int prev = T.row_of_col[(size_t)jc];
      T.row_of_col[(size_t)jc] = i;
      T.col_of_row[(size_t)i] = jc;
      if (jc == j0) break;
      i = prev;
    }
    for (int i : touched) { d[(size_t)i] = inf; pred[(size_t)i] = -1; done[(size_t)i] = 0; }
  }
  T.dr.resize((size_t)n);
  T.dc.resize((size_t)n);
  for (int i = 0; i < n; ++i) T.dr[(size_t)i] = std::exp(u[(size_t)i]);
  for (int j = 0; j < n; ++j) T.dc[(size_t)j] = std::exp(v[(size_t)j]) / cmax[(size_t)j];
  for (int k = 0; k < n; ++k)
    if (!std::isfinite(T.dr[(size_t)k]) || !std::isfinite(T.dc[(size_t)k]) || !(T.dr[(size_t)k] > 0.0) ||
        !(T.dc[(size_t)k] > 0.0))
      return false;  // scalings out of range: leave the matrix to the pivoting fallback
  return true;
}

// B = Dr P A Dc in CSC with sorted row indices: B(col_of_row[i], j) = dr[i] a_ij dc[j]
inline void permuted_scaled_csc(int n, const int *Ap, const int *Ai, const double *Ax, const Transversal &T,
                                std::vector<int> &Bp, std::vector<int> &Bi, std::vector<double> &Bx) {
  Bp.assign(Ap, Ap + n + 1);
  const int64_t nnz = Ap[n];
  Bi.resize((size_t)nnz);
  Bx.resize((size_t)nnz);
  std::vector<std::pair<int, double>> col;
  for (int j = 0; j < n; ++j) {
    col.clear();
    for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
      const int i = Ai[p];
      col.emplace_back(T.col_of_row[(size_t)i], T.dr[(size_t)i] * Ax[p] * T.dc[(size_t)j]);
    }
    std::sort(col.begin(), col.end(), [](const std::pair<int, double> &a, const std::pair<int, double> &b) { return a.first < b.first; });
    for (size_t k = 0; k < col.size(); ++k) {
      Bi[(size_t)Ap[j] + k] = col[k].first;
      Bx[(size_t)Ap[j] + k] = col[k].second;
    }
  }
}

}  // namespace sp
}  // namespace spl

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
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
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
  // Shortest augmenting paths for the rest: one Dijkstra search on reduced costs per unmatched column, SEVERAL AT A
  // TIME (round 4).  A search only reads the duals and the matching of the rows it reaches, and what it would do
  // to them — dual updates on the rows it settled, the augmentation along its path — stays inside that set.  So the
  // next `batch` unmatched columns are searched side by side by a team of threads against the state as it is, each with
  // distances, predecessors and a heap of its own; then the results are applied one after the other in column order,
  // and a result is applied only if no row it reached has been changed by a result applied before it in this round
  // (it is then exactly what a search started now would find); the others stay unmatched and are searched again.
  // Searches are local — a few hundred rows of 10^5 .. 10^6 while many columns are unmatched — so most results stand;
  // towards the end they grow and collide, and the batch shrinks with the share that stood.  What is applied, and in
  // which order, depends on the results only, never on which thread produced them or when: the matching and the
  // scalings are the same for every team size (tests/test_static_pivot.py), and still a minimum-cost perfect matching
  // with its dual variables — a sequence of shortest augmenting paths, in another order of the columns.
  // Per row the shared state is one 16-byte record (dual, match) and a thread's own another (distance, predecessor,
  // settled): a search is a chain of cache misses, one per array it touches per row (round 3 had five arrays); the
  // heap is a 4-ary array heap with lazy deletion, emptied by forgetting its length.
  struct Shared {
    double u;
    int col, dirty;  // dirty: the round in which a result last changed this row
  };
  struct Own {
    double d;
    int pred, done;
  };
  struct Item {
    double d;
    int i;
  };
  struct Heap {
    std::vector<Item> a;
    void clear() { a.clear(); }
    bool empty() const { return a.empty(); }
    void push(double dd, int i) {
      size_t k = a.size();
      a.push_back(Item{dd, i});
      while (k > 0) {
        const size_t par = (k - 1) >> 2;
        if (a[par].d <= dd) break;
        a[k] = a[par];
        k = par;
      }
      a[k] = Item{dd, i};
    }
    Item pop() {
      const Item top = a[0];
      const Item last = a.back();
      a.pop_back();
      const size_t sz = a.size();
      if (sz > 0) {
        size_t k = 0;
        for (;;) {
          const size_t c0 = 4 * k + 1;
          if (c0 >= sz) break;
          size_t best = c0;
          const size_t ce = c0 + 4 < sz ? c0 + 4 : sz;
          for (size_t q = c0 + 1; q < ce; ++q)
            if (a[q].d < a[best].d) best = q;
          if (a[best].d >= last.d) break;
          a[k] = a[best];
          k = best;
        }
        a[k] = last;
      }
      return top;
    }
  };
  struct Result {
    int j0 = -1, end_row = -1;
    double L = 0.0;
    std::vector<int> touched;                      // every row the search reached
    std::vector<std::pair<int, double>> settled;   // (row, distance)
    std::vector<int> path;                         // rows of the augmenting path, end row first
    std::vector<int> path_col;                     // path_col[k]: the column path[k] will be matched to
    int64_t work = 0;
  };
  struct Scratch {
    std::vector<Own> own;
    Heap heap;
  };
  std::vector<Shared> R((size_t)n);
  for (int i = 0; i < n; ++i) R[(size_t)i] = Shared{u[(size_t)i], T.col_of_row[(size_t)i], -1};
  std::vector<int> pending;
  for (int j = 0; j < n; ++j)
    if (T.row_of_col[(size_t)j] < 0) pending.push_back(j);
  // The order of the columns is free.  Ascending, neighbours in the numbering — neighbours in a mesh — would be searched
  // in the same round and collide; a fixed pseudo-random order (the same for every run and team) spreads a round over
  // the whole matrix.
  {
    uint64_t state = 0x9E3779B97F4A7C15ull;
    for (size_t k = pending.size(); k > 1; --k) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      std::swap(pending[k - 1], pending[(size_t)((state >> 33) % k)]);
    }
  }
  auto search = [&](int j0, Scratch &S, Result &out) {
    out.j0 = j0;
    out.end_row = -1;
    out.touched.clear();
    out.settled.clear();
    out.path.clear();
    out.path_col.clear();
    out.work = 0;
    S.heap.clear();
    Own *own = S.own.data();
    int j = j0;
    double lowest = 0.0;     // distance at which column j was reached
    double best_free = inf;  // shortest distance to a free row seen so far: nothing longer can be the answer
    for (;;) {
      const double base = lowest - v[(size_t)j];
      // (round 5: the records of all rows of the column requested before the first is looked at — a search is a chain of
      // cache misses, 85 % of its cycles in this loop and 15 % in the heap: 26.4 -> 22.7 s on the 1e6-unknown 3-D mesh, 8 threads)
      for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
        __builtin_prefetch(&own[Ai[p]]);
        __builtin_prefetch(&R[(size_t)Ai[p]]);
      }
      for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
        const int i = Ai[p];
        Own &o = own[i];
        if (o.done) continue;
        const Shared &r = R[(size_t)i];
        const double dn = base + (c[(size_t)p] - r.u);  // (an infinite cost stays infinite: never below best_free)
        if (!(dn < best_free)) continue;
        if (r.col < 0) best_free = dn;
        if (dn < o.d) {
          if (o.d == inf) out.touched.push_back(i);
          o.d = dn;
          o.pred = j;
          S.heap.push(dn, i);
        }
      }
      int i = -1;
      while (!S.heap.empty()) {
        const Item it = S.heap.pop();
        const Own &o = own[it.i];
        if (!o.done && it.d <= o.d) { i = it.i; break; }
      }
      if (i < 0) break;  // no augmenting path
      Own &o = own[i];
      o.done = 1;
      out.settled.emplace_back(i, o.d);
      ++out.work;
      lowest = o.d;
      if (R[(size_t)i].col < 0) { out.end_row = i; break; }
      j = R[(size_t)i].col;
    }
    if (out.end_row >= 0) {
      out.L = own[out.end_row].d;
      for (int i = out.end_row;;) {
        const int jc = own[i].pred;
        out.path.push_back(i);
        out.path_col.push_back(jc);
        if (jc == j0) break;
        i = T.row_of_col[(size_t)jc];
      }
    }
    for (int i : out.touched) own[i] = Own{inf, -1, 0};
  };
  // the team: SPL_SP_THREADS, by default the hardware's threads up to 8 (measured on the GPU box's host, 1e6-unknown
  // 3-D mesh with a useless diagonal: 27.6 s with one thread, 19.7 with four or eight, 22.3 with sixteen — the last
  // hundreds of columns search most of the matrix each and collide, so the end is sequential whatever the team);
  // small problems are not worth threads
  int team = 1;
  {
    const char *e = getenv("SPL_SP_THREADS");
    const unsigned hw = std::thread::hardware_concurrency();
    team = e ? atoi(e) : (int)std::min<unsigned>(hw ? hw : 1u, 8u);
    if (team < 1) team = 1;
    if (!e && pending.size() < 2000) team = 1;
  }
  constexpr int kMaxBatch = 64, kMinBatch = 1;
  std::vector<Scratch> scratch((size_t)team);
  for (Scratch &S : scratch) S.own.assign((size_t)n, Own{inf, -1, 0});
  std::vector<Result> results((size_t)kMaxBatch);
  // round state shared with the helpers (spinning on atomics: a round lasts microseconds to milliseconds)
  std::atomic<int> round_id{0}, next_item{0}, done_items{0}, stop{0};
  int batch_n = 0;          // written before round_id is advanced (release), read after (acquire)
  const int *batch_cols = nullptr;
  auto work_on_round = [&](Scratch &S) {
    for (;;) {
      const int k = next_item.fetch_add(1, std::memory_order_relaxed);
      if (k >= batch_n) break;
      search(batch_cols[k], S, results[(size_t)k]);
      done_items.fetch_add(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> helpers;
  for (int t = 1; t < team; ++t) {
    try {
      helpers.emplace_back([&, t] {
        int seen = 0;
        for (;;) {
          int now;
          int spins = 0;
          while ((now = round_id.load(std::memory_order_acquire)) == seen) {
            if (stop.load(std::memory_order_acquire)) return;
            if (++spins > 2000) { std::this_thread::yield(); spins = 0; }
          }
          seen = now;
          work_on_round(scratch[(size_t)t]);
        }
      });
    } catch (...) {
      break;  // fewer helpers than asked for: the rounds do not depend on their number
    }
  }
  struct StopHelpers {
    std::atomic<int> &stop;
    std::vector<std::thread> &helpers;
    ~StopHelpers() {
      stop.store(1, std::memory_order_release);
      for (std::thread &th : helpers) th.join();
    }
  } stop_helpers{stop, helpers};
  // Time budget: a matrix that needs more than `max_seconds` is left to the caller's next fallback rather than
  // holding a solve call for many minutes.
  const auto t_start = std::chrono::steady_clock::now();
  int64_t work = 0, next_check = 1 << 20;
  int batch = kMaxBatch, round = 0;
  size_t head = 0;  // pending[head ..) are the unmatched columns
  std::vector<int> retry;
  while (head < pending.size()) {
    if (work > next_check) {
      next_check = work + (1 << 20);
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > max_seconds) return false;
    }
    ++round;
    batch_n = (int)std::min<size_t>((size_t)batch, pending.size() - head);
    batch_cols = pending.data() + head;
    next_item.store(0, std::memory_order_relaxed);
    done_items.store(0, std::memory_order_relaxed);
    round_id.store(round, std::memory_order_release);
    work_on_round(scratch[0]);
    while (done_items.load(std::memory_order_acquire) < batch_n) {
    }
    // apply the results in column order
    retry.clear();
    int stood = 0;
    for (int k = 0; k < batch_n; ++k) {
      Result &res = results[(size_t)k];
      work += res.work;
      if (res.end_row < 0) return false;  // no augmenting path: structurally singular (no applied result can create one)
      bool clean = true;
      for (int i : res.touched)
        if (R[(size_t)i].dirty == round) { clean = false; break; }
      if (!clean) { retry.push_back(res.j0); continue; }
      ++stood;
      const double L = res.L;
      // dual update: settled rows u_i += d_i - L; their matched columns (and j0) v_j += L - (distance of j)
      for (const std::pair<int, double> &sd : res.settled) {
        Shared &r = R[(size_t)sd.first];
        if (r.col >= 0) v[(size_t)r.col] += L - sd.second;
        r.u += sd.second - L;
        r.dirty = round;
      }
      v[(size_t)res.j0] += L;
      for (size_t q = 0; q < res.path.size(); ++q) {
        T.row_of_col[(size_t)res.path_col[q]] = res.path[q];
        R[(size_t)res.path[q]].col = res.path_col[q];
      }
    }
    // the columns that have to be searched again go back to the front of the queue, in order
    head += (size_t)batch_n - retry.size();
    std::copy(retry.begin(), retry.end(), pending.begin() + (int64_t)head);
    batch = std::max(kMinBatch, std::min(kMaxBatch, 2 * stood));
  }
  for (int i = 0; i < n; ++i) { u[(size_t)i] = R[(size_t)i].u; T.col_of_row[(size_t)i] = R[(size_t)i].col; }
  if (getenv("SPL_SP_VERBOSE")) fprintf(stderr, "[transversal] %d threads, %d rounds, %lld rows settled\n", team, round, (long long)work);
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

// umfpack_zi.hip — the complex (`zi`) half of the UMFPACK link-time ABI
// (suitesparse/src/Numeric/LinearAlgebra/Umfpack/Internal.hs:69-135).
//
// Round 3: where the analysis chooses the multifrontal tree (and the tree has work to halve), the factors are NATIVE
// COMPLEX fronts on the tree of the complex pattern (csrc/multifrontal.hip, TreeView::zm; csrc/dense_lu_kernels.hpp,
// the *_z functions).  The object still holds the real embedding described below — packed complex vectors are its
// real vectors, so the complex factors solve it — for residuals, refinement and every fallback; the band path and
// small trees factor it as before.
//
// A complex n x n system  (R + iI)(x + iy) = b + ic  is solved as the real 2n x 2n system with
// interleaved unknowns (x0, y0, x1, y1, ...):  block (i,j) of the embedding E is [[R, -I], [I, R]].
// Packed complex vectors (the only form the reference uses: Az = Xz = Bz = NULL,
// Internal.hs:124-132) are then exactly the real vectors of the embedded system, and E^T is the
// embedding of A^H, so sys = 1 (UMFPACK_At, the conjugate transpose) maps to the real transposed
// solve.  Symbolic / Numeric handles are the `di` handles of E; all arithmetic runs in the same
// GPU kernels as the real path (band LU, banded solves, SpMV-based refinement); values differ from a complex-
// arithmetic LU only in rounding, and `ident <\> v == v` (suitesparse/tests/test-umfpack.hs:16-19,
// on Vector (Complex Double)) holds exactly.
//
// Complex SYMMETRIC matrices (A == A^T, not Hermitian: FEAST's z B - A for real symmetric A, B) get a SYMMETRIC real
// embedding, so that the multifrontal tree runs in its L D L^T mode (half the flops, dense_lu_kernels.hpp Band::sym):
// with M(a) = [[re a, -im a], [im a, re a]] and C = diag(1, -1), the block C M(w) = [[re w, -im w], [-im w, -re w]]
// is a symmetric 2 x 2 matrix for every complex w, hence E' = (C M(w_ij)) is symmetric whenever w_ij = w_ji.  w is
// A under the congruence D A D with a unit-modulus diagonal D = diag(u_r), u_r^2 = conj(a_rr) / |a_rr|: the diagonal
// of D A D is real and positive, so the scalar pivots of a diagonal block start as +|a_rr|, -|a_rr| (the role the
// swap of the two equations plays in the general case).  E' = T E W with T = blockdiag(C M(u_r)), W =
// blockdiag(M(u_r)):  A x = b  is  E' x' = T b, x = W x';  A^H y = c  is  E' y' = W^T c, y = T^T y'.
#include <chrono>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

#include "common.hpp"
#include "../../include/umfpack_hip.h"

namespace {

// CSC arrays of the embedding E (2n x 2n, 4 entries per complex entry, rows ascending)
struct Embedded {
  // plain arrays, NOT std::vector: resize() would zero 240 MB on one thread (40 ms at 10^6 unknowns) before the
  // threads of embed() write every element anyway
  template <typename T>
  struct Raw {
    std::unique_ptr<T[]> mem;
    void resize(size_t n) { mem.reset(new T[n ? n : 1]); }
    T *data() const { return mem.get(); }
    T &operator[](size_t k) const { return mem[k]; }
  };
  Raw<int> p, i;
  Raw<double> x;
};

// swap[r] != 0: the two real rows of complex row r change places (static pivoting, see numeric)
// unit != nullptr: the symmetric embedding C M(u_r a_rj u_j) of a complex symmetric matrix (see the head of this file)
bool embed(int n, const int *Ap, const int *Ai, const double *Ax, const double *Az, bool values, Embedded &E,
           const char *swap = nullptr, const double *unit = nullptr) {
  const long nnz = Ap[n];
  if (4 * nnz >= 0x7fffffffL) return false;
  E.p.resize((size_t)2 * n + 1);
  E.i.resize((size_t)4 * nnz);
  if (values) E.x.resize((size_t)4 * nnz);
  // column 2j starts at 4 Ap[j], column 2j + 1 one block column (2 entries per stored entry) further: no running
  // counter, so the columns are shared out among threads (20 million entries at 10^6 unknowns: 48 ms on one core,
  // a third of a FEAST-style refactorisation)
  auto fill = [&](int j0, int j1) {
    for (int j = j0; j < j1; ++j) {
      const long len = Ap[j + 1] - Ap[j];
      for (int half = 0; half < 2; ++half) {
        long q = 4 * (long)Ap[j] + 2 * len * half;
        E.p[(size_t)2 * j + half] = (int)q;
        for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
          const int r = Ai[p];
          E.i[(size_t)q] = 2 * r;
          E.i[(size_t)q + 1] = 2 * r + 1;
          if (values) {
            const double re = Az ? Ax[p] : Ax[2 * (size_t)p];
            const double im = Az ? Az[p] : Ax[2 * (size_t)p + 1];
            if (unit) {
              // w = (u_lo u_hi) a with the two units taken in index order: entries (r, j) and (j, r) of a symmetric
              // matrix go through the same operations on the same operands and get the same bits
              const int lo = r < j ? r : j, hi = r < j ? j : r;
              const double ar = unit[2 * (size_t)lo], ai = unit[2 * (size_t)lo + 1];
              const double br = unit[2 * (size_t)hi], bi = unit[2 * (size_t)hi + 1];
              const double pr = ar * br - ai * bi, pi = ar * bi + ai * br;
              const double wr = pr * re - pi * im, wi = pr * im + pi * re;
              E.x[(size_t)q] = half == 0 ? wr : -wi;
              E.x[(size_t)q + 1] = half == 0 ? -wi : -wr;
              q += 2;
              continue;
            }
            const double top = half == 0 ? re : -im, bottom = half == 0 ? im : re;
            const bool sw = swap && swap[r];
            E.x[(size_t)q] = sw ? bottom : top;
            E.x[(size_t)q + 1] = sw ? top : bottom;
          }
          q += 2;
        }
      }
    }
  };
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt ? (nt > 8 ? 8 : nt) : 1;
  if (nnz < 200000 || nt < 2) {
    fill(0, n);
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) {
      const int j0 = (int)((long)n * t / nt), j1 = (int)((long)n * (t + 1) / nt);
      try {
        pool.emplace_back(fill, j0, j1);
      } catch (...) {  // a thread could not be started: its columns are done here
        fill(j0, j1);
      }
    }
    fill(0, (int)((long)n / nt));
    for (std::thread &th : pool) th.join();
  }
  E.p[(size_t)2 * n] = (int)(4 * nnz);
  return true;
}

// v <- Q v: the entries 2r, 2r+1 of a packed complex vector change places where swap[r] is set
void swap_pairs(const std::vector<char> &swap, double *v) {
  for (size_t r = 0; r < swap.size(); ++r)
    if (swap[r]) std::swap(v[2 * r], v[2 * r + 1]);
}

// device form of swap_pairs: one thread per complex entry of n x k packed vectors
__global__ __launch_bounds__(256) void swap_pairs_kernel(const char *__restrict__ flags, double *__restrict__ v,
                                                         size_t n, size_t total) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total || !flags[t % n]) return;
  const double a = v[2 * t], b = v[2 * t + 1];
  v[2 * t] = b;
  v[2 * t + 1] = a;
}

// A == A^T exactly (pattern and bits of the values)?  Every off-diagonal entry looks its partner up by bisection.
bool complex_symmetric(int n, const int *Ap, const int *Ai, const double *Ax, const double *Az) {
  std::atomic<bool> ok{true};
  auto same = [&](int p, int q) {
    if (Az) return std::memcmp(&Ax[p], &Ax[q], sizeof(double)) == 0 && std::memcmp(&Az[p], &Az[q], sizeof(double)) == 0;
    return std::memcmp(&Ax[2 * (size_t)p], &Ax[2 * (size_t)q], 2 * sizeof(double)) == 0;
  };
  auto check = [&](int j0, int j1) {
    for (int j = j0; j < j1; ++j) {
      if (!ok.load(std::memory_order_relaxed)) return;
      for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
        const int r = Ai[p];
        if (r == j) continue;
        const int *first = Ai + Ap[r], *last = Ai + Ap[r + 1];
        const int *it = std::lower_bound(first, last, j);
        if (it == last || *it != j || !same(p, (int)(it - Ai))) {
          ok.store(false, std::memory_order_relaxed);
          return;
        }
      }
    }
  };
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt ? (nt > 8 ? 8 : nt) : 1;
  if (Ap[n] < 200000 || nt < 2) {
    check(0, n);
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) {
      const int j0 = (int)((long)n * t / nt), j1 = (int)((long)n * (t + 1) / nt);
      try {
        pool.emplace_back(check, j0, j1);
      } catch (...) {
        check(j0, j1);
      }
    }
    check(0, (int)((long)n / nt));
    for (std::thread &th : pool) th.join();
  }
  return ok.load();
}

// The 2 x 2 transforms of the symmetric embedding on a packed complex vector v of n entries (u: n unit-modulus pairs):
//   0: v <- C (u v)   right-hand side of A x = b         1: v <- u v              its solution
//   2: v <- conj(u) v right-hand side of A^H y = c       3: v <- conj(u) conj(v)  its solution
__host__ __device__ inline void unit_pair(int mode, double ur, double ui, double &vr, double &vi) {
  if (mode >= 2) ui = -ui;
  const double xr = vr, xi = mode == 3 ? -vi : vi;
  const double tr = ur * xr - ui * xi, ti = ur * xi + ui * xr;
  vr = tr;
  vi = mode == 0 ? -ti : ti;
}
void unit_pairs(const std::vector<double> &u, int mode, double *v) {
  for (size_t r = 0; r < u.size() / 2; ++r) unit_pair(mode, u[2 * r], u[2 * r + 1], v[2 * r], v[2 * r + 1]);
}
__global__ __launch_bounds__(256) void unit_pairs_kernel(const double *__restrict__ u, int mode, double *__restrict__ v,
                                                         size_t n, size_t total) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const size_t r = t % n;
  double vr = v[2 * t], vi = v[2 * t + 1];
  unit_pair(mode, u[2 * r], u[2 * r + 1], vr, vi);
  v[2 * t] = vr;
  v[2 * t + 1] = vi;
}

struct ZiSymbolic {  // remembers n so that numeric can rebuild the embedding, and the complex pattern it analysed
  unsigned magic = 0x5A53594Du;
  int n = 0;
  void *di = nullptr;
  std::vector<int> Ap;
  uint64_t ai_hash = 0;
};

}  // namespace

extern "C" {

int umfpack_zi_symbolic(int n_row, int n_col, const int Ap[], const int Ai[], const double Ax[],
                        const double Az[], void **Symbolic, const double Control[], double Info[]) {
  (void)Ax; (void)Az;
  if (!Symbolic) return UMFPACK_ERROR_argument_missing;
  *Symbolic = nullptr;
  if (!Ap) return UMFPACK_ERROR_argument_missing;
  if (n_row <= 0 || n_col <= 0) return UMFPACK_ERROR_n_nonpositive;
  if (Ap[0] != 0 || Ap[n_col] < 0 || (Ap[n_col] > 0 && !Ai)) return UMFPACK_ERROR_invalid_matrix;
  for (int j = 0; j < n_col; ++j) {
    if (Ap[j] > Ap[j + 1]) return UMFPACK_ERROR_invalid_matrix;
    for (int p = Ap[j]; p < Ap[j + 1]; ++p)
      if (Ai[p] < 0 || Ai[p] >= n_row || (p > Ap[j] && Ai[p] <= Ai[p - 1])) return UMFPACK_ERROR_invalid_matrix;
  }
  if (n_row != n_col) {  // rectangular: shape and pattern only (umfpack.hip, Symbolic::rectangular)
    try {
      ZiSymbolic *S = new ZiSymbolic();
      S->n = n_col;
      const int st = spl::symbolic_rectangular(n_row, n_col, Ap, Ai, &S->di);
      if (st < 0) { delete S; return st; }
      *Symbolic = S;
      return st;
    } catch (...) {
      return UMFPACK_ERROR_out_of_memory;
    }
  }
  try {
    Embedded E;
    if (!embed(n_col, Ap, Ai, nullptr, nullptr, false, E)) return UMFPACK_ERROR_out_of_memory;
    ZiSymbolic *S = new ZiSymbolic();
    S->n = n_col;
    S->Ap.assign(Ap, Ap + n_col + 1);
    S->ai_hash = spl::pattern_hash(Ai, Ap[n_col]);
    (void)Control; (void)Info;
    // ordered on the complex pattern (half the vertices, a quarter of the edges of the embedding), then expanded
    const int st = spl::symbolic_of_embedding(n_col, Ap, Ai, E.p.data(), E.i.data(), &S->di);
    if (st < 0) { delete S; return st; }
    *Symbolic = S;
    return st;
  } catch (const std::bad_alloc &) {
    return UMFPACK_ERROR_out_of_memory;
  } catch (...) {  // nothing may cross the C ABI
    return UMFPACK_ERROR_internal_error;
  }
}

int umfpack_zi_numeric(const int Ap[], const int Ai[], const double Ax[], const double Az[],
                       void *Symbolic, void **Numeric, const double Control[], double Info[]) {
  if (!Numeric) return UMFPACK_ERROR_argument_missing;
  *Numeric = nullptr;
  ZiSymbolic *S = static_cast<ZiSymbolic *>(Symbolic);
  if (!S || S->magic != 0x5A53594Du) return UMFPACK_ERROR_invalid_Symbolic_object;
  if (!Ap || !Ai || !Ax) return UMFPACK_ERROR_argument_missing;
  if (spl::symbolic_is_rectangular(S->di)) {
    try {
      const int nnz = Ap[S->n];
      if (nnz < 0) return UMFPACK_ERROR_different_pattern;
      std::vector<char> nonzero((size_t)nnz);
      for (int p = 0; p < nnz; ++p) {
        const double re = Az ? Ax[p] : Ax[2 * (size_t)p], im = Az ? Az[p] : Ax[2 * (size_t)p + 1];
        nonzero[(size_t)p] = ((re != 0.0 || im != 0.0) && re == re && im == im) ? 1 : 0;
      }
      return spl::numeric_rectangular_of(S->di, Ap, Ai, nonzero, Numeric, Ax, Az ? Az : Ax + 1, Az ? 1 : 2);
    } catch (...) {
      return UMFPACK_ERROR_out_of_memory;
    }
  }
  try {
    // Static pivoting inside the 2 x 2 blocks of the diagonal.  The scalar factorisation of the
    // embedding pivots on the REAL part of a complex diagonal entry first; where the imaginary part
    // is the larger one (a shift z I - A close to the real axis of A's diagonal: pivot ~ 0), the two
    // real equations of that complex row change places, so that the first pivot is the imaginary
    // part.  A row permutation Q of the system: (Q E) x = Q b; E^T y = c is (Q E)^T (Q y) = c.  The
    // pattern of the embedding does not change (every block is a full 2 x 2).
    const int n = S->n;
    // UMFPACK_ERROR_different_pattern, decided on the complex pattern (the embedding's is four times as long)
    if (!std::equal(S->Ap.begin(), S->Ap.end(), Ap) || spl::pattern_hash(Ai, Ap[n]) != S->ai_hash)
      return UMFPACK_ERROR_different_pattern;
    const bool timing = getenv("SPL_MF_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    // A complex symmetric matrix (FEAST's z B - A for real symmetric A, B) is embedded symmetrically and factored as
    // L D L^T (head of this file) where the tree has work to halve: below ~1e12 flops (2-D meshes of 10^6 unknowns, 3-D ones below ~50^3) the
    // factorisation is launch-bound and the symmetry check on the host costs more than the flops saved.
    // SPL_ZI_SYMMETRIC=0: the general embedding for every matrix; =1: the symmetric one whenever A == A^T
    // Native complex fronts (csrc/multifrontal.hip, TreeView::zm) take the PLAIN embedding — no swapped pairs, no
    // congruence: a complex pivot is as good as its modulus — whenever the analysis chose the tree; a complex symmetric
    // matrix is then factored as L D L^T in complex arithmetic.  The embeddings below serve the band path, and
    // SPL_ZI_NATIVE=0.
    const bool native = spl::symbolic_has_complex_tree(S->di);
    const char *zs = getenv("SPL_ZI_SYMMETRIC");
    const bool wanted = native || (zs ? zs[0] != '0' : spl::symbolic_tree_flops(S->di) >= 1e12);
    const bool is_sym = wanted && n > 1 && complex_symmetric(n, Ap, Ai, Ax, Az);
    const bool symmetric = is_sym && !native;
    std::vector<char> swap;
    std::vector<double> unit;
    bool any = false;
    if (symmetric) {
      unit.assign((size_t)2 * n, 0.0);
      for (int j = 0; j < n; ++j) {
        unit[(size_t)2 * j] = 1.0;
        const int *first = Ai + Ap[j], *last = Ai + Ap[j + 1];
        const int *it = std::lower_bound(first, last, j);
        if (it == last || *it != j) continue;
        const size_t p = (size_t)(it - Ai);
        const double re = Az ? Ax[p] : Ax[2 * p], im = Az ? Az[p] : Ax[2 * p + 1];
        const double mod = std::hypot(re, im);
        if (!std::isfinite(mod) || mod == 0.0) continue;
        // u = sqrt(conj(a) / |a|), the root with the non-negative real part, without cancellation
        const double c = re / mod, sn = -im / mod;
        double ur, ui;
        if (c >= 0.0) {
          ur = std::sqrt(0.5 * (1.0 + c));
          ui = sn / (2.0 * ur);
        } else {
          ui = std::copysign(std::sqrt(0.5 * (1.0 - c)), sn);
          ur = sn / (2.0 * ui);
        }
        unit[(size_t)2 * j] = ur;
        unit[(size_t)2 * j + 1] = ui;
      }
    } else if (!native) {
      swap.assign((size_t)n, 0);
      for (int j = 0; j < n; ++j)
        for (int p = Ap[j]; p < Ap[j + 1]; ++p)
          if (Ai[p] == j) {
            const double re = Az ? Ax[p] : Ax[2 * (size_t)p], im = Az ? Az[p] : Ax[2 * (size_t)p + 1];
            if (std::fabs(im) > std::fabs(re)) { swap[(size_t)j] = 1; any = true; }
          }
    }
    Embedded E;
    if (!embed(n, Ap, Ai, Ax, Az, true, E, any ? swap.data() : nullptr, symmetric ? unit.data() : nullptr))
      return UMFPACK_ERROR_out_of_memory;
    if (timing)
      fprintf(stderr, "[zi numeric] %s embedding built on the host %8.2f ms\n",
              native ? (is_sym ? "plain (native complex fronts, symmetric)" : "plain (native complex fronts)")
                     : symmetric ? "symmetric" : "general",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    (void)Control; (void)Info;
    const int st = spl::numeric_of_embedding(E.p.data(), E.i.data(), E.x.data(), S->di, Numeric,
                                             native ? (is_sym ? 2 : 1) : 0);
    if (st >= 0 && any) spl::numeric_set_pair_swap(*Numeric, std::move(swap));
    if (st >= 0 && symmetric) spl::numeric_set_pair_unit(*Numeric, std::move(unit));
    return st;
  } catch (const std::bad_alloc &) {
    return UMFPACK_ERROR_out_of_memory;
  } catch (...) {  // nothing may cross the C ABI
    return UMFPACK_ERROR_internal_error;
  }
}

static int zi_solve(int sys, const int Ap[], const int Ai[], const double Ax[], const double Az[],
                    double Xx[], double Xz[], const double Bx[], const double Bz[], void *Numeric,
                    const double Control[], double Info[]);

int umfpack_zi_solve(int sys, const int Ap[], const int Ai[], const double Ax[], const double Az[],
                     double Xx[], double Xz[], const double Bx[], const double Bz[], void *Numeric,
                     const double Control[], double Info[]) {
  const int st = zi_solve(sys, Ap, Ai, Ax, Az, Xx, Xz, Bx, Bz, Numeric, Control, Info);
  if (Info && st < 0) Info[0] = st;  // Info[UMFPACK_STATUS] on the error returns as well
  return st;
}

static int zi_solve(int sys, const int Ap[], const int Ai[], const double Ax[], const double Az[],
                    double Xx[], double Xz[], const double Bx[], const double Bz[], void *Numeric,
                    const double Control[], double Info[]) {
  (void)Az;
  if (spl::numeric_is_rectangular(Numeric)) return UMFPACK_ERROR_invalid_system;
  if (!Xx || !Bx) return UMFPACK_ERROR_argument_missing;
  if (!Ap || !Ai || !Ax) return UMFPACK_ERROR_argument_missing;
  // the Numeric object holds device copies of E and E^T (residuals use those, like the `di` path)
  const std::vector<char> *swap = spl::numeric_pair_swap(Numeric);
  const std::vector<double> *unit = spl::numeric_pair_unit(Numeric);
  if (!Xz && !Bz && !swap && !unit) return umfpack_di_solve(sys, Ap, Ai, Ax, Xx, Bx, Numeric, Control, Info);
  // split real / imaginary arrays and / or swapped row pairs: interleave, solve, de-interleave
  try {
    // UMFPACK's solve takes no dimension argument: it lives in the Numeric object
    const int n2 = spl_umfpack_dimension(Numeric);
    if (n2 <= 0) return UMFPACK_ERROR_invalid_Numeric_object;
    const int n = n2 / 2;
    const bool packed = !Xz && !Bz;
    std::vector<double> b((size_t)n2), x((size_t)n2);
    for (int k = 0; k < n; ++k) {
      b[(size_t)2 * k] = packed ? Bx[(size_t)2 * k] : Bx[k];
      b[(size_t)2 * k + 1] = packed ? Bx[(size_t)2 * k + 1] : (Bz ? Bz[k] : 0.0);
    }
    if (swap && sys == UMFPACK_A) swap_pairs(*swap, b.data());          // (Q E) x = Q b
    if (unit) unit_pairs(*unit, sys == UMFPACK_A ? 0 : 2, b.data());    // E' x' = T b  /  E' y' = W^T c
    const int st = umfpack_di_solve(sys, Ap, Ai, Ax, x.data(), b.data(), Numeric, Control, Info);
    if (swap && sys != UMFPACK_A) swap_pairs(*swap, x.data());          // y = Q w
    if (unit) unit_pairs(*unit, sys == UMFPACK_A ? 1 : 3, x.data());    // x = W x'  /  y = T^T y'
    for (int k = 0; k < n; ++k) {
      if (packed) { Xx[(size_t)2 * k] = x[(size_t)2 * k]; Xx[(size_t)2 * k + 1] = x[(size_t)2 * k + 1]; }
      else { Xx[k] = x[(size_t)2 * k]; if (Xz) Xz[k] = x[(size_t)2 * k + 1]; }
    }
    return st;
  } catch (const std::bad_alloc &) {
    return UMFPACK_ERROR_out_of_memory;
  } catch (...) {  // nothing may cross the C ABI
    return UMFPACK_ERROR_internal_error;
  }
}

// batched complex linearSolve: nrhs right-hand sides, each packed (re, im) pairs when the
// imaginary pointers are NULL, else split arrays of n x nrhs (column-major)
int spl_umfpack_zi_solve_many(int sys, const int Ap[], const int Ai[], const double Ax[], const double Az[],
                              int nrhs, double Xx[], double Xz[], const double Bx[], const double Bz[],
                              void *Numeric) {
  (void)Az;
  if (spl::numeric_is_rectangular(Numeric)) return UMFPACK_ERROR_invalid_system;
  if (nrhs < 0) return UMFPACK_ERROR_argument_missing;
  if (!Ap || !Ai || !Ax) return UMFPACK_ERROR_argument_missing;
  const int n2 = spl_umfpack_dimension(Numeric);
  if (n2 < 0 || (n2 == 0 && !Numeric)) return UMFPACK_ERROR_invalid_Numeric_object;
  if (nrhs > 0 && n2 > 0 && (!Xx || !Bx)) return UMFPACK_ERROR_argument_missing;
  const std::vector<char> *swap = spl::numeric_pair_swap(Numeric);
  const std::vector<double> *unit = spl::numeric_pair_unit(Numeric);
  if (!Xz && !Bz && !swap && !unit) return spl_umfpack_di_solve_many(sys, Ap, Ai, Ax, nrhs, Xx, Bx, Numeric);
  try {
    const size_t n = (size_t)n2 / 2, tot = (size_t)n2 * (size_t)nrhs;
    const bool packed = !Xz && !Bz;
    std::vector<double> b(tot), x(tot);
    for (size_t c = 0; c < (size_t)nrhs; ++c) {
      for (size_t k = 0; k < n; ++k) {
        b[c * n2 + 2 * k] = packed ? Bx[c * n2 + 2 * k] : Bx[c * n + k];
        b[c * n2 + 2 * k + 1] = packed ? Bx[c * n2 + 2 * k + 1] : (Bz ? Bz[c * n + k] : 0.0);
      }
      if (swap && sys == UMFPACK_A) swap_pairs(*swap, b.data() + c * n2);
      if (unit) unit_pairs(*unit, sys == UMFPACK_A ? 0 : 2, b.data() + c * n2);
    }
    const int st = spl_umfpack_di_solve_many(sys, Ap, Ai, Ax, nrhs, x.data(), b.data(), Numeric);
    for (size_t c = 0; c < (size_t)nrhs; ++c) {
      if (swap && sys != UMFPACK_A) swap_pairs(*swap, x.data() + c * n2);
      if (unit) unit_pairs(*unit, sys == UMFPACK_A ? 1 : 3, x.data() + c * n2);
      for (size_t k = 0; k < n; ++k) {
        if (packed) { Xx[c * n2 + 2 * k] = x[c * n2 + 2 * k]; Xx[c * n2 + 2 * k + 1] = x[c * n2 + 2 * k + 1]; }
        else { Xx[c * n + k] = x[c * n2 + 2 * k]; if (Xz) Xz[c * n + k] = x[c * n2 + 2 * k + 1]; }
      }
    }
    return st;
  } catch (const std::bad_alloc &) {
    return UMFPACK_ERROR_out_of_memory;
  } catch (...) {  // nothing may cross the C ABI
    return UMFPACK_ERROR_internal_error;
  }
}

// packed complex right-hand sides and solutions in device memory (see umfpack_hip.h)
int spl_umfpack_zi_solve_many_dev(int sys, const int Ap[], const int Ai[], const double Ax[], int nrhs, double *d_X,
                                  const double *d_B, void *Numeric) {
  if (spl::numeric_is_rectangular(Numeric)) return UMFPACK_ERROR_invalid_system;
  if (nrhs < 0) return UMFPACK_ERROR_argument_missing;
  const int n2 = spl_umfpack_dimension(Numeric);
  if (n2 < 0 || (n2 == 0 && !Numeric)) return UMFPACK_ERROR_invalid_Numeric_object;
  if (nrhs > 0 && n2 > 0 && (!d_X || !d_B)) return UMFPACK_ERROR_argument_missing;
  const std::vector<char> *swap = spl::numeric_pair_swap(Numeric);
  const std::vector<double> *unit = spl::numeric_pair_unit(Numeric);
  if ((!swap && !unit) || nrhs == 0 || n2 == 0) return spl_umfpack_di_solve_many_dev(sys, Ap, Ai, Ax, nrhs, d_X, d_B, Numeric);
  try {
    const size_t n = (size_t)n2 / 2, total = n * (size_t)nrhs;
    if (unit) {  // symmetric embedding: E' x' = T b, x = W x'  /  E' y' = W^T c, y = T^T y'
      spl::DBuf<double> u(2 * n), b(2 * total);
      SPL_HIP(hipMemcpy(u.get(), unit->data(), 2 * n * sizeof(double), hipMemcpyHostToDevice));
      SPL_HIP(hipMemcpy(b.get(), d_B, 2 * total * sizeof(double), hipMemcpyDeviceToDevice));
      const dim3 grid((unsigned)((total + 255) / 256));
      hipLaunchKernelGGL(unit_pairs_kernel, grid, dim3(256), 0, nullptr, u.get(), sys == UMFPACK_A ? 0 : 2, b.get(), n, total);
      SPL_HIP(hipDeviceSynchronize());
      const int st = spl_umfpack_di_solve_many_dev(sys, Ap, Ai, Ax, nrhs, d_X, b.get(), Numeric);
      if (st < 0) return st;
      hipLaunchKernelGGL(unit_pairs_kernel, grid, dim3(256), 0, nullptr, u.get(), sys == UMFPACK_A ? 1 : 3, d_X, n, total);
      SPL_HIP(hipDeviceSynchronize());
      return st;
    }
    spl::DBuf<char> flags(n);
    SPL_HIP(hipMemcpy(flags.get(), swap->data(), n, hipMemcpyHostToDevice));
    const dim3 grid((unsigned)((total + 255) / 256));
    if (sys == UMFPACK_A) {  // (Q E) x = Q b
      spl::DBuf<double> b(2 * total);
      SPL_HIP(hipMemcpy(b.get(), d_B, 2 * total * sizeof(double), hipMemcpyDeviceToDevice));
      hipLaunchKernelGGL(swap_pairs_kernel, grid, dim3(256), 0, nullptr, flags.get(), b.get(), n, total);
      SPL_HIP(hipDeviceSynchronize());
      return spl_umfpack_di_solve_many_dev(sys, Ap, Ai, Ax, nrhs, d_X, b.get(), Numeric);
    }
    const int st = spl_umfpack_di_solve_many_dev(sys, Ap, Ai, Ax, nrhs, d_X, d_B, Numeric);
    if (st < 0) return st;
    hipLaunchKernelGGL(swap_pairs_kernel, grid, dim3(256), 0, nullptr, flags.get(), d_X, n, total);  // y = Q w
    SPL_HIP(hipDeviceSynchronize());
    return st;
  } catch (const spl::DeviceError &e) {
    return e.status == SPL_ERROR_out_of_memory ? UMFPACK_ERROR_out_of_memory : UMFPACK_ERROR_internal_error;
  } catch (const std::bad_alloc &) {
    return UMFPACK_ERROR_out_of_memory;
  } catch (...) {  // nothing may cross the C ABI
    return UMFPACK_ERROR_internal_error;
  }
}

void umfpack_zi_free_symbolic(void **Symbolic) {
  if (!Symbolic || !*Symbolic) return;
  ZiSymbolic *S = static_cast<ZiSymbolic *>(*Symbolic);
  *Symbolic = nullptr;
  if (S->magic != 0x5A53594Du) return;
  S->magic = 0;
  umfpack_di_free_symbolic(&S->di);
  delete S;
}

void umfpack_zi_free_numeric(void **Numeric) { umfpack_di_free_numeric(Numeric); }

void umfpack_zi_report_status(const double Control[], int status) { umfpack_di_report_status(Control, status); }

}  // extern "C"

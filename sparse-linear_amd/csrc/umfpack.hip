// umfpack.hip — the umfpack_di_* link-time ABI (include/umfpack_hip.h) on the MI355X.
//
// Reference call sites: analyze / factor / linearSolve_ (suitesparse/src/Numeric/
// LinearAlgebra/Umfpack.hs:60-102) through the imports at Umfpack/Internal.hs:137-148.
// All arithmetic of that step lives in third-party UMFPACK in the reference (un-vendored);
// parity is defined on the SOLUTION (ident <\> v == v exactly, residual checks otherwise).
//
//   symbolic (host): reverse Cuthill-McKee ordering of the pattern of A + A^T (bandwidths kl, ku)
//                    and, unless the band is narrow, a nested-dissection ordering with its frontal
//                    tree (mf_symbolic.hpp); the cheaper one by flop count is used (UMFPACK also
//                    orders on the CPU);
//   numeric  (GPU):  LU of B = P A P^T without row interchanges — blocked in band storage
//                    (band_nopiv.hip) or multifrontal on the tree (multifrontal.hip), both on the
//                    fp64 matrix cores — when A is diagonally dominant by columns, and as a checked
//                    speculation otherwise; fallback: LAPACK band storage AB[2kl+ku+1][n] with
//                    partial pivoting inside the band (this file), numerically the same as dense
//                    partial pivoting; a zero pivot sets the singular-matrix warning;
//   solve    (GPU):  permute, forward/back substitution (or U^T, L^T for sys = 1) through the band
//                    or the tree, un-permute, then up to 2 steps of iterative refinement with the
//                    residual computed by the SpMV kernels (UMFPACK's default irstep = 2) and
//                    UMFPACK's stopping rules on the componentwise backward error; all right-hand
//                    sides of a batched call share the passes.
// The pivoting band LU here updates one column at a time (rank-1 updates, HBM-bound): it is the
// safety net, not the fast path.
#include <stdio.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <future>
#include <mutex>
#include <vector>

#include <complex>
#include "common.hpp"
#include "mf_symbolic.hpp"
#include "static_pivot.hpp"
#include "../../include/umfpack_hip.h"

namespace spl {

namespace {

constexpr uint32_t kSymMagic = 0x53594D42u;  // "SYMB"
constexpr uint32_t kNumMagic = 0x4E554D52u;  // "NUMR"

struct Symbolic {
  uint32_t magic = kSymMagic;
  int n = 0;
  int nnz = 0;
  int kl = 0, ku = 0;
  std::vector<int> perm;  // new -> old (reverse Cuthill-McKee: the band paths)
  std::vector<int> inv;   // old -> new
  std::vector<int> Ap;    // pattern check in numeric (UMFPACK_ERROR_different_pattern)
  uint64_t ai_hash = 0;   // ... together with a hash of the row indices
  // nested-dissection tree of the multifrontal path, present when that path is the cheaper one
  std::shared_ptr<const mf::Tree> tree;
  // embeddings of complex matrices (umfpack_zi.hip): the tree of the COMPLEX pattern itself, from the same dissection
  // (tree is its expansion): what the native complex fronts are built on
  std::shared_ptr<const mf::Tree> ztree;
  bool have_band = false;  // perm / inv / kl / ku are set (large matrices whose tree wins by a lower bound skip them)
  // Rectangular matrices (round 4).  UMFPACK analyses and factors them and refuses to SOLVE with them
  // (UMFPACK_ERROR_invalid_system: "the matrix is not square"); through the reference's binding nothing of such a
  // factorisation is observable but the statuses (Umfpack.hs:60-102 binds symbolic, numeric, solve and the frees).
  // Here: the analysis records the shape and the pattern, the numeric call checks the pattern and reports whether a
  // full set of min(n_row, n_col) non-zero pivots exists at all — the structural rank over the non-zero entries, where
  // UMFPACK counts the non-zero pivots it found (UMFPACK_WARNING_singular_matrix otherwise) — and holds no factors;
  // the solve returns UMFPACK_ERROR_invalid_system as UMFPACK's does.
  int n_row = 0, n_col = 0;
  bool rectangular = false;
};

struct Numeric {
  uint32_t magic = kNumMagic;
  int device = 0;
  int n = 0, kl = 0, ku = 0, ldab = 1;
  int singular = 0;
  int rectangular = 0;  // 1: of a rectangular matrix (Symbolic::rectangular): no factors, solves return invalid_system
  int nopiv = 0;  // 1: blocked factorisation without interchanges
  int mf_sym = 0;  // 1: the multifrontal factors held are those of a symmetric matrix (L D L^T: half the update flops)
  // Native complex fronts (round 3).  This object holds the real embedding E of a complex matrix (umfpack_zi.hip) for
  // residuals, refinement and every fallback; with zfront = 1 the multifrontal factors are those of the COMPLEX matrix
  // on the tree of its own pattern (ztree: half the unknowns, complex fronts in two planes, multifrontal.hip) — a
  // solve with them is a solve with E (packed complex vectors ARE the real vectors of the embedding), at half the
  // flops and bytes.  zsym: the complex matrix is symmetric (A == A^T): L D L^T.
  std::shared_ptr<const mf::Tree> ztree;
  int zfront = 0, zsym = 0;
  // Threshold pivoting inside the diagonal blocks of the fronts (Band::piv): on for every matrix that is not
  // diagonally dominant by columns — its factors without interchanges are a speculation, and the rows of a pivot block
  // are free to change places.  A symmetric matrix is first tried as L D L^T (no interchanges: half the flops); if the
  // check of a solve rejects those factors, the same tree is factored once more as LU with block pivoting
  // (block_pivot_retry) before static pivoting takes over.  SPL_LU_BLOCK_PIVOT=0: never.
  int dominant = 0, mf_piv = 0, block_pivot_retry = 0;
  DBuf<double> rscale;  // row scales of the block pivoting (new ordering of the tree in use)
  // set when a refactorisation failed after the previous factors were released: the object holds no
  // usable factors any more and every later solve returns an error instead of launching kernels
  std::atomic<int> broken{0};
  // 1: the matrix is NOT diagonally dominant by columns and the no-interchange factors are a
  // speculation; solve checks the backward error it computes anyway and, if it is not at
  // rounding level, refactors with partial pivoting (under `mu`) and solves again
  std::atomic<int> speculative{0};
  std::mutex mu;
  DBuf<double> AB;
  DBuf<double> blkinv;  // no-pivot path: inv(L11), inv(U11) of every diagonal block
  DBuf<int> ipiv, perm, inv;
  // multifrontal factors (then AB is empty and perm/inv hold the nested-dissection ordering); the
  // band ordering is kept for the pivoting fallback
  mf::Factors *mfact = nullptr;
  std::shared_ptr<const mf::Tree> tree;
  std::vector<int> band_perm, band_inv;
  Matrix *A = nullptr;   // rows of A   (residual b - A x)
  Matrix *At = nullptr;  // rows of A^T (residual b - A^T x)
  // set by the `zi` wrapper (umfpack_zi.hip): rows 2r, 2r+1 of the real embedding were swapped
  std::vector<char> pair_swap;
  std::vector<double> pair_unit;  // zi wrapper, complex symmetric matrices: unit-modulus u_r (re, im) of the congruence D A D
  // Static pivoting (static_pivot.hpp): 0 not tried, 1 the factors held are those of B = Dr P A Dc on B's own
  // tree (still a checked speculation), 2 tried and given up.  spA / spAt: rows of B / of B^T on the device
  // (what mf_factor scatters); sp_idx / sp_scale: the permutations and scalings around a solve with B's factors,
  // composed with B's nested-dissection ordering — [0] before, [1] after A x = b; [2] before, [3] after A^T x = b.
  int sp_stage = 0;
  // the most recent solve call that finished on this object (spl_umfpack_solve_report; UMFPACK reports the like in
  // Info[UMFPACK_IR_TAKEN .. UMFPACK_OMEGA1]): walks over the factors (first solve + refinement steps, whatever path),
  // refinement steps kept / attempted, largest componentwise backward error among the delivered columns
  std::atomic<int> last_walks{0}, last_ir_taken{0}, last_ir_attempted{0};
  std::atomic<double> last_omega{0.0};
  Matrix *spA = nullptr, *spAt = nullptr;
  DBuf<int> sp_idx[4];
  DBuf<double> sp_scale[4];
  ~Numeric() {
    delete A;
    delete At;
    delete spA;
    delete spAt;
    if (mfact) mf_free(mfact);
  }
};

thread_local bool t_pattern_vouched = false;  // set around umfpack_di_numeric by spl::numeric_of_embedding
thread_local int t_native_complex = 0;  // ... 1: the embedding is plain (no swapped pairs): native complex fronts may serve it; 2: and A == A^T

// 64-bit hash of the row indices: the second half of the pattern check in numeric (the pointers are
// compared exactly; a pattern with the same column counts but other rows would be scattered with a
// stale ordering, out of the band / the fronts)
uint64_t hash_indices(const int *Ai, int64_t nnz) {
  uint64_t h[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
  int64_t p = 0;
  for (; p + 4 <= nnz; p += 4)
    for (int u = 0; u < 4; ++u) {
      h[u] ^= (uint64_t)(uint32_t)Ai[p + u];
      h[u] *= 0x100000001B3ull;
      h[u] ^= h[u] >> 29;
    }
  for (; p < nnz; ++p) { h[0] ^= (uint64_t)(uint32_t)Ai[p]; h[0] *= 0x100000001B3ull; h[0] ^= h[0] >> 29; }
  return (h[0] * 31 + h[1]) * 31 + (h[2] * 31 + h[3]) + (uint64_t)nnz;
}

// ---- reverse Cuthill-McKee on the pattern of A + A^T (host) --------------------------------
void rcm_order(int n, const int *Ap, const int *Ai, std::vector<int> &perm) {
  std::vector<int64_t> ptr((size_t)n + 1, 0);
  for (int j = 0; j < n; ++j)
    for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
      const int i = Ai[p];
      if (i != j) { ++ptr[(size_t)i + 1]; ++ptr[(size_t)j + 1]; }
    }
  for (int i = 0; i < n; ++i) ptr[(size_t)i + 1] += ptr[(size_t)i];
  std::vector<int> adj((size_t)ptr[(size_t)n]);
  std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
  for (int j = 0; j < n; ++j)
    for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
      const int i = Ai[p];
      if (i != j) { adj[(size_t)cur[(size_t)i]++] = j; adj[(size_t)cur[(size_t)j]++] = i; }
    }
  auto degree = [&](int v) { return (int)(ptr[(size_t)v + 1] - ptr[(size_t)v]); };
  std::vector<int> order;
  order.reserve((size_t)n);
  std::vector<char> visited((size_t)n, 0);
  std::vector<int> level((size_t)n, -1), queue;
  // BFS from `root` restricted to unvisited vertices; returns the last vertex of minimum
  // degree in the deepest level and the depth
  auto bfs_levels = [&](int root, int &far, int &depth) {
    queue.clear();
    queue.push_back(root);
    level[(size_t)root] = 0;
    size_t head = 0;
    while (head < queue.size()) {
      const int v = queue[head++];
      for (int64_t p = ptr[(size_t)v]; p < ptr[(size_t)v + 1]; ++p) {
        const int u = adj[(size_t)p];
        if (!visited[(size_t)u] && level[(size_t)u] < 0) {
          level[(size_t)u] = level[(size_t)v] + 1;
          queue.push_back(u);
        }
      }
    }
    depth = level[(size_t)queue.back()];
    far = queue.back();
    for (size_t t = queue.size(); t-- > 0;) {
      const int v = queue[t];
      if (level[(size_t)v] != depth) break;
      if (degree(v) < degree(far)) far = v;
    }
    for (int v : queue) level[(size_t)v] = -1;
  };
  std::vector<int> nbr;
  for (int start = 0; start < n; ++start) {
    if (visited[(size_t)start]) continue;
    // pseudo-peripheral root (George & Liu): walk to the far end until the depth stops growing
    int root = start, far = start, depth = 0, best = -1;
    for (int it = 0; it < 8; ++it) {
      bfs_levels(root, far, depth);
      if (depth <= best) break;
      best = depth;
      root = far;
    }
    // Cuthill-McKee: BFS, neighbours by increasing degree
    const size_t first = order.size();
    order.push_back(root);
    visited[(size_t)root] = 1;
    size_t head = first;
    while (head < order.size()) {
      const int v = order[head++];
      nbr.clear();
      for (int64_t p = ptr[(size_t)v]; p < ptr[(size_t)v + 1]; ++p) {
        const int u = adj[(size_t)p];
        if (!visited[(size_t)u]) { visited[(size_t)u] = 1; nbr.push_back(u); }
      }
      std::sort(nbr.begin(), nbr.end(), [&](int a, int b) {
        const int da = degree(a), db = degree(b);
        return da != db ? da < db : a < b;
      });
      order.insert(order.end(), nbr.begin(), nbr.end());
    }
  }
  std::reverse(order.begin(), order.end());
  perm = order;
}

// ---- device kernels --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_scatter_kernel(int n, const int *__restrict__ Ap,
                                                           const int *__restrict__ Ai,
                                                           const double *__restrict__ Ax,
                                                           const int *__restrict__ inv, int kv, int ldab,
                                                           double *__restrict__ AB) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= n) return;
  const int nj = inv[j];
  for (int p = Ap[j] + lane; p < Ap[j + 1]; p += 64) {
    const int ni = inv[Ai[p]];
    atomicAdd(&AB[(size_t)(kv + ni - nj) + (size_t)nj * ldab], Ax[p]);
  }
}

struct BandState {
  int ju;
  int singular;
};

// pivot search + row interchange + scaling of column j (one workgroup)
__device__ inline void band_pivot_column(int j, int n, int kl, int ku, int ldab, double *AB, int *ipiv,
                                         BandState *st, double *red_val, int *red_idx) {
  const int kv = kl + ku;
  const int tid = threadIdx.x, nt = blockDim.x;
  double *col = AB + (size_t)j * ldab;
  const int km = min(kl, n - 1 - j);
  double best = -1.0;
  int bi = 0;
  for (int i = tid; i <= km; i += nt) {
    const double a = fabs(col[kv + i]);
    if (a > best) { best = a; bi = i; }  // strided: smaller i seen first per thread
  }
  red_val[tid] = best;
  red_idx[tid] = bi;
  __syncthreads();
  for (int sft = nt >> 1; sft > 0; sft >>= 1) {
    if (tid < sft) {
      const double ov = red_val[tid + sft];
      const int oi = red_idx[tid + sft];
      if (ov > red_val[tid] || (ov == red_val[tid] && oi < red_idx[tid])) { red_val[tid] = ov; red_idx[tid] = oi; }
    }
    __syncthreads();
  }
  const int jp = red_idx[0];
  const double pmax = red_val[0];
  int ju = st->ju;
  const int ju_new = max(ju, min(j + ku + jp, n - 1));
  __syncthreads();
  if (tid == 0) {
    ipiv[j] = j + jp;
    st->ju = ju_new;
    if (!(pmax > 0.0)) st->singular = 1;
  }
  ju = ju_new;
  if (jp != 0) {
    for (int c = j + tid; c <= ju; c += nt) {
      double *a = AB + (size_t)(kv + j - c) + (size_t)c * ldab;
      double *b = a + jp;
      const double t = *a;
      *a = *b;
      *b = t;
    }
  }
  __syncthreads();
  if (pmax > 0.0) {
    const double piv = col[kv];
    for (int i = 1 + tid; i <= km; i += nt) col[kv + i] = col[kv + i] / piv;
  }
}

__global__ __launch_bounds__(256) void band_pivot_kernel(int j, int n, int kl, int ku, int ldab,
                                                         double *AB, int *ipiv, BandState *st) {
  __shared__ double red_val[256];
  __shared__ int red_idx[256];
  band_pivot_column(j, n, kl, ku, ldab, AB, ipiv, st, red_val, red_idx);
}

// rank-1 update of the trailing band: A(j+i, c) -= L(j+i, j) * U(j, c),  i = 1..km, c = j+1..ju
__global__ __launch_bounds__(256) void band_update_kernel(int j, int n, int kl, int ku, int ldab,
                                                          double *__restrict__ AB,
                                                          const BandState *__restrict__ st) {
  const int kv = kl + ku;
  const int km = min(kl, n - 1 - j);
  const int c = j + 1 + blockIdx.y;
  if (c > st->ju) return;
  const int i = 1 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i > km) return;
  const double l = AB[(size_t)(kv + i) + (size_t)j * ldab];
  const double u = AB[(size_t)(kv + j - c) + (size_t)c * ldab];
  AB[(size_t)(kv + i + j - c) + (size_t)c * ldab] -= l * u;
}

// whole factorisation in one workgroup (narrow bands: launch overhead would dominate)
__global__ __launch_bounds__(1024) void band_lu_fused_kernel(int n, int kl, int ku, int ldab, double *AB,
                                                             int *ipiv, BandState *st) {
  __shared__ double red_val[1024];
  __shared__ int red_idx[1024];
  const int kv = kl + ku;
  for (int j = 0; j < n; ++j) {
    band_pivot_column(j, n, kl, ku, ldab, AB, ipiv, st, red_val, red_idx);
    __syncthreads();
    const int km = min(kl, n - 1 - j);
    const int ju = st->ju;
    const int width = ju - j;
    const int total = km * width;
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
      const int i = 1 + t % km, c = j + 1 + t / km;
      const double l = AB[(size_t)(kv + i) + (size_t)j * ldab];
      const double u = AB[(size_t)(kv + j - c) + (size_t)c * ldab];
      AB[(size_t)(kv + i + j - c) + (size_t)c * ldab] -= l * u;
    }
    __syncthreads();
  }
}

// column blockIdx.y: out[k] = in[perm[k]]
// conj (vectors of interleaved (re, im) pairs): 1 = the entries with an odd SOURCE index change sign, 2 = those with an
// odd destination index — the complex conjugate of the vector in its original ordering, on the way in or out
__global__ void gather_perm_kernel(int n, const int *__restrict__ perm, const double *__restrict__ in,
                                   double *__restrict__ out, size_t stride, int conj) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t col = (size_t)blockIdx.y * stride;
  if (k >= n) return;
  const int src = perm[k];
  const double v = in[col + src];
  out[col + k] = ((conj == 1 && (src & 1)) || (conj == 2 && (k & 1))) ? -v : v;
}

// column blockIdx.y: out[k] = scale[k] * in[idx[k]]
__global__ void gather_scale_kernel(int n, const int *__restrict__ idx, const double *__restrict__ scale,
                                    const double *__restrict__ in, double *__restrict__ out, size_t stride) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t col = (size_t)blockIdx.y * stride;
  if (k < n) out[col + k] = scale[k] * in[col + idx[k]];
}

// banded solves with the factors, one workgroup, vector c in HBM (L2-resident)
__global__ __launch_bounds__(1024) void band_solve_kernel(int sys, int n, int kl, int ku, int ldab,
                                                          const double *__restrict__ AB,
                                                          const int *__restrict__ ipiv, double *c) {
  __shared__ double red[1024];
  const int kv = kl + ku;
  const int tid = threadIdx.x, nt = blockDim.x;
  if (sys == 0) {
    // L: forward with the row interchanges
    for (int j = 0; j < n; ++j) {
      const int lm = min(kl, n - 1 - j);
      const int l = ipiv[j];
      if (tid == 0 && l != j) { const double t = c[l]; c[l] = c[j]; c[j] = t; }
      __syncthreads();
      const double cj = c[j];
      const double *col = AB + (size_t)j * ldab + kv;
      for (int i = 1 + tid; i <= lm; i += nt) c[j + i] -= cj * col[i];
      __syncthreads();
    }
    // U: backward
    for (int j = n - 1; j >= 0; --j) {
      const double *col = AB + (size_t)j * ldab + kv;
      if (tid == 0) c[j] = c[j] / col[0];
      __syncthreads();
      const double zj = c[j];
      const int lo = max(0, j - kv);
      for (int i = lo + tid; i < j; i += nt) c[i] -= zj * col[i - j];
      __syncthreads();
    }
  } else {
    // U^T y = c
    for (int j = 0; j < n; ++j) {
      const double *col = AB + (size_t)j * ldab + kv;
      const int lo = max(0, j - kv);
      double s = 0.0;
      for (int i = lo + tid; i < j; i += nt) s += col[i - j] * c[i];
      red[tid] = s;
      __syncthreads();
      for (int sft = nt >> 1; sft > 0; sft >>= 1) {
        if (tid < sft) red[tid] += red[tid + sft];
        __syncthreads();
      }
      if (tid == 0) c[j] = (c[j] - red[0]) / col[0];
      __syncthreads();
    }
    // L^T, interchanges in reverse
    for (int j = n - 2; j >= 0; --j) {
      const int lm = min(kl, n - 1 - j);
      const double *col = AB + (size_t)j * ldab + kv;
      double s = 0.0;
      for (int i = 1 + tid; i <= lm; i += nt) s += col[i] * c[j + i];
      red[tid] = s;
      __syncthreads();
      for (int sft = nt >> 1; sft > 0; sft >>= 1) {
        if (tid < sft) red[tid] += red[tid + sft];
        __syncthreads();
      }
      if (tid == 0) {
        c[j] -= red[0];
        const int l = ipiv[j];
        if (l != j) { const double t = c[l]; c[l] = c[j]; c[j] = t; }
      }
      __syncthreads();
    }
  }
}

// column blockIdx.y: absax[i] = sum_k |a_ik| |x_k| over the rows of a CSR image, 8 lanes per row
__global__ __launch_bounds__(256) void abs_spmv_kernel(int n, const int64_t *__restrict__ rowptr,
                                                       const int *__restrict__ colidx,
                                                       const double *__restrict__ val, const double *__restrict__ x,
                                                       double *__restrict__ absax, size_t stride) {
  const int i = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 3), part = threadIdx.x & 7;
  const size_t col = (size_t)blockIdx.y * stride;
  double s = 0.0;
  if (i < n)
    for (int64_t p = rowptr[i] + part; p < rowptr[i + 1]; p += 8) s += fabs(val[p]) * fabs(x[col + colidx[p]]);
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (i < n && part == 0) absax[col + i] = s;
}

// column blockIdx.y: r = b - ax and the componentwise backward error
// omega = max_i |r_i| / (|A||x| + |b|)_i  (Arioli, Demmel & Duff; the quantity UMFPACK's refinement monitors)
__global__ void residual_kernel(int n, const double *__restrict__ b, const double *__restrict__ ax,
                                const double *__restrict__ absax, double *__restrict__ r,
                                double *__restrict__ omega, size_t stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t col = (size_t)blockIdx.y * stride;
  double a = 0.0;
  if (i < n) {
    const double v = b[col + i] - ax[col + i];
    r[col + i] = v;
    const double den = absax[col + i] + fabs(b[col + i]);
    a = fabs(v);
    if (a > 0.0) a = den > 0.0 ? a / den : 1e300 * 1e300;
    if (!(a == a)) a = 1e300 * 1e300;  // NaN counts as +inf
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) a = fmax(a, __shfl_xor(a, d, 64));
  // (non-negative doubles order like their bit patterns.  The maximum so far is read first: a quarter of a million
  // wavefronts per batch of 16 columns on 16 addresses took 2.3 ms in atomics that change nothing; a stale value only
  // costs an atomic that was not needed)
  if ((threadIdx.x & 63) == 0 && a > 0.0) {
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(omega + blockIdx.y);
    const unsigned long long mine = (unsigned long long)__double_as_longlong(a);
    if (mine > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, mine);
  }
}

// The same two quantities in ONE pass over the rows of op, with the residual accumulated in twice the working
// precision (round 4): every product a_ik x_k is taken with its rounding error (fma), every addition with its own
// (Knuth's two-sum), the errors summed beside the main sum, and the eight partial sums of a row meet the same way.
// r_i is then the correctly rounded residual up to one rounding of the final sum, not the 1 .. (row length) roundings of
// a plain b - A x.  Why it matters: the componentwise backward error of a refined solution sits at eps / 2 or below,
// while a plainly evaluated residual carries about a row length of roundings of its own — the measured omega then
// stalls at 1.2 .. 1.3 eps (3-D Poisson, 8e6 unknowns: 2.76e-16 after the first refinement step), UMFPACK's test
// omega < eps never fires, and a second refinement step — a whole walk over the factors, 65 ms of 230 at C5 — is spent
// to find out that it changes nothing.  UMFPACK's rules are unchanged; what they look at is more accurate.
// (SPL_LU_RESIDUAL=plain: the three-kernel form above.)
__device__ __forceinline__ void two_sum(double a, double b, double &s, double &e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}
// LPR lanes share a row (8; 2 for matrices of short rows — a mesh row of 7 entries spread over 8 lanes is one load per lane
// and a three-step exchange for 56 bytes: 0.52 -> 0.29 ms per residual at config C5 with two lanes walking it)
template <int LPR>
__global__ __launch_bounds__(256) void residual_dd_kernel(int n, const int64_t *__restrict__ rowptr,
                                                          const int *__restrict__ colidx, const double *__restrict__ val,
                                                          const double *__restrict__ x, const double *__restrict__ b,
                                                          double *__restrict__ r, double *__restrict__ omega, size_t stride) {
  const int i = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) / LPR), part = threadIdx.x % LPR;
  const size_t col = (size_t)blockIdx.y * stride;
  double hi = 0.0, lo = 0.0, ab = 0.0;
  if (i < n) {
    if (part == 0) hi = b[col + i];
    for (int64_t p = rowptr[i] + part; p < rowptr[i + 1]; p += LPR) {
      const double a = val[p], xv = x[col + colidx[p]];
      const double ph = a * xv, pl = __builtin_fma(a, xv, -ph);
      double sm, e;
      two_sum(hi, -ph, sm, e);
      hi = sm;
      lo += e - pl;
      ab += fabs(a) * fabs(xv);
    }
  }
#pragma unroll
  for (int d = 1; d < LPR; d <<= 1) {
    const double oh = __shfl_xor(hi, d, 64), ol = __shfl_xor(lo, d, 64);
    double sm, e;
    two_sum(hi, oh, sm, e);
    hi = sm;
    lo += ol + e;
    ab += __shfl_xor(ab, d, 64);
  }
  double a = 0.0;
  if (i < n && part == 0) {
    const double v = hi + lo;
    r[col + i] = v;
    const double den = ab + fabs(b[col + i]);
    a = fabs(v);
    if (a > 0.0) a = den > 0.0 ? a / den : 1e300 * 1e300;
    if (!(a == a)) a = 1e300 * 1e300;  // NaN counts as +inf
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) a = fmax(a, __shfl_xor(a, d, 64));
  if ((threadIdx.x & 63) == 0 && a > 0.0) {
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(omega + blockIdx.y);
    const unsigned long long mine = (unsigned long long)__double_as_longlong(a);
    if (mine > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, mine);
  }
}

__global__ void add_kernel(size_t n, double *__restrict__ x, const double *__restrict__ d) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += d[i];
}

// vector helpers of the FGMRES polish (gmres_polish): <a, b> in two stages — every workgroup writes the sum of its
// share to part[blockIdx.x] (fixed strides, fixed tree), one workgroup adds the parts in index order: the same bits on
// every run (an atomic sum made polished solutions differ from run to run: ADVICE r3);  y = alpha * x + beta * y
__global__ __launch_bounds__(256) void dot_part_kernel(size_t n, const double *__restrict__ a, const double *__restrict__ b,
                                                       double *__restrict__ part) {
  __shared__ double red[4];
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i] * b[i];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void dot_final_kernel(int nparts, const double *__restrict__ part, double *__restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 128; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}
__global__ __launch_bounds__(256) void axpby_kernel(size_t n, double alpha, const double *__restrict__ x, double beta,
                                                    double *__restrict__ y) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = alpha * x[i] + (beta == 0.0 ? 0.0 : beta * y[i]);
}

inline Symbolic *as_symbolic(void *p) {
  Symbolic *s = static_cast<Symbolic *>(p);
  return (s && s->magic == kSymMagic) ? s : nullptr;
}
inline Numeric *as_numeric(void *p) {
  Numeric *s = static_cast<Numeric *>(p);
  return (s && s->magic == kNumMagic) ? s : nullptr;
}

int validate_host_csc(int n_row, int n_col, const int *Ap, const int *Ai) {
  if (Ap[0] != 0) return UMFPACK_ERROR_invalid_matrix;
  for (int j = 0; j < n_col; ++j) {
    if (Ap[j] > Ap[j + 1]) return UMFPACK_ERROR_invalid_matrix;
    for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
      if (Ai[p] < 0 || Ai[p] >= n_row) return UMFPACK_ERROR_invalid_matrix;
      if (p > Ap[j] && Ai[p] <= Ai[p - 1]) return UMFPACK_ERROR_invalid_matrix;  // unsorted / duplicate
    }
  }
  return UMFPACK_OK;
}

// columns of c (device, permuted order, column r at d_c + r * stride) <- solutions of B z = c or
// B^T z = c; kalloc >= k columns are allocated (a multiple of kSolveGroup when k > 1)
void band_solve(const Numeric *N, int sys, double *d_c, int k, size_t stride, hipStream_t s) {
  if (N->n == 0 || k == 0) return;
  if (N->mfact) {
    mf_solve(N->mfact, sys, d_c, k, stride, s);
    return;
  }
  if (N->nopiv) {
    band_nopiv_solve(sys, N->n, N->kl, N->ku, N->ldab, N->AB.get(), N->blkinv.get(), d_c, k, stride, s);
    return;
  }
  for (int c = 0; c < k; ++c)
    hipLaunchKernelGGL(band_solve_kernel, dim3(1), dim3(1024), 0, s, sys, N->n, N->kl, N->ku, N->ldab,
                       N->AB.get(), N->ipiv.get(), d_c + (size_t)c * stride);
}

// d_x (device, original order) <- op(A)^-1 d_b using the factors only, k columns
void factor_solve(const Numeric *N, int sys, const double *d_b, double *d_x, double *d_work, int k, size_t stride,
                  hipStream_t s) {
  const int n = N->n;
  if (n == 0 || k == 0) return;
  const dim3 g((unsigned)((n + 255) / 256), (unsigned)k);
  if (N->sp_stage == 1) {
    // factors of B = Dr P A Dc:  A x = b  <=>  B (Dc^-1 x) = Dr P b;  A^T x = b  <=>  B^T (Dr^-1 P x) = Dc b
    // (the row permutation, the scalings and B's ordering are composed into one gather on either side)
    const int a = sys == UMFPACK_A ? 0 : 2;
    hipLaunchKernelGGL(gather_scale_kernel, g, dim3(256), 0, s, n, N->sp_idx[a].get(), N->sp_scale[a].get(), d_b, d_work,
                       stride);
    band_solve(N, sys, d_work, k, stride, s);
    hipLaunchKernelGGL(gather_scale_kernel, g, dim3(256), 0, s, n, N->sp_idx[a + 1].get(), N->sp_scale[a + 1].get(), d_work,
                       d_x, stride);
    return;
  }
  // B = P A P^T  =>  A x = b  <=>  B (P x) = P b ; (P v)[k] = v[perm[k]]
  // Factors of a SYMMETRIC matrix (L D L^T on the tree): the transposed system is the same system, so it takes the
  // untransposed kernels (the faster ones: their panels run down the contiguous direction).  On native complex fronts
  // sys = At asks for the CONJUGATE transpose: A^H = conj(A) for A == A^T, and conj(A) x = b <=> A conj(x) = conj(b).
  const bool same = sys != UMFPACK_A && N->mfact && N->mf_sym && !N->mf_piv;
  const bool conj = same && N->zfront;
  hipLaunchKernelGGL(gather_perm_kernel, g, dim3(256), 0, s, n, N->perm.get(), d_b, d_work, stride, conj ? 1 : 0);
  band_solve(N, same ? UMFPACK_A : sys, d_work, k, stride, s);
  hipLaunchKernelGGL(gather_perm_kernel, g, dim3(256), 0, s, n, N->inv.get(), d_work, d_x, stride, conj ? 2 : 0);  // x[i] = z[inv[i]]
}

// install an ordering (new -> old, old -> new) on the device
void set_ordering(Numeric *N, const std::vector<int> &perm, const std::vector<int> &inv, hipStream_t s) {
  const size_t n = (size_t)N->n;
  SPL_HIP(hipMemcpyAsync(N->perm.get(), perm.data(), n * sizeof(int), hipMemcpyHostToDevice, s));
  SPL_HIP(hipMemcpyAsync(N->inv.get(), inv.data(), n * sizeof(int), hipMemcpyHostToDevice, s));
  SPL_HIP(hipStreamSynchronize(s));
}

// A == A^T exactly?  The row-major images of A and of A^T (both with ascending columns inside a row) are then the same
// arrays, bit for bit.
__global__ __launch_bounds__(256) void same_csr_kernel(int64_t n, int64_t nnz, const int *__restrict__ p1, const int *__restrict__ p2,
                                                       const int *__restrict__ i1, const int *__restrict__ i2,
                                                       const double *__restrict__ x1, const double *__restrict__ x2,
                                                       int *__restrict__ differ) {
  bool bad = false;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz + n + 1; k += (int64_t)gridDim.x * blockDim.x) {
    if (k < nnz) bad |= i1[k] != i2[k] || __double_as_longlong(x1[k]) != __double_as_longlong(x2[k]);
    else bad |= p1[k - nnz] != p2[k - nnz];
  }
  if (bad) *differ = 1;
}

static bool matrix_is_symmetric(const Numeric *N, hipStream_t s) {
  const char *e = getenv("SPL_LU_SYMMETRIC");
  if (e && e[0] == '0') return false;
  const Matrix *A = N->A, *At = N->At;
  if (!A || !At || A->nnz != At->nnz || !A->rowptr.get() || !At->rowptr.get()) return false;
  DBuf<int> differ(1);
  SPL_HIP(hipMemsetAsync(differ.get(), 0, sizeof(int), s));
  const int64_t total = A->nnz + (int64_t)N->n + 1;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(same_csr_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (int64_t)N->n, A->nnz, A->rowptr.get(),
                     At->rowptr.get(), A->colidx.get(), At->colidx.get(), A->val.get(), At->val.get(), differ.get());
  int h = 1;
  SPL_HIP(hipMemcpyAsync(&h, differ.get(), sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  return h == 0;
}

// rs[g] = 1 / sum_j |A(perm[step g], j)| (1 if the row is empty): the row scales of the block pivoting, in the new
// ordering; step = 2 for the native complex tree on an embedding (row 2g of E stands for complex row g)
__global__ __launch_bounds__(256) void row_scale_kernel(int64_t m, int step, const int *__restrict__ perm,
                                                        const int64_t *__restrict__ rowptr, const double *__restrict__ val,
                                                        double *__restrict__ rs) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= m) return;
  const int i = perm[step * g] ;
  double sum = 0.0;
  for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) sum += fabs(val[p]);
  rs[g] = sum > 0.0 && sum < 1e300 ? 1.0 / sum : 1.0;
}

// resident bytes of the tree factors held (or about to be built)
static size_t tree_factor_bytes(const Numeric *N) {
  if (N->zfront && N->ztree) return mf_device_bytes(*N->ztree, 2);
  return N->tree ? mf_device_bytes(*N->tree, 1) : 0;
}

// multifrontal factors without interchanges on the nested-dissection tree (multifrontal.hip)
void factor_multifrontal(Numeric *N, hipStream_t s) {
  N->AB.release();
  N->blkinv.release();
  if (N->mfact) { mf_free(N->mfact); N->mfact = nullptr; }
  const size_t free_b = device_free_bytes();
  N->zfront = N->ztree ? 1 : 0;
  if (tree_factor_bytes(N) > free_b - free_b / 8) throw DeviceError{SPL_ERROR_out_of_memory};
  N->nopiv = 1;
  if (!N->A->rowptr.get()) throw DeviceError{SPL_ERROR_index_overflow};  // int32 row pointers at this seam
  const char *bp = getenv("SPL_LU_BLOCK_PIVOT");
  const bool pivot = !N->dominant && !(bp && bp[0] == '0');
  if (pivot && N->A->rowptr64.get()) {
    const int step = N->zfront ? 2 : 1;
    const int64_t m = (int64_t)N->n / step;
    if ((int64_t)N->rscale.n != m) N->rscale.alloc((size_t)m);
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, m, step, N->perm.get(),
                       N->A->rowptr64.get(), N->A->val.get(), N->rscale.get());
  }
  const double *rscale = pivot && N->rscale.get() && N->A->rowptr64.get() ? N->rscale.get() : nullptr;
  if (N->zfront) {
    // the embedding of a complex matrix: complex fronts on the tree of the complex pattern, assembled from E's own
    // arrays (column 2j / row 2j of E hold column / row j of the complex matrix); perm / inv are the expanded ordering
    N->mf_sym = N->zsym && !N->block_pivot_retry;
    N->mf_piv = pivot && !N->mf_sym;
    N->mfact = mf_factor(N->ztree, N->At->rowptr.get(), N->At->colidx.get(), N->At->val.get(), N->A->rowptr.get(),
                         N->A->colidx.get(), N->A->val.get(), N->perm.get(), N->inv.get(), s, N->mf_sym != 0, true,
                         N->mf_piv != 0, rscale);
    N->singular = mf_singular(N->mfact);
    return;
  }
  // a symmetric matrix (exactly: A == A^T) is factored as L D L^T on the same fronts: the trailing updates only
  // compute the tiles on and below the diagonal (Band::sym, csrc/dense_lu_kernels.hpp); SPL_LU_SYMMETRIC=0: plain LU
  N->mf_sym = !N->block_pivot_retry && matrix_is_symmetric(N, s) ? 1 : 0;
  N->mf_piv = pivot && !N->mf_sym;
  N->mfact = mf_factor(N->tree, N->At->rowptr.get(), N->At->colidx.get(), N->At->val.get(), N->A->rowptr.get(),
                       N->A->colidx.get(), N->A->val.get(), N->perm.get(), N->inv.get(), s, N->mf_sym != 0, false,
                       N->mf_piv != 0, rscale);
  N->singular = mf_singular(N->mfact);
}

// Static pivoting (static_pivot.hpp): factor B = Dr P A Dc — the maximum-product transversal on the diagonal,
// |b_jj| = 1 >= |b_ij| — without interchanges on B's own nested-dissection tree.  Returns false when it
// cannot be done (structurally singular, scalings out of range, does not fit, zero pivot): the caller then
// falls back to the band factorisation with partial pivoting.  The object is left consistent either way.
bool factor_static_pivot(Numeric *N, const int *Ap, const int *Ai, const double *Ax, hipStream_t s) {
  const int n = N->n;
  if (N->sp_stage != 0 || n < 2) return false;
  const char *off = getenv("SPL_LU_STATIC_PIVOT");
  if (off && off[0] == '0') return false;
  N->sp_stage = 2;  // whatever happens below, it is not tried twice
  const bool timing = getenv("SPL_MF_TIMING") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
  sp::Transversal T;
  double match_seconds = 180.0;  // SPL_LU_MATCH_SECONDS: how long the (sequential, host) matching may take
  if (const char *ev = getenv("SPL_LU_MATCH_SECONDS")) match_seconds = atof(ev);
  if (!sp::max_product_transversal(n, Ap, Ai, Ax, T, match_seconds)) return false;
  if (timing) fprintf(stderr, "[static pivot] transversal done at %.1f ms\n", since());
  std::vector<int> Bp, Bi;
  std::vector<double> Bx;
  sp::permuted_scaled_csc(n, Ap, Ai, Ax, T, Bp, Bi, Bx);
  std::shared_ptr<mf::Tree> tree = std::make_shared<mf::Tree>();
  mf::build_tree(n, Bp.data(), Bi.data(), 256, *tree);
  if (timing) fprintf(stderr, "[static pivot] B and its tree (%d fronts, %.3g flops) at %.1f ms\n", tree->nfronts, tree->flops, since());
  {
    size_t held = (N->AB.n + N->blkinv.n) * sizeof(double);
    if (N->mfact) held += tree_factor_bytes(N);
    const size_t avail = device_free_bytes() + held;
    if (mf_device_bytes(*tree, 1) + (size_t)Bp[(size_t)n] * 24 > avail - avail / 8) return false;
  }
  void *hBt = nullptr, *hB = nullptr;
  if (spl_matrix_create_csr(n, n, 0, n, Bp.data(), Bi.data(), Bx.data(), &hBt) != SPL_OK) return false;
  if (spl_matrix_create(n, n, Bp.data(), Bi.data(), Bx.data(), &hB) != SPL_OK) { spl_matrix_free(&hBt); return false; }
  std::unique_ptr<Matrix> Bt(static_cast<Matrix *>(hBt)), B(static_cast<Matrix *>(hB));
  if (!Bt->rowptr.get() || !B->rowptr.get()) return false;
  // the gathers around a solve (see factor_solve): rowof[r] = the row of A that became row r of B
  const std::vector<int> &pnd = tree->perm, &ind = tree->inv;
  std::vector<int> idx[4];
  std::vector<double> sc[4];
  for (int a = 0; a < 4; ++a) { idx[a].resize((size_t)n); sc[a].resize((size_t)n); }
  for (int k = 0; k < n; ++k) {
    const int r = pnd[(size_t)k];                 // row / column of B at position k of the ordering
    const int i = T.row_of_col[(size_t)r];        // row of A that became row r of B
    idx[0][(size_t)k] = i;  sc[0][(size_t)k] = T.dr[(size_t)i];   // work[k] = dr_i b_i
    idx[2][(size_t)k] = r;  sc[2][(size_t)k] = T.dc[(size_t)r];   // work[k] = dc_r b_r      (A^T x = b)
  }
  for (int j = 0; j < n; ++j) {
    idx[1][(size_t)j] = ind[(size_t)j];  sc[1][(size_t)j] = T.dc[(size_t)j];                           // x_j = dc_j z[inv[j]]
    idx[3][(size_t)j] = ind[(size_t)T.col_of_row[(size_t)j]];  sc[3][(size_t)j] = T.dr[(size_t)j];     // x_i = dr_i z[inv[newrow(i)]]
  }
  // from here on the old factors are released: failures leave the object without factors (broken)
  try {
    N->AB.release();
    N->blkinv.release();
    if (N->mfact) { mf_free(N->mfact); N->mfact = nullptr; }
    N->tree = tree;
    N->ztree.reset();  // B = Dr P E Dc is not the embedding of a complex matrix: real fronts from here on
    N->zfront = 0;
    set_ordering(N, tree->perm, tree->inv, s);
    for (int a = 0; a < 4; ++a) {
      N->sp_idx[a].alloc((size_t)n);
      N->sp_scale[a].alloc((size_t)n);
      SPL_HIP(hipMemcpyAsync(N->sp_idx[a].get(), idx[a].data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
      SPL_HIP(hipMemcpyAsync(N->sp_scale[a].get(), sc[a].data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    }
    SPL_HIP(hipStreamSynchronize(s));
    delete N->spA;
    delete N->spAt;
    N->spA = B.release();
    N->spAt = Bt.release();
    N->nopiv = 1;
    N->mf_sym = 0;  // B = Dr P A Dc is not symmetric
    N->mfact = mf_factor(N->tree, N->spAt->rowptr.get(), N->spAt->colidx.get(), N->spAt->val.get(), N->spA->rowptr.get(),
                         N->spA->colidx.get(), N->spA->val.get(), N->perm.get(), N->inv.get(), s, false, false);
    if (mf_singular(N->mfact)) {  // a zero pivot: these factors are useless; the band fallback rebuilds everything
      N->sp_stage = 2;
      return false;
    }
    N->singular = 0;
    N->sp_stage = 1;
    if (timing) { (void)hipStreamSynchronize(s); fprintf(stderr, "[static pivot] factored at %.1f ms\n", since()); }
    return true;
  } catch (...) {
    N->broken = 1;
    throw;
  }
}

// (re)build the band factors of P A P^T from the device copy of A^T's rows (= the CSC arrays);
// nopiv selects the blocked no-interchange factorisation (band_nopiv.hip) or LAPACK-style
// partial pivoting.  Throws DeviceError; sets N->singular.
// The reverse Cuthill-McKee ordering of an object whose analysis skipped it (symbolic_common: the tree won by a lower
// bound), from the pattern the object holds: the rows of A^T are the columns of A.
static void ensure_band_ordering(Numeric *N, hipStream_t s) {
  const int n = N->n;
  if ((int)N->band_perm.size() == n || n == 0) return;
  if (!N->At || !N->At->rowptr.get()) throw DeviceError{SPL_ERROR_index_overflow};
  const size_t nnz = (size_t)N->At->nnz;
  std::vector<int> p((size_t)n + 1), i(nnz);
  SPL_HIP(hipMemcpyAsync(p.data(), N->At->rowptr.get(), ((size_t)n + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipMemcpyAsync(i.data(), N->At->colidx.get(), nnz * sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  std::vector<int> perm, inv((size_t)n, 0);
  rcm_order(n, p.data(), i.data(), perm);
  for (int k = 0; k < n; ++k) inv[(size_t)perm[(size_t)k]] = k;
  int kl = 0, ku = 0;
  for (int j = 0; j < n; ++j) {
    const int nj = inv[(size_t)j];
    for (int q = p[(size_t)j]; q < p[(size_t)j + 1]; ++q) {
      const int ni = inv[(size_t)i[(size_t)q]];
      kl = std::max(kl, ni - nj);
      ku = std::max(ku, nj - ni);
    }
  }
  N->band_perm.swap(perm);
  N->band_inv.swap(inv);
  N->kl = kl;
  N->ku = ku;
}

void factor_band(Numeric *N, bool nopiv, hipStream_t s) {
  const int n = N->n;
  ensure_band_ordering(N, s);
  // Will the band fit?  Decided BEFORE anything of the current factors is released: a band that does not
  // fit (a large 3-D matrix whose speculation failed) must leave the object as it was — its speculative
  // factors still answer solves, which then report the error again instead of reading freed memory.
  const int ldab_new = nopiv ? band_nopiv_ldab(N->kl, N->ku) : (2 * N->kl + N->ku + 1);
  const size_t band_elems = (size_t)ldab_new * (size_t)n;
  {
    size_t held = (N->AB.n + N->blkinv.n) * sizeof(double);
    if (N->mfact) held += tree_factor_bytes(N);
    const size_t avail = device_free_bytes() + held;
    if (band_elems * sizeof(double) > avail - avail / 8) throw DeviceError{SPL_ERROR_out_of_memory};  // too wide
  }
  try {
    if (N->mfact || N->tree) {  // leaving the multifrontal path: the band paths use the RCM ordering
      if (N->mfact) { mf_free(N->mfact); N->mfact = nullptr; }
      N->tree.reset();
      N->ztree.reset();
      N->zfront = 0;
      set_ordering(N, N->band_perm, N->band_inv, s);
    }
    if (N->sp_stage == 1) N->sp_stage = 2;  // the factors of B are gone: solves use the plain gathers again
    delete N->spA; N->spA = nullptr;
    delete N->spAt; N->spAt = nullptr;
    N->nopiv = nopiv ? 1 : 0;
    N->ldab = ldab_new;
    N->AB.release();
    N->blkinv.release();
    N->AB.alloc(band_elems);
    SPL_HIP(hipMemsetAsync(N->AB.get(), 0, band_elems * sizeof(double), s));
    if (nopiv) N->blkinv.alloc(band_nopiv_inverse_elems(n));
  } catch (...) {
    N->broken = 1;  // the old factors are gone and the new ones could not be built
    throw;
  }
  if (nopiv) {
    N->singular = band_nopiv_factor(n, N->kl, N->ku, N->ldab, N->AB.get(), N->blkinv.get(), N->At->rowptr.get(),
                                    N->At->colidx.get(), N->At->val.get(), N->inv.get(), s);
    return;
  }
  // scatter P A P^T into band storage (reads the CSC arrays the At handle already holds)
  const int kv = N->kl + N->ku;
  if (N->At->nnz > 0)
    hipLaunchKernelGGL(band_scatter_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, n, N->At->rowptr.get(),
                       N->At->colidx.get(), N->At->val.get(), N->inv.get(), kv, N->ldab, N->AB.get());
  DBuf<BandState> state(1);
  SPL_HIP(hipMemsetAsync(state.get(), 0, sizeof(BandState), s));
  const int64_t work_per_col = (int64_t)(N->kl) * (int64_t)(kv + 1);
  if (work_per_col <= 16384) {
    hipLaunchKernelGGL(band_lu_fused_kernel, dim3(1), dim3(1024), 0, s, n, N->kl, N->ku, N->ldab, N->AB.get(),
                       N->ipiv.get(), state.get());
  } else {
    for (int j = 0; j < n; ++j) {
      hipLaunchKernelGGL(band_pivot_kernel, dim3(1), dim3(256), 0, s, j, n, N->kl, N->ku, N->ldab, N->AB.get(),
                         N->ipiv.get(), state.get());
      const int km = std::min(N->kl, n - 1 - j);
      const int width = std::min(kv, n - 1 - j);
      if (km > 0 && width > 0)
        hipLaunchKernelGGL(band_update_kernel, dim3((unsigned)((km + 255) / 256), (unsigned)width), dim3(256), 0, s,
                           j, n, N->kl, N->ku, N->ldab, N->AB.get(), state.get());
    }
  }
  BandState hs;
  SPL_HIP(hipMemcpyAsync(&hs, state.get(), sizeof(BandState), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  SPL_HIP(hipGetLastError());
  N->singular = hs.singular;
}

}  // namespace
}  // namespace spl

using namespace spl;

extern "C" {

// The analysis behind umfpack_di_symbolic and, with mult = 2, umfpack_zi_symbolic: (Ap, Ai) is the n x n pattern
// that is ORDERED; the object describes the (n mult) x (n mult) matrix of dense mult x mult blocks whose CSC
// pattern is (Ep, Ei) — what numeric will be handed and checks against.  mult = 1: Ep = Ap, Ei = Ai.
static int symbolic_common(int n, const int *Ap, const int *Ai, int mult, const int *Ep, const int *Ei,
                           void **SymbolicOut) {
  try {
    std::unique_ptr<Symbolic> S(new Symbolic());
    S->n = n * mult;
    S->nnz = Ep[S->n];
    S->Ap.assign(Ep, Ep + S->n + 1);
    // The two orderings are independent host work: for anything that is not tiny the nested
    // dissection runs on its own thread(s) while this one does the band ordering.
    // Multifrontal or band?  The band factorisation costs about 2 n kl ku flops and n (kl+ku+1)
    // entries; nested dissection is far cheaper on 2-D / 3-D meshes and no better on narrow bands.
    // SPL_LU_METHOD=mf / band forces the choice (tests).
    const char *method = getenv("SPL_LU_METHOD");
    const bool force_mf = method && method[0] == 'm', force_band = method && method[0] == 'b';
    std::future<std::shared_ptr<mf::Tree>> tree_job;
    std::shared_ptr<mf::Tree> small_tree = mult > 1 ? std::make_shared<mf::Tree>() : nullptr;  // of the complex pattern
    const bool want_tree = !force_band && (force_mf || S->n >= 1024);
    const int pattern_symmetric = want_tree ? (mf::detail::structurally_symmetric(n, Ap, Ai) ? 1 : 0) : -1;
    // HIP's current device belongs to the THREAD and is 0 in a new one: the job takes the caller's along, so that the
    // level service (nd_levels.hip: a slab of 1 GiB and more, every traversal kernel) lands on the device the caller
    // selected — with one rank per GPU every rank's analysis would pile onto device 0 otherwise
    int caller_device = -1;
    if (hipGetDevice(&caller_device) != hipSuccess) {
      caller_device = -1;
      (void)hipGetLastError();
    }
    // Streams for the level structures of this analysis (4) and for the factorisation that will follow it on this thread
    // (17): created beside the first host work of the analysis instead of in front of the first traversal and of the first
    // front (15 ms for the first stream of a process, 0.3 - 0.6 ms for every other one).  Joined before this call returns.
    std::future<void> warm_streams;
    if (want_tree && caller_device >= 0 && !(getenv("SPL_PREWARM_STREAMS") && atoi(getenv("SPL_PREWARM_STREAMS")) == 0)) {
      try {
        warm_streams = std::async(std::launch::async, [caller_device] { pooled_streams_prewarm(caller_device, 21); });
      } catch (...) {  // no thread to be had: the streams are made where they are needed
      }
    }
    struct JoinWarm {
      std::future<void> &f;
      ~JoinWarm() {
        if (f.valid()) f.wait();
      }
    } join_warm{warm_streams};
    if (want_tree)
      tree_job = std::async(std::launch::async, [n, Ap, Ai, mult, small_tree, pattern_symmetric, caller_device] {
        if (caller_device >= 0 && hipSetDevice(caller_device) != hipSuccess) (void)hipGetLastError();
        std::shared_ptr<mf::Tree> T = std::make_shared<mf::Tree>();
        mf::build_tree(n, Ap, Ai, 256 / mult, *T, mult, small_tree.get(), pattern_symmetric, &make_gpu_level_service);
        return T;
      });
    const bool timing = getenv("SPL_MF_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    try {
      S->ai_hash = hash_indices(Ei, S->nnz);  // (beside the dissection)
    } catch (...) {
      if (tree_job.valid()) tree_job.wait();
      throw;
    }
    // the band ordering (reverse Cuthill-McKee) and its bandwidths
    auto band_ordering = [&] {
      std::vector<int> perm, inv((size_t)n, 0);
      rcm_order(n, Ap, Ai, perm);
      if (timing) fprintf(stderr, "[symbolic] RCM done at %.1f ms\n", since());
      for (int k = 0; k < n; ++k) inv[(size_t)perm[(size_t)k]] = k;
      int kl = 0, ku = 0;
      for (int j = 0; j < n; ++j) {
        const int nj = inv[(size_t)j];
        for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
          const int ni = inv[(size_t)Ai[p]];
          kl = std::max(kl, ni - nj);
          ku = std::max(ku, nj - ni);
        }
      }
      if (mult == 1) {
        S->perm.swap(perm);
        S->inv.swap(inv);
        S->kl = kl;
        S->ku = ku;
      } else {  // block (i, j) covers rows mult i .. mult i + mult - 1, columns mult j .. mult j + mult - 1
        S->perm.resize((size_t)S->n);
        S->inv.resize((size_t)S->n);
        for (int k = 0; k < n; ++k)
          for (int h = 0; h < mult; ++h) S->perm[(size_t)k * mult + h] = perm[(size_t)k] * mult + h;
        for (int k = 0; k < S->n; ++k) S->inv[(size_t)S->perm[(size_t)k]] = k;
        S->kl = mult * kl + (mult - 1);
        S->ku = mult * ku + (mult - 1);
      }
      S->have_band = true;
    };
    // Measured model of the two factorisations (MI355X, tools/bench_band_vs_tree.py): the band
    // is a chain of n / 64 block steps of a few launches each, about 1.4 us per column however
    // narrow it is, plus its flops at the rate of its large windows; the tree costs a few
    // launches per level (about 2.5 ms for a whole tree) plus its flops at a lower rate (many
    // small fronts).  The break-even is near n = 2 000 on 2-D and 3-D meshes alike.
    auto band_seconds = [&](double kl, double ku) { return 1.4e-6 * S->n + 2.0 * S->n * kl * ku / 3e13; };
    // Large matrices (round 3): the dissection first, with every core (the band ordering used to run beside it and
    // took a thread and its share of the memory system: 1.5 s at config C5, behind which a shorter dissection would
    // only have waited).  The level structure the root region was cut with bounds the band from below — a breadth-
    // first search has at most diameter + 1 levels, and no ordering has a bandwidth below (n - 1) / diameter — so
    // where even that band loses to the tree, the band ordering is not computed at all; a fallback that needs it
    // later (factor_band) computes it then, from the pattern the Numeric object holds.
    // (structurally symmetric patterns only: there kl = ku >= that bandwidth; an unsymmetric pattern may have one narrow
    // side, and keeps the two orderings side by side)
    const bool tree_first = tree_job.valid() && S->n >= 100000 && pattern_symmetric == 1 && !getenv("SPL_LU_ALWAYS_RCM");
    try {
      if (!tree_first) band_ordering();
    } catch (...) {
      if (tree_job.valid()) tree_job.wait();  // Ap / Ai are borrowed: nobody may outlive this call
      throw;
    }
    if (tree_job.valid()) {
      if (timing && !tree_first) fprintf(stderr, "[symbolic] bandwidths done at %.1f ms\n", since());
      std::shared_ptr<mf::Tree> T = tree_job.get();
      if (timing) fprintf(stderr, "[symbolic] nested dissection tree ready at %.1f ms\n", since());
      const double t_tree = 2.5e-3 + T->flops / 2e13;
      bool take_tree = force_mf;
      if (tree_first && !take_tree) {
        const mf::Tree &Ts = small_tree && small_tree->nfronts > 0 ? *small_tree : *T;  // the graph that was dissected
        // A lower bound on the bandwidth of ANY symmetric reordering: with bandwidth w a path of length d reaches at most
        // 1 + d w vertices, so w >= (n - 1) / diameter; the root's level structure has eccentricity root_levels - 1 and
        // diameter <= 2 eccentricity (dividing by the eccentricity itself is up to twice too large)
        const double bw = Ts.root_levels >= 2 ? std::ceil((double)(n - 1) / (2.0 * (double)(Ts.root_levels - 1))) : 0.0;
        const double kmin = mult * bw;  // both half-bandwidths of a structurally symmetric reordering of A + A^T's pattern
        if (kmin > 0 && t_tree < band_seconds(kmin, kmin)) take_tree = true;
        if (timing) fprintf(stderr, "[symbolic] %d levels: band >= %.0f wide, %.3g s at least; tree %.3g s\n", Ts.root_levels, kmin, band_seconds(kmin, kmin), t_tree);
      }
      if (!take_tree) {
        if (!S->have_band) band_ordering();
        take_tree = t_tree < band_seconds((double)S->kl, (double)S->ku);
      }
      if (take_tree) {
        S->tree = T;
        if (small_tree && small_tree->nfronts > 0) S->ztree = small_tree;
      }
    }
    *SymbolicOut = S.release();
    return UMFPACK_OK;
  } catch (const std::bad_alloc &) {
    return UMFPACK_ERROR_out_of_memory;
  } catch (...) {  // e.g. std::system_error from a thread that could not be started: never across the C ABI
    return UMFPACK_ERROR_internal_error;
  }
}

}  // extern "C"

namespace spl {
// analysis of a rectangular matrix: shape and pattern only (Symbolic::rectangular)
int symbolic_rectangular(int n_row, int n_col, const int *Ap, const int *Ai, void **SymbolicOut) {
  try {
    std::unique_ptr<Symbolic> S(new Symbolic());
    S->rectangular = true;
    S->n_row = n_row;
    S->n_col = n_col;
    S->n = n_col;
    S->nnz = Ap[n_col];
    S->Ap.assign(Ap, Ap + n_col + 1);
    S->ai_hash = hash_indices(Ai, S->nnz);
    *SymbolicOut = S.release();
    return UMFPACK_OK;
  } catch (const std::bad_alloc &) {
    return UMFPACK_ERROR_out_of_memory;
  } catch (...) {
    return UMFPACK_ERROR_internal_error;
  }
}
// size of a maximum matching of the columns to the rows over the entries keep[p] != 0: augmenting paths by depth-first
// search with a look-ahead for free rows (Duff's MC21), iterative; O(n_col nnz) at worst, near-linear on what occurs
int structural_rank(int n_row, int n_col, const int *Ap, const int *Ai, const std::vector<char> &keep) {
  std::vector<int> col_of_row((size_t)n_row, -1), look((size_t)n_col), seen((size_t)n_row, -1);
  for (int j = 0; j < n_col; ++j) look[(size_t)j] = Ap[j];
  std::vector<int> stack_col, stack_pos, row_of;
  int rank = 0;
  for (int j0 = 0; j0 < n_col; ++j0) {
    stack_col.assign(1, j0);
    stack_pos.assign(1, Ap[j0]);
    row_of.assign(1, -1);  // the row through which a column on the stack was reached
    bool found = false;
    while (!stack_col.empty() && !found) {
      const int j = stack_col.back();
      int free_row = -1;
      for (int &p = look[(size_t)j]; p < Ap[j + 1]; ++p)  // look-ahead: every entry of a column is tested once for a free row
        if (keep[(size_t)p] && col_of_row[(size_t)Ai[p]] < 0) { free_row = Ai[p]; ++p; break; }
      bool pushed = false;
      if (free_row < 0)
        for (int &p = stack_pos.back(); p < Ap[j + 1]; ++p) {
          const int i = Ai[p];
          if (!keep[(size_t)p] || seen[(size_t)i] == j0) continue;
          seen[(size_t)i] = j0;
          const int next_col = col_of_row[(size_t)i];
          ++p;
          if (next_col < 0) {  // (matched rows stay matched, so the look-ahead has seen every free one; kept for safety)
            free_row = i;
            break;
          }
          stack_col.push_back(next_col);
          stack_pos.push_back(Ap[next_col]);
          row_of.push_back(i);
          pushed = true;
          break;
        }
      if (free_row >= 0) {
        // augment along the stack: the last column takes the free row, every column before it the row that led on from it
        int r = free_row;
        for (size_t k = stack_col.size(); k-- > 0;) {
          col_of_row[(size_t)r] = stack_col[k];
          r = row_of[k];
        }
        found = true;
        break;
      }
      if (!pushed) {
        stack_col.pop_back();
        stack_pos.pop_back();
        row_of.pop_back();
      }
    }
    rank += found ? 1 : 0;
  }
  return rank;
}
// Numerical rank of a SMALL rectangular matrix on the host (the structural rank over the non-zero entries
// calls the 3 x 2 matrix of ones regular; UMFPACK meets an exactly zero pivot there and warns): Gaussian elimination
// with row pivoting, column by column, a column without a non-zero candidate is dependent and skipped.  As in UMFPACK a
// pivot counts unless it is exactly zero (or NaN).  Dense, so only where rows x columns x min(rows, columns) stays
// below kDenseRankWork; -1: too large, the caller keeps the structural rank (documented in include/umfpack_hip.h).
// re, im: the values (im may be null; packed complex: im = re + 1 with stride 2).
constexpr double kDenseRankWork = 4e8;
int dense_numeric_rank(int n_row, int n_col, const int *Ap, const int *Ai, const double *re, const double *im, int vstride) {
  const double work = (double)n_row * (double)n_col * (double)std::min(n_row, n_col);
  if (work > kDenseRankWork) return -1;
  std::vector<std::complex<double>> a((size_t)n_row * (size_t)n_col);  // column-major
  for (int j = 0; j < n_col; ++j)
    for (int p = Ap[j]; p < Ap[j + 1]; ++p)
      a[(size_t)Ai[p] + (size_t)j * (size_t)n_row] = std::complex<double>(re[(size_t)p * vstride], im ? im[(size_t)p * vstride] : 0.0);
  std::vector<char> used((size_t)n_row, 0);
  int rank = 0;
  for (int j = 0; j < n_col && rank < n_row; ++j) {
    std::complex<double> *cj = a.data() + (size_t)j * (size_t)n_row;
    int piv = -1;
    double best = 0.0;
    for (int i = 0; i < n_row; ++i) {
      if (used[(size_t)i]) continue;
      const double mag = std::abs(cj[i]);
      if (mag > best) { best = mag; piv = i; }  // (NaN never compares greater: no pivot)
    }
    if (piv < 0) continue;  // dependent on the columns before it
    used[(size_t)piv] = 1;
    ++rank;
    for (int c = j + 1; c < n_col; ++c) {
      std::complex<double> *cc = a.data() + (size_t)c * (size_t)n_row;
      if (cc[piv] == std::complex<double>(0.0, 0.0)) continue;
      const std::complex<double> f = cc[piv] / cj[piv];
      for (int i = 0; i < n_row; ++i)
        if (!used[(size_t)i]) cc[i] -= f * cj[i];
    }
  }
  return rank;
}

// "factorisation" of a rectangular matrix: the pattern check and the status (Symbolic::rectangular)
int numeric_rectangular(Symbolic *S, const int *Ap, const int *Ai, const std::vector<char> &nonzero, void **NumericOut,
                        const double *re = nullptr, const double *im = nullptr, int vstride = 1) {
  if (Ap[S->n_col] != S->nnz || !std::equal(S->Ap.begin(), S->Ap.end(), Ap) || hash_indices(Ai, S->nnz) != S->ai_hash)
    return UMFPACK_ERROR_different_pattern;
  try {
    std::unique_ptr<Numeric> N(new Numeric());
    N->rectangular = 1;
    N->n = 0;
    if (hipGetDevice(&N->device) != hipSuccess) {
      (void)hipGetLastError();
      N->device = 0;
    }
    int rank = structural_rank(S->n_row, S->n_col, Ap, Ai, nonzero);
    if (re && rank == std::min(S->n_row, S->n_col)) {  // structurally regular: small matrices are eliminated for real
      const int numeric = dense_numeric_rank(S->n_row, S->n_col, Ap, Ai, re, im, vstride);
      if (numeric >= 0) rank = numeric;
    }
    N->singular = rank < std::min(S->n_row, S->n_col) ? 1 : 0;
    const int st = N->singular ? UMFPACK_WARNING_singular_matrix : UMFPACK_OK;
    *NumericOut = N.release();
    return st;
  } catch (const std::bad_alloc &) {
    return UMFPACK_ERROR_out_of_memory;
  } catch (...) {
    return UMFPACK_ERROR_internal_error;
  }
}
bool symbolic_is_rectangular(void *SymbolicIn) {
  Symbolic *S = as_symbolic(SymbolicIn);
  return S && S->rectangular;
}
int numeric_rectangular_of(void *SymbolicIn, const int *Ap, const int *Ai, const std::vector<char> &nonzero, void **NumericOut,
                           const double *re, const double *im, int vstride) {
  Symbolic *S = as_symbolic(SymbolicIn);
  if (!S || !S->rectangular) return UMFPACK_ERROR_invalid_Symbolic_object;
  return numeric_rectangular(S, Ap, Ai, nonzero, NumericOut, re, im, vstride);
}
bool numeric_is_rectangular(void *NumericIn) {
  Numeric *N = as_numeric(NumericIn);
  return N && N->rectangular;
}
}  // namespace spl

extern "C" {

int umfpack_di_symbolic(int n_row, int n_col, const int Ap[], const int Ai[], const double Ax[],
                        void **SymbolicOut, const double Control[], double Info[]) {
  (void)Ax; (void)Control; (void)Info;
  if (!SymbolicOut) return UMFPACK_ERROR_argument_missing;
  *SymbolicOut = nullptr;
  if (!Ap || (!Ai && n_col > 0 && Ap[n_col] > 0)) return UMFPACK_ERROR_argument_missing;
  if (n_row <= 0 || n_col <= 0) return UMFPACK_ERROR_n_nonpositive;
  if (Ap[n_col] < 0) return UMFPACK_ERROR_invalid_matrix;
  int st = validate_host_csc(n_row, n_col, Ap, Ai);
  if (st != UMFPACK_OK) return st;
  if (n_row != n_col) return symbolic_rectangular(n_row, n_col, Ap, Ai, SymbolicOut);
  return symbolic_common(n_col, Ap, Ai, 1, Ap, Ai, SymbolicOut);
}

}  // extern "C"

namespace spl {
uint64_t pattern_hash(const int *Ai, int64_t nnz) { return hash_indices(Ai, nnz); }

// numeric factorisation of an embedding whose pattern the caller has already checked against its own record
// native: 0 the embedding has swapped pairs or scaled blocks (real fronts only); 1 it is the plain embedding of a complex
// matrix (native complex fronts may serve it); 2 and that matrix is symmetric
int numeric_of_embedding(const int *Ep, const int *Ei, const double *Ex, void *Symbolic, void **Numeric, int native) {
  struct Vouch {
    explicit Vouch(int native) { t_pattern_vouched = true; t_native_complex = native; }
    ~Vouch() { t_pattern_vouched = false; t_native_complex = 0; }
  } vouch(native);
  return umfpack_di_numeric(Ep, Ei, Ex, Symbolic, Numeric, nullptr, nullptr);
}
// Native complex fronts serve this analysis?  They need the tree (of the complex pattern), and pay where the tree has
// flops to halve: below ~1e12 flops of the embedding's tree (2-D meshes up to 10^6 unknowns) a factorisation is
// launch-bound either way and a batch of right-hand sides takes twice the passes over the tree (four complex columns
// per pass against eight real ones).  SPL_ZI_NATIVE=1 / 0: always / never.
bool symbolic_has_complex_tree(void *SymbolicIn) {
  Symbolic *S = as_symbolic(SymbolicIn);
  if (!S || !S->tree || !S->ztree) return false;
  const char *zn = getenv("SPL_ZI_NATIVE");
  if (zn && zn[0] == '0') return false;
  if (zn && zn[0] == '1') return true;
  return S->tree->flops >= 1e12;
}
// analysis of the real embedding of an n x n complex matrix from the complex pattern itself (umfpack_zi.hip);
// (Ep, Ei): the pattern of the embedding, 2n x 2n, interleaved unknowns
int symbolic_of_embedding(int n, const int *Ap, const int *Ai, const int *Ep, const int *Ei, void **SymbolicOut) {
  return symbolic_common(n, Ap, Ai, 2, Ep, Ei, SymbolicOut);
}
double symbolic_tree_flops(void *SymbolicIn) {
  Symbolic *S = as_symbolic(SymbolicIn);
  return S && S->tree ? S->tree->flops : 0.0;
}
}  // namespace spl

extern "C" {

int umfpack_di_numeric(const int Ap[], const int Ai[], const double Ax[], void *SymbolicIn,
                       void **NumericOut, const double Control[], double Info[]) {
  (void)Control; (void)Info;
  if (!NumericOut) return UMFPACK_ERROR_argument_missing;
  *NumericOut = nullptr;
  Symbolic *S = as_symbolic(SymbolicIn);
  if (!S) return UMFPACK_ERROR_invalid_Symbolic_object;
  if (!Ap || !Ai || !Ax) return UMFPACK_ERROR_argument_missing;
  if (S->rectangular) {
    if (Ap[S->n_col] != S->nnz) return UMFPACK_ERROR_different_pattern;
    std::vector<char> nonzero((size_t)S->nnz);
    for (int p = 0; p < S->nnz; ++p) nonzero[(size_t)p] = (Ax[p] != 0.0 && Ax[p] == Ax[p]) ? 1 : 0;  // (NaN: no pivot)
    return numeric_rectangular(S, Ap, Ai, nonzero, NumericOut, Ax, nullptr, 1);
  }
  const int n = S->n;
  // (the `zi` wrapper has compared the complex pattern — a quarter of the embedding's — and vouches for the rest)
  if (!t_pattern_vouched &&
      (Ap[n] != S->nnz || !std::equal(S->Ap.begin(), S->Ap.end(), Ap) || hash_indices(Ai, S->nnz) != S->ai_hash))
    return UMFPACK_ERROR_different_pattern;
  Numeric *N = nullptr;
  try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      set_last_error_text("no HIP device visible");
      return UMFPACK_ERROR_internal_error;
    }
    N = new Numeric();
    SPL_HIP(hipGetDevice(&N->device));
    // the calling thread's default stream: ordered after and before work on the legacy default stream (the caller's
    // torch kernels, this library's other entry points) like the legacy stream itself, but factorisations and solves
    // issued by different host threads — the contour points of a FEAST iteration — overlap on the device
    hipStream_t s = hipStreamPerThread;
    const bool timing = getenv("SPL_MF_TIMING") != nullptr;  // phase times on stderr (diagnostic)
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!timing) return;
      (void)hipStreamSynchronize(s);
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[numeric] %-26s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
      t_last = now;
    };
    N->n = n;
    N->kl = S->kl;
    N->ku = S->ku;
    // rows of A^T == the CSC arrays as they are: needed first, for the dominance test
    void *hAt = nullptr;
    {
      int stc = spl_matrix_create_csr(n, n, 0, n, Ap, Ai, Ax, &hAt);
      if (stc != SPL_OK) {
        delete N;
        return stc == SPL_ERROR_out_of_memory ? UMFPACK_ERROR_out_of_memory
               : stc == SPL_ERROR_invalid_matrix ? UMFPACK_ERROR_invalid_matrix : UMFPACK_ERROR_internal_error;
      }
    }
    N->At = static_cast<Matrix *>(hAt);
    if (!N->At->rowptr.get()) { delete N; return UMFPACK_ERROR_out_of_memory; }
    lap("rows of A^T (upload)");
    N->ipiv.alloc((size_t)n);
    N->perm.alloc((size_t)n);
    N->inv.alloc((size_t)n);
    N->band_perm = S->perm;  // (empty when the analysis skipped the band ordering: ensure_band_ordering)
    N->band_inv = S->inv;
    N->tree = S->tree;
    {
      const char *zn = getenv("SPL_ZI_NATIVE");  // 0: the real fronts of the embedding also for complex matrices
      if (t_native_complex && S->tree && S->ztree && !(zn && zn[0] == '0')) {
        N->ztree = S->ztree;
        N->zsym = t_native_complex == 2 ? 1 : 0;
      }
    }
    set_ordering(N, N->tree ? N->tree->perm : N->band_perm, N->tree ? N->tree->inv : N->band_inv, s);
    // device copies of A for the residuals of the refinement: rows of A (transposed on the
    // device) and rows of A^T (the CSC arrays as they are)
    // (by transposing the copy just uploaded, on the device: spl_matrix_create would send the same arrays over PCIe
    // a second time and transpose them with the same kernel — 0.7 GB at config C5, a fifth of a FEAST refactorisation
    // of 10^6 complex unknowns)
    void *hA = nullptr;
    int st = spl_matrix_transpose(hAt, &hA);
    if (st != SPL_OK) {
      spl_matrix_free(&hA);
      delete N;
      return st == SPL_ERROR_out_of_memory ? UMFPACK_ERROR_out_of_memory
             : st == SPL_ERROR_invalid_matrix ? UMFPACK_ERROR_invalid_matrix : UMFPACK_ERROR_internal_error;
    }
    N->A = static_cast<Matrix *>(hA);
    lap("ordering, rows of A");
    // Path: SPL_LU_FORCE_PIVOT=1 -> partial pivoting; =0 -> only provably safe no-interchange
    // factors (column diagonal dominance); default -> no-interchange factors for every matrix,
    // as a speculation when dominance does not hold (checked by every solve, see Numeric).
    const char *force = getenv("SPL_LU_FORCE_PIVOT");
    const bool force_pivot = force && force[0] == '1', no_speculation = force && force[0] == '0';
    bool dominant = false;
    if (!force_pivot)
      dominant = band_is_column_dominant(n, N->At->rowptr.get(), N->At->colidx.get(), N->At->val.get(), s);
    N->dominant = dominant ? 1 : 0;
    if (!force_pivot && (dominant || !no_speculation)) {
      bool fits = true;
      try {
        if (!dominant && getenv("SPL_LU_TEST_SPECULATION_OOM")) throw DeviceError{SPL_ERROR_out_of_memory};  // tests: as if it did not fit
        // the analysis left a region without separators as one giant leaf (Tree::gave_up): that front is not for
        // factoring — straight to static pivoting, whose transversal gives the matrix its mesh pattern back
        if (!dominant && N->tree && N->tree->gave_up) throw DeviceError{SPL_ERROR_out_of_memory};
        if (N->tree) factor_multifrontal(N, s); else factor_band(N, true, s);
      } catch (const DeviceError &e) {
        // The speculation's own ordering does not fit the device: the pattern of A + A^T has no separators — e.g. a
        // mesh matrix whose rows arrive in random order (tools/fuzz_lu_scale.py, family perm2d: 195 364 unknowns
        // returned UMFPACK_ERROR_out_of_memory where SuperLU solved).  The transversal of static pivoting puts the
        // large entries back on the diagonal and B = Dr P A Dc is ordered on its own pattern.
        if (dominant || e.status != SPL_ERROR_out_of_memory) throw;
        fits = false;
      }
      N->speculative = dominant ? 0 : 1;
      if (!fits || (N->speculative && N->singular)) {  // a zero pivot without interchanges proves nothing
        if (!factor_static_pivot(N, Ap, Ai, Ax, s)) {
          N->speculative = 0;
          factor_band(N, false, s);
        }
      }
    } else {
      factor_band(N, false, s);
    }
    lap("factorisation");
    *NumericOut = N;
    return N->singular ? UMFPACK_WARNING_singular_matrix : UMFPACK_OK;
  } catch (const DeviceError &e) {
    delete N;
    return e.status == SPL_ERROR_out_of_memory ? UMFPACK_ERROR_out_of_memory : UMFPACK_ERROR_internal_error;
  } catch (const std::bad_alloc &) {
    delete N;
    return UMFPACK_ERROR_out_of_memory;
  } catch (...) {
    delete N;
    return UMFPACK_ERROR_internal_error;
  }
}

// k systems op(A) X(:,c) = B(:,c) with the factors of N; X, B are n x k column-major on the host.
// Iterative refinement with UMFPACK's defaults and stopping rules, column by column (irstep = 2;
// umf_solve): stop when the componentwise backward error is below machine epsilon, or when a
// step does not at least halve it (a step that raises it is undone).  All columns share every
// pass over the factors.
// Static pivoting from a solve: the matrix is taken from the object's own device copy (the columns of A are the
// rows of N->At), not from the caller's Ap / Ai / Ax — a `zi` caller passes the complex matrix there, whose arrays
// are not those of the embedding this object factors (umfpack_zi.hip).
static bool factor_static_pivot_of_copy(Numeric *N, hipStream_t s) {
  if (N->sp_stage != 0 || !N->At || !N->At->rowptr.get() || N->At->nnz <= 0) return false;
  const size_t n = (size_t)N->n, nnz = (size_t)N->At->nnz;
  std::vector<int> p(n + 1), i(nnz);
  std::vector<double> x(nnz);
  SPL_HIP(hipMemcpyAsync(p.data(), N->At->rowptr.get(), (n + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipMemcpyAsync(i.data(), N->At->colidx.get(), nnz * sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipMemcpyAsync(x.data(), N->At->val.get(), nnz * sizeof(double), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  return factor_static_pivot(N, p.data(), i.data(), x.data(), s);
}

// Flexible GMRES on op x = b, right-preconditioned by the factors held (the solves `factor_solve` runs), from the
// iterate in x: what static pivoting needs when plain refinement stalls.  The factors of B = Dr P A Dc without
// interchanges are exact for a nearby matrix; on matrices with values over many orders of magnitude that matrix is
// not near enough for refinement — a stationary iteration — to reach rounding level, but as a preconditioner it
// leaves A M^-1 with a few outlying eigenvalues, which a Krylov method removes in as many steps.  One cycle of at most
// `m` steps (modified Gram-Schmidt; the small least-squares problem by Givens rotations on the host).  x is updated
// in place; the caller measures the backward error afterwards.
static void gmres_polish(Numeric *N, int sys, const Matrix *op, const double *b, double *x, double *work, int m,
                         hipStream_t s) {
  const size_t n = (size_t)N->n;
  const unsigned gv = (unsigned)((n + 255) / 256), gd = gv < 1024 ? gv : 1024;
  DBuf<double> V((size_t)(m + 1) * n), Z((size_t)m * n), w(n), dsc(1 + (size_t)gd);
  auto dot = [&](const double *a, const double *c) {
    double h = 0.0;
    hipLaunchKernelGGL(dot_part_kernel, dim3(gd), dim3(256), 0, s, n, a, c, dsc.get() + 1);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, s, (int)gd, dsc.get() + 1, dsc.get());
    SPL_HIP(hipMemcpyAsync(&h, dsc.get(), sizeof(double), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    return h;
  };
  auto axpby = [&](double alpha, const double *xx, double beta, double *yy) {
    hipLaunchKernelGGL(axpby_kernel, dim3(gv), dim3(256), 0, s, n, alpha, xx, beta, yy);
  };
  // r = b - op x
  int st = launch_spmv(op, x, w.get(), 0, s);
  if (st != SPL_OK) throw DeviceError{st};
  axpby(1.0, b, -1.0, w.get());
  const double beta = std::sqrt(dot(w.get(), w.get()));
  if (!(beta > 0.0)) return;
  axpby(1.0 / beta, w.get(), 0.0, V.get());
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs((size_t)m, 0.0), sn((size_t)m, 0.0), g((size_t)m + 1, 0.0);
  g[0] = beta;
  int steps = 0;
  for (int j = 0; j < m; ++j) {
    double *zj = Z.get() + (size_t)j * n, *vj = V.get() + (size_t)j * n, *vn = V.get() + (size_t)(j + 1) * n;
    factor_solve(N, sys, vj, zj, work, 1, n, s);  // z_j = M^-1 v_j
    st = launch_spmv(op, zj, vn, 0, s);           // w = op z_j
    if (st != SPL_OK) throw DeviceError{st};
    for (int i = 0; i <= j; ++i) {
      const double h = dot(vn, V.get() + (size_t)i * n);
      H[(size_t)i * m + j] = h;
      axpby(-h, V.get() + (size_t)i * n, 1.0, vn);
    }
    const double hn = std::sqrt(dot(vn, vn));
    H[(size_t)(j + 1) * m + j] = hn;
    for (int i = 0; i < j; ++i) {  // earlier rotations on the new column
      const double t = cs[(size_t)i] * H[(size_t)i * m + j] + sn[(size_t)i] * H[(size_t)(i + 1) * m + j];
      H[(size_t)(i + 1) * m + j] = -sn[(size_t)i] * H[(size_t)i * m + j] + cs[(size_t)i] * H[(size_t)(i + 1) * m + j];
      H[(size_t)i * m + j] = t;
    }
    const double a = H[(size_t)j * m + j], c2 = H[(size_t)(j + 1) * m + j], rr = std::hypot(a, c2);
    steps = j + 1;
    if (!(rr > 0.0)) break;
    cs[(size_t)j] = a / rr;
    sn[(size_t)j] = c2 / rr;
    H[(size_t)j * m + j] = rr;
    H[(size_t)(j + 1) * m + j] = 0.0;
    g[(size_t)j + 1] = -sn[(size_t)j] * g[(size_t)j];
    g[(size_t)j] = cs[(size_t)j] * g[(size_t)j];
    if (std::fabs(g[(size_t)j + 1]) <= 1e-16 * beta || !(hn > 0.0)) break;  // converged, or the Krylov space is exhausted
    axpby(1.0 / hn, vn, 0.0, vn);
  }
  // y from the triangular system, x += Z y
  std::vector<double> y((size_t)steps, 0.0);
  for (int i = steps - 1; i >= 0; --i) {
    double t = g[(size_t)i];
    for (int l = i + 1; l < steps; ++l) t -= H[(size_t)i * m + l] * y[(size_t)l];
    const double d = H[(size_t)i * m + i];
    y[(size_t)i] = d != 0.0 ? t / d : 0.0;
  }
  for (int i = 0; i < steps; ++i) axpby(y[(size_t)i], Z.get() + (size_t)i * n, 1.0, x);
  SPL_HIP(hipStreamSynchronize(s));
}

// device_io: X and B are device pointers (spl_umfpack_*_solve_many_dev), else host
static int solve_columns(Numeric *N, int sys, int k, double *X, const double *B, const int *Ap, const int *Ai,
                         const double *Ax, bool device_io = false, double *Info = nullptr) {
  const int n = N->n;
  if (N->broken) return UMFPACK_ERROR_invalid_Numeric_object;  // a failed refactorisation left no factors
  try {
    DeviceGuard g(N->device);
    // the calling thread's default stream: ordered after and before work on the legacy default stream (the caller's
    // torch kernels, this library's other entry points) like the legacy stream itself, but factorisations and solves
    // issued by different host threads — the contour points of a FEAST iteration — overlap on the device
    hipStream_t s = hipStreamPerThread;
    if (n == 0 || k == 0) return N->singular ? UMFPACK_WARNING_singular_matrix : UMFPACK_OK;
    const bool timing = getenv("SPL_MF_TIMING") != nullptr;  // phase times on stderr (diagnostic)
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!timing) return;
      (void)hipStreamSynchronize(s);
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[solve] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
      t_last = now;
    };
    const size_t stride = (size_t)n;
    const int kalloc = k == 1 ? 1 : (k + kSolveGroup - 1) / kSolveGroup * kSolveGroup;
    const size_t total = stride * (size_t)kalloc, used = stride * (size_t)k;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)k);
    // one allocation for all work vectors of the call (each hipMalloc / hipFree is a device round trip)
    struct Span {
      double *p;
      double *get() const { return p; }
    };
    DBuf<double> slab(9 * total + (size_t)k);
    Span db{slab.get()}, dx{slab.get() + total}, dwork{slab.get() + 2 * total}, dr{slab.get() + 3 * total},
        dd{slab.get() + 4 * total}, dax{slab.get() + 5 * total}, dabs{slab.get() + 6 * total},
        dxn{slab.get() + 7 * total}, drn{slab.get() + 8 * total}, domega{slab.get() + 9 * total};
    for (const Span *buf : {&db, &dx, &dwork, &dr, &dd, &dxn, &drn})
      if (kalloc > k) SPL_HIP(hipMemsetAsync(buf->get() + used, 0, (total - used) * sizeof(double), s));
    SPL_HIP(hipMemcpyAsync(db.get(), B, used * sizeof(double),
                           device_io ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    // speculative factors may be replaced below: solves on such an object take turns
    std::unique_lock<std::mutex> turn(N->mu, std::defer_lock);
    if (N->speculative) turn.lock();
    std::vector<double> omega((size_t)k, 0.0), on((size_t)k, 0.0);
    bool polished = false;
    int walks = 0, ir_taken = 0, ir_attempted = 0;
    double delivered = 0.0;
    lap("buffers, upload of b");
  again:
    factor_solve(N, sys, db.get(), dx.get(), dwork.get(), k, stride, s);
    ++walks;
    lap("solve with the factors");
    if (!N->singular) {
      const Matrix *op = sys == UMFPACK_A ? N->A : N->At;
      const char *rs = getenv("SPL_LU_RESIDUAL");
      const bool plain_residual = op->vw != 1 || (rs && rs[0] == 'p');
      auto backward_error = [&](const double *x, double *r, std::vector<double> &out) {
        if (!plain_residual) {
          SPL_HIP(hipMemsetAsync(domega.get(), 0, (size_t)k * sizeof(double), s));
          if (op->nnz <= 12 * (int64_t)n)  // short rows
            hipLaunchKernelGGL(residual_dd_kernel<2>, dim3((unsigned)(((size_t)n * 2 + 255) / 256), (unsigned)k), dim3(256), 0, s,
                               n, op->rowptr64.get(), op->colidx.get(), op->val.get(), x, db.get(), r, domega.get(), stride);
          else
            hipLaunchKernelGGL(residual_dd_kernel<8>, dim3((unsigned)(((size_t)n * 8 + 255) / 256), (unsigned)k), dim3(256), 0, s,
                               n, op->rowptr64.get(), op->colidx.get(), op->val.get(), x, db.get(), r, domega.get(), stride);
          SPL_HIP(hipMemcpyAsync(out.data(), domega.get(), (size_t)k * sizeof(double), hipMemcpyDeviceToHost, s));
          SPL_HIP(hipStreamSynchronize(s));
          return;
        }
        for (int c = 0; c < k; ++c) {
          int st = launch_spmv(op, x + (size_t)c * stride, dax.get() + (size_t)c * stride, 0, s);
          if (st != SPL_OK) throw DeviceError{st};
        }
        hipLaunchKernelGGL(abs_spmv_kernel, dim3((unsigned)(((size_t)n * 8 + 255) / 256), (unsigned)k), dim3(256), 0,
                           s, n, op->rowptr64.get(), op->colidx.get(), op->val.get(), x, dabs.get(), stride);
        SPL_HIP(hipMemsetAsync(domega.get(), 0, (size_t)k * sizeof(double), s));
        hipLaunchKernelGGL(residual_kernel, grid, dim3(256), 0, s, n, db.get(), dax.get(), dabs.get(), r,
                           domega.get(), stride);
        SPL_HIP(hipMemcpyAsync(out.data(), domega.get(), (size_t)k * sizeof(double), hipMemcpyDeviceToHost, s));
        SPL_HIP(hipStreamSynchronize(s));
      };
      const double eps = 2.220446049250313e-16;
      backward_error(dx.get(), dr.get(), omega);
      lap("residual, backward error");
      std::vector<char> active((size_t)k);
      int nactive = 0;
      for (int c = 0; c < k; ++c) nactive += (active[(size_t)c] = omega[(size_t)c] >= eps);
      // UMFPACK's default is two steps.  Factors kept as a speculation (no interchanges, no
      // diagonal dominance) may take more while each step still halves the backward error: static
      // pivoting with refinement, before the factors are given up for pivoted ones below.
      const int max_steps = N->speculative ? 10 : 2;
      const double stall = 0.5;
      for (int it = 0; it < max_steps && nactive > 0; ++it) {
        SPL_HIP(hipMemcpyAsync(dxn.get(), dx.get(), used * sizeof(double), hipMemcpyDeviceToDevice, s));
        ++walks;
        ++ir_attempted;
        if (nactive == k) {
          factor_solve(N, sys, dr.get(), dd.get(), dwork.get(), k, stride, s);
          hipLaunchKernelGGL(add_kernel, dim3((unsigned)((used + 255) / 256)), dim3(256), 0, s, used, dxn.get(),
                             dd.get());
        } else {
          // Only the columns still being refined go through the factors, side by side: a walk over the tree (or the
          // band) costs the same for a group of kSolveGroup columns whatever they hold, and the columns of a batch
          // rarely all need the second step (drn is free until the backward error below).
          int ka = 0;
          for (int c = 0; c < k; ++c)
            if (active[(size_t)c])
              SPL_HIP(hipMemcpyAsync(drn.get() + (size_t)ka++ * stride, dr.get() + (size_t)c * stride,
                                     stride * sizeof(double), hipMemcpyDeviceToDevice, s));
          const int ka_alloc = ka == 1 ? 1 : (ka + kSolveGroup - 1) / kSolveGroup * kSolveGroup;
          if (ka_alloc > ka) {
            SPL_HIP(hipMemsetAsync(drn.get() + (size_t)ka * stride, 0, (size_t)(ka_alloc - ka) * stride * sizeof(double), s));
            SPL_HIP(hipMemsetAsync(dwork.get() + (size_t)ka * stride, 0, (size_t)(ka_alloc - ka) * stride * sizeof(double), s));
          }
          factor_solve(N, sys, drn.get(), dd.get(), dwork.get(), ka, stride, s);
          ka = 0;
          for (int c = 0; c < k; ++c)
            if (active[(size_t)c])
              hipLaunchKernelGGL(add_kernel, dim3((unsigned)((stride + 255) / 256)), dim3(256), 0, s, stride,
                                 dxn.get() + (size_t)c * stride, dd.get() + (size_t)ka++ * stride);
        }
        backward_error(dxn.get(), drn.get(), on);
        if (timing) {
          double lo = 1e300, hi = 0.0, lo2 = 1e300, hi2 = 0.0;
          for (int c = 0; c < k; ++c)
            if (active[(size_t)c]) {
              lo = std::min(lo, omega[(size_t)c]), hi = std::max(hi, omega[(size_t)c]);
              lo2 = std::min(lo2, on[(size_t)c]), hi2 = std::max(hi2, on[(size_t)c]);
            }
          fprintf(stderr, "[solve] columns refined in this step: %d of %d, backward errors %.2e .. %.2e -> %.2e .. %.2e\n",
                  nactive, k, lo, hi, lo2, hi2);
        }
        lap("refinement step");
        bool kept = false;
        for (int c = 0; c < k; ++c) kept |= active[(size_t)c] && on[(size_t)c] <= omega[(size_t)c];
        ir_taken += kept ? 1 : 0;
        for (int c = 0; c < k; ++c) {
          if (!active[(size_t)c]) continue;
          const double o_new = on[(size_t)c], o_old = omega[(size_t)c];
          if (!(o_new <= o_old)) {  // worse (or NaN): keep the previous iterate of this column
            active[(size_t)c] = 0;
            --nactive;
            continue;
          }
          const size_t off = (size_t)c * stride;
          SPL_HIP(hipMemcpyAsync(dx.get() + off, dxn.get() + off, stride * sizeof(double), hipMemcpyDeviceToDevice, s));
          SPL_HIP(hipMemcpyAsync(dr.get() + off, drn.get() + off, stride * sizeof(double), hipMemcpyDeviceToDevice, s));
          omega[(size_t)c] = o_new;
          // UMFPACK's rule: a step that does not halve the backward error ends the refinement (letting the factors of
          // static pivoting go on while a step still gained a tenth changed nothing: their refinement stops because a
          // step makes things WORSE, at 1e-10 .. 5e-10 on matrices with values over six orders of magnitude — what
          // gets them to rounding level is the Krylov polish below)
          if (o_new > o_old * stall || o_new < eps) {  // stagnated, or converged
            active[(size_t)c] = 0;
            --nactive;
          }
        }
        // columns that are done must not move any more: their residual no longer drives a step
        for (int c = 0; c < k; ++c)
          if (!active[(size_t)c])
            SPL_HIP(hipMemsetAsync(dr.get() + (size_t)c * stride, 0, stride * sizeof(double), s));
      }
      // no-interchange factors of a matrix without diagonal dominance: accepted only if every
      // refined solution is backward stable to rounding level, as pivoted factors would make it
      double worst = 0.0;
      for (int c = 0; c < k; ++c) worst = (omega[(size_t)c] <= worst) ? worst : omega[(size_t)c];  // NaN -> worst
      delivered = worst;
      if (turn.owns_lock() && N->speculative && !(worst <= 1e-13)) {
        // first the static-pivoting stage (stays on the tree, still checked by this very loop), then, if that
        // fails too, the band factorisation with partial pivoting
        // a symmetric matrix whose L D L^T speculation failed: the same tree once more as LU with threshold pivoting
        // inside the diagonal blocks (no host work), then the static-pivoting stage
        if (N->mfact && N->mf_sym && !N->mf_piv && !N->block_pivot_retry && N->sp_stage == 0 && N->tree) {
          const char *bp = getenv("SPL_LU_BLOCK_PIVOT");
          if (!(bp && bp[0] == '0')) {
            N->block_pivot_retry = 1;
            try {
              factor_multifrontal(N, s);
            } catch (const DeviceError &) {
              N->broken = 1;
              throw;
            }
            if (!N->singular) goto again;
          }
        }
        if (factor_static_pivot_of_copy(N, s)) goto again;
        if (N->sp_stage == 1 && !polished) {
          // the factors held ARE those of static pivoting and refinement has stalled: a Krylov polish preconditioned
          // by them (gmres_polish), column by column, then the same test again
          polished = true;
          const char *pe = getenv("SPL_LU_GMRES");
          if (!(pe && pe[0] == '0')) {
            for (int c = 0; c < k; ++c)
              if (!(omega[(size_t)c] <= 1e-13))
                for (int cycle = 0; cycle < 3; ++cycle) {
                  // the polish writes the column in place: keep the iterate it started from, and put it back when the
                  // cycle made the backward error worse — omega[c] is then the error of what is delivered
                  const size_t off = (size_t)c * stride;
                  SPL_HIP(hipMemcpyAsync(dxn.get() + off, dx.get() + off, stride * sizeof(double), hipMemcpyDeviceToDevice, s));
                  gmres_polish(N, sys, op, db.get() + off, dx.get() + off, dwork.get(), 20, s);
                  backward_error(dx.get(), dr.get(), on);
                  const bool better = on[(size_t)c] < 0.5 * omega[(size_t)c];
                  if (on[(size_t)c] <= omega[(size_t)c]) omega[(size_t)c] = on[(size_t)c];
                  else SPL_HIP(hipMemcpyAsync(dx.get() + off, dxn.get() + off, stride * sizeof(double), hipMemcpyDeviceToDevice, s));
                  if (timing) fprintf(stderr, "[solve] FGMRES cycle %d, column %d: backward error %.2e\n", cycle, c, on[(size_t)c]);
                  if (omega[(size_t)c] <= 1e-13 || !better) break;
                }
            worst = 0.0;
            for (int c = 0; c < k; ++c) worst = (omega[(size_t)c] <= worst) ? worst : omega[(size_t)c];
            delivered = worst;
            lap("FGMRES polish");
            if (worst <= 1e-13) goto deliver;
          }
        }
        try {
          factor_band(N, false, s);
        } catch (const DeviceError &e) {
          // The band with partial pivoting does not fit (factor_band decides that before it releases anything: the
          // factors held are intact).  Nothing better can be built on this device; the refined solution at hand
          // is returned when its componentwise backward error is at the level threshold pivoting itself reaches on
          // such matrices (tools/fuzz_lu_scale.py, family perm2d: static pivoting + refinement 1e-12 .. 1e-10,
          // SuperLU 3e-11), as UMFPACK returns whatever its pivoting achieved; anything worse stays an error.
          if (e.status != SPL_ERROR_out_of_memory || !(worst <= 1e-9)) throw;
          if (getenv("SPL_MF_TIMING")) fprintf(stderr, "[solve] pivoted band does not fit: keeping the factors held, backward error %.2e\n", worst);
          goto deliver;
        }
        N->speculative = 0;  // only now may other threads use the object without taking turns
        goto again;
      }
    }
  deliver:
    SPL_HIP(hipMemcpyAsync(X, dx.get(), used * sizeof(double),
                           device_io ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    SPL_HIP(hipGetLastError());
    lap("download of x");
    N->last_walks = walks;
    N->last_ir_taken = ir_taken;
    N->last_ir_attempted = ir_attempted;
    N->last_omega = delivered;
    if (Info) {  // what umfpack_*_solve reports there (umfpack.h: UMFPACK_IR_TAKEN 80, _IR_ATTEMPTED 81, _OMEGA1 82, _OMEGA2 83)
      Info[0] = N->singular ? UMFPACK_WARNING_singular_matrix : UMFPACK_OK;
      Info[80] = ir_taken;
      Info[81] = ir_attempted;
      Info[82] = delivered;  // one quantity here: max_i |r_i| / (|A||x| + |b|)_i over all rows (omega2's rows do not occur:
      Info[83] = 0.0;        // the denominator of a row with entries never vanishes unless its whole numerator does)
    }
    return N->singular ? UMFPACK_WARNING_singular_matrix : UMFPACK_OK;
  } catch (const DeviceError &e) {
    return e.status == SPL_ERROR_out_of_memory ? UMFPACK_ERROR_out_of_memory : UMFPACK_ERROR_internal_error;
  } catch (const std::bad_alloc &) {
    return UMFPACK_ERROR_out_of_memory;
  } catch (...) {
    return UMFPACK_ERROR_internal_error;
  }
}

int umfpack_di_solve(int sys, const int Ap[], const int Ai[], const double Ax[], double X[],
                     const double B[], void *NumericIn, const double Control[], double Info[]) {
  (void)Control;
  auto status = [&]() -> int {
    Numeric *N = as_numeric(NumericIn);
    if (!N) return UMFPACK_ERROR_invalid_Numeric_object;
    if (N->rectangular) return UMFPACK_ERROR_invalid_system;  // "the matrix is not square", as UMFPACK's solve
    if (!X || !B) return UMFPACK_ERROR_argument_missing;
    if (!Ap || !Ai || !Ax) return UMFPACK_ERROR_argument_missing;  // UMFPACK needs A for refinement
    if (sys != UMFPACK_A && sys != UMFPACK_At) return UMFPACK_ERROR_invalid_system;
    return solve_columns(N, sys, 1, X, B, Ap, Ai, Ax, false, Info);
  };
  const int st = status();
  if (Info && st < 0) Info[0] = st;  // Info[UMFPACK_STATUS] on the error returns as well
  return st;
}

// batched linearSolve: nrhs right-hand sides in one call (see umfpack_hip.h)
int spl_umfpack_di_solve_many(int sys, const int Ap[], const int Ai[], const double Ax[], int nrhs, double X[],
                              const double B[], void *NumericIn) {
  Numeric *N = as_numeric(NumericIn);
  if (!N) return UMFPACK_ERROR_invalid_Numeric_object;
  if (N->rectangular) return UMFPACK_ERROR_invalid_system;
  if (nrhs < 0) return UMFPACK_ERROR_argument_missing;
  if (nrhs > 0 && N->n > 0 && (!X || !B)) return UMFPACK_ERROR_argument_missing;
  if (!Ap || !Ai || !Ax) return UMFPACK_ERROR_argument_missing;
  if (sys != UMFPACK_A && sys != UMFPACK_At) return UMFPACK_ERROR_invalid_system;
  return solve_columns(N, sys, nrhs, X, B, Ap, Ai, Ax);
}

// the same with X and B in device memory (see umfpack_hip.h)
int spl_umfpack_di_solve_many_dev(int sys, const int Ap[], const int Ai[], const double Ax[], int nrhs, double *d_X,
                                  const double *d_B, void *NumericIn) {
  Numeric *N = as_numeric(NumericIn);
  if (!N) return UMFPACK_ERROR_invalid_Numeric_object;
  if (N->rectangular) return UMFPACK_ERROR_invalid_system;
  if (nrhs < 0) return UMFPACK_ERROR_argument_missing;
  if (nrhs > 0 && N->n > 0 && (!d_X || !d_B)) return UMFPACK_ERROR_argument_missing;
  if (sys != UMFPACK_A && sys != UMFPACK_At) return UMFPACK_ERROR_invalid_system;
  return solve_columns(N, sys, nrhs, d_X, d_B, Ap, Ai, Ax, true);
}

}  // extern "C"

namespace spl {
void numeric_set_pair_swap(void *NumericIn, std::vector<char> &&flags) {
  if (Numeric *N = as_numeric(NumericIn)) N->pair_swap = std::move(flags);
}
const std::vector<char> *numeric_pair_swap(void *NumericIn) {
  Numeric *N = as_numeric(NumericIn);
  return N && !N->pair_swap.empty() ? &N->pair_swap : nullptr;
}
void numeric_set_pair_unit(void *NumericIn, std::vector<double> &&u) {
  if (Numeric *N = as_numeric(NumericIn)) N->pair_unit = std::move(u);
}
const std::vector<double> *numeric_pair_unit(void *NumericIn) {
  Numeric *N = as_numeric(NumericIn);
  return N && !N->pair_unit.empty() ? &N->pair_unit : nullptr;
}
}  // namespace spl

extern "C" {

// dimension of the factored system (used by the `zi` wrappers in split-array mode); 0 if invalid
int spl_umfpack_dimension(void *NumericIn) {
  Numeric *N = as_numeric(NumericIn);
  return N ? N->n : 0;
}

// which factorisation the object holds now: 0 band, partial pivoting; 1 band, no interchanges
// (diagonally dominant matrix); 2 the same as a speculation; 3 / 4 multifrontal, no interchanges
// (dominant / speculation); 5 multifrontal with static pivoting (see umfpack_hip.h); -1 if invalid
int spl_umfpack_path(void *NumericIn) {
  Numeric *N = as_numeric(NumericIn);
  if (!N) return -1;
  if (N->mfact && N->sp_stage == 1) return 5;
  if (N->mfact) return N->speculative ? 4 : 3;
  return N->nopiv ? (N->speculative ? 2 : 1) : 0;
}

int spl_umfpack_stats(void *NumericIn, double out[8]) {
  Numeric *N = as_numeric(NumericIn);
  if (!N || !out) return -1;
  for (int i = 0; i < 8; ++i) out[i] = 0.0;
  out[0] = (double)spl_umfpack_path(NumericIn);
  out[1] = (double)N->n;
  if (N->mfact && N->zfront && N->ztree) {
    out[4] = (double)mf_device_bytes(*N->ztree, 2);
    out[5] = 4.0 * (N->mf_sym ? 0.5 * N->ztree->flops : N->ztree->flops);  // a complex multiply-add is four real ones
    out[6] = (double)N->ztree->nfronts;
    out[7] = 1.0 + (N->mf_piv ? 2.0 : 0.0);  // bit 0: native complex fronts; bit 1: threshold pivoting inside the diagonal blocks
  } else if (N->mfact) {
    out[7] = N->mf_piv ? 2.0 : 0.0;
    out[4] = (double)mf_device_bytes(*N->tree, 1);
    out[5] = N->mf_sym ? 0.5 * N->tree->flops : N->tree->flops;  // L D L^T on the same fronts: about half the flops of LU
    out[6] = (double)N->tree->nfronts;
  } else {
    out[2] = (double)N->kl;
    out[3] = (double)N->ku;
    out[4] = (double)(N->AB.n + N->blkinv.n) * sizeof(double);
    out[5] = 2.0 * (double)N->n * (double)N->kl * (double)N->ku;
  }
  return 0;
}

// the most recent solve on this object and the bytes a walk over its factors reads (see umfpack_hip.h)
int spl_umfpack_solve_report(void *NumericIn, double out[8]) {
  Numeric *N = as_numeric(NumericIn);
  if (!N || !out) return -1;
  for (int i = 0; i < 8; ++i) out[i] = 0.0;
  out[0] = (double)N->last_walks.load();
  out[1] = (double)N->last_ir_taken.load();
  out[2] = (double)N->last_ir_attempted.load();
  out[3] = N->last_omega.load();
  // factor entries a walk touches, each once: the tree's panels — np^2 + 2 np nb per front (L11 and U11 share the
  // pivot block, L21, U12; complex fronts: two planes) —, or the band; 8 bytes each, no index arrays (dense panels);
  // plus the right-hand side in and the solution out
  double entries = 0.0;
  const mf::Tree *T = N->mfact ? ((N->zfront && N->ztree) ? N->ztree.get() : N->tree.get()) : nullptr;
  if (T) {
    for (int f = 0; f < T->nfronts; ++f) {
      const double np = T->np[(size_t)f], nb = T->nb[(size_t)f];
      entries += np * np + 2.0 * np * nb;
    }
    if (N->zfront && N->ztree) entries *= 2.0;
  } else {
    entries = (double)N->n * ((double)N->kl + (double)N->ku + 1.0);
  }
  out[4] = 8.0 * entries + 16.0 * (double)N->n;
  if (N->mfact) mf_chain_info(N->mfact, out + 5);  // chain matrices: bytes, milliseconds to build, pivots per block
  return 0;
}

void umfpack_di_free_symbolic(void **SymbolicIO) {
  if (!SymbolicIO || !*SymbolicIO) return;
  Symbolic *S = as_symbolic(*SymbolicIO);
  *SymbolicIO = nullptr;
  if (!S) return;
  S->magic = 0;
  delete S;
}

void umfpack_di_free_numeric(void **NumericIO) {
  if (!NumericIO || !*NumericIO) return;
  Numeric *N = as_numeric(*NumericIO);
  *NumericIO = nullptr;
  if (!N) return;
  int prev = -1;
  const bool have_prev = hipGetDevice(&prev) == hipSuccess;
  (void)hipSetDevice(N->device);  // finalizers may run on a thread with another current device
  N->magic = 0;
  delete N;
  if (have_prev) (void)hipSetDevice(prev);
}

void umfpack_di_report_status(const double Control[], int status) {
  // default print level (Control == NULL): errors only, like UMFPACK's prl = 1
  int prl = 1;
  if (Control) prl = (int)Control[0];
  if (status == UMFPACK_OK) {
    if (prl >= 2) fprintf(stderr, "UMFPACK (MI355X backend) status: OK\n");
    return;
  }
  if (status > 0 && prl < 2) return;
  const char *msg = "unknown status";
  switch (status) {
    case UMFPACK_WARNING_singular_matrix: msg = "WARNING: matrix is singular"; break;
    case UMFPACK_ERROR_out_of_memory: msg = "ERROR: out of memory (band profile too wide for HBM)"; break;
    case UMFPACK_ERROR_invalid_Numeric_object: msg = "ERROR: Numeric object is invalid"; break;
    case UMFPACK_ERROR_invalid_Symbolic_object: msg = "ERROR: Symbolic object is invalid"; break;
    case UMFPACK_ERROR_argument_missing: msg = "ERROR: required argument(s) missing"; break;
    case UMFPACK_ERROR_n_nonpositive: msg = "ERROR: dimension (n_row or n_col) must be > 0"; break;
    case UMFPACK_ERROR_invalid_matrix: msg = "ERROR: input matrix is invalid"; break;
    case UMFPACK_ERROR_different_pattern: msg = "ERROR: pattern of matrix (Ap and/or Ai) has changed"; break;
    case UMFPACK_ERROR_invalid_system: msg = "ERROR: system argument invalid (square A x = b or A' x = b only)"; break;
    case UMFPACK_ERROR_internal_error: msg = "ERROR: internal error (HIP device failure)"; break;
  }
  if (prl >= 1) fprintf(stderr, "UMFPACK (MI355X backend) status: %s\n", msg);
}

}  // extern "C"

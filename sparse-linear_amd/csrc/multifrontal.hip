// multifrontal.hip — multifrontal LU without row interchanges on the nested-dissection tree of
// mf_symbolic.hpp, and its solves.  Serves umfpack_di_numeric / umfpack_di_solve for matrices whose
// band profile is too expensive (2-D / 3-D meshes): the work drops from O(n * band^2) to the
// O(n^2) (3-D) / O(n^1.5) (2-D) of nested dissection, the storage from n * band to the fronts.
//
// Every tree node owns a dense column-major frontal matrix F = [pivots | boundary]^2 in one big HBM
// allocation.  Numeric factorisation, level by level from the leaves:
//   assemble   : the entries of P A P^T go to the front of their earlier-eliminated index (one kernel
//                for all fronts); the Schur complements of the children are added into their parent
//                ("extend-add", children in a fixed order, so the result is deterministic);
//   factor     : the first np columns/rows of F are eliminated by the blocked fp64-MFMA kernels of
//                dense_lu_kernels.hpp on a dense view (factor_loop with a pivot limit) — large fronts —
//                or by one workgroup per front running the same device code — the many small ones;
//                the trailing nb x nb block is then the front's Schur complement.
// Fronts stay resident: a solve walks the tree up (L, or U^T) and down (U, or L^T) with one
// workgroup per front and a per-front work vector; children hand their boundary part to the parent
// in the same fixed order.  No interchanges: used under the same rule as the band path (diagonal
// dominance, or a speculation that every solve checks — umfpack.hip).
#include <stdio.h>
#include <chrono>
#include <memory>
#include <vector>

#include "dense_lu_kernels.hpp"
#include "mf_symbolic.hpp"

namespace spl {

namespace {

struct DeviceTree {
  DBuf<int> p0, np, nb, ld, parent, front_of, bidx, rel;
  DBuf<int64_t> bptr, foff, ioff, woff, roff;
};

struct TreeView {  // raw pointers for kernels
  const int *p0, *np, *nb, *ld, *parent, *front_of, *bidx, *rel;
  const int64_t *bptr, *foff, *ioff, *woff, *roff;
};

template <typename T>
void upload_vec(DBuf<T> &d, const std::vector<T> &h, hipStream_t s) {
  d.alloc(h.size());
  if (!h.empty()) SPL_HIP(hipMemcpyAsync(d.get(), h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

// local position of global (new) index g in front f: pivots first, then the sorted boundary
__device__ __forceinline__ int local_pos(const TreeView &t, int f, int g) {
  const int p0 = t.p0[f], np = t.np[f];
  if (g < p0 + np) return g - p0;
  const int *b = t.bidx + t.bptr[f];
  int lo = 0, hi = t.nb[f] - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (b[mid] < g) lo = mid + 1; else hi = mid;
  }
  return np + lo;
}

// entries of A (CSC arrays, original numbering) -> fronts; 8 lanes per column
__global__ __launch_bounds__(256) void assemble_kernel(int n, const int *__restrict__ Ap, const int *__restrict__ Ai,
                                                       const double *__restrict__ Ax, const int *__restrict__ inv,
                                                       TreeView t, double *__restrict__ fronts) {
  const int j = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 3), part = threadIdx.x & 7;
  if (j >= n) return;
  const int gj = inv[j];
  for (int p = Ap[j] + part; p < Ap[j + 1]; p += 8) {
    const int gi = inv[Ai[p]];
    const int f = t.front_of[min(gi, gj)];
    const int r = local_pos(t, f, gi), c = local_pos(t, f, gj);
    fronts[t.foff[f] + (int64_t)r + (int64_t)c * t.ld[f]] = Ax[p];
  }
}

// rel[roff[c] + k] = position of the k-th boundary index of front c inside its parent's front
__global__ __launch_bounds__(256) void rel_kernel(int nfronts, TreeView t, int *__restrict__ rel) {
  const int c = blockIdx.x;
  const int p = t.parent[c];
  if (p < 0) return;
  const int *b = t.bidx + t.bptr[c];
  for (int k = threadIdx.x; k < t.nb[c]; k += blockDim.x) rel[t.roff[c] + k] = local_pos(t, p, b[k]);
}

// parent += Schur complement of the listed children; blockIdx.y = child, a workgroup moves a tile of
// 64 rows x 4 columns (lanes run down the rows: contiguous in the child, nearly so in the parent)
__global__ __launch_bounds__(256) void extend_add_kernel(const int *__restrict__ children, TreeView t,
                                                         double *__restrict__ fronts) {
  const int c = children[blockIdx.y];
  const int nb = t.nb[c];
  const int ntr = (nb + 63) >> 6, ntc = (nb + 3) >> 2;
  if ((int64_t)blockIdx.x >= (int64_t)ntr * ntc) return;
  const int r = (int)(blockIdx.x % ntr) * 64 + (threadIdx.x & 63), cc = (int)(blockIdx.x / ntr) * 4 + (threadIdx.x >> 6);
  if (r >= nb || cc >= nb) return;
  const int p = t.parent[c], npc = t.np[c];
  const int *rel = t.rel + t.roff[c];
  const double v = fronts[t.foff[c] + (int64_t)(npc + r) + (int64_t)(npc + cc) * t.ld[c]];
  fronts[t.foff[p] + (int64_t)rel[r] + (int64_t)rel[cc] * t.ld[p]] += v;
}

// small fronts of a tree level: one workgroup per front runs the whole partial factorisation
__global__ __launch_bounds__(256) void front_factor_kernel(const int *__restrict__ list, TreeView t,
                                                           double *__restrict__ fronts, double *__restrict__ invs,
                                                           int *__restrict__ singular) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  const int f = list[blockIdx.x];
  const int np = t.np[f];
  if (np == 0) return;
  const int fs = np + t.nb[f];
  const Band b{fronts + t.foff[f], fs, fs, fs, t.ld[f] + 1, 0};
  front_factor_by_workgroup(b, np, invs + t.ioff[f], singular, dsm);
}

constexpr int kSmallFront = 1024;  // fronts up to this size are factored by one workgroup each
constexpr int kStreams = 8;        // larger fronts of a level are spread over this many streams

// ---- solves: one workgroup per front -------------------------------------------------------------
// M(i, j) of the triangular system a front contributes: F(i, j), or F(j, i) for the transposed systems
template <bool TRANS>
__device__ __forceinline__ double sys_elem(const double *F, int ld, int i, int j) {
  return TRANS ? F[(size_t)j + (size_t)i * ld] : F[(size_t)i + (size_t)j * ld];
}

// v = T w for the 64 x 64 inverse diagonal block (column-major), T = inv or inv^T; the first 256
// threads of the workgroup, 4 per row
template <bool TRANS>
__device__ __forceinline__ void apply_inverse_block(const double *__restrict__ inv, const double *w, double *v) {
  if (threadIdx.x >= 256) return;
  const int l = threadIdx.x >> 2, q = threadIdx.x & 3;
  double acc = 0.0;
#pragma unroll 4
  for (int u = 0; u < NB / 4; ++u) {
    const int tt = q + 4 * u;
    acc += (TRANS ? inv[tt + l * NB] : inv[l + tt * NB]) * w[tt];
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  if (q == 0) v[l] = acc;
}

// W[i] -= sum_{tt < jb} M(i, c0 + tt) v[tt] for i in [ilo, ihi), by the 256 threads of the workgroup.
// Untransposed the front runs down i (thread = row, 8 loads in flight); transposed it runs along tt
// (a wavefront per row, lanes along tt, butterfly sum).
template <bool TRANS>
__device__ __forceinline__ void couple_block(const double *__restrict__ F, int ld, int ilo, int ihi, int c0, int jb,
                                             const double *v, double *W) {
  if (!TRANS) {
    for (int i = ilo + threadIdx.x; i < ihi; i += blockDim.x) {
      const double *row = F + (size_t)i + (size_t)c0 * ld;
      double a0 = 0.0, a1 = 0.0;
      int tt = 0;
      for (; tt + 8 <= jb; tt += 8) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) e[u] = row[(size_t)(tt + u) * ld];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          a0 += e[u] * v[tt + u];
          a1 += e[u + 1] * v[tt + u + 1];
        }
      }
      for (; tt < jb; ++tt) a0 += row[(size_t)tt * ld] * v[tt];
      W[i] -= a0 + a1;
    }
  } else {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const double vl = lane < jb ? v[lane] : 0.0;
    constexpr int RW = 8;  // rows per wavefront and trip: 8 loads in flight per lane
    for (int i0 = ilo + wave * RW; i0 < ihi; i0 += nw * RW) {
      double e[RW];
#pragma unroll
      for (int u = 0; u < RW; ++u)
        e[u] = (i0 + u < ihi && lane < jb) ? F[(size_t)(c0 + lane) + (size_t)(i0 + u) * ld] * vl : 0.0;
#pragma unroll
      for (int u = 0; u < RW; ++u) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) e[u] += __shfl_xor(e[u], m, 64);
        if (lane == 0 && i0 + u < ihi) W[i0 + u] -= e[u];
      }
    }
  }
}

// W = [rhs at the pivots | 0]
__global__ __launch_bounds__(256) void solve_init_kernel(const int *__restrict__ list, TreeView t,
                                                         const double *__restrict__ c, double *__restrict__ work) {
  const int f = list[blockIdx.x];
  const int np = t.np[f], fs = np + t.nb[f];
  double *W = work + t.woff[f];
  for (int i = threadIdx.x; i < fs; i += blockDim.x) W[i] = i < np ? c[t.p0[f] + i] : 0.0;
}

// W(parent)[rel] += boundary part of W(child), for the listed children
__global__ __launch_bounds__(256) void solve_gather_kernel(const int *__restrict__ children, TreeView t,
                                                           double *__restrict__ work) {
  const int c = children[blockIdx.x];
  const int p = t.parent[c], npc = t.np[c];
  const int *rel = t.rel + t.roff[c];
  const double *Wc = work + t.woff[c];
  double *Wp = work + t.woff[p];
  for (int k = threadIdx.x; k < t.nb[c]; k += blockDim.x) Wp[rel[k]] += Wc[npc + k];
}

// forward elimination inside a front: y = M11^-1 W[0:np) block by block (stored inverses of the diagonal
// blocks), every later entry of W loses its coupling with the block just solved
constexpr int kSolveThreads = 1024;  // one workgroup per front, 16 wavefronts for the coupling loops

template <bool TRANS>
__global__ __launch_bounds__(kSolveThreads) void solve_forward_kernel(const int *__restrict__ list, TreeView t,
                                                            const double *__restrict__ fronts,
                                                            const double *__restrict__ invs,
                                                            double *__restrict__ work) {
  __shared__ double w[NB], v[NB];
  const int f = list[blockIdx.x];
  const int np = t.np[f], fs = np + t.nb[f], ld = t.ld[f];
  const double *F = fronts + t.foff[f];
  double *W = work + t.woff[f];
  for (int j0 = 0; j0 < np; j0 += NB) {
    const int jb = min(NB, np - j0);
    if (threadIdx.x < NB) w[threadIdx.x] = threadIdx.x < jb ? W[j0 + threadIdx.x] : 0.0;
    __syncthreads();
    // forward: L (unit lower) or U^T -> inverse of L11, or of U11 transposed
    const double *inv = invs + t.ioff[f] + (size_t)(j0 / NB) * (2 * NB * NB) + (TRANS ? NB * NB : 0);
    apply_inverse_block<TRANS>(inv, w, v);
    __syncthreads();
    if (threadIdx.x < jb) W[j0 + threadIdx.x] = v[threadIdx.x];
    couple_block<TRANS>(F, ld, j0 + jb, fs, j0, jb, v, W);
    __syncthreads();
  }
}

// back substitution inside a front: x_piv = M11^-1 (y - M12 x_bnd), x_bnd read from the solution of
// the ancestors; writes the pivots' part of the solution
template <bool TRANS>
__global__ __launch_bounds__(kSolveThreads) void solve_backward_kernel(const int *__restrict__ list, TreeView t,
                                                             const double *__restrict__ fronts,
                                                             const double *__restrict__ invs,
                                                             double *__restrict__ work, double *__restrict__ x) {
  __shared__ double w[NB], v[NB];
  const int f = list[blockIdx.x];
  const int np = t.np[f], nb = t.nb[f], ld = t.ld[f], p0 = t.p0[f];
  const double *F = fronts + t.foff[f];
  double *W = work + t.woff[f];
  const int *b = t.bidx + t.bptr[f];
  for (int k = threadIdx.x; k < nb; k += blockDim.x) W[np + k] = x[b[k]];
  __syncthreads();
  for (int k0 = 0; k0 < nb; k0 += NB) {  // boundary columns in strips of 64, through LDS
    const int kb = min(NB, nb - k0);
    if (threadIdx.x < NB) v[threadIdx.x] = threadIdx.x < kb ? W[np + k0 + threadIdx.x] : 0.0;
    __syncthreads();
    couple_block<TRANS>(F, ld, 0, np, np + k0, kb, v, W);
    __syncthreads();
  }
  const int nblk = (np + NB - 1) / NB;
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int j0 = blk * NB, jb = min(NB, np - j0);
    if (threadIdx.x < NB) w[threadIdx.x] = threadIdx.x < jb ? W[j0 + threadIdx.x] : 0.0;
    __syncthreads();
    // backward: U or L^T -> inverse of U11, or of L11 transposed
    const double *inv = invs + t.ioff[f] + (size_t)blk * (2 * NB * NB) + (TRANS ? 0 : NB * NB);
    apply_inverse_block<TRANS>(inv, w, v);
    __syncthreads();
    if (threadIdx.x < jb) {
      W[j0 + threadIdx.x] = v[threadIdx.x];
      x[p0 + j0 + threadIdx.x] = v[threadIdx.x];
    }
    couple_block<TRANS>(F, ld, 0, j0, j0, jb, v, W);
    __syncthreads();
  }
}

// ---- large fronts: the same steps spread over many workgroups -----------------------------------
// boundary part of the solution into the front's work vector: W[np + k] = x[bidx[k]]
__global__ __launch_bounds__(256) void front_gather_x_kernel(const int *__restrict__ b, int nb,
                                                             const double *__restrict__ x, double *__restrict__ dst) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nb) dst[k] = x[b[k]];
}

// z[i] -= sum_k M(i, np + k) xb[k], i < np: 64 rows per workgroup
template <bool TRANS>
__global__ __launch_bounds__(256) void front_gemv_kernel(const double *__restrict__ F, int ld, int np, int nb,
                                                         const double *__restrict__ xb, double *__restrict__ z) {
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.x * 64;
  if (!TRANS) {
    const int i = i0 + lane;
    double a0 = 0.0, a1 = 0.0;
    if (i < np) {
      const double *row = F + (size_t)i + (size_t)np * ld;
      int k = wave;
      for (; k + 28 < nb; k += 32) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) e[u] = row[(size_t)(k + 4 * u) * ld];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          a0 += e[u] * xb[k + 4 * u];
          a1 += e[u + 1] * xb[k + 4 * u + 4];
        }
      }
      for (; k < nb; k += 4) a0 += row[(size_t)k * ld] * xb[k];
    }
    part[wave][lane] = a0 + a1;
    __syncthreads();
    if (wave == 0 && i < np) z[i] -= (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  } else {
    for (int r = wave * 4; r < 64; r += 16) {  // 4 rows of the chunk per wavefront and trip
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int k = lane; k < nb; k += 64) {
        const double xk = xb[k];
#pragma unroll
        for (int u = 0; u < 4; ++u)  // M(i, np + k) = F(np + k, i)
          if (i0 + r + u < np) acc[u] += F[(size_t)(np + k) + (size_t)(i0 + r + u) * ld] * xk;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[u] += __shfl_xor(acc[u], m, 64);
        if (lane == 0 && i0 + r + u < np) z[i0 + r + u] -= acc[u];
      }
    }
  }
}

__global__ __launch_bounds__(256) void front_scatter_x_kernel(int p0, int np, const double *__restrict__ src,
                                                              double *__restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < np) x[p0 + i] = src[i];
}

constexpr int kBigSolve = 3072;  // fronts above this size are solved by many workgroups

}  // namespace

namespace mf {

struct Factors {
  std::shared_ptr<const Tree> tree;
  DeviceTree D;
  TreeView view;
  DBuf<double> fronts, invs;
  std::vector<DBuf<int>> level_lists;                // fronts of each depth
  std::vector<DBuf<int>> small_lists;                // ... those factored by one workgroup each
  std::vector<int> small_counts;
  std::vector<DBuf<int>> solve_lists;                // ... those solved by one workgroup each
  std::vector<int> solve_counts;
  std::vector<DBuf<int>> child_lists[2];             // children (by slot) of the fronts of each depth
  std::vector<int> child_counts[2];
  int singular = 0;
};

}  // namespace mf

size_t mf_device_bytes(const mf::Tree &T) {
  return ((size_t)T.front_elems + (size_t)T.inv_elems + (size_t)T.work_elems) * sizeof(double) +
         ((size_t)T.bidx.size() + (size_t)T.rel_elems + 8 * (size_t)T.nfronts + (size_t)T.n) * sizeof(int64_t);
}

void mf_free(mf::Factors *F) { delete F; }

int mf_singular(const mf::Factors *F) { return F->singular; }

// numeric factorisation of P A P^T; d_Ap/d_Ai/d_Ax: CSC arrays of A on the device, d_inv: old -> new
mf::Factors *mf_factor(std::shared_ptr<const mf::Tree> tree, const int *d_Ap, const int *d_Ai, const double *d_Ax,
                       const int *d_inv, hipStream_t s) {
  const mf::Tree &T = *tree;
  const bool timing = getenv("SPL_MF_TIMING") != nullptr;  // phase times on stderr (diagnostic)
  auto clock_now = [] { return std::chrono::steady_clock::now(); };
  auto t_start = clock_now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto now = clock_now();
    fprintf(stderr, "[mf_factor] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_start).count());
    t_start = now;
  };
  std::unique_ptr<mf::Factors> Fp(new mf::Factors());
  mf::Factors &F = *Fp;
  F.tree = tree;
  DeviceTree &D = F.D;
  upload_vec(D.p0, T.p0, s);
  upload_vec(D.np, T.np, s);
  upload_vec(D.nb, T.nb, s);
  upload_vec(D.ld, T.ld, s);
  upload_vec(D.parent, T.parent, s);
  upload_vec(D.front_of, T.front_of, s);
  upload_vec(D.bidx, T.bidx, s);
  upload_vec(D.bptr, T.bptr, s);
  upload_vec(D.foff, T.foff, s);
  upload_vec(D.ioff, T.ioff, s);
  upload_vec(D.woff, T.woff, s);
  upload_vec(D.roff, T.roff, s);
  D.rel.alloc((size_t)T.rel_elems);
  F.view = TreeView{D.p0.get(), D.np.get(), D.nb.get(), D.ld.get(), D.parent.get(), D.front_of.get(), D.bidx.get(),
                    D.rel.get(), D.bptr.get(), D.foff.get(), D.ioff.get(), D.woff.get(), D.roff.get()};
  const int nd = T.maxdepth + 1;
  F.level_lists.resize((size_t)nd);
  F.small_lists.resize((size_t)nd);
  F.small_counts.assign((size_t)nd, 0);
  F.solve_lists.resize((size_t)nd);
  F.solve_counts.assign((size_t)nd, 0);
  for (int sl = 0; sl < 2; ++sl) {
    F.child_lists[sl].resize((size_t)nd);
    F.child_counts[sl].assign((size_t)nd, 0);
  }
  std::vector<std::vector<int>> staged;  // host copies must outlive the asynchronous uploads
  staged.reserve((size_t)4 * nd);
  for (int d = 0; d < nd; ++d) {
    upload_vec(F.level_lists[(size_t)d], T.by_depth[(size_t)d], s);
    {
      std::vector<int> small;
      for (int f : T.by_depth[(size_t)d])
        if (T.np[(size_t)f] > 0 && T.fs(f) <= kSmallFront) small.push_back(f);
      F.small_counts[(size_t)d] = (int)small.size();
      staged.push_back(std::move(small));
      upload_vec(F.small_lists[(size_t)d], staged.back(), s);
      std::vector<int> one_wg;
      for (int f : T.by_depth[(size_t)d])
        if (T.fs(f) <= kBigSolve) one_wg.push_back(f);
      F.solve_counts[(size_t)d] = (int)one_wg.size();
      staged.push_back(std::move(one_wg));
      upload_vec(F.solve_lists[(size_t)d], staged.back(), s);
    }
    if (d + 1 < nd) {
      std::vector<int> ch[2];
      for (int c : T.by_depth[(size_t)d + 1]) ch[T.slot[(size_t)c]].push_back(c);
      for (int sl = 0; sl < 2; ++sl) {
        F.child_counts[sl][(size_t)d] = (int)ch[sl].size();
        staged.push_back(std::move(ch[sl]));
        upload_vec(F.child_lists[sl][(size_t)d], staged.back(), s);
      }
    }
  }
  SPL_HIP(hipStreamSynchronize(s));
  staged.clear();
  lap("tree upload");
  F.fronts.alloc((size_t)T.front_elems);
  F.invs.alloc((size_t)T.inv_elems);
  SPL_HIP(hipMemsetAsync(F.fronts.get(), 0, (size_t)T.front_elems * sizeof(double), s));
  if (T.nfronts > 0)
    hipLaunchKernelGGL(rel_kernel, dim3((unsigned)T.nfronts), dim3(256), 0, s, T.nfronts, F.view, D.rel.get());
  if (T.n > 0)
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)(((size_t)T.n * 8 + 255) / 256)), dim3(256), 0, s, T.n, d_Ap,
                       d_Ai, d_Ax, d_inv, F.view, F.fronts.get());
  lap("alloc + zero + assemble");
  DBuf<int> singular(1);
  SPL_HIP(hipMemsetAsync(singular.get(), 0, sizeof(int), s));
  set_factor_attributes();
  static bool attr_set = false;
  if (!attr_set) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&front_factor_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileBytes)));
    attr_set = true;
  }
  hipStream_t side[kStreams];
  for (int i = 0; i < kStreams; ++i) SPL_HIP(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
  for (int d = nd - 1; d >= 0; --d) {
    // Schur complements of the children, one child slot after the other (two children of a parent
    // may touch the same entry: a fixed order keeps the sums reproducible)
    if (d + 1 < nd) {
      int max_nb = 0;
      for (int c : T.by_depth[(size_t)d + 1]) max_nb = std::max(max_nb, T.nb[(size_t)c]);
      const int64_t ntile = (int64_t)((max_nb + 63) / 64) * ((max_nb + 3) / 4);
      for (int sl = 0; sl < 2; ++sl)
        if (F.child_counts[sl][(size_t)d] > 0 && ntile > 0)
          hipLaunchKernelGGL(extend_add_kernel, dim3((unsigned)ntile, (unsigned)F.child_counts[sl][(size_t)d]),
                             dim3(256), 0, s, F.child_lists[sl][(size_t)d].get(), F.view, F.fronts.get());
    }
    // small fronts: one launch, one workgroup each; large fronts: the multi-launch blocked
    // factorisation, independent fronts spread over side streams
    SPL_HIP(hipStreamSynchronize(s));
    if (F.small_counts[(size_t)d] > 0)
      hipLaunchKernelGGL(front_factor_kernel, dim3((unsigned)F.small_counts[(size_t)d]), dim3(256), 2 * kTileBytes, s,
                         F.small_lists[(size_t)d].get(), F.view, F.fronts.get(), F.invs.get(), singular.get());
    int turn = 0;
    for (int f : T.by_depth[(size_t)d]) {
      if (T.np[(size_t)f] == 0 || T.fs(f) <= kSmallFront) continue;
      const Band b = dense_view(F.fronts.get() + T.foff[(size_t)f], T.fs(f), T.ld[(size_t)f]);
      factor_loop(b, T.np[(size_t)f], F.invs.get() + T.ioff[(size_t)f], singular.get(), side[turn++ % kStreams]);
    }
    if (turn > 0)
      for (int i = 0; i < kStreams && i < turn; ++i) SPL_HIP(hipStreamSynchronize(side[i]));
  }
  lap("levels");
  for (int i = 0; i < kStreams; ++i) (void)hipStreamDestroy(side[i]);
  SPL_HIP(hipMemcpyAsync(&F.singular, singular.get(), sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  SPL_HIP(hipGetLastError());
  return Fp.release();
}

// columns of c (device, new ordering, column r at d_c + r * stride) <- (P A P^T)^-1 c or its transpose
void mf_solve(const mf::Factors *Fp, int sys, double *d_c, int k, size_t stride, hipStream_t s) {
  const mf::Factors &F = *Fp;
  const mf::Tree &T = *F.tree;
  if (T.n == 0 || k == 0) return;
  const int nd = T.maxdepth + 1;
  DBuf<double> work((size_t)T.work_elems), zbuf((size_t)T.work_elems);
  double *fronts = F.fronts.get();
  double *invs = F.invs.get();
  for (int col = 0; col < k; ++col) {
    double *c = d_c + (size_t)col * stride;
    // up the tree: L y = c (or U^T y = c)
    for (int d = nd - 1; d >= 0; --d) {
      const unsigned nf = (unsigned)T.by_depth[(size_t)d].size();
      hipLaunchKernelGGL(solve_init_kernel, dim3(nf), dim3(256), 0, s, F.level_lists[(size_t)d].get(), F.view, c,
                         work.get());
      if (d + 1 < nd)
        for (int sl = 0; sl < 2; ++sl)
          if (F.child_counts[sl][(size_t)d] > 0)
            hipLaunchKernelGGL(solve_gather_kernel, dim3((unsigned)F.child_counts[sl][(size_t)d]), dim3(256), 0, s,
                               F.child_lists[sl][(size_t)d].get(), F.view, work.get());
      if (F.solve_counts[(size_t)d] > 0) {
        const unsigned ns = (unsigned)F.solve_counts[(size_t)d];
        const int *list = F.solve_lists[(size_t)d].get();
        if (sys == 0)
          hipLaunchKernelGGL(solve_forward_kernel<false>, dim3(ns), dim3(kSolveThreads), 0, s, list, F.view, fronts, invs,
                             work.get());
        else
          hipLaunchKernelGGL(solve_forward_kernel<true>, dim3(ns), dim3(kSolveThreads), 0, s, list, F.view, fronts, invs,
                             work.get());
      }
      for (int f : T.by_depth[(size_t)d]) {
        if (T.fs(f) <= kBigSolve || T.np[(size_t)f] == 0) continue;
        const Band b = dense_view(fronts + T.foff[(size_t)f], T.fs(f), T.ld[(size_t)f]);
        double *W = work.get() + T.woff[(size_t)f], *Z = zbuf.get() + T.woff[(size_t)f];
        if (sys == 0) solve_pass<0, 1>(b, invs + T.ioff[(size_t)f], b.n, W, Z, 0, s, T.np[(size_t)f]);
        else solve_pass<2, 1>(b, invs + T.ioff[(size_t)f], b.n, W, Z, 0, s, T.np[(size_t)f]);
      }
    }
    // down the tree: U x = y (or L^T x = y); the pivots' part of x overwrites c
    for (int d = 0; d < nd; ++d) {
      if (F.solve_counts[(size_t)d] > 0) {
        const unsigned ns = (unsigned)F.solve_counts[(size_t)d];
        const int *list = F.solve_lists[(size_t)d].get();
        if (sys == 0)
          hipLaunchKernelGGL(solve_backward_kernel<false>, dim3(ns), dim3(kSolveThreads), 0, s, list, F.view, fronts, invs,
                             work.get(), c);
        else
          hipLaunchKernelGGL(solve_backward_kernel<true>, dim3(ns), dim3(kSolveThreads), 0, s, list, F.view, fronts, invs,
                             work.get(), c);
      }
      for (int f : T.by_depth[(size_t)d]) {
        const int np = T.np[(size_t)f], nb = T.nb[(size_t)f];
        if (T.fs(f) <= kBigSolve || np == 0) continue;
        double *Ff = fronts + T.foff[(size_t)f];
        const int ld = T.ld[(size_t)f];
        double *W = work.get() + T.woff[(size_t)f], *Z = zbuf.get() + T.woff[(size_t)f];
        if (nb > 0) {
          hipLaunchKernelGGL(front_gather_x_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s,
                             F.D.bidx.get() + T.bptr[(size_t)f], nb, c, W + np);
          if (sys == 0)
            hipLaunchKernelGGL(front_gemv_kernel<false>, dim3((unsigned)((np + 63) / 64)), dim3(256), 0, s, Ff, ld, np,
                               nb, W + np, Z);
          else
            hipLaunchKernelGGL(front_gemv_kernel<true>, dim3((unsigned)((np + 63) / 64)), dim3(256), 0, s, Ff, ld, np,
                               nb, W + np, Z);
        }
        const Band b = dense_view(Ff, np, ld);  // the pivot block alone
        if (sys == 0) solve_pass<1, 1>(b, invs + T.ioff[(size_t)f], np, Z, W, 0, s);
        else solve_pass<3, 1>(b, invs + T.ioff[(size_t)f], np, Z, W, 0, s);
        hipLaunchKernelGGL(front_scatter_x_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s,
                           T.p0[(size_t)f], np, W, c);
      }
    }
  }
  SPL_HIP(hipStreamSynchronize(s));  // the work vectors are freed on return
}

}  // namespace spl

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

// parent += Schur complement of the listed children (16 x 16 tiles; blockIdx.y = child)
__global__ __launch_bounds__(256) void extend_add_kernel(const int *__restrict__ children, TreeView t,
                                                         double *__restrict__ fronts) {
  const int c = children[blockIdx.y];
  const int nb = t.nb[c];
  const int ntile = (nb + 15) >> 4;
  if ((int)blockIdx.x >= ntile * ntile) return;
  const int tr = blockIdx.x % ntile, tc = blockIdx.x / ntile;
  const int r = tr * 16 + (threadIdx.x & 15), cc = tc * 16 + (threadIdx.x >> 4);
  if (r >= nb || cc >= nb) return;
  const int p = t.parent[c], npc = t.np[c];
  const int *rel = t.rel + t.roff[c];
  const double v = fronts[t.foff[c] + (int64_t)(npc + r) + (int64_t)(npc + cc) * t.ld[c]];
  fronts[t.foff[p] + (int64_t)rel[r] + (int64_t)rel[cc] * t.ld[p]] += v;
}

// ---- solves: one workgroup per front -------------------------------------------------------------
// M(i, j) of the triangular system a front contributes: F(i, j), or F(j, i) for the transposed systems
template <bool TRANS>
__device__ __forceinline__ double sys_elem(const double *F, int ld, int i, int j) {
  return TRANS ? F[(size_t)j + (size_t)i * ld] : F[(size_t)i + (size_t)j * ld];
}

// v = T w for the 64 x 64 inverse diagonal block (column-major), T = inv or inv^T; 256 threads, 4 per row
template <bool TRANS>
__device__ __forceinline__ void apply_inverse_block(const double *__restrict__ inv, const double *w, double *v) {
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
template <bool TRANS>
__global__ __launch_bounds__(256) void solve_forward_kernel(const int *__restrict__ list, TreeView t,
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
    for (int i = j0 + jb + threadIdx.x; i < fs; i += blockDim.x) {
      double acc = 0.0;
      for (int tt = 0; tt < jb; ++tt) acc += sys_elem<TRANS>(F, ld, i, j0 + tt) * v[tt];
      W[i] -= acc;
    }
    __syncthreads();
  }
}

// back substitution inside a front: x_piv = M11^-1 (y - M12 x_bnd), x_bnd read from the solution of
// the ancestors; writes the pivots' part of the solution
template <bool TRANS>
__global__ __launch_bounds__(256) void solve_backward_kernel(const int *__restrict__ list, TreeView t,
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
  for (int i = threadIdx.x; i < np; i += blockDim.x) {
    double acc = 0.0;
    for (int k = 0; k < nb; ++k) acc += sys_elem<TRANS>(F, ld, i, np + k) * W[np + k];
    W[i] -= acc;
  }
  __syncthreads();
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
    for (int i = threadIdx.x; i < j0; i += blockDim.x) {
      double acc = 0.0;
      for (int tt = 0; tt < jb; ++tt) acc += sys_elem<TRANS>(F, ld, i, j0 + tt) * v[tt];
      W[i] -= acc;
    }
    __syncthreads();
  }
}

}  // namespace

namespace mf {

struct Factors {
  std::shared_ptr<const Tree> tree;
  DeviceTree D;
  TreeView view;
  DBuf<double> fronts, invs;
  std::vector<DBuf<int>> level_lists;                // fronts of each depth
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
  for (int sl = 0; sl < 2; ++sl) {
    F.child_lists[sl].resize((size_t)nd);
    F.child_counts[sl].assign((size_t)nd, 0);
  }
  std::vector<std::vector<int>> staged;  // host copies must outlive the asynchronous uploads
  staged.reserve((size_t)2 * nd);
  for (int d = 0; d < nd; ++d) {
    upload_vec(F.level_lists[(size_t)d], T.by_depth[(size_t)d], s);
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
  F.fronts.alloc((size_t)T.front_elems);
  F.invs.alloc((size_t)T.inv_elems);
  SPL_HIP(hipMemsetAsync(F.fronts.get(), 0, (size_t)T.front_elems * sizeof(double), s));
  if (T.nfronts > 0)
    hipLaunchKernelGGL(rel_kernel, dim3((unsigned)T.nfronts), dim3(256), 0, s, T.nfronts, F.view, D.rel.get());
  if (T.n > 0)
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)(((size_t)T.n * 8 + 255) / 256)), dim3(256), 0, s, T.n, d_Ap,
                       d_Ai, d_Ax, d_inv, F.view, F.fronts.get());
  DBuf<int> singular(1);
  SPL_HIP(hipMemsetAsync(singular.get(), 0, sizeof(int), s));
  for (int d = nd - 1; d >= 0; --d) {
    // Schur complements of the children, one child slot after the other (two children of a parent
    // may touch the same entry: a fixed order keeps the sums reproducible)
    if (d + 1 < nd) {
      int max_nb = 0;
      for (int c : T.by_depth[(size_t)d + 1]) max_nb = std::max(max_nb, T.nb[(size_t)c]);
      const int ntile = (max_nb + 15) / 16;
      for (int sl = 0; sl < 2; ++sl)
        if (F.child_counts[sl][(size_t)d] > 0 && ntile > 0)
          hipLaunchKernelGGL(extend_add_kernel, dim3((unsigned)(ntile * ntile), (unsigned)F.child_counts[sl][(size_t)d]),
                             dim3(256), 0, s, F.child_lists[sl][(size_t)d].get(), F.view, F.fronts.get());
    }
    for (int f : T.by_depth[(size_t)d]) {
      if (T.np[(size_t)f] == 0) continue;
      const Band b = dense_view(F.fronts.get() + T.foff[(size_t)f], T.fs(f), T.ld[(size_t)f]);
      factor_loop(b, T.np[(size_t)f], F.invs.get() + T.ioff[(size_t)f], singular.get(), s);
    }
  }
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
  DBuf<double> work((size_t)T.work_elems);
  for (int col = 0; col < k; ++col) {
    double *c = d_c + (size_t)col * stride;
    for (int d = nd - 1; d >= 0; --d) {
      const unsigned nf = (unsigned)T.by_depth[(size_t)d].size();
      const int *list = F.level_lists[(size_t)d].get();
      hipLaunchKernelGGL(solve_init_kernel, dim3(nf), dim3(256), 0, s, list, F.view, c, work.get());
      if (d + 1 < nd)
        for (int sl = 0; sl < 2; ++sl)
          if (F.child_counts[sl][(size_t)d] > 0)
            hipLaunchKernelGGL(solve_gather_kernel, dim3((unsigned)F.child_counts[sl][(size_t)d]), dim3(256), 0, s,
                               F.child_lists[sl][(size_t)d].get(), F.view, work.get());
      if (sys == 0)
        hipLaunchKernelGGL(solve_forward_kernel<false>, dim3(nf), dim3(256), 0, s, list, F.view, F.fronts.get(),
                           F.invs.get(), work.get());
      else
        hipLaunchKernelGGL(solve_forward_kernel<true>, dim3(nf), dim3(256), 0, s, list, F.view, F.fronts.get(),
                           F.invs.get(), work.get());
    }
    for (int d = 0; d < nd; ++d) {
      const unsigned nf = (unsigned)T.by_depth[(size_t)d].size();
      const int *list = F.level_lists[(size_t)d].get();
      if (sys == 0)
        hipLaunchKernelGGL(solve_backward_kernel<false>, dim3(nf), dim3(256), 0, s, list, F.view, F.fronts.get(),
                           F.invs.get(), work.get(), c);
      else
        hipLaunchKernelGGL(solve_backward_kernel<true>, dim3(nf), dim3(256), 0, s, list, F.view, F.fronts.get(),
                           F.invs.get(), work.get(), c);
    }
  }
  SPL_HIP(hipStreamSynchronize(s));  // work is freed on return
}

}  // namespace spl

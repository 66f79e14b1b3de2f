// assemble.hip — COO -> CSC (`compress`, Sparse.hs:184-255) and the sparse linear
// combination alpha*A + beta*B (`lin` over `glin`, Sparse.hs:401-431) in HBM.
//
// compress: bounds check (rows first, then columns: Sparse.hs:196-212), bucket by
// column (histogram + scan + cursor scatter), sort every column by the 64-bit key
// (row << 32 | input position) — i.e. by row, ties in input order, the order in
// which the oracle's stable sorts leave duplicates — then sum each run of equal
// rows left to right into its first entry (dedupInPlace, Sparse.hs:257-280) and
// compact.  Explicit zeros are kept.
//
// lin: both operands are valid Matrix values, so their columns are strictly
// ascending; the union pattern of a column is a two-pointer merge.  Values are
// evaluated exactly as the reference's SPA does: (0 + alpha*a) + beta*b, each
// operation separately rounded (-ffp-contract=off).
#include "common.hpp"
#include <atomic>

namespace spl {

namespace {

inline unsigned blocks_for(int64_t n, int per_block, int64_t cap = 1 << 20) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

__global__ __launch_bounds__(256) void bounds_kernel(const int *__restrict__ idx, int64_t nnz, int bound,
                                                     unsigned long long *__restrict__ first_bad) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long bad = ~0ull;
  for (; i < nnz; i += stride) {
    const int v = idx[i];
    if ((v < 0 || v >= bound) && (unsigned long long)i < bad) bad = (unsigned long long)i;
  }
  if (bad != ~0ull) atomicMin(first_bad, bad);
}

__global__ __launch_bounds__(256) void count_kernel(const int *__restrict__ idx, int64_t nnz,
                                                    int *__restrict__ counts) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) atomicAdd(&counts[idx[i]], 1);
}

__global__ __launch_bounds__(256) void coo_bucket_kernel(const int *__restrict__ rows,
                                                         const int *__restrict__ cols,
                                                         const double *__restrict__ vals, int64_t nnz,
                                                         const int64_t *__restrict__ colptr,
                                                         int *__restrict__ cursor, int64_t *__restrict__ key,
                                                         double *__restrict__ val) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    const int c = cols[i];
    const int64_t pos = colptr[c] + (int64_t)atomicAdd(&cursor[c], 1);
    key[pos] = ((int64_t)rows[i] << 32) | i;  // nnz < 2^31: the position fits the low word
    val[pos] = vals[i];
  }
}

__global__ __launch_bounds__(256) void mark_col_starts_kernel(const int64_t *__restrict__ colptr,
                                                              int64_t ncols, int *__restrict__ head) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; c < ncols; c += stride)
    if (colptr[c] < colptr[c + 1]) head[colptr[c]] = 1;
}

__global__ __launch_bounds__(256) void mark_row_changes_kernel(const int64_t *__restrict__ key, int64_t nnz,
                                                               int *__restrict__ head) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride)
    if (i == 0 || (key[i] >> 32) != (key[i - 1] >> 32)) head[i] = 1;
}

// every run head sums its run left to right and writes the compacted entry
__global__ __launch_bounds__(256) void dedup_sum_kernel(const int64_t *__restrict__ key,
                                                        const double *__restrict__ val,
                                                        const int *__restrict__ head,
                                                        const int64_t *__restrict__ outpos, int64_t nnz,
                                                        int *__restrict__ out_idx,
                                                        double *__restrict__ out_val) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) {
    if (!head[i]) continue;
    double acc = val[i];
    for (int64_t q = i + 1; q < nnz && !head[q]; ++q) acc = acc + val[q];  // x' + x, Sparse.hs:272-273
    const int64_t o = outpos[i];
    out_idx[o] = (int)(key[i] >> 32);
    out_val[o] = acc;
  }
}

__global__ __launch_bounds__(256) void remap_ptr_kernel(const int64_t *__restrict__ colptr,
                                                        const int64_t *__restrict__ outpos, int64_t ncols,
                                                        int64_t nnz, int *__restrict__ newptr) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; c <= ncols; c += stride) newptr[c] = (int)outpos[colptr[c]];  // outpos has nnz+1 entries
}

// ---- lin ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sorted_check_kernel(const int *__restrict__ ptr,
                                                           const int *__restrict__ idx, int64_t ncols,
                                                           int *__restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= ncols) return;
  bool bad = false;
  for (int k = ptr[c] + 1 + lane; k < ptr[c + 1]; k += 64) bad |= !(idx[k - 1] < idx[k]);
  if (bad) atomicOr(flag, 1);
}

// VW = 1: Double.  VW = 2: Complex Double, values and scalars as (re, im) pairs; products and sums in
// Data.Complex's order — (x:+y)*(x':+y') = (x*x' - y*y') :+ (x*y' + y*x'), sums componentwise.
template <bool FILL, int VW>
__global__ __launch_bounds__(256) void lin_merge_kernel(
    double alpha, double alpha_im, const int *__restrict__ Ap, const int *__restrict__ Ai,
    const double *__restrict__ Ax, double beta, double beta_im, const int *__restrict__ Bp,
    const int *__restrict__ Bi, const double *__restrict__ Bx, int64_t ncols, int *__restrict__ counts,
    const int64_t *__restrict__ Cp, int *__restrict__ Ci, double *__restrict__ Cx) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  int a = Ap[c], b = Bp[c];
  const int ae = Ap[c + 1], be = Bp[c + 1];
  int64_t o = FILL ? Cp[c] : 0;
  int n = 0;
  while (a < ae || b < be) {
    const int ra = a < ae ? Ai[a] : 0x7fffffff;
    const int rb = b < be ? Bi[b] : 0x7fffffff;
    if (FILL) {
      Ci[o] = ra < rb ? ra : rb;
      if (VW == 1) {
        double w = 0.0;                       // SG.reset 0
        if (ra <= rb) w = w + alpha * Ax[a];  // fA: \r a -> r + alpha * a
        if (rb <= ra) w = w + beta * Bx[b];   // fB: \r b -> r + beta * b
        Cx[o] = w;
      } else {
        double wr = 0.0, wi = 0.0;
        if (ra <= rb) {
          const double xr = Ax[2 * (size_t)a], xi = Ax[2 * (size_t)a + 1];
          wr = wr + (alpha * xr - alpha_im * xi);
          wi = wi + (alpha * xi + alpha_im * xr);
        }
        if (rb <= ra) {
          const double xr = Bx[2 * (size_t)b], xi = Bx[2 * (size_t)b + 1];
          wr = wr + (beta * xr - beta_im * xi);
          wi = wi + (beta * xi + beta_im * xr);
        }
        Cx[2 * (size_t)o] = wr;
        Cx[2 * (size_t)o + 1] = wi;
      }
      ++o;
    }
    ++n;
    if (ra <= rb) ++a;
    if (rb <= ra) ++b;
  }
  if (!FILL) counts[c] = n;
}


// Tiled form (round 3): the merge itself is unchanged — one thread walks the two sorted columns exactly as above,
// so the values are the same bits — but a workgroup of kLinCols threads takes kLinCols CONSECUTIVE columns, whose
// entries are one contiguous range of each operand, stages those ranges in LDS with coalesced loads, merges out of
// LDS into an LDS image of the result range and writes that image back with coalesced stores.  The thread-per-column
// kernel above reads and writes a 12-byte entry per lane and instruction at addresses a column apart (330 GB/s at
// 4e7 entries, profiles/r02_assembly_kernel_stats.txt); here every global access is a full line.  Tiles whose two
// ranges exceed the LDS image (very long columns) fall back to the walk in global memory, tile by tile.
constexpr int kLinCols = 64;

template <int VW>
struct LinTile {
  static constexpr int kCap = VW == 1 ? 3072 : 2048;  // staged entries of A + B; the result range has at most as many
};

// n elements global -> LDS (or LDS -> global) by one workgroup, eight independent loads in flight per thread: a
// plain loop would wait for every load before it issues the next (one HBM latency per 64 elements)
template <typename T, typename D, typename S>
__device__ inline void lin_copy(D dst, S src, int n) {
  for (int i0 = 0; i0 < n; i0 += kLinCols * 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * kLinCols + (int)threadIdx.x;
      v[u] = src[i < n ? i : n - 1];  // branch-free: lanes past the end re-read the last element
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * kLinCols + (int)threadIdx.x;
      if (i < n) dst[i] = v[u];
    }
  }
}

template <bool FILL, int VW>
__global__ __launch_bounds__(kLinCols) void lin_tile_kernel(
    double alpha, double alpha_im, const int *__restrict__ Ap, const int *__restrict__ Ai,
    const double *__restrict__ Ax, double beta, double beta_im, const int *__restrict__ Bp,
    const int *__restrict__ Bi, const double *__restrict__ Bx, int64_t ncols, int *__restrict__ counts,
    const int64_t *__restrict__ Cp, int *__restrict__ Ci, double *__restrict__ Cx) {
  constexpr int CAP = LinTile<VW>::kCap;
  extern __shared__ __attribute__((aligned(16))) unsigned char lin_lds[];
  int *sI = reinterpret_cast<int *>(lin_lds);                       // CAP: rows of A's range, then B's
  int *sO = sI + CAP;                                               // CAP: rows of the result range (FILL)
  double *sX = reinterpret_cast<double *>(sO + (FILL ? CAP : 0));   // CAP * VW values in, then CAP * VW out
  double *sY = sX + (size_t)CAP * VW;
  const int64_t c0 = (int64_t)blockIdx.x * kLinCols;
  const int64_t c1 = c0 + kLinCols < ncols ? c0 + kLinCols : ncols;
  const int64_t c = c0 + threadIdx.x;
  const int a0 = Ap[c0], a1 = Ap[c1], b0 = Bp[c0], b1 = Bp[c1];
  const int nA = a1 - a0, nB = b1 - b0;
  const bool staged = nA + nB <= CAP;
  int64_t o0 = 0;
  int nC = 0;
  if (FILL) { o0 = Cp[c0]; nC = (int)(Cp[c1] - o0); }
  if (staged) {
    // rows and values of both ranges in ONE sweep, sixteen entries per thread in flight (a tile costs three HBM
    // round trips instead of one per array and batch)
    const int nT = nA + nB;
    for (int i0 = 0; i0 < nT; i0 += kLinCols * 16) {
      int vi[16];
      double vx[16][VW];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        // no branch around a load (the compiler drains vmcnt at every join): lanes past the end re-read the last entry
        int i = i0 + u * kLinCols + (int)threadIdx.x;
        i = i < nT ? i : nT - 1;
        const bool in_a = i < nA;
        const int k = in_a ? a0 + i : b0 + (i - nA);
        const int *pi = in_a ? Ai : Bi;        // one load with a per-lane address, not two predicated ones
        const double *px = in_a ? Ax : Bx;
        vi[u] = pi[k];
        if (FILL) {
#pragma unroll
          for (int q = 0; q < VW; ++q) vx[u][q] = px[(size_t)k * VW + q];
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = i0 + u * kLinCols + (int)threadIdx.x;
        if (i < nT) {
          sI[i] = vi[u];
          if (FILL) {
#pragma unroll
            for (int q = 0; q < VW; ++q) sX[(size_t)i * VW + q] = vx[u][q];
          }
        }
      }
    }
  }
  __syncthreads();
  if (c < ncols) {
    const int as = Ap[c], bs = Bp[c];
    const int ae = Ap[c + 1], be = Bp[c + 1];
    const int64_t oc = FILL ? Cp[c] : 0;
    // The walk of Sparse.hs:401-431 with the head of each column held in registers: per step ONE round trip to
    // the staged image (the entry behind the one just consumed), the result written without waiting.  get_* /
    // put are the staged (LDS) or the global accessors; the arithmetic and its order are those of lin_merge_kernel.
    auto walk = [&](auto get_ai, auto get_ax, auto get_bi, auto get_bx, auto put) {
      constexpr int kEnd = 0x7fffffff;
      int a = as, b = bs, n = 0;
      int ra = a < ae ? get_ai(a) : kEnd, rb = b < be ? get_bi(b) : kEnd;
      double xa0 = 0.0, xa1 = 0.0, xb0 = 0.0, xb1 = 0.0;
      if (FILL) {
        if (a < ae) get_ax(a, xa0, xa1);
        if (b < be) get_bx(b, xb0, xb1);
      }
      while (ra != kEnd || rb != kEnd) {
        const bool ta = ra <= rb, tb = rb <= ra;
        if (FILL) {
          const int r = ta ? ra : rb;
          if (VW == 1) {
            double w = 0.0;                  // SG.reset 0
            if (ta) w = w + alpha * xa0;     // fA: \r a -> r + alpha * a
            if (tb) w = w + beta * xb0;      // fB: \r b -> r + beta * b
            put(n, r, w, 0.0);
          } else {
            double wr = 0.0, wi = 0.0;
            if (ta) { wr = wr + (alpha * xa0 - alpha_im * xa1); wi = wi + (alpha * xa1 + alpha_im * xa0); }
            if (tb) { wr = wr + (beta * xb0 - beta_im * xb1); wi = wi + (beta * xb1 + beta_im * xb0); }
            put(n, r, wr, wi);
          }
        }
        ++n;
        if (ta) { ++a; ra = a < ae ? get_ai(a) : kEnd; if (FILL && a < ae) get_ax(a, xa0, xa1); }
        if (tb) { ++b; rb = b < be ? get_bi(b) : kEnd; if (FILL && b < be) get_bx(b, xb0, xb1); }
      }
      return n;
    };
    int n;
    if (staged)
      n = walk([&](int k) { return sI[k - a0]; },
               [&](int k, double &re, double &im) { if (VW == 1) re = sX[k - a0]; else { re = sX[2 * (k - a0)]; im = sX[2 * (k - a0) + 1]; } },
               [&](int k) { return sI[nA + (k - b0)]; },
               [&](int k, double &re, double &im) { if (VW == 1) re = sX[nA + (k - b0)]; else { re = sX[2 * (nA + (k - b0))]; im = sX[2 * (nA + (k - b0)) + 1]; } },
               [&](int k, int r, double re, double im) {
                 const int q = (int)(oc - o0) + k;
                 sO[q] = r;
                 if (VW == 1) sY[q] = re; else { sY[2 * q] = re; sY[2 * q + 1] = im; }
               });
    else
      n = walk([&](int k) { return Ai[k]; },
               [&](int k, double &re, double &im) { if (VW == 1) re = Ax[k]; else { re = Ax[2 * (size_t)k]; im = Ax[2 * (size_t)k + 1]; } },
               [&](int k) { return Bi[k]; },
               [&](int k, double &re, double &im) { if (VW == 1) re = Bx[k]; else { re = Bx[2 * (size_t)k]; im = Bx[2 * (size_t)k + 1]; } },
               [&](int k, int r, double re, double im) {
                 const int64_t q = oc + k;
                 Ci[q] = r;
                 if (VW == 1) Cx[q] = re; else { Cx[2 * (size_t)q] = re; Cx[2 * (size_t)q + 1] = im; }
               });
    if (!FILL) counts[c] = n;
  }
  if (FILL && staged) {
    __syncthreads();
    lin_copy<int>(Ci + o0, sO, nC);
    lin_copy<double>(Cx + (size_t)o0 * VW, sY, nC * VW);
  }
}

template <bool FILL, int VW>
size_t lin_tile_lds_bytes() {
  constexpr size_t CAP = LinTile<VW>::kCap;
  return FILL ? CAP * 4 * 2 + CAP * VW * 8 * 2 : CAP * 4;
}

}  // namespace

// COO (device arrays) -> CSC.  Returns SPL_OK / SPL_ERROR_index_out_of_bounds.
// Outputs: d_newptr[ncols+1] (int32), out_idx/out_val allocated with nnz_out entries.
int compress_device(int nrows, int ncols, int64_t nnz, const int *d_rows, const int *d_cols,
                    const double *d_vals, int *d_newptr, DBuf<int> &out_idx, DBuf<double> &out_val,
                    int64_t *nnz_out, int64_t *bad, hipStream_t s, bool check_only) {
  *nnz_out = 0;
  if (nnz == 0) {
    SPL_HIP(hipMemsetAsync(d_newptr, 0, ((size_t)ncols + 1) * sizeof(int), s));
    out_idx.alloc(0);
    out_val.alloc(0);
    SPL_HIP(hipStreamSynchronize(s));
    return SPL_OK;
  }
  DBuf<unsigned long long> first_bad(1);
  for (int pass = 0; pass < 2; ++pass) {  // rows, then columns (Sparse.hs:196-212)
    SPL_HIP(hipMemsetAsync(first_bad.get(), 0xff, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(bounds_kernel, dim3(blocks_for(nnz, 256, 8192)), dim3(256), 0, s,
                       pass == 0 ? d_rows : d_cols, nnz, pass == 0 ? nrows : ncols, first_bad.get());
    unsigned long long h = 0;
    SPL_HIP(hipMemcpyAsync(&h, first_bad.get(), sizeof(h), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    if (h != ~0ull) {
      if (bad) *bad = (int64_t)h;
      return SPL_ERROR_index_out_of_bounds;
    }
  }
  if (check_only) return SPL_OK;  // bounds only (spl_matrix_compress_dev keeps the reference's order of complaints)
  DBuf<int> counts((size_t)ncols);
  DBuf<int64_t> colptr((size_t)ncols + 1);
  SPL_HIP(hipMemsetAsync(counts.get(), 0, (size_t)(ncols ? ncols : 1) * sizeof(int), s));
  hipLaunchKernelGGL(count_kernel, dim3(blocks_for(nnz, 256, 16384)), dim3(256), 0, s, d_cols, nnz,
                     counts.get());
  exclusive_scan_i32_to_i64(counts.get(), colptr.get(), ncols, s);
  SPL_HIP(hipMemsetAsync(counts.get(), 0, (size_t)(ncols ? ncols : 1) * sizeof(int), s));
  DBuf<int64_t> key((size_t)nnz);
  DBuf<double> val((size_t)nnz);
  hipLaunchKernelGGL(coo_bucket_kernel, dim3(blocks_for(nnz, 256, 16384)), dim3(256), 0, s, d_rows, d_cols,
                     d_vals, nnz, colptr.get(), counts.get(), key.get(), val.get());
  segmented_sort_pairs64(colptr.get(), ncols, key.get(), val.get(), s);
  // run heads: first entry of a column, or a row change
  DBuf<int> head((size_t)nnz);
  DBuf<int64_t> outpos((size_t)nnz + 1);
  SPL_HIP(hipMemsetAsync(head.get(), 0, (size_t)nnz * sizeof(int), s));
  hipLaunchKernelGGL(mark_col_starts_kernel, dim3(blocks_for(ncols, 256, 8192)), dim3(256), 0, s,
                     colptr.get(), (int64_t)ncols, head.get());
  hipLaunchKernelGGL(mark_row_changes_kernel, dim3(blocks_for(nnz, 256, 16384)), dim3(256), 0, s, key.get(),
                     nnz, head.get());
  exclusive_scan_i32_to_i64(head.get(), outpos.get(), nnz, s);
  int64_t nz = 0;
  SPL_HIP(hipMemcpyAsync(&nz, outpos.get() + nnz, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  out_idx.alloc((size_t)nz);
  out_val.alloc((size_t)nz);
  hipLaunchKernelGGL(dedup_sum_kernel, dim3(blocks_for(nnz, 256, 16384)), dim3(256), 0, s, key.get(),
                     val.get(), head.get(), outpos.get(), nnz, out_idx.get(), out_val.get());
  hipLaunchKernelGGL(remap_ptr_kernel, dim3(blocks_for(ncols + 1, 256, 8192)), dim3(256), 0, s, colptr.get(),
                     outpos.get(), (int64_t)ncols, nnz, d_newptr);
  SPL_HIP(hipStreamSynchronize(s));
  *nnz_out = nz;
  return SPL_OK;
}

// columns strictly ascending?  (valid Matrix invariant, tests/Test/LinearAlgebra.hs:57-58)
bool columns_sorted(const int *d_ptr, const int *d_idx, int64_t ncols, hipStream_t s) {
  if (ncols == 0) return true;
  DBuf<int> flag(1);
  SPL_HIP(hipMemsetAsync(flag.get(), 0, sizeof(int), s));
  hipLaunchKernelGGL(sorted_check_kernel, dim3(blocks_for(ncols, 4)), dim3(256), 0, s, d_ptr, d_idx, ncols,
                     flag.get());
  int h = 0;
  SPL_HIP(hipMemcpyAsync(&h, flag.get(), sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  return h == 0;
}

// C = alpha A + beta B on device CSC arrays with sorted columns; vw = 2: packed complex values, scalars
// (alpha, alpha_im), (beta, beta_im)
static void lin_device_any(int vw, double alpha, double alpha_im, const int *Ap, const int *Ai, const double *Ax,
                           double beta, double beta_im, const int *Bp, const int *Bi, const double *Bx, int64_t ncols,
                           DBuf<int64_t> &Cp, DBuf<int> &Ci, DBuf<double> &Cx, int64_t *nnzC, hipStream_t s) {
  Cp.alloc((size_t)ncols + 1);
  DBuf<int> counts((size_t)ncols);
  const char *tile_env = getenv("SPL_LIN_TILED");
  const bool tiled = !(tile_env && tile_env[0] == '0');  // SPL_LIN_TILED=0: the thread-per-column kernels (ablation)
  const unsigned grid = tiled ? blocks_for(ncols, kLinCols) : blocks_for(ncols, 256);
  static std::atomic<uint64_t> lds_set{0};
  const size_t lds_count = lin_tile_lds_bytes<false, 1>(), lds_fill1 = lin_tile_lds_bytes<true, 1>(),
               lds_fill2 = lin_tile_lds_bytes<true, 2>();
  int lin_dev = 0;
  SPL_HIP(hipGetDevice(&lin_dev));
  if (tiled && !(lds_set.load(std::memory_order_acquire) >> (lin_dev & 63) & 1u)) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lin_tile_kernel<true, 1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fill1));
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lin_tile_kernel<true, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fill2));
    lds_set.fetch_or(1ull << (lin_dev & 63), std::memory_order_release);
  }
  if (ncols > 0) {
    if (tiled)
      hipLaunchKernelGGL((lin_tile_kernel<false, 1>), dim3(grid), dim3(kLinCols), lds_count, s, alpha,
                         alpha_im, Ap, Ai, Ax, beta, beta_im, Bp, Bi, Bx, ncols, counts.get(), (const int64_t *)nullptr,
                         (int *)nullptr, (double *)nullptr);
    else
      hipLaunchKernelGGL((lin_merge_kernel<false, 1>), dim3(grid), dim3(256), 0, s, alpha, alpha_im, Ap, Ai, Ax, beta,
                         beta_im, Bp, Bi, Bx, ncols, counts.get(), (const int64_t *)nullptr, (int *)nullptr,
                         (double *)nullptr);
  }
  exclusive_scan_i32_to_i64(counts.get(), Cp.get(), ncols, s);
  int64_t nz = 0;
  SPL_HIP(hipMemcpyAsync(&nz, Cp.get() + ncols, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  Ci.alloc((size_t)nz);
  Cx.alloc((size_t)nz * (size_t)vw);
  if (ncols > 0 && nz > 0) {
    if (tiled && vw == 1)
      hipLaunchKernelGGL((lin_tile_kernel<true, 1>), dim3(grid), dim3(kLinCols), lds_fill1, s, alpha,
                         alpha_im, Ap, Ai, Ax, beta, beta_im, Bp, Bi, Bx, ncols, (int *)nullptr, Cp.get(), Ci.get(), Cx.get());
    else if (tiled)
      hipLaunchKernelGGL((lin_tile_kernel<true, 2>), dim3(grid), dim3(kLinCols), lds_fill2, s, alpha,
                         alpha_im, Ap, Ai, Ax, beta, beta_im, Bp, Bi, Bx, ncols, (int *)nullptr, Cp.get(), Ci.get(), Cx.get());
    else if (vw == 1)
      hipLaunchKernelGGL((lin_merge_kernel<true, 1>), dim3(grid), dim3(256), 0, s, alpha, alpha_im, Ap, Ai, Ax, beta,
                         beta_im, Bp, Bi, Bx, ncols, (int *)nullptr, Cp.get(), Ci.get(), Cx.get());
    else
      hipLaunchKernelGGL((lin_merge_kernel<true, 2>), dim3(grid), dim3(256), 0, s, alpha, alpha_im, Ap, Ai, Ax, beta,
                         beta_im, Bp, Bi, Bx, ncols, (int *)nullptr, Cp.get(), Ci.get(), Cx.get());
  }
  SPL_HIP(hipStreamSynchronize(s));
  *nnzC = nz;
}

void lin_device(double alpha, const int *Ap, const int *Ai, const double *Ax, double beta, const int *Bp,
                const int *Bi, const double *Bx, int64_t ncols, DBuf<int64_t> &Cp, DBuf<int> &Ci,
                DBuf<double> &Cx, int64_t *nnzC, hipStream_t s) {
  lin_device_any(1, alpha, 0.0, Ap, Ai, Ax, beta, 0.0, Bp, Bi, Bx, ncols, Cp, Ci, Cx, nnzC, s);
}

void lin_device_z(const double alpha[2], const int *Ap, const int *Ai, const double *Az, const double beta[2],
                  const int *Bp, const int *Bi, const double *Bz, int64_t ncols, DBuf<int64_t> &Cp, DBuf<int> &Ci,
                  DBuf<double> &Cz, int64_t *nnzC, hipStream_t s) {
  lin_device_any(2, alpha[0], alpha[1], Ap, Ai, Az, beta[0], beta[1], Bp, Bi, Bz, ncols, Cp, Ci, Cz, nnzC, s);
}

// ---- kronecker / takeDiag (Sparse.hs:597-648) --------------------------------------------------
namespace {

// lengths of the columns of C = A (x) B: column ja * ncolsB + jb has len(A[:,ja]) * len(B[:,jb]) entries
__global__ void kron_count_kernel(const int *__restrict__ Ap, const int *__restrict__ Bp, int64_t ncolsA,
                                  int64_t ncolsB, int64_t *__restrict__ counts) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncolsA * ncolsB) return;
  const int64_t ja = j / ncolsB, jb = j % ncolsB;
  counts[j] = (int64_t)(Ap[ja + 1] - Ap[ja]) * (int64_t)(Bp[jb + 1] - Bp[jb]);
}

// one wavefront per output column: entry e of the column is (a = e / lenB, b = e % lenB) -> row
// ia * nrowsB + ib (ascending, as the rows of both operands ascend), value b * a
__global__ __launch_bounds__(256) void kron_fill_kernel(const int *__restrict__ Ap, const int *__restrict__ Ai,
                                                        const double *__restrict__ Ax,
                                                        const int *__restrict__ Bp, const int *__restrict__ Bi,
                                                        const double *__restrict__ Bx, int64_t ncolsA,
                                                        int64_t ncolsB, int nrowsB,
                                                        const int64_t *__restrict__ Cp, int *__restrict__ Ci,
                                                        double *__restrict__ Cx) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= ncolsA * ncolsB) return;
  const int64_t ja = j / ncolsB, jb = j % ncolsB;
  const int pa = Ap[ja], pb = Bp[jb], lb = Bp[jb + 1] - pb;
  const int64_t base = Cp[j], len = Cp[j + 1] - base;
  for (int64_t e = lane; e < len; e += 64) {
    const int ea = (int)(e / lb), eb = (int)(e % lb);
    Ci[base + e] = Ai[pa + ea] * nrowsB + Bi[pb + eb];
    Cx[base + e] = Bx[pb + eb] * Ax[pa + ea];  // U.map (* a) bs
  }
}

// d[c] = A[c, c] or 0: 8 lanes search column c (rows ascend, at most one hit)
__global__ __launch_bounds__(256) void take_diag_kernel(const int *__restrict__ Ap, const int *__restrict__ Ai,
                                                        const double *__restrict__ Ax, int n,
                                                        double *__restrict__ d) {
  const int c = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 3), part = threadIdx.x & 7;
  double v = 0.0;
  if (c < n)
    for (int p = Ap[c] + part; p < Ap[c + 1]; p += 8)
      if (Ai[p] == c) v = Ax[p];
  // exactly one lane can hold a hit; OR the bit patterns together (0.0 is all-zero bits)
  unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  bits |= __shfl_xor(bits, 1, 64);
  bits |= __shfl_xor(bits, 2, 64);
  bits |= __shfl_xor(bits, 4, 64);
  if (c < n && part == 0) d[c] = __longlong_as_double((long long)bits);
}

}  // namespace

// C = A (x) B on device CSC arrays; Cp is 64-bit (the caller checks the int32 seam)
void kronecker_device(int nrowsB, const int *Ap, const int *Ai, const double *Ax, int64_t ncolsA, const int *Bp,
                      const int *Bi, const double *Bx, int64_t ncolsB, DBuf<int64_t> &Cp, DBuf<int> &Ci,
                      DBuf<double> &Cx, int64_t *nnzC, hipStream_t s) {
  const int64_t nc = ncolsA * ncolsB;
  Cp.alloc((size_t)nc + 1);
  DBuf<int64_t> counts((size_t)(nc ? nc : 1));
  if (nc > 0)
    hipLaunchKernelGGL(kron_count_kernel, dim3(blocks_for(nc, 256)), dim3(256), 0, s, Ap, Bp, ncolsA, ncolsB,
                       counts.get());
  exclusive_scan_i64(counts.get(), Cp.get(), nc, s);
  int64_t nz = 0;
  SPL_HIP(hipMemcpyAsync(&nz, Cp.get() + nc, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  *nnzC = nz;
  if (nz >= 0x7fffffffLL) return;  // does not fit the int32 seam: the caller reports the overflow
  Ci.alloc((size_t)nz);
  Cx.alloc((size_t)nz);
  if (nz > 0)
    hipLaunchKernelGGL(kron_fill_kernel, dim3(blocks_for(nc, 4)), dim3(256), 0, s, Ap, Ai, Ax, Bp, Bi, Bx, ncolsA,
                       ncolsB, nrowsB, Cp.get(), Ci.get(), Cx.get());
  SPL_HIP(hipStreamSynchronize(s));
}

// ---- block assembly: hcat / vcat / fromBlocks / fromBlocksDiag (Sparse.hs:500-595) -----------------------
// The result's column c is the concatenation, in list order, of column c - col_off[b] of every block b that
// covers it, row indices shifted by row_off[b] (vcat's copyWithOffset, Sparse.hs:551-559; hcat is the case of
// disjoint column ranges, fromBlocks = vcat . map hcat places block (r, c) at the summed heights / widths).
// Blocks that share columns must be listed by ascending row offset (they are: vcat stacks in list order), so
// every result column ascends.
__global__ __launch_bounds__(256) void blocks_count_kernel(int ncols_b, const int *__restrict__ Bp, int col_off,
                                                          int *__restrict__ len) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < ncols_b) len[col_off + c] += Bp[c + 1] - Bp[c];  // one launch per block, in stream order: no race
}

// one wavefront per column of the block: copy it behind what earlier blocks put into that result column
__global__ __launch_bounds__(256) void blocks_copy_kernel(int ncols_b, const int *__restrict__ Bp, const int *__restrict__ Bi,
                                                         const double *__restrict__ Bx, int vw, int row_off, int col_off,
                                                         int64_t *__restrict__ cursor, int *__restrict__ Ci,
                                                         double *__restrict__ Cx) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= ncols_b) return;
  const int s = Bp[c], n = Bp[c + 1] - s;
  const int64_t dst = cursor[col_off + c];
  for (int t = lane; t < n; t += 64) {
    Ci[dst + t] = Bi[s + t] + row_off;
    for (int k = 0; k < vw; ++k) Cx[(dst + t) * vw + k] = Bx[(size_t)(s + t) * vw + k];
  }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) cursor[col_off + c] = dst + n;
}

void blocks_assemble_device(int nblocks, const int *ncols_b, const int *const *d_Bp, const int *const *d_Bi,
                            const double *const *d_Bx, int vw, const int *row_off, const int *col_off, int64_t ncolsC,
                            DBuf<int64_t> &Cp, DBuf<int> &Ci, DBuf<double> &Cx, int64_t *nnzC, hipStream_t s) {
  DBuf<int> len((size_t)ncolsC + 1);
  SPL_HIP(hipMemsetAsync(len.get(), 0, ((size_t)ncolsC + 1) * sizeof(int), s));
  for (int b = 0; b < nblocks; ++b)
    if (ncols_b[b] > 0)
      hipLaunchKernelGGL(blocks_count_kernel, dim3((unsigned)((ncols_b[b] + 255) / 256)), dim3(256), 0, s, ncols_b[b], d_Bp[b],
                         col_off[b], len.get());
  Cp.alloc((size_t)ncolsC + 1);
  exclusive_scan_i32_to_i64(len.get(), Cp.get(), ncolsC, s);
  int64_t nz = 0;
  SPL_HIP(hipMemcpyAsync(&nz, Cp.get() + ncolsC, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  *nnzC = nz;
  Ci.alloc((size_t)nz);
  Cx.alloc((size_t)nz * (size_t)vw);
  if (nz == 0) return;
  DBuf<int64_t> cursor((size_t)ncolsC + 1);
  SPL_HIP(hipMemcpyAsync(cursor.get(), Cp.get(), ((size_t)ncolsC + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
  for (int b = 0; b < nblocks; ++b)
    if (ncols_b[b] > 0)
      hipLaunchKernelGGL(blocks_copy_kernel, dim3((unsigned)((ncols_b[b] + 3) / 4)), dim3(256), 0, s, ncols_b[b], d_Bp[b], d_Bi[b],
                         d_Bx[b], vw, row_off[b], col_off[b], cursor.get(), Ci.get(), Cx.get());
  SPL_HIP(hipGetLastError());
  SPL_HIP(hipStreamSynchronize(s));
}

void take_diag_device(const int *Ap, const int *Ai, const double *Ax, int n, double *d, hipStream_t s) {
  if (n > 0)
    hipLaunchKernelGGL(take_diag_kernel, dim3((unsigned)(((size_t)n * 8 + 255) / 256)), dim3(256), 0, s, Ap, Ai, Ax,
                       n, d);
}

}  // namespace spl

// spgemm_z.hip — mm on `Matrix U.Vector (Complex Double)` (Sparse.hs:691-702 at the second SPECIALIZE
// instance, :456-457).
//
// The pattern of C = A B does not depend on the values, so it comes from the real SpGEMM of spgemm.hip run
// on the two patterns (its values are discarded).  The values are then accumulated where the reference
// accumulates them: for a column j of B, k ascending over its entries, C[i, j] <- C[i, j] + A[i, k] * B[k, j]
// with Data.Complex's product (x*x' - y*y') :+ (x*y' + y*x') and componentwise sum, starting from 0 — so
// every entry is bit-identical to the Haskell code (oracle: orc_mm_z).
//
// Kernel: one wavefront per CHUNK of at most 512 consecutive entries of a column of C.  The chunk's row
// indices and its complex accumulators live in LDS (10 KB per wavefront).  The wavefront walks the entries
// (k, b) of B[:, j] in order — the order of the sums — and for each of them its lanes take the entries of
// A[:, k] that fall into the chunk's row range side by side (distinct rows, hence distinct accumulators;
// the start is found by bisection when the column has more than one chunk), find the row among the chunk's
// indices by bisection in LDS and add the product.  LDS operations of one wavefront execute in program
// order, so the accumulation order per entry is the reference's whatever the lane.  Long columns simply
// have more chunks: no size classes, no accumulators in global memory.
#include "common.hpp"

namespace spl {
namespace {

constexpr int kChunk = 512;
constexpr int kWaves = 4;

__global__ __launch_bounds__(256) void chunk_count_kernel(const int64_t *__restrict__ Cp, int64_t ncols,
                                                          int *__restrict__ counts) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < ncols) counts[j] = (int)((Cp[j + 1] - Cp[j] + kChunk - 1) / kChunk);
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void mm_values_z_kernel(
    const int *__restrict__ Ap, const int *__restrict__ Ai, const double *__restrict__ Az,
    const int *__restrict__ Bp, const int *__restrict__ Bi, const double *__restrict__ Bz, int64_t ncolsB,
    const int64_t *__restrict__ Cp, const int *__restrict__ Ci, const int64_t *__restrict__ chunk_ptr,
    double *__restrict__ Cz) {
  __shared__ double acc[kWaves][2 * kChunk];
  __shared__ int rows[kWaves][kChunk];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t total = chunk_ptr[ncolsB];
  for (int64_t item = (int64_t)blockIdx.x * kWaves + w; item < total; item += (int64_t)gridDim.x * kWaves) {
    // the column of this chunk: the last j with chunk_ptr[j] <= item
    int64_t lo = 0, hi = ncolsB;
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (chunk_ptr[mid] <= item) lo = mid; else hi = mid;
    }
    const int64_t j = lo;
    const int64_t c0 = Cp[j], clen = Cp[j + 1] - c0;
    const int64_t first = c0 + (item - chunk_ptr[j]) * kChunk;
    const int len = (int)((c0 + clen - first) < kChunk ? (c0 + clen - first) : kChunk);
    for (int t = lane; t < len; t += 64) {
      rows[w][t] = Ci[first + t];
      acc[w][2 * t] = 0.0;      // SG.reset 0
      acc[w][2 * t + 1] = 0.0;
    }
    wave_sync();
    const int row_lo = rows[w][0], row_hi = rows[w][len - 1];
    const bool whole = clen <= kChunk;
    for (int q = Bp[j]; q < Bp[j + 1]; ++q) {  // ascending k: the order of the sums
      const int k = Bi[q];
      const double br = Bz[2 * (size_t)q], bi = Bz[2 * (size_t)q + 1];
      int a0 = Ap[k];
      const int a1 = Ap[k + 1];
      if (!whole) {  // first entry of A[:, k] with row >= row_lo
        int l = a0, h = a1;
        while (l < h) {
          const int m = (l + h) >> 1;
          if (Ai[m] < row_lo) l = m + 1; else h = m;
        }
        a0 = l;
      }
      for (int base = a0; base < a1; base += 64) {
        const int p = base + lane;
        const int i = p < a1 ? Ai[p] : 0x7fffffff;
        if (i <= row_hi) {
          int l = 0, h = len - 1;  // the row is one of the chunk's (the pattern of C holds every product)
          while (l < h) {
            const int m = (l + h) >> 1;
            if (rows[w][m] < i) l = m + 1; else h = m;
          }
          if (rows[w][l] == i) {
            const double ar = Az[2 * (size_t)p], ai = Az[2 * (size_t)p + 1];
            const double pr = ar * br - ai * bi;
            const double pi = ar * bi + ai * br;
            acc[w][2 * l] = acc[w][2 * l] + pr;  // \c a -> c + a * b
            acc[w][2 * l + 1] = acc[w][2 * l + 1] + pi;
          }
        }
        if (__ballot(i > row_hi) != 0ull) break;  // rows ascend: nothing further belongs to this chunk
      }
      wave_sync();
    }
    for (int t = lane; t < len; t += 64) {
      Cz[2 * (size_t)(first + t)] = acc[w][2 * t];
      Cz[2 * (size_t)(first + t) + 1] = acc[w][2 * t + 1];
    }
    wave_sync();
  }
}

}  // namespace

// C = A B on packed-complex device CSC arrays with sorted columns; Cz gets 2 * nnz(C) doubles
void spgemm_device_z(int64_t nrowsA, int64_t ncolsA, const int *Ap, const int *Ai, const double *Az, int64_t ncolsB,
                     const int *Bp, const int *Bi, const double *Bz, DBuf<int64_t> &Cp, DBuf<int> &Ci,
                     DBuf<double> &Cz, int64_t *nnzC, hipStream_t s) {
  {
    // pattern: the real kernels on the two patterns; they read nnz doubles of "values" — the first halves of
    // the packed arrays serve, the products are thrown away
    DBuf<double> unused;
    spgemm_device(nrowsA, ncolsA, Ap, Ai, Az, ncolsB, Bp, Bi, Bz, Cp, Ci, unused, nnzC, nullptr, s);
  }
  Cz.alloc((size_t)*nnzC * 2);
  if (*nnzC == 0 || ncolsB == 0) return;
  DBuf<int> counts((size_t)ncolsB);
  DBuf<int64_t> chunk_ptr((size_t)ncolsB + 1);
  hipLaunchKernelGGL(chunk_count_kernel, dim3((unsigned)((ncolsB + 255) / 256)), dim3(256), 0, s, Cp.get(), ncolsB,
                     counts.get());
  exclusive_scan_i32_to_i64(counts.get(), chunk_ptr.get(), ncolsB, s);
  int64_t total = 0;
  SPL_HIP(hipMemcpyAsync(&total, chunk_ptr.get() + ncolsB, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  const int64_t want = (total + kWaves - 1) / kWaves;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
  hipLaunchKernelGGL(mm_values_z_kernel, dim3(grid), dim3(256), 0, s, Ap, Ai, Az, Bp, Bi, Bz, ncolsB, Cp.get(), Ci.get(),
                     chunk_ptr.get(), Cz.get());
  SPL_HIP(hipStreamSynchronize(s));
  SPL_HIP(hipGetLastError());
}

}  // namespace spl

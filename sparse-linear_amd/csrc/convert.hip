// convert.hip — format plumbing in HBM: validation of a compressed 5-tuple,
// the order-preserving transpose (reference: Sparse.hs:301-329, which is also
// the CSC -> CSR converter) and the segmented key/value sort it
// and SpGEMM / compress share.
#include "common.hpp"

namespace spl {

namespace {

inline unsigned blocks_for(int64_t n, int per_block) {
  int64_t b = (n + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : b);
}

// ---- validation ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void validate_ptr_kernel(const int *__restrict__ ptr, int64_t nmajor,
                                                           int64_t nnz, int *__restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (i == 0) {
    if (ptr[0] != 0) atomicOr(flag, 1);
    if ((int64_t)ptr[nmajor] != nnz) atomicOr(flag, 2);
  }
  for (; i < nmajor; i += stride)
    if (ptr[i] > ptr[i + 1] || ptr[i] < 0) atomicOr(flag, 4);
}

__global__ __launch_bounds__(256) void validate_idx_kernel(const int *__restrict__ idx, int64_t nnz,
                                                           int64_t nminor, int *__restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < nnz; i += stride) {
    const int v = idx[i];
    bad |= (v < 0) || ((int64_t)v >= nminor);
  }
  if (bad) atomicOr(flag, 8);
}

// ---- transpose ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void histogram_kernel(const int *__restrict__ idx, int64_t nnz,
                                                        int *__restrict__ counts) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) atomicAdd(&counts[idx[i]], 1);
}

// one wavefront per major slice: scatter its entries to the cursor of their minor index
__global__ __launch_bounds__(256) void transpose_fill_kernel(
    const int *__restrict__ ptr, const int *__restrict__ idx, const double *__restrict__ val,
    int64_t nmajor, const int64_t *__restrict__ out_ptr, int *__restrict__ cursor,
    int *__restrict__ out_idx, double *__restrict__ out_val) {
  const int lane = threadIdx.x & 63;
  const int64_t slice = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (slice >= nmajor) return;
  const int s = ptr[slice], e = ptr[slice + 1];
  for (int k = s + lane; k < e; k += 64) {
    const int r = idx[k];
    const int64_t pos = out_ptr[r] + (int64_t)atomicAdd(&cursor[r], 1);
    out_idx[pos] = (int)slice;
    out_val[pos] = val[k];
  }
}

// ---- segmented sort -----------------------------------------------------------------
// ascending-only bitonic network: stage sizes k = 2,4,..; first substage pairs
// i with i ^ (k-1) ("flip"), later substages i with i ^ j.  Positions >= len act
// as +inf, so a pair whose upper index is >= len is simply skipped.

template <typename K>
__device__ inline K key_max();
template <>
__device__ inline int key_max<int>() { return 0x7fffffff; }
template <>
__device__ inline int64_t key_max<int64_t>() { return 0x7fffffffffffffffLL; }
template <>
__device__ inline unsigned key_max<unsigned>() { return 0xffffffffu; }

// len <= 64: one wavefront, registers + shuffles
template <typename K>
__device__ inline void wave_sort_segment(K *key, double *val, int64_t base, int len) {
  const int lane = threadIdx.x & 63;
  K k = lane < len ? key[base + lane] : key_max<K>();
  double v = lane < len ? val[base + lane] : 0.0;
  int n2 = 2;
  while (n2 < len) n2 <<= 1;
  for (int size = 2; size <= n2; size <<= 1) {
    {
      const int partner = lane ^ (size - 1);
      const K pk = __shfl(k, partner, 64);
      const double pv = __shfl(v, partner, 64);
      const bool lower = lane < partner;
      const bool take = lower ? (pk < k) : (pk > k);
      if (take) { k = pk; v = pv; }
    }
    for (int j = size >> 2; j > 0; j >>= 1) {
      const int partner = lane ^ j;
      const K pk = __shfl(k, partner, 64);
      const double pv = __shfl(v, partner, 64);
      const bool lower = lane < partner;
      const bool take = lower ? (pk < k) : (pk > k);
      if (take) { k = pk; v = pv; }
    }
  }
  if (lane < len) { key[base + lane] = k; val[base + lane] = v; }
}

template <typename K>
__global__ __launch_bounds__(256) void segsort_small_kernel(const int64_t *__restrict__ ptr,
                                                            int64_t nseg, K *__restrict__ key,
                                                            double *__restrict__ val,
                                                            int64_t *__restrict__ big_list,
                                                            int *__restrict__ big_count, int64_t max_len) {
  const int lane = threadIdx.x & 63;
  const int64_t seg = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (seg >= nseg) return;
  const int64_t s = ptr[seg], e = ptr[seg + 1];
  const int64_t len = e - s;
  if (len <= 1) return;
  if (len <= 64) {
    wave_sort_segment(key, val, s, (int)len);
  } else if (lane == 0 && len <= max_len) {
    big_list[atomicAdd(big_count, 1)] = seg;
  }
}

constexpr int kSortLdsCap = 4096;

// len > 64: one workgroup per listed segment; LDS when it fits, global otherwise
template <typename K>
__global__ __launch_bounds__(256) void segsort_big_kernel(const int64_t *__restrict__ ptr,
                                                          const int64_t *__restrict__ big_list,
                                                          K *__restrict__ key,
                                                          double *__restrict__ val) {
  __shared__ K skey[kSortLdsCap];
  __shared__ double sval[kSortLdsCap];
  const int64_t seg = big_list[blockIdx.x];
  const int64_t s = ptr[seg];
  const int64_t len = ptr[seg + 1] - s;
  int64_t n2 = 2;
  while (n2 < len) n2 <<= 1;
  const bool in_lds = len <= kSortLdsCap;
  if (in_lds) {
    for (int64_t i = threadIdx.x; i < len; i += blockDim.x) { skey[i] = key[s + i]; sval[i] = val[s + i]; }
  }
  __syncthreads();
  for (int64_t size = 2; size <= n2; size <<= 1) {
    for (int64_t j = size >> 1; j > 0; j >>= 1) {
      const bool flip = (j == (size >> 1));
      for (int64_t t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
        // t-th pair of this substage: lower index lo has bit j clear
        const int64_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int64_t hi = flip ? (lo ^ (size - 1)) : (lo | j);
        if (hi < len) {
          if (in_lds) {
            const K a = skey[lo], b = skey[hi];
            if (b < a) {
              skey[lo] = b; skey[hi] = a;
              const double va = sval[lo]; sval[lo] = sval[hi]; sval[hi] = va;
            }
          } else {
            const K a = key[s + lo], b = key[s + hi];
            if (b < a) {
              key[s + lo] = b; key[s + hi] = a;
              const double va = val[s + lo]; val[s + lo] = val[s + hi]; val[s + hi] = va;
            }
          }
        }
      }
      __syncthreads();
    }
  }
  if (in_lds) {
    for (int64_t i = threadIdx.x; i < len; i += blockDim.x) { key[s + i] = skey[i]; val[s + i] = sval[i]; }
  }
}

__global__ __launch_bounds__(256) void max_len_kernel(const int64_t *__restrict__ ptr, int64_t nseg,
                                                      unsigned long long *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long m = 0;
  for (; i < nseg; i += stride) {
    const unsigned long long l = (unsigned long long)(ptr[i + 1] - ptr[i]);
    m = l > m ? l : m;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const unsigned long long o = __shfl_xor(m, d, 64);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

// share of entries whose 128-byte x line (16 doubles) is not touched by the previous row:
// ~1 for uniformly random columns, ~1/16 for banded / stencil rows
__global__ __launch_bounds__(256) void new_line_kernel(const int64_t *__restrict__ ptr,
                                                       const int *__restrict__ col, int64_t nrows,
                                                       int64_t stride_rows,
                                                       unsigned long long *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = 1 + i * stride_rows;
  unsigned long long fresh = 0, total = 0;
  if (r < nrows) {
    const int64_t ps = ptr[r - 1], pe = ptr[r], ce = ptr[r + 1];
    int64_t q = ps;
    for (int64_t k = pe; k < ce; ++k) {
      const int line = col[k] >> 4;
      while (q < pe && (col[q] >> 4) < line) ++q;
      fresh += !(q < pe && (col[q] >> 4) == line);
      ++total;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    fresh += __shfl_xor(fresh, d, 64);
    total += __shfl_xor(total, d, 64);
  }
  if ((threadIdx.x & 63) == 0 && total) { atomicAdd(&out[0], fresh); atomicAdd(&out[1], total); }
}

}  // namespace

int validate_compressed(const int *d_ptr, const int *d_idx, int64_t nmajor, int64_t nminor,
                        int64_t nnz, hipStream_t s) {
  DBuf<int> flag(1);
  SPL_HIP(hipMemsetAsync(flag.get(), 0, sizeof(int), s));
  unsigned g1 = blocks_for(nmajor, 256);
  if (g1 > 4096) g1 = 4096;
  hipLaunchKernelGGL(validate_ptr_kernel, dim3(g1), dim3(256), 0, s, d_ptr, nmajor, nnz, flag.get());
  int h = 0;
  SPL_HIP(hipMemcpyAsync(&h, flag.get(), sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  if (h) return SPL_ERROR_invalid_matrix;  // do not touch idx with untrusted pointers
  if (nnz > 0) {
    unsigned g2 = blocks_for(nnz, 256);
    if (g2 > 8192) g2 = 8192;
    hipLaunchKernelGGL(validate_idx_kernel, dim3(g2), dim3(256), 0, s, d_idx, nnz, nminor, flag.get());
    SPL_HIP(hipMemcpyAsync(&h, flag.get(), sizeof(int), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
  }
  return h ? SPL_ERROR_invalid_matrix : SPL_OK;
}

template <typename K>
static void segmented_sort_impl(const int64_t *d_ptr64, int64_t nseg, K *d_key, double *d_val,
                                hipStream_t s, int64_t max_len = 0x7fffffffffffffffLL) {
  if (nseg <= 0) return;
  DBuf<int64_t> big_list((size_t)nseg);
  DBuf<int> big_count(1);
  SPL_HIP(hipMemsetAsync(big_count.get(), 0, sizeof(int), s));
  hipLaunchKernelGGL(segsort_small_kernel<K>, dim3(blocks_for(nseg, 4)), dim3(256), 0, s, d_ptr64, nseg,
                     d_key, d_val, big_list.get(), big_count.get(), max_len);
  int nbig = 0;
  SPL_HIP(hipMemcpyAsync(&nbig, big_count.get(), sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  if (nbig > 0) {
    hipLaunchKernelGGL(segsort_big_kernel<K>, dim3((unsigned)nbig), dim3(256), 0, s, d_ptr64,
                       big_list.get(), d_key, d_val);
    SPL_HIP(hipStreamSynchronize(s));
  }
}

void segmented_sort_pairs(const int64_t *d_ptr64, int64_t nseg, int *d_key, double *d_val,
                          hipStream_t s) {
  segmented_sort_impl<int>(d_ptr64, nseg, d_key, d_val, s);
}
void segmented_sort_pairs_capped(const int64_t *d_ptr64, int64_t nseg, int *d_key, double *d_val,
                                 int64_t max_len, hipStream_t s) {
  segmented_sort_impl<int>(d_ptr64, nseg, d_key, d_val, s, max_len);
}
void segmented_sort_pairs64(const int64_t *d_ptr64, int64_t nseg, int64_t *d_key, double *d_val,
                            hipStream_t s) {
  segmented_sort_impl<int64_t>(d_ptr64, nseg, d_key, d_val, s);
}
void segmented_sort_pairs_u32(const int64_t *d_ptr64, int64_t nseg, unsigned *d_key, double *d_val,
                              hipStream_t s) {
  segmented_sort_impl<unsigned>(d_ptr64, nseg, d_key, d_val, s);
}

void transpose_compressed(const int *d_ptr, const int *d_idx, const double *d_val, int64_t nmajor,
                          int64_t nminor, int64_t nnz, int64_t *out_ptr64, int *out_idx,
                          double *out_val, hipStream_t s) {
  DBuf<int> counts((size_t)nminor);
  SPL_HIP(hipMemsetAsync(counts.get(), 0, (size_t)(nminor ? nminor : 1) * sizeof(int), s));
  if (nnz > 0) {
    unsigned g = blocks_for(nnz, 256);
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(histogram_kernel, dim3(g), dim3(256), 0, s, d_idx, nnz, counts.get());
  }
  exclusive_scan_i32_to_i64(counts.get(), out_ptr64, nminor, s);
  if (nnz == 0) return;
  SPL_HIP(hipMemsetAsync(counts.get(), 0, (size_t)(nminor ? nminor : 1) * sizeof(int), s));
  hipLaunchKernelGGL(transpose_fill_kernel, dim3(blocks_for(nmajor, 4)), dim3(256), 0, s, d_ptr, d_idx,
                     d_val, nmajor, out_ptr64, counts.get(), out_idx, out_val);
  // atomics hand out slots in arbitrary order; restore ascending old-major order
  segmented_sort_pairs(out_ptr64, nminor, out_idx, out_val, s);
  SPL_HIP(hipStreamSynchronize(s));
}

void finalize_matrix(Matrix *m, hipStream_t s) {
  const int64_t nl = m->nrows_local;
  // SPL_FORCE_PTR64=1 exercises the 64-bit row-pointer kernels on small matrices (tests only)
  const char *force64 = getenv("SPL_FORCE_PTR64");
  if (m->nnz < (int64_t)0x7fffffff && !(force64 && force64[0] == '1')) {
    m->rowptr.alloc((size_t)nl + 1);
    narrow_i64_to_i32(m->rowptr64.get(), m->rowptr.get(), nl + 1, s);
  }
  DBuf<unsigned long long> mx(1);
  SPL_HIP(hipMemsetAsync(mx.get(), 0, sizeof(unsigned long long), s));
  if (nl > 0) {
    unsigned g = blocks_for(nl, 256);
    if (g > 256) g = 256;  // one atomicMax per wavefront on ONE address, ~12 ns apiece: 4 096 workgroups took 0.19 ms
    hipLaunchKernelGGL(max_len_kernel, dim3(g), dim3(256), 0, s, m->rowptr64.get(), nl, mx.get());
  }
  unsigned long long h = 0;
  SPL_HIP(hipMemcpyAsync(&h, mx.get(), sizeof(h), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  m->max_row_len = (int64_t)h;
  m->new_line_fraction = -1.0;  // measured when a kernel choice first needs it (measure_locality)
}

// samples about 64k rows for the locality estimate that steers choose_blocking(); once per matrix
void measure_locality(Matrix *m, hipStream_t s) {
  if (m->new_line_fraction >= 0.0) return;
  const int64_t nl = m->nrows_local;
  m->new_line_fraction = 0.0;
  if (nl > 1 && m->nnz > 0) {
    DBuf<unsigned long long> acc(2);
    SPL_HIP(hipMemsetAsync(acc.get(), 0, 2 * sizeof(unsigned long long), s));
    const int64_t stride = nl / 65536 > 0 ? nl / 65536 : 1;
    const int64_t samples = (nl - 1 + stride - 1) / stride;
    hipLaunchKernelGGL(new_line_kernel, dim3(blocks_for(samples, 256)), dim3(256), 0, s, m->rowptr64.get(),
                       m->colidx.get(), nl, stride, acc.get());
    unsigned long long ha[2] = {0, 0};
    SPL_HIP(hipMemcpyAsync(ha, acc.get(), sizeof(ha), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    if (ha[1]) m->new_line_fraction = (double)ha[0] / (double)ha[1];
  }
}

}  // namespace spl

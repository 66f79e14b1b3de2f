// spgemm.hip — C = A * B on CSC operands (reference: mm, Sparse.hs:691-702, with
// the dense sparse accumulator of Data/Vector/Sparse/ScatterGather.hs).
//
// Reference semantics kept exactly:
//   * column j of C has the UNION pattern of { A[:,k] : B[k,j] stored }; numerical
//     cancellation keeps a stored zero (the mask is set regardless of the value);
//   * row indices ascend inside a column; pointers are the exclusive prefix sum;
//   * C[i,j] = fold (\acc k -> acc + A[i,k]*B[k,j]) 0 over k ascending, each
//     multiply and add separately rounded — the kernels below walk k SEQUENTIALLY
//     per output column and parallelise over the (distinct) rows of A[:,k], so a
//     given accumulator receives its contributions in ascending-k order: values
//     are bit-identical to the reference order, not merely within tolerance.
//
// MI355X design: the reference's O(nrows) dense accumulator per column is replaced
// by an accumulator sized to the column's work.  Columns are binned by their
// number of intermediate products (an upper bound of their nnz):
//   bin S  (<= 256 products)  one wavefront per column, 512-slot hash table in LDS
//   bin M  (<= 4096 products) one workgroup per column, 8192-slot hash table in LDS
//   bin L  (more)             one workgroup per column, dense accumulator in HBM from
//                             a small pool (the reference's own data structure, but only
//                             for the heavy columns), gathered in row order
// Two passes (symbolic count, exclusive scan to 64-bit pointers, numeric fill),
// then a segment sort by row index of the hash-table columns.  HBM-bound /
// latency-bound integer + fp64 work; no MFMA (no dense contraction).
#include "common.hpp"

namespace spl {

namespace {

constexpr int kEmpty = -1;
constexpr int kSmallProducts = 256, kSmallTable = 512;
constexpr int kMediumProducts = 4096, kMediumTable = 8192;
constexpr int kMaxPool = 512;

inline unsigned blocks_for(int64_t n, int per_block) {
  int64_t b = (n + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : b);
}

struct Csc {
  const int *p;
  const int *i;
  const double *x;
};

__global__ __launch_bounds__(256) void products_kernel(Csc A, Csc B, int64_t ncolsB,
                                                       int64_t *__restrict__ nprod,
                                                       int64_t *__restrict__ medium_list,
                                                       int64_t *__restrict__ large_list,
                                                       int *__restrict__ list_counts) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncolsB) return;
  int64_t n = 0;
  for (int q = B.p[j]; q < B.p[j + 1]; ++q) {
    const int k = B.i[q];
    n += A.p[k + 1] - A.p[k];
  }
  nprod[j] = n;
  if (n > kMediumProducts) large_list[atomicAdd(&list_counts[1], 1)] = j;
  else if (n > kSmallProducts) medium_list[atomicAdd(&list_counts[0], 1)] = j;
}

template <int TABLE>
__device__ inline int hash_slot(int row) {
  return (int)(((unsigned)row * 0x9E3779B1u) >> 7) & (TABLE - 1);
}

// insert `row`, return its slot (linear probing; the table never fills: TABLE >= 2*products)
template <int TABLE>
__device__ inline int hash_insert(int *keys, int row) {
  int slot = hash_slot<TABLE>(row);
  while (true) {
    const int old = atomicCAS(&keys[slot], kEmpty, row);
    if (old == kEmpty || old == row) return slot;
    slot = (slot + 1) & (TABLE - 1);
  }
}

// bin S: one wavefront per column of B
template <bool NUMERIC>
__global__ __launch_bounds__(256) void spgemm_wave_kernel(Csc A, Csc B, int64_t ncolsB,
                                                          const int64_t *__restrict__ nprod,
                                                          int *__restrict__ counts,
                                                          const int64_t *__restrict__ Cp,
                                                          int *__restrict__ Ci, double *__restrict__ Cx) {
  __shared__ int keys_all[4][kSmallTable];
  __shared__ double vals_all[NUMERIC ? 4 : 1][NUMERIC ? kSmallTable : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 4 + wave;
  if (j >= ncolsB) return;
  const int64_t np = nprod[j];
  if (np > kSmallProducts) return;
  if (np == 0) {
    if (!NUMERIC && lane == 0) counts[j] = 0;
    return;
  }
  int *keys = keys_all[wave];
  double *vals = vals_all[NUMERIC ? wave : 0];
  for (int t = lane; t < kSmallTable; t += 64) {
    keys[t] = kEmpty;
    if (NUMERIC) vals[t] = 0.0;  // SG.reset 0
  }
  __builtin_amdgcn_wave_barrier();
  const int qs = B.p[j], qe = B.p[j + 1];
  for (int q = qs; q < qe; ++q) {  // k ascending: the reference's iforM_ colB
    const int k = B.i[q];
    const double b = NUMERIC ? B.x[q] : 0.0;
    const int ps = A.p[k], pe = A.p[k + 1];
    for (int p = ps + lane; p < pe; p += 64) {
      const int slot = hash_insert<kSmallTable>(keys, A.i[p]);
      if (NUMERIC) vals[slot] = vals[slot] + A.x[p] * b;  // \c a -> c + a * b
    }
    __builtin_amdgcn_wave_barrier();
  }
  // extraction in slot order; rows are sorted afterwards
  int64_t base = NUMERIC ? Cp[j] : 0;
  int total = 0;
  for (int t0 = 0; t0 < kSmallTable; t0 += 64) {
    const int key = keys[t0 + lane];
    const bool used = key != kEmpty;
    const unsigned long long m = __ballot(used);
    if (NUMERIC && used) {
      const int off = __popcll(m & ((1ull << lane) - 1ull));
      Ci[base + total + off] = key;
      Cx[base + total + off] = vals[t0 + lane];
    }
    total += __popcll(m);
  }
  if (!NUMERIC && lane == 0) counts[j] = total;
}

// bin M: one workgroup per listed column
template <bool NUMERIC>
__global__ __launch_bounds__(256) void spgemm_block_kernel(Csc A, Csc B,
                                                           const int64_t *__restrict__ list,
                                                           int *__restrict__ counts,
                                                           const int64_t *__restrict__ Cp,
                                                           int *__restrict__ Ci, double *__restrict__ Cx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int *keys = reinterpret_cast<int *>(smem);
  double *vals = reinterpret_cast<double *>(smem + kMediumTable * sizeof(int));
  __shared__ int wave_counts[4];
  __shared__ int running;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j = list[blockIdx.x];
  for (int t = threadIdx.x; t < kMediumTable; t += 256) {
    keys[t] = kEmpty;
    if (NUMERIC) vals[t] = 0.0;
  }
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  const int qs = B.p[j], qe = B.p[j + 1];
  for (int q = qs; q < qe; ++q) {
    const int k = B.i[q];
    const double b = NUMERIC ? B.x[q] : 0.0;
    const int ps = A.p[k], pe = A.p[k + 1];
    for (int p = ps + (int)threadIdx.x; p < pe; p += 256) {
      const int slot = hash_insert<kMediumTable>(keys, A.i[p]);
      if (NUMERIC) vals[slot] = vals[slot] + A.x[p] * b;
    }
    if (NUMERIC) __syncthreads();  // next k may hit the same accumulator from another wavefront
  }
  __syncthreads();
  const int64_t base = NUMERIC ? Cp[j] : 0;
  for (int t0 = 0; t0 < kMediumTable; t0 += 256) {
    const int key = keys[t0 + threadIdx.x];
    const bool used = key != kEmpty;
    const unsigned long long m = __ballot(used);
    if (lane == 0) wave_counts[wave] = __popcll(m);
    __syncthreads();
    int off = running;
    for (int w = 0; w < wave; ++w) off += wave_counts[w];
    if (NUMERIC && used) {
      off += __popcll(m & ((1ull << lane) - 1ull));
      Ci[base + off] = key;
      Cx[base + off] = vals[t0 + threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) running += wave_counts[0] + wave_counts[1] + wave_counts[2] + wave_counts[3];
    __syncthreads();
  }
  if (!NUMERIC && threadIdx.x == 0) counts[j] = running;
}

// bin L: persistent workgroups, each owning one dense accumulator of the pool
template <bool NUMERIC>
__global__ __launch_bounds__(256) void spgemm_dense_kernel(Csc A, Csc B, int64_t nrowsA,
                                                           const int64_t *__restrict__ list, int nlist,
                                                           unsigned char *__restrict__ pool_flags,
                                                           double *__restrict__ pool_vals,
                                                           int *__restrict__ counts,
                                                           const int64_t *__restrict__ Cp,
                                                           int *__restrict__ Ci, double *__restrict__ Cx) {
  __shared__ int wave_counts[4];
  __shared__ int64_t running;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char *flags = pool_flags + (size_t)blockIdx.x * (size_t)nrowsA;
  double *w = NUMERIC ? pool_vals + (size_t)blockIdx.x * (size_t)nrowsA : nullptr;
  for (int li = blockIdx.x; li < nlist; li += gridDim.x) {
    const int64_t j = list[li];
    if (threadIdx.x == 0) running = 0;
    const int qs = B.p[j], qe = B.p[j + 1];
    for (int q = qs; q < qe; ++q) {
      const int k = B.i[q];
      const double b = NUMERIC ? B.x[q] : 0.0;
      const int ps = A.p[k], pe = A.p[k + 1];
      for (int p = ps + (int)threadIdx.x; p < pe; p += 256) {
        const int r = A.i[p];
        flags[r] = 1;
        if (NUMERIC) w[r] = w[r] + A.x[p] * b;
      }
      __syncthreads();
    }
    // gather in row order (ScatterGather.hs:97-147), clearing the accumulator as we go
    const int64_t base = NUMERIC ? Cp[j] : 0;
    for (int64_t r0 = 0; r0 < nrowsA; r0 += 256) {
      const int64_t r = r0 + threadIdx.x;
      const bool used = r < nrowsA && flags[r] != 0;
      const unsigned long long m = __ballot(used);
      if (lane == 0) wave_counts[wave] = __popcll(m);
      __syncthreads();
      int64_t off = running;
      for (int ww = 0; ww < wave; ++ww) off += wave_counts[ww];
      if (used) {
        if (NUMERIC) {
          off += __popcll(m & ((1ull << lane) - 1ull));
          Ci[base + off] = (int)r;
          Cx[base + off] = w[r];
          w[r] = 0.0;
        }
        flags[r] = 0;
      }
      __syncthreads();
      if (threadIdx.x == 0) running += wave_counts[0] + wave_counts[1] + wave_counts[2] + wave_counts[3];
      __syncthreads();
    }
    if (!NUMERIC && threadIdx.x == 0) counts[j] = (int)running;
    __syncthreads();
  }
}

}  // namespace

// C = A B; all pointers are device pointers.  Cp is 64-bit (nnz(C) may exceed 2^31).
void spgemm_device(int64_t nrowsA, int64_t ncolsA, const int *Ap, const int *Ai, const double *Ax,
                   int64_t ncolsB, const int *Bp, const int *Bi, const double *Bx, DBuf<int64_t> &Cp,
                   DBuf<int> &Ci, DBuf<double> &Cx, int64_t *nnzC, int64_t *products, hipStream_t s) {
  (void)ncolsA;
  Csc A{Ap, Ai, Ax}, B{Bp, Bi, Bx};
  Cp.alloc((size_t)ncolsB + 1);
  *nnzC = 0;
  if (products) *products = 0;
  if (ncolsB == 0) {
    SPL_HIP(hipMemsetAsync(Cp.get(), 0, sizeof(int64_t), s));
    Ci.alloc(0);
    Cx.alloc(0);
    SPL_HIP(hipStreamSynchronize(s));
    return;
  }
  DBuf<int64_t> nprod((size_t)ncolsB), medium_list((size_t)ncolsB), large_list((size_t)ncolsB);
  DBuf<int> list_counts(2), counts((size_t)ncolsB);
  SPL_HIP(hipMemsetAsync(list_counts.get(), 0, 2 * sizeof(int), s));
  hipLaunchKernelGGL(products_kernel, dim3(blocks_for(ncolsB, 256)), dim3(256), 0, s, A, B, ncolsB,
                     nprod.get(), medium_list.get(), large_list.get(), list_counts.get());
  int hc[2] = {0, 0};
  SPL_HIP(hipMemcpyAsync(hc, list_counts.get(), 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  const int nmedium = hc[0], nlarge = hc[1];
  if (products) {
    DBuf<int64_t> pscan((size_t)ncolsB + 1);
    exclusive_scan_i64(nprod.get(), pscan.get(), ncolsB, s);
    SPL_HIP(hipMemcpy(products, pscan.get() + ncolsB, sizeof(int64_t), hipMemcpyDeviceToHost));
  }
  const size_t medium_lds_sym = kMediumTable * sizeof(int);
  const size_t medium_lds_num = kMediumTable * (sizeof(int) + sizeof(double));
  static bool attr_set = false;
  if (!attr_set) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spgemm_block_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)medium_lds_num));
    attr_set = true;
  }
  int pool = nlarge < kMaxPool ? nlarge : kMaxPool;
  {  // keep the accumulator pool under ~8 GB
    const int64_t cap = (int64_t)8e9 / (9 * (nrowsA > 0 ? nrowsA : 1));
    if (pool > cap) pool = (int)(cap < 1 ? 1 : cap);
  }
  DBuf<unsigned char> pool_flags;
  DBuf<double> pool_vals;
  if (nlarge > 0) {
    pool_flags.alloc((size_t)pool * (size_t)nrowsA);
    pool_vals.alloc((size_t)pool * (size_t)nrowsA);
    SPL_HIP(hipMemsetAsync(pool_flags.get(), 0, (size_t)pool * (size_t)nrowsA, s));
    SPL_HIP(hipMemsetAsync(pool_vals.get(), 0, (size_t)pool * (size_t)nrowsA * sizeof(double), s));
  }

  // ---- symbolic: nnz per column
  hipLaunchKernelGGL(spgemm_wave_kernel<false>, dim3(blocks_for(ncolsB, 4)), dim3(256), 0, s, A, B, ncolsB,
                     nprod.get(), counts.get(), (const int64_t *)nullptr, (int *)nullptr, (double *)nullptr);
  if (nmedium > 0)
    hipLaunchKernelGGL(spgemm_block_kernel<false>, dim3((unsigned)nmedium), dim3(256), medium_lds_sym, s, A, B,
                       medium_list.get(), counts.get(), (const int64_t *)nullptr, (int *)nullptr,
                       (double *)nullptr);
  if (nlarge > 0)
    hipLaunchKernelGGL(spgemm_dense_kernel<false>, dim3((unsigned)pool), dim3(256), 0, s, A, B, nrowsA,
                       large_list.get(), nlarge, pool_flags.get(), (double *)nullptr, counts.get(),
                       (const int64_t *)nullptr, (int *)nullptr, (double *)nullptr);
  exclusive_scan_i32_to_i64(counts.get(), Cp.get(), ncolsB, s);
  int64_t nz = 0;
  SPL_HIP(hipMemcpyAsync(&nz, Cp.get() + ncolsB, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  Ci.alloc((size_t)nz);
  Cx.alloc((size_t)nz);
  *nnzC = nz;
  if (nz == 0) return;

  // ---- numeric
  hipLaunchKernelGGL(spgemm_wave_kernel<true>, dim3(blocks_for(ncolsB, 4)), dim3(256), 0, s, A, B, ncolsB,
                     nprod.get(), (int *)nullptr, Cp.get(), Ci.get(), Cx.get());
  if (nmedium > 0)
    hipLaunchKernelGGL(spgemm_block_kernel<true>, dim3((unsigned)nmedium), dim3(256), medium_lds_num, s, A, B,
                       medium_list.get(), (int *)nullptr, Cp.get(), Ci.get(), Cx.get());
  if (nlarge > 0)
    hipLaunchKernelGGL(spgemm_dense_kernel<true>, dim3((unsigned)pool), dim3(256), 0, s, A, B, nrowsA,
                       large_list.get(), nlarge, pool_flags.get(), pool_vals.get(), (int *)nullptr, Cp.get(),
                       Ci.get(), Cx.get());
  SPL_HIP(hipGetLastError());
  // hash-table columns come out in slot order: sort them by row (dense-bin columns
  // are already ascending and longer ones are skipped by the cap)
  segmented_sort_pairs_capped(Cp.get(), ncolsB, Ci.get(), Cx.get(), kMediumProducts, s);
  SPL_HIP(hipStreamSynchronize(s));
}

}  // namespace spl

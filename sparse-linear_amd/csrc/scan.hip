// scan.hip — exclusive prefix sums (computePtrs' `scanl (+) 0`, Sparse.hs:282-291,
// and every pointer array this backend builds).  Three-phase block scan,
// 64-wide wavefront shuffles, 64-bit accumulation throughout.
#include "common.hpp"

namespace spl {

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;  // 2048 items per workgroup

__device__ inline int64_t wave_inclusive_scan(int64_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int64_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// phase 1: local exclusive scan of one tile, tile total to sums[blockIdx]
template <typename TIn>
__global__ __launch_bounds__(kScanThreads) void scan_tiles(const TIn *__restrict__ in,
                                                           int64_t *__restrict__ out,
                                                           int64_t *__restrict__ sums, int64_t n) {
  __shared__ int64_t wave_tot[kScanThreads / 64];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int64_t item[kScanItems];
  int64_t tsum = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    int64_t idx = base + i;
    item[i] = idx < n ? (int64_t)in[idx] : 0;
    tsum += item[i];
  }
  int64_t incl = wave_inclusive_scan(tsum);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int64_t wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    if (w < wave) wave_off += wave_tot[w];
    total += wave_tot[w];
  }
  int64_t run = wave_off + incl - tsum;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    int64_t idx = base + i;
    if (idx < n) out[idx] = run;
    run += item[i];
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// phase 3: add the scanned tile offsets; last entry = grand total
__global__ __launch_bounds__(kScanThreads) void scan_add_offsets(int64_t *__restrict__ out,
                                                                 const int64_t *__restrict__ tile_off,
                                                                 int64_t n, int64_t ntiles) {
  const int64_t tile = blockIdx.x;
  const int64_t off = tile_off[tile];
  const int64_t base = tile * kScanTile;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    int64_t idx = base + (int64_t)i * kScanThreads + threadIdx.x;
    if (idx < n) out[idx] += off;
  }
  if (tile == 0 && threadIdx.x == 0) out[n] = tile_off[ntiles];
}

__global__ void scan_write_total(int64_t *out, const int64_t *sums, int64_t n) { out[n] = sums[0]; }

template <typename TIn>
void scan_impl(const TIn *d_in, int64_t *d_out, int64_t n, hipStream_t s) {
  if (n <= 0) {
    SPL_HIP(hipMemsetAsync(d_out, 0, sizeof(int64_t), s));
    return;
  }
  const int64_t ntiles = (n + kScanTile - 1) / kScanTile;
  DBuf<int64_t> sums((size_t)ntiles);
  hipLaunchKernelGGL(scan_tiles<TIn>, dim3((unsigned)ntiles), dim3(kScanThreads), 0, s, d_in, d_out,
                     sums.get(), n);
  if (ntiles == 1) {
    hipLaunchKernelGGL(scan_write_total, dim3(1), dim3(1), 0, s, d_out, sums.get(), n);
  } else {
    DBuf<int64_t> offs((size_t)ntiles + 1);
    scan_impl<int64_t>(sums.get(), offs.get(), ntiles, s);
    hipLaunchKernelGGL(scan_add_offsets, dim3((unsigned)ntiles), dim3(kScanThreads), 0, s, d_out,
                       offs.get(), n, ntiles);
    SPL_HIP(hipStreamSynchronize(s));  // offs/sums are freed on return
    return;
  }
  SPL_HIP(hipStreamSynchronize(s));
}

template <typename A, typename B>
__global__ void convert_kernel(const A *__restrict__ in, B *__restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = (B)in[i];
}

inline unsigned grid_for(int64_t n, int threads) {
  int64_t b = (n + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > 8192) b = 8192;
  return (unsigned)b;
}

}  // namespace

void exclusive_scan_i32_to_i64(const int *d_in, int64_t *d_out, int64_t n, hipStream_t s) {
  scan_impl<int>(d_in, d_out, n, s);
}
void exclusive_scan_i64(const int64_t *d_in, int64_t *d_out, int64_t n, hipStream_t s) {
  scan_impl<int64_t>(d_in, d_out, n, s);
}
void narrow_i64_to_i32(const int64_t *d_in, int *d_out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL((convert_kernel<int64_t, int>), dim3(grid_for(n, 256)), dim3(256), 0, s, d_in,
                     d_out, n);
}
void widen_i32_to_i64(const int *d_in, int64_t *d_out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL((convert_kernel<int, int64_t>), dim3(grid_for(n, 256)), dim3(256), 0, s, d_in,
                     d_out, n);
}

}  // namespace spl

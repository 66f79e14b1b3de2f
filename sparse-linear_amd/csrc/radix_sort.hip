// radix_sort.hip — least-significant-digit radix sort of 64-bit keys, hand-written (round 4: it replaces the one
// vendor primitive the product had taken in, a library radix sort in nd_levels.hip).
//
// One pass per 8-bit digit, three launches per pass, no allocation and no host synchronisation inside (the caller owns
// the scratch: radix_sort_u64_temp_bytes):
//   radix_hist_kernel     a workgroup per tile of 4096 keys counts its keys per digit value: hist[digit][tile]
//   radix_binscan_kernel  a workgroup per digit value: exclusive sums over the tiles, the value's total to bin_total
//   radix_scatter_kernel  a workgroup per tile: where value d of this tile starts = (exclusive sum of bin_total)[d] +
//                         hist[d][tile]; the keys then move in memory order.  A wavefront owns 1024 consecutive keys and
//                         takes them 64 at a time: the lanes that hold the same digit value find each other with eight
//                         ballots, a key's place is the wavefront's running count of its value plus the number of such
//                         lanes below it — the order of equal digits is the order in memory, which is what makes the
//                         least-significant-digit scheme a sort, and nothing depends on the order atomics complete in.
// Rate: three reads and one write of the keys per pass; the level structures of the nested dissection sort
// (level, vertex) keys of 25 - 35 bits: 4 - 5 passes.
#include "common.hpp"

namespace spl {
namespace {

constexpr int kRadixBits = 8, kRadixBins = 1 << kRadixBits;
constexpr int kRadixThreads = 256, kRadixWaves = kRadixThreads / 64;
constexpr int kRadixPerWave = 1024;                       // consecutive keys a wavefront owns
constexpr int kRadixTile = kRadixWaves * kRadixPerWave;   // 4096 keys per workgroup

__global__ __launch_bounds__(kRadixThreads) void radix_hist_kernel(const unsigned long long *__restrict__ keys, int64_t n,
                                                                   int shift, int ntiles, int *__restrict__ hist) {
  __shared__ int cnt[kRadixBins];
  cnt[threadIdx.x] = 0;  // (kRadixThreads == kRadixBins)
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRadixTile;
  for (int i = threadIdx.x; i < kRadixTile; i += kRadixThreads) {
    const int64_t at = base + i;
    if (at < n) atomicAdd(&cnt[(int)((keys[at] >> shift) & (kRadixBins - 1))], 1);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * (size_t)ntiles + blockIdx.x] = cnt[threadIdx.x];
}

// exclusive sums over the tiles of digit value blockIdx.x, in place; its total to bin_total
__global__ __launch_bounds__(kRadixThreads) void radix_binscan_kernel(int *__restrict__ hist, int ntiles,
                                                                      int *__restrict__ bin_total) {
  __shared__ int wave_tot[kRadixWaves];
  int *row = hist + (size_t)blockIdx.x * (size_t)ntiles;
  const int per = (ntiles + kRadixThreads - 1) / kRadixThreads;
  const int lo = min((int)threadIdx.x * per, ntiles), hi = min(lo + per, ntiles);
  int mine = 0;
  for (int i = lo; i < hi; ++i) mine += row[i];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int before = incl - mine, total = 0;
#pragma unroll
  for (int w = 0; w < kRadixWaves; ++w) {
    if (w < wave) before += wave_tot[w];
    total += wave_tot[w];
  }
  for (int i = lo; i < hi; ++i) {
    const int v = row[i];
    row[i] = before;
    before += v;
  }
  if (threadIdx.x == 0) bin_total[blockIdx.x] = total;
}

__global__ __launch_bounds__(kRadixThreads) void radix_scatter_kernel(const unsigned long long *__restrict__ keys,
                                                                      unsigned long long *__restrict__ out, int64_t n,
                                                                      int shift, int ntiles, const int *__restrict__ hist,
                                                                      const int *__restrict__ bin_total) {
  __shared__ int cnt[kRadixWaves][kRadixBins];  // per wavefront: counts, then the next free place of each digit value
  __shared__ int wave_tot[kRadixWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < kRadixWaves; ++w) cnt[w][threadIdx.x] = 0;
  // where digit value threadIdx.x starts overall: exclusive sum of the totals
  const int tot = bin_total[threadIdx.x];
  int incl = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int start = incl - tot;
#pragma unroll
  for (int w = 0; w < kRadixWaves; ++w)
    if (w < wave) start += wave_tot[w];
  start += hist[(size_t)threadIdx.x * (size_t)ntiles + blockIdx.x];  // ... and inside it, of this tile
  // the wavefronts count their own keys
  const int64_t base = (int64_t)blockIdx.x * kRadixTile + (int64_t)wave * kRadixPerWave;
  for (int r = 0; r < kRadixPerWave / 64; ++r) {
    const int64_t at = base + r * 64 + lane;
    if (at < n) atomicAdd(&cnt[wave][(int)((keys[at] >> shift) & (kRadixBins - 1))], 1);
  }
  __syncthreads();
  {
    int run = start;  // thread = digit value: the first place of each wavefront's keys of that value
#pragma unroll
    for (int w = 0; w < kRadixWaves; ++w) {
      const int c = cnt[w][threadIdx.x];
      cnt[w][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
  // in memory order, 64 keys at a time; only this wavefront touches cnt[wave][*] from here on
  for (int r = 0; r < kRadixPerWave / 64; ++r) {
    const int64_t at = base + r * 64 + lane;
    const bool valid = at < n;
    const unsigned long long key = valid ? keys[at] : 0ull;
    const int d = (int)((key >> shift) & (kRadixBins - 1));
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < kRadixBits; ++b) {
      const unsigned long long has = __ballot((d >> b) & 1);
      same &= ((d >> b) & 1) ? has : ~has;
    }
    const int rank = __popcll(same & ((1ull << lane) - 1ull));
    int place = 0;
    if (valid) place = cnt[wave][d];
    __builtin_amdgcn_wave_barrier();  // every lane has read its value's place before the first of them moves it on
    if (valid && rank == 0) cnt[wave][d] = place + __popcll(same);
    __builtin_amdgcn_wave_barrier();
    if (valid) out[place + rank] = key;
  }
}

}  // namespace

size_t radix_sort_u64_temp_bytes(int64_t n) {
  const int64_t ntiles = (n + kRadixTile - 1) / kRadixTile;
  return ((size_t)kRadixBins * (size_t)(ntiles > 0 ? ntiles : 1) + (size_t)kRadixBins) * sizeof(int);
}

// keys[0 .. n) sorted ascending by their bits [0, nbits) (higher bits must be equal or irrelevant); `alt` is a second
// buffer of n keys, `temp` radix_sort_u64_temp_bytes(n) bytes.  Returns the buffer that holds the result (keys or alt).
// Everything is enqueued on `s`; n < 2^31.
unsigned long long *radix_sort_u64(unsigned long long *keys, unsigned long long *alt, int64_t n, int nbits, void *temp,
                                   hipStream_t s) {
  if (n <= 1 || nbits <= 0) return keys;
  const int ntiles = (int)((n + kRadixTile - 1) / kRadixTile);
  int *hist = static_cast<int *>(temp), *bin_total = hist + (size_t)kRadixBins * (size_t)ntiles;
  unsigned long long *in = keys, *out = alt;
  for (int shift = 0; shift < nbits; shift += kRadixBits) {
    hipLaunchKernelGGL(radix_hist_kernel, dim3((unsigned)ntiles), dim3(kRadixThreads), 0, s, in, n, shift, ntiles, hist);
    hipLaunchKernelGGL(radix_binscan_kernel, dim3(kRadixBins), dim3(kRadixThreads), 0, s, hist, ntiles, bin_total);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3((unsigned)ntiles), dim3(kRadixThreads), 0, s, in, out, n, shift, ntiles,
                       hist, bin_total);
    std::swap(in, out);
  }
  return in;
}

}  // namespace spl

// diagnostics / tests: sorts d_keys[0 .. n) in place by their low nbits bits (scratch allocated here)
extern "C" int spl_debug_sort_u64(unsigned long long *d_keys, long long n, int nbits, void *stream) {
  if (n < 0 || n >= (1ll << 31) || nbits < 0 || nbits > 64 || (n > 0 && !d_keys)) return SPL_ERROR_argument_missing;
  if (n <= 1) return SPL_OK;
  try {
    hipStream_t s = spl::as_stream(stream);
    spl::DBuf<unsigned long long> alt((size_t)n);
    spl::DBuf<char> temp(spl::radix_sort_u64_temp_bytes(n));
    unsigned long long *res = spl::radix_sort_u64(d_keys, alt.get(), n, nbits, temp.get(), s);
    if (res != d_keys) SPL_HIP(hipMemcpyAsync(d_keys, res, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    SPL_HIP(hipStreamSynchronize(s));
    return hipGetLastError() == hipSuccess ? SPL_OK : SPL_ERROR_device;
  } catch (const spl::DeviceError &e) {
    return e.status;
  } catch (...) {
    return SPL_ERROR_internal;
  }
}

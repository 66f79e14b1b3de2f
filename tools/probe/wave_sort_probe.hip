// Probe 7 (round 3): how fast can ONE WAVEFRONT sort 1024 / 2048 32-bit keys held in registers (16 / 32 per lane,
// element e = r * 64 + lane) with a bitonic network whose cross-lane exchanges are DPP moves (xor 1, 2, 4, 8), lane
// swaps of gfx950 (v_permlane16_swap / v_permlane32_swap) or ds_bpermute (SHUF = 0: every cross-lane step through the
// LDS crossbar), and whose in-lane exchanges (distance >= 64) are plain min / max between registers?  This is what a
// wavefront-per-column SpGEMM (sort the products of a column of ~1000 products by (row, k) without LDS round trips or
// workgroup barriers) would spend most of its instructions on.  Prints sorts per second chip-wide and checks order.
// build: hipcc -O3 --offload-arch=gfx950 wave_sort_probe.hip -o wave_sort_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int J, int SHUF>
__device__ __forceinline__ unsigned xorshuf(unsigned v) {
  if (SHUF == 0) return (unsigned)__shfl_xor((int)v, J, 64);
  if (J == 1) return __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  if (J == 2) return __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  if (J == 4) {
    unsigned p = __builtin_amdgcn_update_dpp(0u, v, 0x104, 0xF, 0x5, false);      // row_shl:4 into banks 0, 2
    return __builtin_amdgcn_update_dpp(p, v, 0x114, 0xF, 0xA, false);             // row_shr:4 into banks 1, 3
  }
  if (J == 8) return __builtin_amdgcn_update_dpp(0u, v, 0x128, 0xF, 0xF, false);  // row_ror:8
  if (J == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // r[0] = rows [0,0,2,2], r[1] = rows [1,1,3,3]
    return (threadIdx.x & 16) ? r[0] : r[1];
  }
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);    // r[0] = halves [lo, lo], r[1] = [hi, hi]
  return (threadIdx.x & 32) ? r[0] : r[1];
}

template <int NR, int K, int J, int SHUF>
__device__ __forceinline__ void stage(unsigned (&k)[NR]) {
  if constexpr (J >= 64) {
    constexpr int M = J / 64;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if ((r & M) == 0) {
        const unsigned a = k[r], b = k[r | M];
        const bool asc = ((r * 64) & K) == 0;  // compile-time: K > J >= 64
        k[r] = asc ? min(a, b) : max(a, b);
        k[r | M] = asc ? max(a, b) : min(a, b);
      }
    }
  } else {
    const int lane = threadIdx.x & 63;
    const bool upper = (lane & J) != 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const unsigned a = k[r], p = xorshuf<J, SHUF>(a);
      // descending block: (e & K) != 0 with e = r * 64 + lane
      const bool desc = K >= 64 ? (((r * 64) & K) != 0) : ((lane & K) != 0);
      k[r] = (upper != desc) ? max(a, p) : min(a, p);
    }
  }
}

template <int NR, int K, int J, int SHUF>
__device__ __forceinline__ void merge_level(unsigned (&k)[NR]) {
  stage<NR, K, J, SHUF>(k);
  if constexpr (J > 1) merge_level<NR, K, J / 2, SHUF>(k);
}
template <int NR, int K, int SHUF>
__device__ __forceinline__ void sort_levels(unsigned (&k)[NR]) {
  if constexpr (K > 2) sort_levels<NR, K / 2, SHUF>(k);
  merge_level<NR, K, K / 2, SHUF>(k);
}

template <int NR, int SHUF>
__global__ __launch_bounds__(256) void sort_kernel(int iters, unsigned *__restrict__ bad, unsigned *__restrict__ sink) {
  const int lane = threadIdx.x & 63;
  unsigned k[NR];
  unsigned seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0, violations = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < NR; ++r) { seed = seed * 1664525u + 1013904223u; k[r] = seed ^ (seed >> 15); }
    sort_levels<NR, 64 * NR, SHUF>(k);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const unsigned nxt_lane = (unsigned)__shfl_down((int)k[r], 1, 64);
      const unsigned nxt_reg = r + 1 < NR ? (unsigned)__shfl((int)k[r + 1 < NR ? r + 1 : r], 0, 64) : 0xffffffffu;
      const unsigned nxt = lane == 63 ? nxt_reg : nxt_lane;
      violations += k[r] > nxt ? 1u : 0u;
      acc ^= k[r];
    }
  }
  if (violations) atomicAdd(bad, violations);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int NR, int SHUF>
void run(int wgs, int iters, unsigned *bad, unsigned *sink) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipMemset(bad, 0, 4);
  hipLaunchKernelGGL((sort_kernel<NR, SHUF>), dim3(wgs), dim3(256), 0, 0, 2, bad, sink);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((sort_kernel<NR, SHUF>), dim3(wgs), dim3(256), 0, 0, iters, bad, sink);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  unsigned h = 0;
  (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  const double sorts = (double)wgs * 4 * iters;
  printf("%4d keys per wavefront, %s, %4d workgroups of 4 wavefronts: %8.3f ms for %.0f sorts = %7.1f M sorts/s (1.05 M sorts: %.2f ms), order violations %u\n",
         64 * NR, SHUF ? "DPP / lane swaps" : "ds_bpermute     ", wgs, ms, sorts, sorts / ms / 1e3, 1048576.0 / (sorts / ms), h);
}

int main() {
  unsigned *bad, *sink;
  (void)hipMalloc(&bad, 4);
  (void)hipMalloc(&sink, 4);
  for (int wgs : {256, 512, 1024}) {
    run<16, 1>(wgs, 200, bad, sink);
    run<32, 1>(wgs, 100, bad, sink);
  }
  run<16, 0>(1024, 200, bad, sink);
  run<32, 0>(1024, 100, bad, sink);
  return 0;
}

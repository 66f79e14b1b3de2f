// Probe: how many 8-byte gathers per clock can a CU pull from an L2-resident table when every gather
// misses L1 (random 128-byte lines)?  Sets the ceiling for the x gathers of the column-blocked SpMV.
// build: hipcc -O3 --offload-arch=gfx950 l1_gather_probe.hip -o l1_gather_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int U>
__global__ __launch_bounds__(1024) void gather_kernel(const double *__restrict__ table, uint32_t mask, int iters,
                                                      double *__restrict__ out) {
  uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s = s * 1664525u + 1013904223u;
      v[u] = table[(s >> 7) & mask];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

// MIX of the U gathers of a trip go through the L2 atomic unit instead (fetch-or with 0 returns the
// value; it does not allocate a line in L1): does that path add to the L1-miss path's throughput?
template <int U, int MIX>
__global__ __launch_bounds__(1024) void gather_mix_kernel(double *__restrict__ table, uint32_t mask, int iters,
                                                          double *__restrict__ out) {
  uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  double acc = 0.0;
  unsigned long long *t64 = reinterpret_cast<unsigned long long *>(table);
  for (int it = 0; it < iters; ++it) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s = s * 1664525u + 1013904223u;
      const uint32_t i = (s >> 7) & mask;
      if (u < MIX)
        v[u] = __longlong_as_double((long long)__hip_atomic_fetch_or(t64 + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      else
        v[u] = table[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int U, int MIX>
double run_mix(double *table, uint32_t mask, int wgs, int iters, double *out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((gather_mix_kernel<U, MIX>), dim3(wgs), dim3(1024), 0, 0, table, mask, 4, out);
  hipEventRecord(a);
  hipLaunchKernelGGL((gather_mix_kernel<U, MIX>), dim3(wgs), dim3(1024), 0, 0, table, mask, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

template <int U>
double run(const double *table, uint32_t mask, int wgs, int iters, double *out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(gather_kernel<U>, dim3(wgs), dim3(1024), 0, 0, table, mask, 4, out);
  hipEventRecord(a);
  hipLaunchKernelGGL(gather_kernel<U>, dim3(wgs), dim3(1024), 0, 0, table, mask, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  const int wgs = 256;  // one 16-wave workgroup per CU
  double *out;
  hipMalloc(&out, (size_t)wgs * 1024 * sizeof(double));
  for (int logn : {12, 15, 18, 21, 24}) {  // table of 2^logn doubles: 32 KB (L1), 256 KB, 2 MB (L2), 16 MB, 128 MB
    const size_t n = (size_t)1 << logn;
    double *table;
    hipMalloc(&table, n * sizeof(double));
    std::vector<double> h(n, 1.0);
    hipMemcpy(table, h.data(), n * sizeof(double), hipMemcpyHostToDevice);
    const int iters = 400;
    const double ms8 = run<8>(table, (uint32_t)(n - 1), wgs, iters, out);
    const double ms16 = run<16>(table, (uint32_t)(n - 1), wgs, iters / 2, out);
    const double g = (double)wgs * 1024 * iters * 8;
    printf("table %8.0f KB: U=8 %.3f ms = %.1f Ggather/s = %.3f gathers/clk/CU (2.4 GHz) | U=16 %.3f ms = %.3f gathers/clk/CU\n",
           n * 8 / 1024.0, ms8, g / ms8 / 1e6, g / (ms8 * 1e-3) / 256 / 2.4e9, ms16, g / (ms16 * 1e-3) / 256 / 2.4e9);
    if (logn == 18) {
      const double per = (double)wgs * 1024 * iters * 8 / 1e-3 / 256 / 2.4e9;
      printf("  through the L2 atomic unit (fetch-or 0), of 8 gathers per trip: 8: %.3f  4: %.3f  2: %.3f  1: %.3f gathers/clk/CU\n",
             per / run_mix<8, 8>(table, (uint32_t)(n - 1), wgs, iters, out), per / run_mix<8, 4>(table, (uint32_t)(n - 1), wgs, iters, out),
             per / run_mix<8, 2>(table, (uint32_t)(n - 1), wgs, iters, out), per / run_mix<8, 1>(table, (uint32_t)(n - 1), wgs, iters, out));
    }
    hipFree(table);
  }
  return 0;
}

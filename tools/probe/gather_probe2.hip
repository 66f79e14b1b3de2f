// Probe 2: what bounds 8-byte gathers from an L2-resident table?
//  (A) random gathers with 256 / 128 / 64 / 32 workgroups (one per CU): if the per-CU rate rises when
//      fewer CUs gather, the bound is shared (L2 channels); if not, it is the CU's own L1 miss path.
//  (B) column-sorted gathers: the 64 lanes of an instruction ascend through the table with a mean
//      spacing of S doubles (S = 25: 0.64 entries per 128-byte line, what a 19.5 K-row panel of C2
//      gives; 12, 50, 100; 16 = every lane its own line, in order): gathers per clock per CU and,
//      with the expected distinct lines per instruction, requests per clock.
//  (C) ds_add_f64 into random rows of a 19.5 K-row LDS image, 16 wavefronts: adds per clock per CU.
// build: hipcc -O3 --offload-arch=gfx950 gather_probe2.hip -o gather_probe2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int U>
__global__ __launch_bounds__(1024) void gather_random(const double *__restrict__ table, uint32_t mask, int iters,
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

// lanes ascend: lane l reads base + l*S + jitter(0..S-1); base random per instruction (wave-uniform)
template <int U>
__global__ __launch_bounds__(1024) void gather_sorted(const double *__restrict__ table, uint32_t mask, int S, int iters,
                                                      double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  uint32_t sw = (blockIdx.x * 16u + (threadIdx.x >> 6)) * 2246822519u + 777u;  // wave-uniform stream
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s = s * 1664525u + 1013904223u;
      sw = sw * 1664525u + 1013904223u;
      const uint32_t base = (sw >> 6) & mask;
      const uint32_t j = (s >> 9) % (uint32_t)S;
      v[u] = table[(base + (uint32_t)(lane * S) + j) & mask];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(1024) void lds_add(int rows, int iters, double *__restrict__ out) {
  extern __shared__ double ylds[];
  for (int i = threadIdx.x; i < rows; i += 1024) ylds[i] = 0.0;
  __syncthreads();
  uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = s * 1664525u + 1013904223u;
      const uint32_t r = (s >> 8) % (uint32_t)rows;
      __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)(ylds + r), 1.0);
    }
  }
  __syncthreads();
  double acc = 0.0;
  for (int i = threadIdx.x; i < rows; i += 1024) acc += ylds[i];
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <typename F>
double timed(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch(4);
  hipEventRecord(a);
  launch(0);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  double *out;
  hipMalloc(&out, (size_t)256 * 1024 * sizeof(double));
  const size_t n = (size_t)1 << 18;  // 2 MiB of doubles
  double *table;
  hipMalloc(&table, n * sizeof(double));
  std::vector<double> h(n, 1.0);
  hipMemcpy(table, h.data(), n * sizeof(double), hipMemcpyHostToDevice);
  const uint32_t mask = (uint32_t)(n - 1);
  const int iters = 400;
  for (int wgs : {256, 192, 128, 64, 32, 8}) {
    const double ms = timed([&](int small) {
      hipLaunchKernelGGL(gather_random<8>, dim3(wgs), dim3(1024), 0, 0, table, mask, small ? small : iters, out);
    });
    const double g = (double)wgs * 1024 * iters * 8;
    printf("A random, %3d workgroups: %.3f ms = %.1f Ggather/s = %.3f gathers/clk/CU-in-use (2.4 GHz)\n", wgs, ms,
           g / ms / 1e6, g / (ms * 1e-3) / wgs / 2.4e9);
  }
  for (int S : {100, 50, 25, 16, 12, 6, 2, 1}) {
    const double ms = timed([&](int small) {
      hipLaunchKernelGGL(gather_sorted<8>, dim3(256), dim3(1024), 0, 0, table, mask, S, small ? small : iters, out);
    });
    const double g = (double)256 * 1024 * iters * 8;
    // expected distinct 128-byte lines per 64-lane instruction: lanes cover 64*S doubles = 4*S lines,
    // entries per line d = 16/S: lines touched = 4 S (1 - exp(-d)) for S >= 16, ~4 S below
    const double lines = S >= 16 ? 4.0 * S * (1.0 - __builtin_exp(-16.0 / S)) : 4.0 * S + 1.0;
    printf("B sorted, spacing %3d doubles: %.3f ms = %.3f gathers/clk/CU, ~%.1f lines per instruction -> %.3f requests/clk/CU\n",
           S, ms, g / (ms * 1e-3) / 256 / 2.4e9, lines, g / 64.0 * lines / (ms * 1e-3) / 256 / 2.4e9);
  }
  {
    hipFuncSetAttribute(reinterpret_cast<const void *>(&lds_add), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int rows = 19532;
    const double ms = timed([&](int small) {
      hipLaunchKernelGGL(lds_add, dim3(256), dim3(1024), rows * sizeof(double), 0, rows, small ? small : 2000, out);
    });
    const double g = (double)256 * 1024 * 2000 * 8;
    printf("C ds_add_f64, random rows of %d, 16 wavefronts: %.3f ms = %.2f adds/clk/CU\n", rows, ms,
           g / (ms * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}

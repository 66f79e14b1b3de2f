// Probe 6 (round 3): why do column-SORTED gathers (the panel kernel's: 64 lanes ascending through x with a mean
// spacing of 25 doubles, ~47 distinct 128-byte lines per instruction) send fewer requests per clock to the L2
// (0.35 per clock and CU, gather_probe2 row B) than uniformly random gathers (0.43)?  If neighbouring lines of x
// share an L2 channel (coarse channel interleave), an instruction's ~47 requests queue on a few channels.
// Test: the same sorted gathers with the LINE index multiplied by an odd constant modulo the table's 2^14 lines
// (a bijection on lines; the 16 doubles of a line stay together, so lanes share requests exactly as before).
// build: hipcc -O3 --offload-arch=gfx950 gather_probe3.hip -o gather_probe3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int U>
__global__ __launch_bounds__(1024) void gather_sorted(const double *__restrict__ table, uint32_t mask, int S, uint32_t mul,
                                                      int iters, double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  uint32_t sw = (blockIdx.x * 16u + (threadIdx.x >> 6)) * 2246822519u + 777u;  // wave-uniform stream
  const uint32_t lmask = mask >> 4;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s = s * 1664525u + 1013904223u;
      sw = sw * 1664525u + 1013904223u;
      const uint32_t base = (sw >> 6) & mask;
      const uint32_t j = (s >> 9) % (uint32_t)S;
      const uint32_t e = (base + (uint32_t)(lane * S) + j) & mask;
      const uint32_t line = ((e >> 4) * mul) & lmask;
      v[u] = table[(line << 4) | (e & 15u)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
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
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int wgs : {256, 8}) {
    for (int S : {25, 50}) {
      for (uint32_t mul : {1u, 3u, 5u, 9u, 17u, 33u, 65u, 129u, 257u, 513u, 1025u, 0x9E5u, 0x2545u}) {
        hipLaunchKernelGGL(gather_sorted<8>, dim3(wgs), dim3(1024), 0, 0, table, mask, S, mul, 4, out);
        hipEventRecord(a);
        hipLaunchKernelGGL(gather_sorted<8>, dim3(wgs), dim3(1024), 0, 0, table, mask, S, mul, iters, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double g = (double)wgs * 1024 * iters * 8;
        const double lines = 4.0 * S * (1.0 - __builtin_exp(-16.0 / S));
        printf("%3d workgroups, spacing %2d doubles, line multiplier %5u: %.3f ms = %.3f gathers/clk/CU, ~%.1f lines per instruction"
               " -> %.3f requests/clk/CU\n", wgs, S, mul, ms, g / (ms * 1e-3) / wgs / 2.4e9, lines,
               g / 64.0 * lines / (ms * 1e-3) / wgs / 2.4e9);
      }
    }
  }
  return 0;
}

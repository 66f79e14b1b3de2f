// Probe 3: how do a CU's HBM stream and its L2-resident gathers share the vector L1 (TCP)?
//  S  pure stream: every wavefront keeps US 512-byte loads (dwordx2 per lane, nt) in flight over a
//     3 GiB buffer; bytes per clock per CU with 8 / 64 / 256 workgroups (one per CU, 16 wavefronts).
//  G  pure random 8-byte gathers from a 2 MiB table (as gather_probe2 A), 256 workgroups.
//  M  mixed, same wavefront: per iteration UG gathers + US stream loads, all issued, then all used.
//  W  mixed, different wavefronts: wavefronts 0..7 stream, 8..15 gather (same totals per CU).
// If the two kinds share a fixed number of miss slots, time(M) ~ time(S alone) + time(G alone) for the
// same amounts; if they ran on independent resources, time(M) ~ max of the two.
// build: hipcc -O3 --offload-arch=gfx950 tcp_mix_probe.hip -o tcp_mix_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int US, int UG>
__global__ __launch_bounds__(1024) void mix_kernel(const double *__restrict__ stream, size_t per_wg_doubles,
                                                   const double *__restrict__ table, uint32_t mask, int iters,
                                                   int split, double *__restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  const double *base = stream + (size_t)blockIdx.x * per_wg_doubles;
  double acc = 0.0;
  // split: wavefronts 0..7 only stream (twice as much each), 8..15 only gather (twice as much each)
  const bool do_s = !split || wave < 8, do_g = !split || wave >= 8;
  const int nsw = split ? 8 : 16;
  const int sw = split ? wave : wave;
  size_t pos = (size_t)(sw % nsw) * 64 + lane;
  const size_t stride = (size_t)nsw * 64;
  const int rep = split ? 2 : 1;
  for (int it = 0; it < iters; ++it) {
    for (int r = 0; r < rep; ++r) {
      double g[UG > 0 ? UG : 1], v[US > 0 ? US : 1];
      if (do_g) {
#pragma unroll
        for (int u = 0; u < UG; ++u) {
          s = s * 1664525u + 1013904223u;
          g[u] = table[(s >> 7) & mask];
        }
      }
      if (do_s) {
#pragma unroll
        for (int u = 0; u < US; ++u) {
          v[u] = __builtin_nontemporal_load(base + pos);
          pos += stride;
          if (pos >= per_wg_doubles) pos -= per_wg_doubles;
        }
      }
      if (do_g) {
#pragma unroll
        for (int u = 0; u < UG; ++u) acc += g[u];
      }
      if (do_s) {
#pragma unroll
        for (int u = 0; u < US; ++u) acc += v[u];
      }
    }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int US, int UG>
float run(int wgs, const double *stream, size_t per_wg, const double *table, uint32_t mask, int iters, int split,
          double *out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((mix_kernel<US, UG>), dim3(wgs), dim3(1024), 0, 0, stream, per_wg, table, mask, 2, split, out);
  hipEventRecord(a);
  hipLaunchKernelGGL((mix_kernel<US, UG>), dim3(wgs), dim3(1024), 0, 0, stream, per_wg, table, mask, iters, split, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  double *out, *table, *stream;
  CHECK(hipMalloc(&out, (size_t)256 * 1024 * sizeof(double)));
  const size_t tn = (size_t)1 << 18;
  CHECK(hipMalloc(&table, tn * sizeof(double)));
  std::vector<double> h(tn, 1.0);
  CHECK(hipMemcpy(table, h.data(), tn * sizeof(double), hipMemcpyHostToDevice));
  const size_t per_wg = (size_t)12 << 17;  // 12 MiB per workgroup -> 3 GiB for 256
  CHECK(hipMalloc(&stream, 256 * per_wg * sizeof(double)));
  CHECK(hipMemset(stream, 0, 256 * per_wg * sizeof(double)));
  const uint32_t mask = (uint32_t)(tn - 1);
  const double clk = 2.4e9;
  // S: pure stream, 16 loads in flight per wavefront; iters * 16 waves * 16 loads * 512 B per workgroup
  for (int wgs : {8, 64, 256}) {
    const int iters = 96;  // 96 * 16 * 16 * 512 B = 12.6 MB per workgroup
    const float ms = run<16, 0>(wgs, stream, per_wg, table, mask, iters, 0, out);
    const double bytes = (double)wgs * iters * 16 * 16 * 512;
    printf("S stream only, %3d workgroups, 16 loads/wave in flight: %.3f ms, %.2f TB/s, %.1f B/clk/CU-in-use\n", wgs, ms,
           bytes / ms / 1e9, bytes / (ms * 1e-3) / wgs / clk);
  }
  {
    const int iters = 96;
    const float ms = run<8, 0>(256, stream, per_wg, table, mask, iters * 2, 0, out);
    const double bytes = (double)256 * iters * 2 * 16 * 8 * 512;
    printf("S stream only, 256 workgroups,  8 loads/wave in flight: %.3f ms, %.2f TB/s, %.1f B/clk/CU\n", ms, bytes / ms / 1e9,
           bytes / (ms * 1e-3) / 256 / clk);
  }
  // amounts per workgroup and iteration: 16 waves x US x 512 B of stream, 16 waves x UG x 64 gathers.
  // C2 per entry: 12 B of stream per gather -> per 64 gathers 768 B = 1.5 stream loads: US : UG = 3 : 2
  const int iters = 200;
  const float g_only = run<0, 8>(256, stream, per_wg, table, mask, iters, 0, out);
  const float s_only = run<12, 0>(256, stream, per_wg, table, mask, iters, 0, out);
  const float mixed = run<12, 8>(256, stream, per_wg, table, mask, iters, 0, out);
  const float split = run<12, 8>(256, stream, per_wg, table, mask, iters, 1, out);
  const double gathers = (double)256 * iters * 16 * 8 * 64, sbytes = (double)256 * iters * 16 * 12 * 512;
  printf("G gathers only (8/wave/iter): %.3f ms = %.3f gathers/clk/CU\n", g_only, gathers / (g_only * 1e-3) / 256 / clk);
  printf("S stream only (12 loads/wave/iter): %.3f ms = %.2f TB/s\n", s_only, sbytes / s_only / 1e9);
  printf("M both in every wavefront: %.3f ms  (sum of the two alone %.3f, max %.3f): %.3f gathers/clk/CU + %.2f TB/s\n", mixed,
         g_only + s_only, g_only > s_only ? g_only : s_only, gathers / (mixed * 1e-3) / 256 / clk, sbytes / mixed / 1e9);
  printf("W wavefronts 0-7 stream, 8-15 gather: %.3f ms: %.3f gathers/clk/CU + %.2f TB/s\n", split,
         gathers / (split * 1e-3) / 256 / clk, sbytes / split / 1e9);
  return 0;
}

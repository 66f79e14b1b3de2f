// two_chains_probe.hip — do two chains of small dependent kernels on two streams overlap?  N launches on one stream against N
// on each of 2 / 4 / 8 streams (each chain ping-pongs between its own two buffers), kernels of 1 and of 128 workgroups, and a
// variant whose kernels spin for ~8 us (a latency-bound kernel of the solves / traversals).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 two_chains_probe.hip -o two_chains_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void step_kernel(const int *in, int *out, int n, int spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int v = i < n ? in[(i + 1) % n] : 0;
  const long long t0 = wall_clock64();
  while (spin > 0 && wall_clock64() - t0 < (long long)spin * 100) { v += 0; }  // wall_clock64: 100 MHz
  if (i < n) out[i] = v + 1;
}
int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 1000;
  const int n = 128 * 256, S = 8;
  int *buf;
  CK(hipMalloc(&buf, (size_t)2 * S * n * 4));
  CK(hipMemset(buf, 0, (size_t)2 * S * n * 4));
  hipStream_t st[S];
  for (int i = 0; i < S; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  for (int spin : {0, 8})
    for (int blocks : {1, 128})
      for (int ns : {1, 2, 4, 8}) {
        auto run = [&] {
          for (int k = 0; k < N; ++k)
            for (int q = 0; q < ns; ++q) {
              int *a = buf + (size_t)(2 * q) * n, *b = a + n;
              hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(256), 0, st[q], (k & 1) ? b : a, (k & 1) ? a : b, blocks * 256, spin);
            }
          for (int q = 0; q < ns; ++q) CK(hipStreamSynchronize(st[q]));
        };
        run();
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
          auto t0 = std::chrono::steady_clock::now();
          run();
          best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        printf("kernels of %3d workgroups spinning %d us: %d streams x %d launches  %9.1f us = %6.2f us per launch of a chain\n", blocks, spin, ns, N, best, best / N);
      }
  return 0;
}

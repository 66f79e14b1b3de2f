// graph_chain_probe.hip — what a dependent launch costs on this platform, by the way it is issued: a chain of N small kernels
// (each reads what the one before wrote) as N launches on one stream, and as ONE launch of a captured hipGraph of the same N
// nodes; grids of 1 and of 128 workgroups.  The solve walks and the level structures of the analysis are such chains.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 graph_chain_probe.hip -o graph_chain_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void step_kernel(const int *in, int *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[(i + 1) % n] + 1;
}
int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 1000;
  int *a, *b;
  const int n = 128 * 256;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int blocks : {1, 128}) {
    auto chain = [&](hipStream_t q) {
      for (int k = 0; k < N; ++k) hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(256), 0, q, (k & 1) ? b : a, (k & 1) ? a : b, blocks * 256);
    };
    chain(s); CK(hipStreamSynchronize(s));
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      chain(s); CK(hipStreamSynchronize(s));
      best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("%3d workgroups: %d launches on a stream      %8.1f us = %.2f us each\n", blocks, N, best, best / N);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("%3d workgroups: one graph of %d kernel nodes   %8.1f us = %.2f us each\n", blocks, N, best, best / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}

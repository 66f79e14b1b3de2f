// diag_bench.hip — where do the 75 - 88 us of the 64 x 64 diagonal-block factorisation go (round 4)?  It sits on the
// chain of every block step of a front (837 steps x 75 us = 63 of the 168 ms of a 100^3 factorisation).
// A copy of diag_block_factor_t with clock64() stamps: [0] entry, [1] after the LU of the block, [2] after the write
// to band storage and the pivots' reciprocals, [3] after both inverses.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../sparse-linear_amd/csrc -I../../include diag_bench.hip -o diag_bench
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dense_lu_kernels.hpp"

namespace spl { void set_last_error(const char *what, hipError_t e) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); } }
using namespace spl;

#include "diag_bench_fn.inc"

__global__ __launch_bounds__(256) void probe_kernel(Band b, int j0, int jb, int *singular, double *invL, double *invU, long long *stamps) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double(*D)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
  double(*lcol)[NB] = reinterpret_cast<double(*)[NB]>(dsm + NB * LDP);
  const int tr = threadIdx.x & 63, tc = threadIdx.x >> 6;
  for (int c = tc; c < NB; c += 4) D[tr][c] = (tr < jb && c < jb) ? b.get(j0 + tr, j0 + c) : (tr == c ? 1.0 : 0.0);
  __syncthreads();
  diag_block_factor_probe<false>(b, j0, jb, D, lcol, singular, invL, invU, stamps);
}

int main() {
  const int n = 64, ld = 80;
  std::vector<double> A((size_t)ld * n, 0.0);
  srand(1);
  for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) A[(size_t)i + (size_t)j * ld] = (i == j) ? 70.0 : (rand() / (double)RAND_MAX - 0.5);
  double *dA, *dinv; int *dsing; long long *dst;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dinv, 2 * 64 * 64 * 8); hipMalloc(&dsing, 4); hipMalloc(&dst, 64);
  hipMemset(dsing, 0, 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileBytes));
  hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_lu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileBytes));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    Band b = dense_view(dA, n, ld);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), 2 * kTileBytes, 0, b, 0, 64, dsing, dinv, dinv + 64 * 64, dst);
    long long st[4];
    hipMemcpy(st, dst, 32, hipMemcpyDeviceToHost);
    printf("probe: LU of the block %.1f us, write-back + reciprocals %.1f us, both inverses %.1f us (clock64 at 100 MHz)\n",
           (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01);
  }
  // the library's own kernel, back to back on one stream (each launch waits for the one before: launch gap included)
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  Band b = dense_view(dA, n, ld);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(diag_lu_kernel, dim3(1), dim3(256), 2 * kTileBytes, 0, b, 0, 64, dsing, dinv, dinv + 64 * 64);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(diag_lu_kernel, dim3(1), dim3(256), 2 * kTileBytes, 0, b, 0, 64, dsing, dinv, dinv + 64 * 64);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("diag_lu_kernel: %.1f us per launch (200 back to back)\n", ms * 1e3 / 200);
  return 0;
}

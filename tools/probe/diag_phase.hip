// diag_phase.hip — clock64() stamps inside a copy of diag_block_factor_blocked (dense_lu_kernels.hpp): LU of the block,
// write to band storage, diagonal 16 x 16 inverses, off-diagonal blocks, inverses to global memory (diag_phase.inc is
// written by make_diag_phase.py).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dense_lu_kernels.hpp"
namespace spl { void set_last_error(const char *what, hipError_t e) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); } }
using namespace spl;
#include "diag_phase.inc"
__global__ __launch_bounds__(256) void probe_kernel(Band b, int jb, int *singular, double *invL, double *invU, long long *st) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double(*D)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
  const int tr = threadIdx.x & 63, tc = threadIdx.x >> 6;
  for (int c = tc; c < NB; c += 4) D[tr][c] = (tr < jb && c < jb) ? b.get(tr, c) : (tr == c ? 1.0 : 0.0);
  __syncthreads();
  diag_block_factor_probe(b, 0, jb, D, singular, invL, invU, st);
}
int main() {
  const int n = 64, ld = 80;
  std::vector<double> A((size_t)ld * n, 0.0);
  srand(1);
  for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) A[(size_t)i + (size_t)j * ld] = (i == j) ? 9.0 : (rand() / (double)RAND_MAX - 0.5);
  double *dA, *dinv; int *dsing; long long *dst;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dinv, 2 * 64 * 64 * 8); hipMalloc(&dsing, 4); hipMalloc(&dst, 128);
  hipMemset(dsing, 0, 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileBytes));
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    Band b = dense_view(dA, n, ld);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), 2 * kTileBytes, 0, b, 64, dsing, dinv, dinv + 64 * 64, dst);
    long long st[9];
    hipMemcpy(st, dst, 72, hipMemcpyDeviceToHost);
    printf("clocks: LU %lld, to band storage %lld, diagonal inverses %lld, off-diagonal blocks %lld, inverses out %lld; total %lld\n",
           st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[5] - st[0]);
    printf("        of the LU: panels in registers %lld, pivot rows of the other columns %lld, trailing blocks %lld\n", st[6], st[7], st[8]);
  }
  return 0;
}

// update_bench.hip — isolates the dense LU kernels of dense_lu_kernels.hpp on one large dense matrix:
// (1) the issue rate of v_mfma_f64_16x16x4 by itself, (2) one K = 128 trailing-update pass,
// (3) a whole blocked factorisation.  Diagnostic tool, not part of the library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../sparse-linear_amd/csrc -I../../include update_bench.hip -o update_bench
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dense_lu_kernels.hpp"

namespace spl { void set_last_error(const char *what, hipError_t e) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); } }
using namespace spl;

__global__ __launch_bounds__(256) void mfma_rate_kernel(double *out, int iters) {
  double4v acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (double4v){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_rate_n_kernel(double *out, int iters) {
  double4v acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (double4v){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void fma_rate_kernel(T *out, int iters) {
  T acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (T)i;
  const T a = (T)(1.0 + threadIdx.x * 1e-9), b = (T)(threadIdx.x * 1e-9);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  T s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void fill_kernel(double *A, size_t n, size_t ld) {
  const size_t j = blockIdx.x;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
    const unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(j * 40503u);
    A[i + j * ld] = (i == j ? 8.0 * 128 : 0.0) + ((h >> 8) & 0xffff) * (1.0 / 65536) - 0.5;
  }
}

static float elapsed(hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  SPL_HIP(hipEventSynchronize(b));
  SPL_HIP(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 16384;
  hipEvent_t e0, e1;
  SPL_HIP(hipEventCreate(&e0));
  SPL_HIP(hipEventCreate(&e1));
  hipStream_t s = nullptr;
  {  // (1)
    double *out;
    SPL_HIP(hipMalloc(&out, sizeof(double) * 256 * 4096));
    for (int wgs_per_cu : {1, 2, 4}) {
      const int iters = 20000, grid = 256 * wgs_per_cu;
      hipLaunchKernelGGL(mfma_rate_kernel, dim3(grid), dim3(256), 0, s, out, 100);
      SPL_HIP(hipEventRecord(e0, s));
      hipLaunchKernelGGL(mfma_rate_kernel, dim3(grid), dim3(256), 0, s, out, iters);
      SPL_HIP(hipEventRecord(e1, s));
      const float ms = elapsed(e0, e1);
      const double flops = (double)grid * 4 * iters * 4.0 * 2048;
      printf("mfma_f64_16x16x4: %d waves/SIMD: %.1f TFLOP/s (%.1f ns per instruction per SIMD)\n", wgs_per_cu,
             flops / ms * 1e-9, ms * 1e6 / ((double)iters * 4 * wgs_per_cu));
    }
    for (int wgs_per_cu : {1, 2}) {
      const int iters = 10000, grid = 256 * wgs_per_cu;
      hipLaunchKernelGGL(mfma_rate_n_kernel<8>, dim3(grid), dim3(256), 0, s, out, 100);
      SPL_HIP(hipEventRecord(e0, s));
      hipLaunchKernelGGL(mfma_rate_n_kernel<8>, dim3(grid), dim3(256), 0, s, out, iters);
      SPL_HIP(hipEventRecord(e1, s));
      const float ms = elapsed(e0, e1);
      printf("mfma_f64_16x16x4, 8 accumulators: %d waves/SIMD: %.1f TFLOP/s\n", wgs_per_cu,
             (double)grid * 4 * iters * 8.0 * 2048 / ms * 1e-9);
    }
    for (int wgs_per_cu : {1, 2, 4}) {
      const int iters = 20000, grid = 256 * wgs_per_cu;
      hipLaunchKernelGGL(fma_rate_kernel<double>, dim3(grid), dim3(256), 0, s, out, 100);
      SPL_HIP(hipEventRecord(e0, s));
      hipLaunchKernelGGL(fma_rate_kernel<double>, dim3(grid), dim3(256), 0, s, out, iters);
      SPL_HIP(hipEventRecord(e1, s));
      float ms = elapsed(e0, e1);
      printf("v_fma_f64: %d waves/SIMD: %.1f TFLOP/s (%.2f ns per instruction per SIMD)\n", wgs_per_cu,
             (double)grid * 4 * iters * 16.0 * 128 / ms * 1e-9, ms * 1e6 / ((double)iters * 16 * wgs_per_cu));
      hipLaunchKernelGGL(fma_rate_kernel<float>, dim3(grid), dim3(256), 0, s, (float *)out, 100);
      SPL_HIP(hipEventRecord(e0, s));
      hipLaunchKernelGGL(fma_rate_kernel<float>, dim3(grid), dim3(256), 0, s, (float *)out, iters);
      SPL_HIP(hipEventRecord(e1, s));
      ms = elapsed(e0, e1);
      printf("v_fma_f32: %d waves/SIMD: %.1f TFLOP/s (%.2f ns per instruction per SIMD)\n", wgs_per_cu,
             (double)grid * 4 * iters * 16.0 * 128 / ms * 1e-9, ms * 1e6 / ((double)iters * 16 * wgs_per_cu));
    }
    SPL_HIP(hipFree(out));
  }
  const size_t ld = (size_t)n + 8;
  double *A, *invs;
  int *sing;
  SPL_HIP(hipMalloc(&A, ld * n * sizeof(double)));
  SPL_HIP(hipMalloc(&invs, inverse_block_elems(n) * sizeof(double)));
  SPL_HIP(hipMalloc(&sing, sizeof(int)));
  SPL_HIP(hipMemset(sing, 0, sizeof(int)));
  hipLaunchKernelGGL(fill_kernel, dim3(n), dim3(256), 0, s, A, (size_t)n, ld);
  set_factor_attributes();
  const Band b = dense_view(A, n, (int)ld);
  {  // (2) one K = 128 pass over the window right/below of the first 128 columns (values: whatever)
    const int origin = 128;
    const int nt = (n - origin + 63) / 64;
    Region g{origin, n, origin, n, 0, 128, 0, nt, 0};  // npiv = 0: no look-ahead
    const size_t lds = kTileBytes + 2 * NB * sizeof(double);
    for (int rep = 0; rep < 2; ++rep) {
      SPL_HIP(hipEventRecord(e0, s));
      for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL(gemm_update_kernel, dim3(nt, nt), dim3(256), lds, s, b, g, sing, invs, invs + NB * NB);
      SPL_HIP(hipEventRecord(e1, s));
      const float ms = elapsed(e0, e1) / 5;
      const double w = (double)(n - origin);
      printf("update pass K=128 on %d^2: %.3f ms, %.1f TFLOP/s, window RMW %.2f TB/s\n", n - origin, ms,
             2.0 * w * w * 128 / ms * 1e-9, 16.0 * w * w / ms * 1e-9);
    }
  }
  {  // (2b) the same pass by the kernel without the look-ahead code (three wavefronts per SIMD)
    const int origin = 128;
    const int nt = (n - origin) / 64;
    Region g2{origin, origin + nt * 64, origin, origin + nt * 64, 0, 128, 0, nt, 0};
    for (int rep = 0; rep < 2; ++rep) {
      SPL_HIP(hipEventRecord(e0, s));
      for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL(gemm_update_bulk_kernel, dim3(nt, nt), dim3(256), kTileBytes + 2 * NB * sizeof(double), s, b, g2, sing);
      SPL_HIP(hipEventRecord(e1, s));
      const float ms = elapsed(e0, e1) / 5;
      const double w = (double)nt * 64;
      printf("bulk (3 wavefronts per SIMD) update pass K=128 on %d^2: %.3f ms, %.1f TFLOP/s\n", nt * 64, ms, 2.0 * w * w * 128 / ms * 1e-9);
    }
  }
  {  // (3)
    hipLaunchKernelGGL(fill_kernel, dim3(n), dim3(256), 0, s, A, (size_t)n, ld);
    SPL_HIP(hipEventRecord(e0, s));
    factor_loop(b, n, invs, sing, s);
    SPL_HIP(hipEventRecord(e1, s));
    const float ms = elapsed(e0, e1);
    {
      hipStream_t main_s, helper;
      SPL_HIP(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
      SPL_HIP(hipStreamCreateWithFlags(&helper, hipStreamNonBlocking));
      hipLaunchKernelGGL(fill_kernel, dim3(n), dim3(256), 0, main_s, A, (size_t)n, ld);
      SPL_HIP(hipEventRecord(e0, main_s));
      factor_loop(b, n, invs, sing, main_s, helper);
      SPL_HIP(hipEventRecord(e1, main_s));
      const float msf = elapsed(e0, e1);
      printf("factor_loop dense %d, look-ahead tile on a helper stream: %.1f ms, %.1f TFLOP/s\n", n, msf,
             2.0 / 3.0 * (double)n * n * n / msf * 1e-9);
      SPL_HIP(hipStreamSynchronize(helper));
      (void)hipStreamDestroy(helper);
      (void)hipStreamDestroy(main_s);
    }
    int h = 0;
    SPL_HIP(hipMemcpy(&h, sing, sizeof(int), hipMemcpyDeviceToHost));
    printf("factor_loop dense %d: %.1f ms, %.1f TFLOP/s (singular flag %d)\n", n, ms,
           2.0 / 3.0 * (double)n * n * n / ms * 1e-9, h);
    hipLaunchKernelGGL(fill_kernel, dim3(n), dim3(256), 0, s, A, (size_t)n, ld);
    SPL_HIP(hipEventRecord(e0, s));
    factor_loop(b, n / 2, invs, sing, s);
    SPL_HIP(hipEventRecord(e1, s));
    const float ms2 = elapsed(e0, e1);
    const double np = n / 2, nb = n - n / 2;
    printf("factor_loop front %d (np %d): %.1f ms, %.1f TFLOP/s\n", n, n / 2, ms2,
           (2.0 / 3.0 * np * np * np + 2 * np * np * nb + 2 * np * nb * nb) / ms2 * 1e-9);
  }
  return 0;
}

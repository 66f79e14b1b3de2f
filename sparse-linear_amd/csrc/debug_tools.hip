// debug_tools.hip — diagnostics that are not part of the path: a stand-in for a communication kernel.
//
// spl_debug_occupy launches `blocks` workgroups of `threads` threads that copy a buffer onto itself in a loop
// until `milliseconds` have passed: what a collective's channel kernels look like to the CU dispatcher (few
// workgroups, no LDS, steady loads and stores).  tools/bench_reserved_cus.py runs it on one stream and times the
// SpMV on another, with and without CUs reserved for it (spl_matrix_set_reserved_cus).
#include "common.hpp"

namespace spl {
namespace {

__global__ void occupy_kernel(double *__restrict__ buf, size_t n, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  while (wall_clock64() - t0 < ticks) {  // every wave reaches the exit: the clock only moves forward
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
      buf[i] = __builtin_nontemporal_load(buf + i) + 0.0;
  }
}

}  // namespace
}  // namespace spl

extern "C" int spl_debug_occupy(int blocks, int threads, double milliseconds, double *d_buf, size_t count, void *stream) {
  if (blocks <= 0 || threads <= 0 || threads > 1024 || !d_buf || count == 0 || !(milliseconds >= 0.0) || milliseconds > 2000.0)
    return SPL_ERROR_argument_missing;
  // wall_clock64 counts at 100 MHz on gfx9
  const unsigned long long ticks = (unsigned long long)(milliseconds * 1e5);
  hipLaunchKernelGGL(spl::occupy_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), 0, spl::as_stream(stream), d_buf,
                     count, ticks);
  return hipGetLastError() == hipSuccess ? SPL_OK : SPL_ERROR_device;
}

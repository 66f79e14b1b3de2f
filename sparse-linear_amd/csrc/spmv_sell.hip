// spmv_sell.hip — sliced-ELL (SELL-64) SpMV image for matrices with regular rows and column
// locality (banded, stencil: every discretised PDE operator of the C1 / C5 configs and the
// banded variant of C2).
//
// Why: in the CSR-stream kernel consecutive lanes hold consecutive entries of ONE row, i.e. 20
// different x lines per gather instruction on the banded matrix; ablation shows that those
// gathers cost 0.13 of its 0.59 ms although they all hit cache (the L1 processes a gather line by
// line).  Here lane l owns ROW l of a 64-row slice and the slice is stored entry-major
// (col/val[slice_base + k*64 + l]), so the k-th entries of 64 neighbouring rows are read with one
// coalesced load each and their x gathers fall into 4-8 consecutive lines.  No LDS, no cross-lane
// step: lane l folds a*x + acc over k = 0, 1, ... — ascending column order, separately rounded
// multiply and add — so results are bit-identical to the reference order (Sparse.hs:447-451) for
// every row length.  Rows shorter than the slice's widest row are padded (col 0, val 0) and
// masked by the row length; the image is only built when padding stays under 1/8 of nnz.
#include "common.hpp"

namespace spl {

namespace {

inline unsigned blocks_for(int64_t n, int per_block) {
  int64_t b = (n + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : b);
}

__global__ __launch_bounds__(256) void sell_width_kernel(int64_t nrows, const int64_t *__restrict__ rowptr,
                                                         int64_t nslices, int *__restrict__ width64) {
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= nslices) return;
  const int64_t r = s * 64 + lane;
  int len = r < nrows ? (int)(rowptr[r + 1] - rowptr[r]) : 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) len = max(len, __shfl_xor(len, d, 64));
  if (lane == 0) width64[s] = len;  // entries per row of the slice; scanned in units of rows-of-64 below
}

__global__ __launch_bounds__(256) void sell_fill_kernel(int64_t nrows, const int64_t *__restrict__ rowptr,
                                                        const int *__restrict__ colidx,
                                                        const double *__restrict__ val, int64_t nslices,
                                                        const int64_t *__restrict__ sliceoff,  // in units of 64 entries
                                                        int *__restrict__ scol, double *__restrict__ sval) {
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= nslices) return;
  const int64_t base = sliceoff[s] * 64;
  const int width = (int)(sliceoff[s + 1] - sliceoff[s]);
  const int64_t r = s * 64 + lane;
  const int64_t p0 = r < nrows ? rowptr[r] : 0;
  const int len = r < nrows ? (int)(rowptr[r + 1] - p0) : 0;
  for (int k = 0; k < width; ++k) {
    const bool ok = k < len;
    scol[base + (int64_t)k * 64 + lane] = ok ? colidx[p0 + k] : 0;
    sval[base + (int64_t)k * 64 + lane] = ok ? val[p0 + k] : 0.0;
  }
}

template <int U>
__global__ __launch_bounds__(256) void spmv_sell_kernel(int64_t nrows, int64_t nslices,
                                                        const int64_t *__restrict__ rowptr,
                                                        const int64_t *__restrict__ sliceoff,
                                                        const int *__restrict__ scol,
                                                        const double *__restrict__ sval,
                                                        const double *__restrict__ x, double *__restrict__ y,
                                                        int accumulate) {
  const int lane = threadIdx.x & 63;
  // XCD-aware remap, as in the CSR-stream kernel: neighbouring slices share x lines
  const int64_t nblocks = (nslices + 3) / 4;
  const int64_t per_xcd = gridDim.x >> 3;
  const int64_t rb = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (rb >= nblocks) return;
  const int64_t s = rb * 4 + (threadIdx.x >> 6);
  if (s >= nslices) return;
  const int64_t base = sliceoff[s] * 64 + lane;
  const int width = (int)(sliceoff[s + 1] - sliceoff[s]);
  const int64_t r = s * 64 + lane;
  const bool valid = r < nrows;
  const int len = valid ? (int)(rowptr[r + 1] - rowptr[r]) : 0;
  double acc = (accumulate && valid) ? y[r] : 0.0;
  for (int k0 = 0; k0 < width; k0 += U) {
    int c[U];
    double a[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = k0 + u < width;  // wave-uniform
      c[u] = in ? __builtin_nontemporal_load(scol + base + (int64_t)(k0 + u) * 64) : 0;
      a[u] = in ? __builtin_nontemporal_load(sval + base + (int64_t)(k0 + u) * 64) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = x[c[u]];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k0 + u < len) acc = a[u] * xv[u] + acc;  // a * x + y, ascending column
  }
  if (valid) y[r] = acc;
}

}  // namespace

// padded entries the SELL image of m would hold (decides whether building it pays)
int64_t sell_padded_entries(const Matrix *m, hipStream_t s) {
  const int64_t nslices = (m->nrows_local + 63) / 64;
  if (nslices == 0) return 0;
  DBuf<int> width((size_t)nslices);
  DBuf<int64_t> off((size_t)nslices + 1);
  hipLaunchKernelGGL(sell_width_kernel, dim3(blocks_for(nslices, 4)), dim3(256), 0, s, m->nrows_local,
                     m->rowptr64.get(), nslices, width.get());
  exclusive_scan_i32_to_i64(width.get(), off.get(), nslices, s);
  int64_t total = 0;
  SPL_HIP(hipMemcpyAsync(&total, off.get() + nslices, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  return total * 64;
}

void build_sell_image(Matrix *m, hipStream_t s) {
  auto img = new SellImage();
  try {
    img->nslices = (m->nrows_local + 63) / 64;
    DBuf<int> width((size_t)img->nslices);
    img->sliceoff.alloc((size_t)img->nslices + 1);
    if (img->nslices > 0)
      hipLaunchKernelGGL(sell_width_kernel, dim3(blocks_for(img->nslices, 4)), dim3(256), 0, s, m->nrows_local,
                         m->rowptr64.get(), img->nslices, width.get());
    exclusive_scan_i32_to_i64(width.get(), img->sliceoff.get(), img->nslices, s);
    int64_t total = 0;
    SPL_HIP(hipMemcpyAsync(&total, img->sliceoff.get() + img->nslices, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    img->entries = total * 64;
    img->col.alloc((size_t)img->entries);
    img->val.alloc((size_t)img->entries);
    if (img->nslices > 0)
      hipLaunchKernelGGL(sell_fill_kernel, dim3(blocks_for(img->nslices, 4)), dim3(256), 0, s, m->nrows_local,
                         m->rowptr64.get(), m->colidx.get(), m->val.get(), img->nslices, img->sliceoff.get(),
                         img->col.get(), img->val.get());
    SPL_HIP(hipStreamSynchronize(s));
    SPL_HIP(hipGetLastError());
  } catch (...) {
    delete img;
    throw;
  }
  delete m->sell;
  m->sell = img;
}

int launch_spmv_sell(const Matrix *m, const double *d_x, double *d_y, int accumulate, hipStream_t s) {
  const SellImage *g = m->sell;
  if (!g) return SPL_ERROR_internal;
  if (g->nslices == 0) return SPL_OK;
  const int64_t nblocks = (g->nslices + 3) / 4;
  const int64_t grid = ((nblocks + 7) / 8) * 8;
  if (grid > 0x7fffffffLL) return SPL_ERROR_internal;
  static int unroll = -1;  // SPL_SELL_UNROLL (tuning): entries in flight per lane
  if (unroll < 0) {
    const char *ev = getenv("SPL_SELL_UNROLL");
    unroll = ev ? atoi(ev) : 4;
  }
#define SPL_LAUNCH_SELL(U)                                                                                   \
  hipLaunchKernelGGL(spmv_sell_kernel<U>, dim3((unsigned)grid), dim3(256), 0, s, m->nrows_local, g->nslices, \
                     m->rowptr64.get(), g->sliceoff.get(), g->col.get(), g->val.get(), d_x, d_y, accumulate)
  switch (unroll) {
    case 2: SPL_LAUNCH_SELL(2); break;
    case 8: SPL_LAUNCH_SELL(8); break;
    case 10: SPL_LAUNCH_SELL(10); break;
    case 20: SPL_LAUNCH_SELL(20); break;
    default: SPL_LAUNCH_SELL(4); break;
  }
#undef SPL_LAUNCH_SELL
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("spmv_sell launch", e); return SPL_ERROR_device; }
  return SPL_OK;
}

}  // namespace spl

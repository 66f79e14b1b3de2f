// generate.hip — synthetic workloads of SURVEY.md §8d generated directly in
// HBM (include/spl_synth.h is the specification; the CPU oracle restates it on
// the host and tests compare the two bit for bit).  Each rank of a multi-GPU
// run generates only its own row block; nothing is transferred.
#include "common.hpp"
#include "../../include/spl_synth.h"

namespace spl {

namespace {

enum { kRandom = 0, kBanded = 1, kPoisson2d = 2, kPoisson3d = 3 };

struct GenParams {
  int kind;
  int K;
  int64_t n;   // matrix dimension
  int64_t m;   // grid edge for the Poisson kinds
  uint64_t seed;
  int64_t row0, nrows_local;
};

// one row of random(n, K): K draws, stable-sorted by column, duplicates summed
// in draw order.  Returns the number of distinct columns.
__device__ int random_row(const GenParams &g, int64_t r, int *c_out, double *v_out) {
  uint64_t c[SPL_MAX_DRAWS];
  double v[SPL_MAX_DRAWS];
  for (int k = 0; k < g.K; ++k) {
    uint64_t ck = spl_random_col(g.seed, (uint64_t)r, (uint64_t)k, (uint64_t)g.n);
    double vk = spl_random_val(g.seed, (uint64_t)r, (uint64_t)k);
    int j = k - 1;
    while (j >= 0 && c[j] > ck) { c[j + 1] = c[j]; v[j + 1] = v[j]; --j; }
    c[j + 1] = ck;
    v[j + 1] = vk;
  }
  int m = 0;
  uint64_t last = ~0ull;
  double acc = 0.0;
  for (int k = 0; k < g.K; ++k) {
    if (m > 0 && c[k] == last) {
      acc = acc + v[k];
    } else {
      if (m > 0 && v_out) v_out[m - 1] = acc;
      if (c_out) c_out[m] = (int)c[k];
      last = c[k];
      acc = v[k];
      ++m;
    }
  }
  if (m > 0 && v_out) v_out[m - 1] = acc;
  return m;
}

__device__ inline int stencil_row(const GenParams &g, int64_t r, int *c_out, double *v_out) {
  const int64_t m = g.m;
  int cnt = 0;
  if (g.kind == kPoisson2d) {
    const int64_t iy = r / m, ix = r % m;
    const int64_t cc[5] = {r - m, r - 1, r, r + 1, r + m};
    const bool ok[5] = {iy > 0, ix > 0, true, ix < m - 1, iy < m - 1};
#pragma unroll
    for (int t = 0; t < 5; ++t)
      if (ok[t]) {
        if (c_out) { c_out[cnt] = (int)cc[t]; v_out[cnt] = (t == 2) ? 4.0 : -1.0; }
        ++cnt;
      }
  } else {
    const int64_t iz = r / (m * m), iy = (r / m) % m, ix = r % m;
    const int64_t cc[7] = {r - m * m, r - m, r - 1, r, r + 1, r + m, r + m * m};
    const bool ok[7] = {iz > 0, iy > 0, ix > 0, true, ix < m - 1, iy < m - 1, iz < m - 1};
#pragma unroll
    for (int t = 0; t < 7; ++t)
      if (ok[t]) {
        if (c_out) { c_out[cnt] = (int)cc[t]; v_out[cnt] = (t == 3) ? 6.0 : -1.0; }
        ++cnt;
      }
  }
  return cnt;
}

__device__ inline int banded_row(const GenParams &g, int64_t r, int *c_out, double *v_out) {
  int cnt = 0;
  for (int d = 0; d < SPL_BAND_DIAGS; ++d) {
    const int64_t c = r + spl_band_offset(d);
    if (c < 0 || c >= g.n) continue;
    if (c_out) { c_out[cnt] = (int)c; v_out[cnt] = spl_random_val(g.seed, (uint64_t)r, (uint64_t)d); }
    ++cnt;
  }
  return cnt;
}

__global__ __launch_bounds__(256) void gen_count(GenParams g, int *__restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.nrows_local) return;
  const int64_t r = g.row0 + i;
  int m;
  if (g.kind == kRandom) m = random_row(g, r, nullptr, nullptr);
  else if (g.kind == kBanded) m = banded_row(g, r, nullptr, nullptr);
  else m = stencil_row(g, r, nullptr, nullptr);
  counts[i] = m;
}

__global__ __launch_bounds__(256) void gen_fill(GenParams g, const int64_t *__restrict__ rowptr,
                                                int *__restrict__ colidx, double *__restrict__ val) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.nrows_local) return;
  const int64_t r = g.row0 + i;
  int c[SPL_MAX_DRAWS];
  double v[SPL_MAX_DRAWS];
  int m;
  if (g.kind == kRandom) m = random_row(g, r, c, v);
  else if (g.kind == kBanded) m = banded_row(g, r, c, v);
  else m = stencil_row(g, r, c, v);
  const int64_t p = rowptr[i];
  for (int k = 0; k < m; ++k) { colidx[p + k] = c[k]; val[p + k] = v[k]; }
}

__global__ __launch_bounds__(256) void gen_vector_kernel(uint64_t seed, int64_t j0, int64_t count,
                                                         double *__restrict__ x) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) x[i] = spl_vector_entry(seed, (uint64_t)(j0 + i));
}

__global__ __launch_bounds__(256) void gen_rmat_kernel(uint64_t seed, int scale, uint32_t ta, uint32_t tb,
                                                       uint32_t tc, int64_t nedges, int *__restrict__ rows,
                                                       int *__restrict__ cols, double *__restrict__ vals) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < nedges; e += stride) {
    uint64_t r, c;
    spl_rmat_edge(seed, (uint64_t)e, scale, ta, tb, tc, &r, &c);
    rows[e] = (int)r;
    cols[e] = (int)c;
    vals[e] = spl_uniform_value(spl_hash(seed ^ SPL_VAL_SALT, (uint64_t)e, 63));
  }
}

}  // namespace

void generate_rmat_coo(uint64_t seed, int scale, uint32_t ta, uint32_t tb, uint32_t tc, int64_t nedges,
                       int *d_rows, int *d_cols, double *d_vals, hipStream_t s) {
  if (nedges <= 0) return;
  int64_t b = (nedges + 255) / 256;
  if (b > 16384) b = 16384;
  hipLaunchKernelGGL(gen_rmat_kernel, dim3((unsigned)b), dim3(256), 0, s, seed, scale, ta, tb, tc, nedges,
                     d_rows, d_cols, d_vals);
  SPL_HIP(hipGetLastError());
}

void generate_synthetic(Matrix *mat, int kind, int64_t n_or_m, int K, uint64_t seed, hipStream_t s) {
  GenParams g;
  g.kind = kind;
  g.K = K;
  g.seed = seed;
  g.m = n_or_m;
  g.n = kind == kPoisson2d ? n_or_m * n_or_m
        : kind == kPoisson3d ? n_or_m * n_or_m * n_or_m
                             : n_or_m;
  g.row0 = mat->row0;
  g.nrows_local = mat->nrows_local;
  const int64_t nl = mat->nrows_local;
  DBuf<int> counts((size_t)nl);
  mat->rowptr64.alloc((size_t)nl + 1);
  const unsigned grid = (unsigned)((nl + 255) / 256 > 0 ? (nl + 255) / 256 : 1);
  hipLaunchKernelGGL(gen_count, dim3(grid), dim3(256), 0, s, g, counts.get());
  exclusive_scan_i32_to_i64(counts.get(), mat->rowptr64.get(), nl, s);
  int64_t nnz = 0;
  SPL_HIP(hipMemcpyAsync(&nnz, mat->rowptr64.get() + nl, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  mat->nnz = nnz;
  mat->colidx.alloc((size_t)nnz);
  mat->val.alloc((size_t)nnz);
  hipLaunchKernelGGL(gen_fill, dim3(grid), dim3(256), 0, s, g, mat->rowptr64.get(), mat->colidx.get(),
                     mat->val.get());
  SPL_HIP(hipGetLastError());
}

void generate_vector(uint64_t seed, int64_t j0, int64_t j1, double *d_x, hipStream_t s) {
  const int64_t count = j1 - j0;
  if (count <= 0) return;
  int64_t b = (count + 255) / 256;
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(gen_vector_kernel, dim3((unsigned)b), dim3(256), 0, s, seed, j0, count, d_x);
  SPL_HIP(hipGetLastError());
}

}  // namespace spl

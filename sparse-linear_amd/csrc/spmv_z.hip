// spmv_z.hip — Complex Double CSR SpMV, native (the reference's second SPECIALIZE instance of axpy_ / mulV,
// sparse-linear/src/Data/Matrix/Sparse.hs:456-457, 465-466).
//
// Round 1 ran complex products through the real 2n x 2n embedding: 4 stored real entries of 12 bytes per
// complex entry.  Here the values are packed (re, im) pairs next to one int32 column index — 20 bytes per
// entry — and x, y are packed complex vectors: one 16-byte gather brings both parts of x[c].
// Semantics, per stored entry in ascending column order:  y[r] <- a * x[c] + y[r]  with base's Data.Complex
// arithmetic,  (a :+ b) * (c :+ d) = (a*c - b*d) :+ (a*d + b*c)  and componentwise (+), every real operation
// separately rounded (this directory is compiled with -ffp-contract=off): bit-identical to the reference order
// for every row shorter than one LDS chunk (longer rows: wavefront tree sum, tolerance-checked), as the real
// CSR-stream kernel of spmv.hip whose structure this kernel shares: one wavefront owns 64 consecutive rows, the
// entries are streamed coalesced and non-temporal, products are staged in a wavefront-private LDS chunk, lane l
// folds row l sequentially.
#include "common.hpp"

namespace spl {

namespace {

typedef double double2v __attribute__((ext_vector_type(2)));
constexpr int kWavesPerBlockZ = 4;
constexpr int kRowsPerBlockZ = kWavesPerBlockZ * 64;

template <int EPL, typename PtrT>
__global__ __launch_bounds__(kWavesPerBlockZ * 64) void spmv_stream_z(
    int64_t nrows, int64_t nblocks, const PtrT *__restrict__ rowptr, const int *__restrict__ colidx,
    const double2v *__restrict__ val, const double2v *__restrict__ x, double2v *__restrict__ y, int accumulate) {
  constexpr int CH = 64 * EPL;
  __shared__ double2v prod_all[kWavesPerBlockZ][CH];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t per_xcd = gridDim.x >> 3;  // XCD-aware remap as in spmv_stream (speed only)
  const int64_t rb = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (rb >= nblocks) return;
  const int64_t r0 = rb * kRowsPerBlockZ + (int64_t)wave * 64;
  if (r0 >= nrows) return;
  double2v *prod = prod_all[wave];
  const int64_t r = r0 + lane;
  const bool valid = r < nrows;
  const int64_t rc = valid ? r : nrows - 1;
  PtrT my_s = rowptr[rc];
  PtrT my_e = rowptr[rc + 1];
  const int nvalid = (nrows - r0) < 64 ? (int)(nrows - r0) : 64;
  const PtrT S = __shfl(my_s, 0, 64);
  const PtrT E = __shfl(my_e, nvalid - 1, 64);
  if (!valid) { my_s = E; my_e = E; }
  double2v acc;
  acc.x = 0.0;
  acc.y = 0.0;
  if (accumulate && valid) acc = y[r];
  for (PtrT b0 = S; b0 < E; b0 += CH) {
    double2v p[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const PtrT k = b0 + (PtrT)(i * 64 + lane);
      p[i].x = 0.0;
      p[i].y = 0.0;
      if (k < E) {
        const int c = __builtin_nontemporal_load(colidx + k);
        const double2v a = __builtin_nontemporal_load(val + k);
        const double2v xv = x[c];
        p[i].x = a.x * xv.x - a.y * xv.y;  // (a :+ b) * (c :+ d) = (a*c - b*d) :+ (a*d + b*c)
        p[i].y = a.x * xv.y + a.y * xv.x;
      }
    }
    // one long row covers the whole chunk: wavefront-wide reduction (order differs: tolerance only)
    const bool covers = (my_s <= b0) && (my_e >= b0 + CH);
    if (__ballot(covers) != 0ull) {
      double pr = 0.0, pi = 0.0;
#pragma unroll
      for (int e = 0; e < EPL; ++e) { pr += p[e].x; pi += p[e].y; }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) { pr += __shfl_xor(pr, d, 64); pi += __shfl_xor(pi, d, 64); }
      if (covers) { acc.x = pr + acc.x; acc.y = pi + acc.y; }
      continue;
    }
#pragma unroll
    for (int i = 0; i < EPL; ++i) prod[i * 64 + lane] = p[i];
    __builtin_amdgcn_wave_barrier();
    const PtrT lo = (my_s > b0 ? my_s : b0) - b0;
    const PtrT hi = (my_e < b0 + CH ? my_e : b0 + CH) - b0;
    for (PtrT t = lo; t < hi; ++t) {
      const double2v q = prod[t];
      acc.x = q.x + acc.x;  // a * x + y, componentwise
      acc.y = q.y + acc.y;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (valid) y[r] = acc;
}

__global__ __launch_bounds__(256) void iota_f64_kernel(int64_t n, double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)i;
}

__global__ __launch_bounds__(256) void gather_z_kernel(int64_t n, const double *__restrict__ pos, const double2v *__restrict__ in,
                                                       double2v *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[(int64_t)pos[i]];
}

}  // namespace

int launch_spmv_z(const Matrix *m, const double *d_x, double *d_y, int accumulate, hipStream_t s) {
  if (m->nrows_local == 0) return SPL_OK;
  const int64_t nblocks = (m->nrows_local + kRowsPerBlockZ - 1) / kRowsPerBlockZ;
  const int64_t grid = ((nblocks + 7) / 8) * 8;
  if (grid > 0x7fffffffLL) return SPL_ERROR_internal;
  const double2v *val = reinterpret_cast<const double2v *>(m->val.get());
  const double2v *x = reinterpret_cast<const double2v *>(d_x);
  double2v *y = reinterpret_cast<double2v *>(d_y);
  if (m->rowptr.get())
    hipLaunchKernelGGL((spmv_stream_z<4, int>), dim3((unsigned)grid), dim3(kWavesPerBlockZ * 64), 0, s, m->nrows_local,
                       nblocks, m->rowptr.get(), m->colidx.get(), val, x, y, accumulate);
  else
    hipLaunchKernelGGL((spmv_stream_z<4, int64_t>), dim3((unsigned)grid), dim3(kWavesPerBlockZ * 64), 0, s,
                       m->nrows_local, nblocks, m->rowptr64.get(), m->colidx.get(), val, x, y, accumulate);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("spmv_z launch", e); return SPL_ERROR_device; }
  return SPL_OK;
}

// positions 0 .. n-1 as doubles (exact below 2^53): the payload that turns the real-valued transpose into a
// permutation, along which the packed complex values are then gathered
void fill_positions(int64_t n, double *d_out, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(iota_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, d_out);
}
void gather_complex_values(int64_t n, const double *d_pos, const double *d_in, double *d_out, hipStream_t s) {
  if (n > 0)
    hipLaunchKernelGGL(gather_z_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, d_pos,
                       reinterpret_cast<const double2v *>(d_in), reinterpret_cast<double2v *>(d_out));
}

}  // namespace spl

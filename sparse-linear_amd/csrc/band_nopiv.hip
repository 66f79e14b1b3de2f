// band_nopiv.hip — blocked band LU WITHOUT row interchanges, and its blocked solves.
//
// Used by umfpack_di_numeric when the permuted matrix is diagonally dominant by columns
// (|a_jj| >= sum_{i != j} |a_ij| for every column, a_jj != 0): elimination then never needs a
// row interchange (partial pivoting would keep the diagonal, up to ties), the growth factor is
// <= 2, and the factors keep the band (kl, ku) of the matrix.  Every discretised Poisson /
// M-matrix of the C5 ladder is of this kind; anything else takes the pivoting path in
// umfpack.hip.  (UMFPACK's own "symmetric strategy" prefers the diagonal for such matrices.)
//
// Storage: AB[(ku + i - j) + j*ldab], ldab >= kl + ku + 1 (band_nopiv_ldab pads it), i.e.
// A(i,j) = AB[ku + i + j*(ldab-1)]: any sub-block inside the band is a dense column-major matrix
// with leading dimension ldab-1.
// Right-looking, block size NB = 64, block steps taken in pairs:
//   diag   : LU of the 64 x 64 diagonal block + explicit inverses of its two factors, one
//            workgroup, the block held in registers (diag_block_factor).  Runs as a look-ahead
//            inside the update kernel (the workgroup of tile (0,0)), off the critical path of
//            the other tiles; the inverses are kept per block for the solves.
//   trsm   : L21 = A21 inv(U11), U12 = inv(L11) A12 as 64x64x64 products on the matrix cores
//   update : A22 -= L21 U12 on 64x64 tiles, K staged through LDS in slices of 32.  The first
//            block step of a pair only updates the panels of the second (an L-shaped strip);
//            then both update the rest of the window in one pass with K = 128, which halves
//            the read-modify-write traffic of the window.  Entries outside the band are read
//            as zero and never written.
// fp64 dense-kernel work: these products are the one contraction-shaped step of the whole backend
// (outside the headline metric) and run on the fp64 matrix cores (v_mfma_f64_16x16x4).
// The solves run in super blocks of 256 unknowns, one launch each: the in-block part is four
// 64x64 matrix-vector products with the stored inverses, the rest a band matrix-vector update.
#include "dense_lu_kernels.hpp"

namespace spl {

namespace {

// ---- dominance test on the CSC arrays (permuted indices do not matter for this property) ----
__global__ __launch_bounds__(256) void col_dominance_kernel(int n, const int *__restrict__ Ap,
                                                            const int *__restrict__ Ai,
                                                            const double *__restrict__ Ax,
                                                            int *__restrict__ not_dominant) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= n) return;
  // the answer is known once one column fails: later wavefronts leave at once, and only the first few raise the flag
  // (2e6 atomics on one word took 15 ms on the embedding of a complex shift at 100^3)
  if (__hip_atomic_load(not_dominant, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  double off = 0.0, diag = 0.0;
  for (int p = Ap[j] + lane; p < Ap[j + 1]; p += 64) {
    const double a = fabs(Ax[p]);
    if (Ai[p] == j) diag += a; else off += a;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    off += __shfl_xor(off, d, 64);
    diag += __shfl_xor(diag, d, 64);
  }
  if (lane == 0 && !(diag > 0.0 && diag >= off)) atomicOr(not_dominant, 1);
}

__global__ __launch_bounds__(256) void band2_scatter_kernel(int n, const int *__restrict__ Ap,
                                                            const int *__restrict__ Ai,
                                                            const double *__restrict__ Ax,
                                                            const int *__restrict__ inv, Band b) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= n) return;
  const int nj = inv[j];
  for (int p = Ap[j] + lane; p < Ap[j + 1]; p += 64) atomicAdd(&b.at(inv[Ai[p]], nj), Ax[p]);
}

}  // namespace

bool band_is_column_dominant(int n, const int *d_Ap, const int *d_Ai, const double *d_Ax, hipStream_t s) {
  if (n == 0) return true;
  DBuf<int> flag(1);
  SPL_HIP(hipMemsetAsync(flag.get(), 0, sizeof(int), s));
  hipLaunchKernelGGL(col_dominance_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, n, d_Ap, d_Ai, d_Ax,
                     flag.get());
  int h = 0;
  SPL_HIP(hipMemcpyAsync(&h, flag.get(), sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  return h == 0;
}

// scatter P A P^T into AB (zeroed, ldab = band_nopiv_ldab(kl, ku)) and factor it in place; returns the singular flag
size_t band_nopiv_inverse_elems(int n) { return inverse_block_elems(n); }

int band_nopiv_factor(int n, int kl, int ku, int ldab, double *d_AB, double *d_invs, const int *d_Ap,
                      const int *d_Ai, const double *d_Ax, const int *d_inv, hipStream_t s) {
  const Band b = band_view(d_AB, n, kl, ku, ldab);
  hipLaunchKernelGGL(band2_scatter_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, n, d_Ap, d_Ai, d_Ax,
                     d_inv, b);
  DBuf<int> singular(1);
  SPL_HIP(hipMemsetAsync(singular.get(), 0, sizeof(int), s));
  // wide bands: the look-ahead tile of every window on a helper stream (factor_loop)
  hipStream_t helper = nullptr;
  if (std::min(kl, ku) / 64 + 1 >= kSplitTiles) SPL_HIP(hipStreamCreateWithFlags(&helper, hipStreamNonBlocking));
  int h = 0;
  try {
    factor_loop(b, n, d_invs, singular.get(), s, helper);
    SPL_HIP(hipMemcpyAsync(&h, singular.get(), sizeof(int), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    SPL_HIP(hipGetLastError());
  } catch (...) {
    if (helper) { (void)hipStreamSynchronize(helper); (void)hipStreamDestroy(helper); }
    throw;
  }
  if (helper) (void)hipStreamDestroy(helper);  // joined into s by factor_loop's events, and s is idle
  return h;
}

// columns of c (device, permuted order, column r at d_c + r * stride) <- B^-1 c (sys 0) or B^-T c
// (sys 1) with the no-pivot factors.  Right-hand sides go through the band four at a time; the
// caller pads to a multiple of kSolveGroup columns when nrhs > 1.
void band_nopiv_solve(int sys, int n, int kl, int ku, int ldab, const double *d_AB, const double *d_invs,
                      double *d_c, int nrhs, size_t stride, hipStream_t s) {
  if (n == 0 || nrhs == 0) return;
  const Band b = band_view(const_cast<double *>(d_AB), n, kl, ku, ldab);
  if (nrhs == 1) {
    DBuf<double> z((size_t)n);
    solve_group<1>(sys, b, d_invs, d_c, z.get(), stride, s);
    SPL_HIP(hipStreamSynchronize(s));  // z is freed on return
    return;
  }
  DBuf<double> z((size_t)kSolveGroup * stride);
  for (int c0 = 0; c0 < nrhs; c0 += kSolveGroup)
    solve_group<kSolveGroup>(sys, b, d_invs, d_c + (size_t)c0 * stride, z.get(), stride, s);
  SPL_HIP(hipStreamSynchronize(s));
}

}  // namespace spl

// band_nopiv.hip — blocked band LU WITHOUT row interchanges, and its blocked solves.
//
// Used by umfpack_di_numeric when the permuted matrix is diagonally dominant by columns
// (|a_jj| >= sum_{i != j} |a_ij| for every column, a_jj != 0): elimination then never needs a
// row interchange (partial pivoting would keep the diagonal, up to ties), the growth factor is
// <= 2, and the factors keep the band (kl, ku) of the matrix.  Every discretised Poisson /
// M-matrix of the C5 ladder is of this kind; anything else takes the pivoting path in
// umfpack.hip.  (UMFPACK's own "symmetric strategy" prefers the diagonal for such matrices.)
//
// Storage: AB[(ku + i - j) + j*ldab], ldab = kl + ku + 1, i.e. A(i,j) = AB[ku + i + j*(ldab-1)]:
// any sub-block inside the band is a dense column-major matrix with leading dimension ldab-1.
// Right-looking, block size NB = 32:
//   diag   : LU of the NB x NB diagonal block in LDS (one workgroup)
//   trsm_L : L21 = A21 U11^-1      one thread per row   (rows below the block, <= kl+NB-1 of them)
//   trsm_U : U12 = L11^-1 A12      one thread per column
//   gemm   : A22 -= L21 U12        64x64 tiles, K = NB, operands staged in LDS; entries outside
//            the band are read as zero and never written
// fp64 dense-kernel work: this GEMM is the one contraction-shaped step of the whole backend
// (it is outside the headline metric); it uses plain fp64 FMAs here, an MFMA-f64 tile is the
// obvious follow-up.  The solves are blocked the same way (diagonal block in LDS, then one
// thread per affected row / one workgroup per affected column for the transposed forms).
#include <algorithm>

#include "common.hpp"

// dense-kernel work with no bit-parity contract (parity of the solve step is defined on the
// solution): allow fused multiply-adds here although the library default is -ffp-contract=off
#pragma clang fp contract(fast)

namespace spl {

namespace {

constexpr int NB = 32;

struct Band {
  double *AB;
  int n, kl, ku, ldab;
  __device__ __forceinline__ bool in_band(int i, int j) const { return i - j <= kl && j - i <= ku; }
  __device__ __forceinline__ double &at(int i, int j) const {
    return AB[(size_t)(ku + i) + (size_t)j * (size_t)(ldab - 1)];
  }
  __device__ __forceinline__ double get(int i, int j) const { return in_band(i, j) ? at(i, j) : 0.0; }
};

// ---- dominance test on the CSC arrays (permuted indices do not matter for this property) ----
__global__ __launch_bounds__(256) void col_dominance_kernel(int n, const int *__restrict__ Ap,
                                                            const int *__restrict__ Ai,
                                                            const double *__restrict__ Ax,
                                                            int *__restrict__ not_dominant) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= n) return;
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

// ---- factorisation ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diag_lu_kernel(Band b, int j0, int jb, int *__restrict__ singular) {
  __shared__ double D[NB][NB + 1];
  const int tid = threadIdx.x;
  for (int t = tid; t < NB * NB; t += 256) {
    const int r = t % NB, c = t / NB;
    D[r][c] = (r < jb && c < jb) ? b.get(j0 + r, j0 + c) : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int k = 0; k < jb; ++k) {
    const double piv = D[k][k];
    if (piv == 0.0) {
      if (tid == 0) atomicOr(singular, 1);
    } else {
      if (tid > k && tid < jb) D[tid][k] = D[tid][k] / piv;
    }
    __syncthreads();
    const int m = jb - k - 1;
    for (int t = tid; t < m * m; t += 256) {
      const int r = k + 1 + t % m, c = k + 1 + t / m;
      D[r][c] -= D[r][k] * D[k][c];
    }
    __syncthreads();
  }
  for (int t = tid; t < jb * jb; t += 256) {
    const int r = t % jb, c = t / jb;
    if (b.in_band(j0 + r, j0 + c)) b.at(j0 + r, j0 + c) = D[r][c];
  }
}

// L21 = A21 * U11^-1 (thread per row) and U12 = L11^-1 * A12 (thread per column) in one launch
__global__ __launch_bounds__(256) void trsm_kernel(Band b, int j0, int jb, int nrows_below, int ncols_right) {
  __shared__ double D[NB][NB + 1];
  const int tid = threadIdx.x;
  for (int t = tid; t < NB * NB; t += 256) {
    const int r = t % NB, c = t / NB;
    D[r][c] = (r < jb && c < jb) ? b.get(j0 + r, j0 + c) : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  const int g = blockIdx.x * 256 + tid;
  if (g < nrows_below) {
    const int i = j0 + jb + g;
    double x[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) x[t] = (t < jb) ? b.get(i, j0 + t) : 0.0;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      double acc = x[t];
#pragma unroll
      for (int s = 0; s < NB; ++s)
        if (s < t) acc -= x[s] * D[s][t];
      x[t] = acc / D[t][t];
    }
#pragma unroll
    for (int t = 0; t < NB; ++t)
      if (t < jb && b.in_band(i, j0 + t)) b.at(i, j0 + t) = x[t];
  } else if (g - nrows_below < ncols_right) {
    const int j = j0 + jb + (g - nrows_below);
    double u[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) u[t] = (t < jb) ? b.get(j0 + t, j) : 0.0;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      double acc = u[t];
#pragma unroll
      for (int s = 0; s < NB; ++s)
        if (s < t) acc -= D[t][s] * u[s];
      u[t] = acc;  // unit lower triangular
    }
#pragma unroll
    for (int t = 0; t < NB; ++t)
      if (t < jb && b.in_band(j0 + t, j)) b.at(j0 + t, j) = u[t];
  }
}

// A22 -= L21 * U12 on 64x64 tiles (4x4 outputs per thread), K = jb <= NB
__global__ __launch_bounds__(256) void gemm_update_kernel(Band b, int j0, int jb, int nrows_below,
                                                          int ncols_right) {
  const int r0 = j0 + jb + blockIdx.x * 64, c0 = j0 + jb + blockIdx.y * 64;
  // the whole tile lies outside the band: nothing to do
  if (r0 - (c0 + 63) > b.kl || c0 - (r0 + 63) > b.ku) return;
  __shared__ double Ls[NB][64 + 1];  // Ls[t][r] = L(r0 + r, j0 + t)
  __shared__ double Us[NB][64 + 1];  // Us[t][c] = U(j0 + t, c0 + c)
  const int tid = threadIdx.x;
  const int rend = j0 + jb + nrows_below, cend = j0 + jb + ncols_right;
  for (int t = tid; t < NB * 64; t += 256) {
    const int r = t % 64, k = t / 64;
    const int i = r0 + r;
    Ls[k][r] = (k < jb && i < rend) ? b.get(i, j0 + k) : 0.0;
  }
  for (int t = tid; t < NB * 64; t += 256) {
    const int k = t % NB, c = t / NB;
    const int j = c0 + c;
    Us[k][c] = (k < jb && j < cend) ? b.get(j0 + k, j) : 0.0;
  }
  __syncthreads();
  const int tr = (tid % 16) * 4, tc = (tid / 16) * 4;
  double acc[4][4] = {};
#pragma unroll 8
  for (int k = 0; k < NB; ++k) {
    double l[4], u[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { l[a] = Ls[k][tr + a]; u[a] = Us[k][tc + a]; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][c] += l[a] * u[c];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int i = r0 + tr + a, j = c0 + tc + c;
      if (i < rend && j < cend && b.in_band(i, j)) b.at(i, j) -= acc[a][c];
    }
}

// ---- blocked solves -----------------------------------------------------------------------------
// mode 0: L (unit lower) forward, 1: U backward, 2: U^T forward, 3: L^T (unit) backward.
// One workgroup: fold in the contributions of already-solved entries for modes 2/3 ("left-
// looking" dots, coalesced down a column), solve the diagonal block in LDS, write the block.
__global__ __launch_bounds__(256) void solve_diag_kernel(Band b, int mode, int j0, int jb, double *c) {
  __shared__ double D[NB][NB + 1];
  __shared__ double v[NB];
  const int tid = threadIdx.x;
  for (int t = tid; t < NB * NB; t += 256) {
    const int r = t % NB, cc = t / NB;
    D[r][cc] = (r < jb && cc < jb) ? b.get(j0 + r, j0 + cc) : (r == cc ? 1.0 : 0.0);
  }
  if (tid < NB) v[tid] = tid < jb ? c[j0 + tid] : 0.0;
  __syncthreads();
  if (mode == 2 || mode == 3) {
    // v[t] -= sum_i M(i, j0+t) * c[i] over the solved entries i adjacent to the block:
    // 8 threads per block column t, consecutive threads on consecutive i (contiguous down a column)
    const int t = tid >> 3, part = tid & 7;
    double s = 0.0;
    if (t < jb) {
      const int j = j0 + t;
      if (mode == 2) {
        const int lo = max(0, j - b.ku);
        for (int i = lo + part; i < j0; i += 8) s += b.at(i, j) * c[i];  // U(i,j), i < j0
      } else {
        const int hi = min(b.n - 1, j + b.kl);
        for (int i = j0 + jb + part; i <= hi; i += 8) s += b.at(i, j) * c[i];  // L(i,j) beyond the block
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (part == 0 && t < jb) v[t] -= s;
    __syncthreads();
  }
  if (tid < 64) {  // NB x NB triangular solve by one wavefront: lane l owns v[l], columns broadcast by shuffle
    const int l = tid;
    double x = l < NB ? v[l] : 0.0;
    if (mode == 0) {  // L unit lower, forward
      for (int t = 0; t < jb; ++t) {
        const double xt = __shfl(x, t, 64);
        if (l > t && l < jb) x -= D[l][t] * xt;
      }
    } else if (mode == 1) {  // U, backward
      for (int t = jb - 1; t >= 0; --t) {
        if (l == t) x = x / D[t][t];
        const double xt = __shfl(x, t, 64);
        if (l < t) x -= D[l][t] * xt;
      }
    } else if (mode == 2) {  // U^T lower, forward
      for (int t = 0; t < jb; ++t) {
        if (l == t) x = x / D[t][t];
        const double xt = __shfl(x, t, 64);
        if (l > t && l < jb) x -= D[t][l] * xt;
      }
    } else {  // L^T unit upper, backward
      for (int t = jb - 1; t >= 0; --t) {
        const double xt = __shfl(x, t, 64);
        if (l < t) x -= D[t][l] * xt;
      }
    }
    if (l < jb) c[j0 + l] = x;
  }
}

// right-looking update after a diagonal solve (modes 0 and 1): one thread per affected row
__global__ __launch_bounds__(256) void solve_update_kernel(Band b, int mode, int j0, int jb, double *c) {
  __shared__ double v[NB];
  if (threadIdx.x < NB) v[threadIdx.x] = threadIdx.x < jb ? c[j0 + threadIdx.x] : 0.0;
  __syncthreads();
  const int g = blockIdx.x * 256 + threadIdx.x;
  int i;
  if (mode == 0) {
    i = j0 + jb + g;
    if (i >= b.n || i > j0 + jb - 1 + b.kl) return;
  } else {
    i = j0 - 1 - g;
    if (i < 0 || i < j0 - b.ku) return;
  }
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < NB; ++t)
    if (t < jb) s += b.get(i, j0 + t) * v[t];
  c[i] -= s;
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

// scatter P A P^T into AB (zeroed, ldab = kl+ku+1) and factor it in place; returns the singular flag
int band_nopiv_factor(int n, int kl, int ku, double *d_AB, const int *d_Ap, const int *d_Ai,
                      const double *d_Ax, const int *d_inv, hipStream_t s) {
  Band b{d_AB, n, kl, ku, kl + ku + 1};
  hipLaunchKernelGGL(band2_scatter_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, n, d_Ap, d_Ai, d_Ax,
                     d_inv, b);
  DBuf<int> singular(1);
  SPL_HIP(hipMemsetAsync(singular.get(), 0, sizeof(int), s));
  for (int j0 = 0; j0 < n; j0 += NB) {
    const int jb = std::min(NB, n - j0);
    hipLaunchKernelGGL(diag_lu_kernel, dim3(1), dim3(256), 0, s, b, j0, jb, singular.get());
    const int below = std::max(0, std::min(n, j0 + jb + kl) - (j0 + jb));   // rows with any in-band entry
    const int right = std::max(0, std::min(n, j0 + jb + ku) - (j0 + jb));
    if (below + right > 0)
      hipLaunchKernelGGL(trsm_kernel, dim3((unsigned)((below + right + 255) / 256)), dim3(256), 0, s, b, j0, jb,
                         below, right);
    if (below > 0 && right > 0)
      hipLaunchKernelGGL(gemm_update_kernel, dim3((unsigned)((below + 63) / 64), (unsigned)((right + 63) / 64)),
                         dim3(256), 0, s, b, j0, jb, below, right);
  }
  int h = 0;
  SPL_HIP(hipMemcpyAsync(&h, singular.get(), sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  SPL_HIP(hipGetLastError());
  return h;
}

// c (device, permuted order) <- B^-1 c (sys 0) or B^-T c (sys 1) with the no-pivot factors
void band_nopiv_solve(int sys, int n, int kl, int ku, const double *d_AB, double *d_c, hipStream_t s) {
  if (n == 0) return;
  Band b{const_cast<double *>(d_AB), n, kl, ku, kl + ku + 1};
  const int nblk = (n + NB - 1) / NB;
  if (sys == 0) {
    for (int k = 0; k < nblk; ++k) {  // L forward
      const int j0 = k * NB, jb = std::min(NB, n - j0);
      hipLaunchKernelGGL(solve_diag_kernel, dim3(1), dim3(256), 0, s, b, 0, j0, jb, d_c);
      const int rows = std::max(0, std::min(n, j0 + jb + kl) - (j0 + jb));
      if (rows > 0)
        hipLaunchKernelGGL(solve_update_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, b, 0, j0, jb,
                           d_c);
    }
    for (int k = nblk - 1; k >= 0; --k) {  // U backward
      const int j0 = k * NB, jb = std::min(NB, n - j0);
      hipLaunchKernelGGL(solve_diag_kernel, dim3(1), dim3(256), 0, s, b, 1, j0, jb, d_c);
      const int rows = std::min(j0, ku);
      if (rows > 0)
        hipLaunchKernelGGL(solve_update_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, b, 1, j0, jb,
                           d_c);
    }
  } else {
    for (int k = 0; k < nblk; ++k) {  // U^T forward
      const int j0 = k * NB, jb = std::min(NB, n - j0);
      hipLaunchKernelGGL(solve_diag_kernel, dim3(1), dim3(256), 0, s, b, 2, j0, jb, d_c);
    }
    for (int k = nblk - 1; k >= 0; --k) {  // L^T backward
      const int j0 = k * NB, jb = std::min(NB, n - j0);
      hipLaunchKernelGGL(solve_diag_kernel, dim3(1), dim3(256), 0, s, b, 3, j0, jb, d_c);
    }
  }
}

}  // namespace spl

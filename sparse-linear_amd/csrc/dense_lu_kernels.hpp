// dense_lu_kernels.hpp — the blocked LU kernels WITHOUT row interchanges and their blocked solves,
// shared by the band path (band_nopiv.hip) and the multifrontal fronts (multifrontal.hip).  Included
// into each of those translation units; everything lives in an anonymous namespace.  See
// band_nopiv.hip for the algorithm notes.
#pragma once

#include <algorithm>

#include "common.hpp"

// dense-kernel work with no bit-parity contract (parity of the solve step is defined on the
// solution): allow fused multiply-adds here although the library default is -ffp-contract=off
#pragma clang fp contract(fast)

namespace spl {
namespace {

constexpr int NB = 64;   // panel width

// View of the matrix being factored.  Band storage: A(i,j) = AB[(ku + i) + j*(ldab-1)], doff = ku.
// A dense column-major matrix with leading dimension ld is the same thing with doff = 0,
// ldab = ld + 1 and kl = ku = n (every entry "in band"): the multifrontal fronts use that.
struct Band {
  double *AB;
  int n, kl, ku, ldab, doff;
  // sym = 1 (multifrontal fronts of a matrix with A == A^T, round 3): LU without interchanges of a symmetric matrix
  // is L D L^T, U = D L^T.  Then the pivot rows are not solved for but written as the scaled transposes of the pivot
  // columns (trsm_tile), and the trailing update only computes the tiles on and below the diagonal (update_tile): half
  // the flops of the update, which is 89 % of the factorisation at config C5.  Entries above the diagonal TILES of a
  // trailing block are stale from then on; nothing reads them (extend_add_kernel mirrors the lower triangle).
  int sym = 0;
  // zoff != 0: a COMPLEX dense front in two planes (round 3, native `zi` fronts): the real parts at AB, the imaginary
  // parts zoff doubles further, same leading dimension.  The *_z device functions below read both; every index
  // computation is the one of the real kernels.
  size_t zoff = 0;
  // piv = 1 (multifrontal fronts of a matrix that is not diagonally dominant, round 3): threshold partial pivoting
  // INSIDE every 64 x 64 diagonal block — the rows of a pivot block are fully summed, so any of them may serve as the
  // pivot of a column of the block.  The interchange stays local: P_b A_bb = L_bb U_bb, and P_b is folded into the
  // stored inverse (inv(L11) P_b), through which the panel solves and the triangular solves see the block; nothing
  // outside diag_block_factor knows.  A pivot is taken out of its natural turn only when it is below 0.1 of the largest
  // candidate (UMFPACK's default threshold), so well-behaved blocks give the factors they gave without it.
  int piv = 0;
  // ... candidates are compared AFTER scaling by rscale[row of the view] when given (1 / sum of the absolute values of
  // that row of the matrix: UMFPACK's default row scaling) — magnitudes of rows with different scales say nothing
  const double *rscale = nullptr;
  __device__ __forceinline__ bool in_band(int i, int j) const { return i - j <= kl && j - i <= ku; }
  __device__ __forceinline__ double &at(int i, int j) const {
    return AB[(size_t)(doff + i) + (size_t)j * (size_t)(ldab - 1)];
  }
  __device__ __forceinline__ double get(int i, int j) const { return in_band(i, j) ? at(i, j) : 0.0; }
  // the same with a streaming hint: factor panels in a triangular solve are read once per walk
  __device__ __forceinline__ double get_nt(int i, int j) const { return in_band(i, j) ? __builtin_nontemporal_load(&at(i, j)) : 0.0; }
};
// Band::piv of a factorisation WITHOUT interchanges: 0 (diagonal blocks by panels of 16 on the matrix cores, round 4), or 2
// with SPL_LU_DIAG=plain in the environment (the unblocked form of rounds 1 - 3: ablation)
inline int diag_form_without_interchanges() {
  const char *e = getenv("SPL_LU_DIAG");
  return (e && e[0] == 'p') ? 2 : 0;
}
inline Band band_view(double *AB, int n, int kl, int ku, int ldab) {
  Band b{AB, n, kl, ku, ldab, ku};
  b.piv = diag_form_without_interchanges();
  return b;
}
inline Band dense_view(double *F, int n, int ld, int sym = 0, size_t zoff = 0, int piv = 0, const double *rscale = nullptr) {
  return Band{F, n, n, n, ld + 1, 0, sym, zoff, piv, rscale};
}

// ---- factorisation ----------------------------------------------------------------------------
typedef double double4v __attribute__((ext_vector_type(4)));
constexpr int LDP = NB + 1;                                  // padded leading dimension in LDS
constexpr size_t kTileBytes = (size_t)NB * LDP * sizeof(double);

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// LU of the diagonal block held in LDS tile D (identity-padded beyond jb), then the explicit
// inverses of its two triangular factors: the triangular solves of the panel then become small
// GEMMs on the matrix cores, and a block of the triangular solve a matrix-vector product.
// Diagonally dominant blocks are well conditioned, so explicit inverses are safe here.
//
// This chain of 64 dependent pivots sits on the critical path of every block step, so it is kept
// in registers: 256 threads, lane = row, wave w owns the columns c = w mod 4 (16 per thread, all
// indices static after unrolling).  Row k of a rank-1 update comes from lane k of the same wave
// (v_readlane); only the multiplier column crosses waves, through LDS, one barrier per pivot.
// The two inverses are built the same way (right-looking substitution on an identity), U^-1 with
// k descending and L^-1 with k ascending in the same loop.
// Results: LU'd block to band storage and tile D, inverses (NB x NB column-major) to invL / invU.
constexpr double kPivotThreshold = 0.1;

template <bool PIV>
__device__ __forceinline__ void diag_block_factor_t(const Band &b, int j0, int jb, double (*D)[LDP],
                                                    double (*lcol)[NB], int *__restrict__ singular,
                                                    double *__restrict__ invL, double *__restrict__ invU) {
  const int tid = threadIdx.x;
  const int tr = tid & 63, tc = tid >> 6;
  constexpr int NS = NB / 4;  // register slots per thread
  double a[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) a[u] = D[tr][4 * u + tc];
  // PIV: implicit row interchanges.  `done`: this row has been a pivot; `mypos`: in which step.  The pivot row of step
  // k travels to the other wavefronts through the padding word of row k of tile D.
  bool done = false;
  int mypos = tr;
  const double rs = (PIV && b.rscale && tr < jb) ? b.rscale[j0 + tr] : 1.0;
  // The pivot loops stay rolled (straight-line code of this size would run at instruction-fetch
  // speed).  Register slots must be indexed statically, so the slots are shifted down by one
  // after every group of 4 pivots: in group g slot j holds column 4 (g + j) + wave, the pivot
  // columns of the group are always slot 0, and finished columns leave through tile D.
#pragma unroll 1
  for (int g = 0; g < NS; ++g) {
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) {
      const int k = 4 * g + kw;
      if (tc == kw) {
        int prow = k;
        if (PIV) {
          // largest candidate among the rows that have not been pivots yet (ties: the smallest row), and the row whose
          // natural turn it is: row k, or the first unused one if row k went earlier
          double best = done ? -1.0 : fabs(a[0]) * rs;
          int brow = tr;
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int orow = __shfl_xor(brow, off, 64);
            if (ob > best || (ob == best && orow < brow)) { best = ob; brow = orow; }
          }
          const unsigned long long unused = __ballot(!done);
          const int nat = ((unused >> k) & 1ull) ? k : (int)__ffsll((long long)unused) - 1;
          const double anat = fabs(readlane_f64(a[0], nat)) * readlane_f64(rs, nat);
          prow = (anat >= kPivotThreshold * best) ? nat : brow;
          prow = __builtin_amdgcn_readfirstlane(prow);
        }
        const double piv = readlane_f64(a[0], prow);
        if (piv == 0.0) {
          if (tr == 0) atomicOr(singular, 1);
        } else if (PIV ? (!done && tr != prow) : (tr > k)) {
          a[0] = a[0] / piv;
        }
        lcol[k & 1][tr] = a[0];
        if (PIV && tr == 0) reinterpret_cast<int &>(D[k][NB]) = prow;
      }
      __syncthreads();
      int prow = k;
      if (PIV) {
        prow = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int &>(D[k][NB]));
        if (tr == prow) { done = true; mypos = k; }
      }
      const double l = (PIV ? !done : (tr > k)) ? lcol[k & 1][tr] : 0.0;
      if (tc > kw) a[0] -= l * readlane_f64(a[0], prow);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (g + 4 * qd < NS) {  // some slot of this quarter still holds a live column
#pragma unroll
          for (int j = (qd == 0 ? 1 : 4 * qd); j < 4 * qd + 4; ++j) a[j] -= l * readlane_f64(a[j], prow);
        }
      }
    }
    D[tr][4 * g + tc] = a[0];
#pragma unroll
    for (int j = 0; j + 1 < NS; ++j) a[j] = a[j + 1];
    a[NS - 1] = 0.0;
  }
  __syncthreads();
  if (PIV) {
    // rows into pivot order: D'[step in which the row was the pivot] = D[row]; the padding words keep the pivot rows
    double v[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) v[u] = D[tr][4 * u + tc];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NS; ++u) D[mypos][4 * u + tc] = v[u];
    __syncthreads();
  }
  for (int c = tc; c < jb; c += 4)
    if (tr < jb && b.in_band(j0 + tr, j0 + c)) b.at(j0 + tr, j0 + c) = D[tr][c];
  // reciprocals of the pivots, once
  double *dinv = &lcol[0][0];
  if (tid < NB) {
    const double d = D[tid][tid];
    dinv[tid] = d != 0.0 ? 1.0 / d : 0.0;
  }
  __syncthreads();
  double x[NS], y[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) x[u] = y[u] = (tr == 4 * u + tc) ? 1.0 : 0.0;
#pragma unroll 1
  for (int s = 0; s < NB; ++s) {
    {  // U^-1: row k final after scaling by 1/U(k,k); rows above lose U(i,k) * row k
      const int k = NB - 1 - s, ks = k >> 2;
      const double dk = dinv[k];
      const double uik = tr < k ? D[tr][k] : 0.0;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (4 * qd + 3 >= ks) {  // X(k, c) is zero for c < k
#pragma unroll
          for (int u = 4 * qd; u < 4 * qd + 4; ++u) {
            const double xk = readlane_f64(x[u], k) * dk;
            x[u] = (tr == k) ? xk : x[u] - uik * xk;
          }
        }
      }
    }
    {  // L^-1 (unit diagonal): rows below lose L(i,k) * row k
      const int k = s, ks = k >> 2;
      const double lik = tr > k ? D[tr][k] : 0.0;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (4 * qd <= ks) {  // Y(k, c) is zero for c > k
#pragma unroll
          for (int u = 4 * qd; u < 4 * qd + 4; ++u) y[u] -= lik * readlane_f64(y[u], k);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int c = 4 * u + tc;
    invU[tr + c * NB] = x[u];
    // PIV: inv(L11) P_b — column c of inv(L11) multiplies the row that was the pivot of step c
    invL[tr + (PIV ? reinterpret_cast<const int &>(D[c][NB]) : c) * NB] = y[u];
  }
}

// The same without interchanges, BLOCKED by 16 (round 4).  The form above pays about 1 000 cycles per pivot — a
// division, an LDS round trip and a workgroup barrier, because column k belongs to wavefront k mod 4, then 16 dependent
// v_readlane + multiply-add pairs — and as many per step of the two inverses: 27 + 27 us per 64 x 64 block
// (tools/probe/diag_bench.hip), on the chain of every block step of a front (a 100^3 factorisation spends 63 of its
// 168 ms in 837 such steps, GPU busy but at a concurrency of 1.3: tools/trace_gaps.py).  Here:
//   LU      four panels of 16 columns.  One wavefront factors a panel in registers (lane = row, 16 columns, the
//           multiplier stays in its lane, the pivot row comes by v_readlane: no barrier inside a panel); a second one
//           solves the 16 pivot rows of the remaining columns (lane = column, the 120 multipliers as broadcast LDS
//           reads); the trailing 16 x 16 blocks are updated on the matrix cores (v_mfma_f64_16x16x4: lane l supplies
//           A[l % 16][k0 + l / 16] and B[k0 + l / 16][l % 16], holds C[l / 16 + 4 r][l % 16]).  Three barriers a panel.
//   inverses  by the block recurrence of a triangular inverse, in place in the tile: the four diagonal 16 x 16 blocks
//           of both factors by substitution inside a wavefront (lanes 0 - 15 rows of L_ww, 16 - 31 rows of U_ww with
//           their columns mirrored, so that both halves walk the same registers), then block column by block column —
//           L from the right, U from the left, one pair per stage — X_ij = -(sum_k X_ik T_k + M_ii T_i) with
//           T_k = L_kj M_jj: sixteen 16 x 16 x 16 products per factor on the matrix cores.  An MFMA result is laid out
//           exactly as the B operand of the next product wants it, so the T_k never leave the registers.
// Sums are associated differently from the form above (the matrix cores accumulate four products at a time, fused):
// factors and inverses agree with it to rounding level, not bit for bit.
__device__ __forceinline__ double4v mfma16(double a, double b, double4v acc) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
}
// acc += Ablk * T, Ablk = the 16 x 16 block of tile D at (r0, c0) — masked: 0 full block, 1 unit lower triangle (ones on
// the diagonal, zeros above), 2 upper triangle with its diagonal — and T a 16 x 16 block in MFMA result layout
__device__ __forceinline__ double4v block_times(const double (*D)[LDP], int r0, int c0, int mask, double4v T, double4v acc) {
  const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = 4 * r + kq;
    double a = D[r0 + i][c0 + k];
    if (mask == 1) a = i > k ? a : (i == k ? 1.0 : 0.0);
    if (mask == 2) a = i <= k ? a : 0.0;
    acc = mfma16(a, T[r], acc);
  }
  return acc;
}
// the 16 x 16 block of tile D at (r0, c0) as an MFMA B operand / result layout: element r = blk[lane / 16 + 4 r][lane % 16]
__device__ __forceinline__ double4v block_as_b(const double (*D)[LDP], int r0, int c0, int mask) {
  const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
  double4v T;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = 4 * r + kq;
    double v = D[r0 + k][c0 + j];
    if (mask == 1) v = k > j ? v : (k == j ? 1.0 : 0.0);
    if (mask == 2) v = k <= j ? v : 0.0;
    T[r] = v;
  }
  return T;
}

__device__ __forceinline__ void diag_block_factor_blocked(const Band &b, int j0, int jb, double (*D)[LDP],
                                                          int *__restrict__ singular, double *__restrict__ invL,
                                                          double *__restrict__ invU) {
  constexpr int PB = 16, NP = NB / PB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- LU, panel by panel
#pragma unroll 1
  for (int p = 0; p < NP; ++p) {
    const int c0 = p * PB;
    if (wave == 0) {
      double a[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) a[c] = D[lane][c0 + c];
      bool zero_pivot = false;
#pragma unroll
      for (int kk = 0; kk < PB; ++kk) {
        // straight-line code: with a branch on the (uniform) zero test the compiler copies the sixteen registers of the
        // row around every pivot; a zero pivot divides by one instead (exact) and is reported after the panel
        const int k = c0 + kk;
        const double piv = readlane_f64(a[kk], k);
        const bool zero = piv == 0.0;
        zero_pivot |= zero;
        // (a true division: a refined reciprocal — v_rcp_f64 + two Newton steps — saves 0.7 of 34 us and costs half an
        // ulp per multiplier, which a static-pivoting case at the edge of what refinement recovers did not forgive)
        const double q = a[kk] / (zero ? 1.0 : piv);
        const double l = lane > k ? q : 0.0;
        a[kk] = lane > k ? q : a[kk];
#pragma unroll
        for (int c = kk + 1; c < PB; ++c) a[c] = __builtin_fma(-l, readlane_f64(a[c], k), a[c]);  // (ds_bpermute instead of v_readlane: slower, 34 -> 37 us)
      }
      if (zero_pivot && lane == 0) atomicOr(singular, 1);
#pragma unroll
      for (int c = 0; c < PB; ++c) D[lane][c0 + c] = a[c];
    }
    __syncthreads();
    if (p + 1 == NP) break;
    if (wave == 1) {  // the pivot rows of the columns to the right: U12 = inv(L_pp) A12, lane = column
      const int col = c0 + PB + lane;
      if (col < NB) {
        double u[PB];
#pragma unroll
        for (int r = 0; r < PB; ++r) u[r] = D[c0 + r][col];
#pragma unroll
        for (int k = 0; k + 1 < PB; ++k)
#pragma unroll
          for (int r = k + 1; r < PB; ++r) u[r] = __builtin_fma(-D[c0 + r][c0 + k], u[k], u[r]);
#pragma unroll
        for (int r = 1; r < PB; ++r) D[c0 + r][col] = u[r];
      }
    }
    __syncthreads();
    const int nbk = NP - 1 - p;  // trailing blocks per side
    for (int t = wave; t < nbk * nbk; t += 4) {
      const int r0 = (p + 1 + t / nbk) * PB, q0 = (p + 1 + t % nbk) * PB;
      double4v acc = {0.0, 0.0, 0.0, 0.0};
      acc = block_times(D, r0, c0, 0, block_as_b(D, c0, q0, 0), acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) D[r0 + (lane >> 4) + 4 * r][q0 + (lane & 15)] -= acc[r];
    }
    __syncthreads();
  }
  // the factors to band storage
  for (int c = wave; c < jb; c += 4)
    if (lane < jb && b.in_band(j0 + lane, j0 + c)) b.at(j0 + lane, j0 + c) = D[lane][c];
  __syncthreads();  // (the inverses overwrite the tile)
  // ---- inverses, in place.  Diagonal blocks: wavefront w takes block w of both factors
  {
    const int d0 = wave * PB, half = lane >> 4, r = lane & 15;  // half 0: a row of L_ww, 1: of U_ww (2, 3: idle)
    // z[c]: L rows hold column c of the inverse, U rows column 15 - c (both halves then need columns 0 .. s at step s)
    double z[PB], co[PB];  // co[s]: this row's coefficient at step s: L(r, s), or U(r, 15 - s)
#pragma unroll
    for (int c = 0; c < PB; ++c) {
      z[c] = ((half == 0 ? c : PB - 1 - c) == r) ? 1.0 : 0.0;
      co[c] = half < 2 ? D[d0 + r][d0 + (half == 0 ? c : PB - 1 - c)] : 0.0;
    }
    double dinv = 1.0;  // 1 / U(r, r) in the U rows
    if (half == 1) {
      const double d = D[d0 + r][d0 + r];
      dinv = d != 0.0 ? 1.0 / d : 0.0;
    }
#pragma unroll
    for (int s = 0; s < PB; ++s) {
      // L: row s of inv(L_ww) is final (unit diagonal); rows below lose L(r, s) x it.  U: row k = 15 - s is final after
      // scaling by 1 / U(k, k); rows above lose U(r, k) x it.
      const double dk = readlane_f64(dinv, 16 + PB - 1 - s);
      const bool below = half == 0 && r > s, above = half == 1 && r < PB - 1 - s, isk = half == 1 && r == PB - 1 - s;
      const double cf = (below || above) ? co[s] : 0.0;
#pragma unroll
      for (int c = 0; c <= s; ++c) {
        const double vl = readlane_f64(z[c], s), vu = readlane_f64(z[c], 16 + PB - 1 - s) * dk;
        const double v = half == 0 ? vl : vu;
        z[c] = isk ? vu : __builtin_fma(-cf, v, z[c]);
      }
    }
    if (half == 0) {
#pragma unroll
      for (int c = 0; c < PB; ++c)
        if (c < r) D[d0 + r][d0 + c] = z[c];
    } else if (half == 1) {
#pragma unroll
      for (int c = 0; c < PB; ++c)
        if (PB - 1 - c >= r) D[d0 + r][d0 + PB - 1 - c] = z[c];
    }
  }
  __syncthreads();
  // off-diagonal blocks: stage st pairs block column jl = 2 - st of inv(L) (rows below it) with ju = 1 + st of inv(U)
#pragma unroll 1
  for (int st = 0; st < NP - 1; ++st) {
    const int jl = NP - 2 - st, ju = 1 + st, nout = st + 1;  // nout outputs per factor
    // outputs 0 .. nout-1: L blocks (i = jl + 1 + t, jl); nout .. 2 nout - 1: U blocks (i = t', ju), t' = 0 .. ju - 1;
    // a wavefront takes outputs wave and wave + 4 (at most six per stage)
    auto output = [&](int t, int &i, int &j) -> double4v {
      const bool isL = t < nout;
      j = isL ? jl : ju;
      i = isL ? jl + 1 + t : t - nout;
      const double4v Mjj = block_as_b(D, j * PB, j * PB, isL ? 1 : 2);
      double4v acc = {0.0, 0.0, 0.0, 0.0};
      if (isL) {
        // X_ij = -( sum_{k = j+1}^{i-1} X_ik T_k + M_ii T_i ),  T_k = L_kj M_jj
        for (int k = j + 1; k <= i; ++k) {
          double4v T = {0.0, 0.0, 0.0, 0.0};
          T = block_times(D, k * PB, j * PB, 0, Mjj, T);
          acc = block_times(D, i * PB, k * PB, k == i ? 1 : 0, T, acc);
        }
      } else {
        // X_ij = -( M_ii T_i + sum_{k = i+1}^{j-1} X_ik T_k ),  T_k = U_kj M_jj
        for (int k = i; k < j; ++k) {
          double4v T = {0.0, 0.0, 0.0, 0.0};
          T = block_times(D, k * PB, j * PB, 0, Mjj, T);
          acc = block_times(D, i * PB, k * PB, k == i ? 2 : 0, T, acc);
        }
      }
      return acc;
    };
    const bool has0 = wave < 2 * nout, has1 = wave + 4 < 2 * nout;
    double4v X0 = {0.0, 0.0, 0.0, 0.0}, X1 = {0.0, 0.0, 0.0, 0.0};
    int i0 = 0, q0 = 0, i1 = 0, q1 = 0;
    if (has0) X0 = output(wave, i0, q0);
    if (has1) X1 = output(wave + 4, i1, q1);
    __syncthreads();  // every T of this stage has been formed from the original blocks of these two block columns
    if (has0)
#pragma unroll
      for (int r = 0; r < 4; ++r) D[i0 * PB + (lane >> 4) + 4 * r][q0 * PB + (lane & 15)] = -X0[r];
    if (has1)
#pragma unroll
      for (int r = 0; r < 4; ++r) D[i1 * PB + (lane >> 4) + 4 * r][q1 * PB + (lane & 15)] = -X1[r];
    __syncthreads();
  }
  for (int c = wave; c < NB; c += 4) {
    const double v = D[lane][c];
    invL[lane + c * NB] = lane > c ? v : (lane == c ? 1.0 : 0.0);
    invU[lane + c * NB] = lane <= c ? v : 0.0;
  }
}

// Band::piv: 0 no interchanges (the blocked form), 1 threshold pivoting inside the block, 2 no interchanges in the
// unblocked form of rounds 1 - 3 (ablation: SPL_LU_DIAG=plain)
__device__ __forceinline__ void diag_block_factor(const Band &b, int j0, int jb, double (*D)[LDP],
                                                  double (*lcol)[NB], int *__restrict__ singular,
                                                  double *__restrict__ invL, double *__restrict__ invU) {
  if (b.piv == 1) diag_block_factor_t<true>(b, j0, jb, D, lcol, singular, invL, invU);
  else if (b.piv == 2) diag_block_factor_t<false>(b, j0, jb, D, lcol, singular, invL, invU);
  else diag_block_factor_blocked(b, j0, jb, D, singular, invL, invU);
}

__global__ __launch_bounds__(256) void diag_lu_kernel(Band b, int j0, int jb, int *__restrict__ singular,
                                                      double *__restrict__ invL, double *__restrict__ invU) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double(*D)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
  double(*lcol)[NB] = reinterpret_cast<double(*)[NB]>(dsm + NB * LDP);
  const int tr = threadIdx.x & 63, tc = threadIdx.x >> 6;
  for (int c = tc; c < NB; c += 4)
    D[tr][c] = (tr < jb && c < jb) ? b.get(j0 + tr, j0 + c) : (tr == c ? 1.0 : 0.0);
  __syncthreads();
  diag_block_factor(b, j0, jb, D, lcol, singular, invL, invU);
}

// 64 x 64 x 64 product of two LDS tiles on the fp64 matrix cores:
//     acc[a][c][r] = sum_k Cs[k][qc + a*16 + lane/16 + 4r] * Rs[k][qr + c*16 + lane%16]
// Cs[k][.] is indexed by the output COLUMN, Rs[k][.] by the output ROW, so a lane's 16 neighbours
// hold 16 consecutive rows of one column: 128 contiguous bytes of band storage per access.  Each
// of the 4 wavefronts owns a 32 x 32 quadrant = 2 x 2 MFMA tiles.  v_mfma_f64_16x16x4_f64 layout
// probed on gfx950 (tools/probe/mfma_f64_probe.hip): lane l supplies A[l%16][l/16] and
// B[l/16][l%16] and holds C[(l/16) + 4*r][l%16] in element r.
struct TilePos {
  int qr, qc, li, lk;
  __device__ TilePos() {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    qr = (wave & 1) * 32, qc = (wave >> 1) * 32, li = lane & 15, lk = lane >> 4;
  }
  __device__ int row(int c) const { return qr + c * 16 + li; }
  __device__ int col(int a, int r) const { return qc + a * 16 + lk + 4 * r; }
};

template <int KD>
__device__ __forceinline__ void mfma_tile_64(const double (*Cs)[LDP], const double (*Rs)[LDP], const TilePos &p,
                                             double4v (&acc)[2][2]) {
#pragma unroll 4
  for (int k0 = 0; k0 < KD; k0 += 4) {
    double af[2], bf[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[a] = Cs[k0 + p.lk][p.qc + a * 16 + p.li];
#pragma unroll
    for (int c = 0; c < 2; ++c) bf[c] = Rs[k0 + p.lk][p.qr + c * 16 + p.li];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[c], acc[a][c], 0, 0, 0);
  }
}

__device__ __forceinline__ void zero_acc(double4v (&acc)[2][2]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = (double4v){0.0, 0.0, 0.0, 0.0};
}

// panel solves as GEMMs: tiles [0, ntile_l): L21 tile (64 rows) <- A21 tile * invU;
// tiles [ntile_l, ...): U12 tile (64 columns) <- invL * A12 tile.  In place.  Both operands are
// requested from memory at once (32 loads in flight per thread: the kernel is a latency chain
// between the diagonal block and the update) and go through LDS in two K-halves of 32, so that
// four workgroups fit a CU.
constexpr size_t kTrsmLds = (size_t)2 * 32 * (NB + 1) * sizeof(double);
__device__ __forceinline__ void trsm_tile(const Band &b, int j0, int jb, int nrows_below, int ncols_right,
                                          const double *__restrict__ invL, const double *__restrict__ invU,
                                          int tile, double *dsm) {
  constexpr int KH = 32;
  double(*Cs)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);            // [k][output column]
  double(*Rs)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + KH * LDP);  // [k][output row]
  const int tid = threadIdx.x;
  const int ntile_l = (nrows_below + 63) / 64;
  const bool is_l = tile < ntile_l;
  if (b.sym && !is_l) return;  // the pivot rows come from the L tiles below (U12 = D L21^T)
  const int rend = j0 + jb + nrows_below, cend = j0 + jb + ncols_right;
  const int lo = tid % 64, hi = tid / 64;  // t = tid + 256 u  ->  t % 64 = lo, t / 64 = hi + 4 u
  int r0, c0;
  double rv[16], cv[16];  // Rs[hi + 4 u][lo] and Cs[lo][hi + 4 u] of this thread
  if (is_l) {
    r0 = j0 + jb + tile * 64, c0 = j0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = hi + 4 * u;  // Rs[k][r] = A21(r0 + r, j0 + k)
      rv[u] = (k < jb && r0 + lo < rend) ? b.get(r0 + lo, j0 + k) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) cv[u] = invU[tid + 256 * u];  // Cs[k][c] = invU(k, c), k = lo, c = hi + 4 u
  } else {
    r0 = j0, c0 = j0 + jb + (tile - ntile_l) * 64;
#pragma unroll
    for (int u = 0; u < 16; ++u) rv[u] = invL[tid + 256 * u];  // Rs[k][r] = invL(r, k), r = lo, k = hi + 4 u
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = hi + 4 * u;  // Cs[k][c] = A12(j0 + k, c0 + c), k = lo
      cv[u] = (lo < jb && c0 + c < cend) ? b.get(j0 + lo, c0 + c) : 0.0;
    }
  }
  const TilePos p;
  double4v acc[2][2];
  zero_acc(acc);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) Rs[hi + 4 * u][lo] = rv[8 * h + u];  // k = hi + 4 (8 h + u) = 32 h + (hi + 4 u)
    if (lo / KH == h) {
#pragma unroll
      for (int u = 0; u < 16; ++u) Cs[lo - KH * h][hi + 4 * u] = cv[u];
    }
    __syncthreads();
    mfma_tile_64<KH>(Cs, Rs, p, acc);
  }
  const int rlim = is_l ? rend : j0 + jb, clim = is_l ? j0 + jb : cend;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = r0 + p.row(c), j = c0 + p.col(a, r);
        if (i < rlim && j < clim && b.in_band(i, j)) {
          b.at(i, j) = acc[a][c][r];
          if (b.sym) b.at(j, i) = b.at(j, j) * acc[a][c][r];  // U(j, i) = d_j L(i, j), d_j = U11(j, j) (diag_block_factor stored it)
        }
      }
}

__global__ __launch_bounds__(256) void trsm_gemm_kernel(Band b, int j0, int jb, int nrows_below, int ncols_right,
                                                        const double *__restrict__ invL,
                                                        const double *__restrict__ invU) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  trsm_tile(b, j0, jb, nrows_below, ncols_right, invL, invU, (int)blockIdx.x, dsm);
}

// Trailing update  A(rows, cols) -= L(rows, kb .. kb+klen) * U(kb .. kb+klen, cols)  on 64 x 64
// tiles of the region rows [rb, re) x columns [cb, ce), rb == cb on the diagonal.  K is staged
// through LDS in slices of KS = 32, software-pipelined through registers (the loads of the next
// slice are in flight during the MFMAs of this one; 2 workgroups per CU); klen is 64 for a
// single block step and 128 when two block steps share one pass over the window, which halves
// the read-modify-write traffic of the window.  Entries outside the band read as zero and are
// never written.
//   lshape = 0: 2-D grid over the whole region.
//   lshape = 1: 1-D grid over the first tile column and the first tile row only (the panels the
//               second block step of a pair needs before the shared pass).
// Look-ahead: the workgroup of tile (0,0) -- the next diagonal block -- goes on to factor and
// invert it while the other tiles are still being updated, which takes the diagonal-block chain
// off the critical path.
constexpr int KS = 32;
constexpr int kSplitTiles = 64;  // windows of at least this many tiles a side split off the look-ahead tile (factor_loop)
constexpr int kAheadTiles = 80;  // ... and beyond this many tiles a side past the next pair, the pair's chain runs beside the bulk pass
struct Region {
  int rb, re, cb, ce, kb, klen, lshape, ntile_rows;
  int npiv;  // pivots of the (partial) factorisation: the look-ahead only factors blocks below it
};

// one 64 x 64 tile (tx, ty) of the update; LOOKAHEAD: tile (0,0) goes on to factor the next block
// (a template parameter: the one-workgroup front kernel calls this without the look-ahead and must
// not carry the registers of the diagonal-block factorisation through its update tiles)
template <bool LOOKAHEAD>
__device__ __forceinline__ void update_tile(const Band &b, const Region &g, int tx, int ty,
                                            int *__restrict__ singular, double *__restrict__ next_invL,
                                            double *__restrict__ next_invU, double *dsm) {
  const int r0 = g.rb + tx * 64, c0 = g.cb + ty * 64;
  if (b.sym && c0 > r0) return;  // symmetric fronts: only the tiles on and below the diagonal (rb == cb)
  if (r0 - (c0 + 63) > b.kl || c0 - (r0 + 63) > b.ku) return;  // tile entirely outside the band
  double(*Us)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);            // Us[k][c] = U(kb + k, c0 + c)
  double(*Ls)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + KS * LDP);  // Ls[k][r] = L(r0 + r, kb + k)
  const int tid = threadIdx.x;
  // Interior tile: the tile of A22 and both operand panels lie wholly inside the update range, the
  // matrix and the band, and K is a whole number of slices.  Then nothing is predicated and every
  // address is a pointer plus a constant stride (the predicated form spends more VALU cycles on
  // addresses and masks than the matrix cores spend on the products).
  const int klast = g.kb + g.klen - 1;
  const bool interior = r0 + 64 <= g.re && c0 + 64 <= g.ce && g.klen % KS == 0 &&            // range
                        r0 + 63 - c0 <= b.kl && c0 + 63 - r0 <= b.ku &&                        // C in band
                        r0 + 63 - g.kb <= b.kl && klast - r0 <= b.ku &&                        // L panel in band
                        klast - c0 <= b.kl && c0 + 63 - g.kb <= b.ku;                          // U panel in band
  const size_t cs = (size_t)(b.ldab - 1);  // column stride
  // the tile of A22 is requested first, so that its HBM latency overlaps the staging and the MFMAs
  // (entries outside this step's update range get a zero product; tile (0,0) needs them below)
  const TilePos p;
  double cold[2][2][4];
  if (interior) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double *src = &b.at(r0 + p.row(c), c0 + p.col(a, 0));
        // (the window is read once and written once per pass: streaming hints keep it from pushing the panel tiles,
        // which every tile of a row / column of the window reads again, out of the L2 — the bulk passes of a 100^3
        // factorisation read 183 -> 157 GB, 137 -> 134.5 ms.  An XCD-aware tile order cut the panel re-reads by another
        // 75 % and was SLOWER, 139 ms: the pass is not bound by HBM traffic — profiles/r04_concurrency_experiments.txt)
#pragma unroll
        for (int r = 0; r < 4; ++r) cold[a][c][r] = __builtin_nontemporal_load(&src[(size_t)(4 * r) * cs]);  // col(a, r) = col(a, 0) + 4 r
      }
  } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = r0 + p.row(c), j = c0 + p.col(a, r);
          cold[a][c][r] = (i < b.n && j < b.n && b.in_band(i, j)) ? b.at(i, j) : 0.0;  // also beyond re/ce
        }
  }
  double4v acc[2][2];
  zero_acc(acc);
  // K slices are software-pipelined through registers: the global loads of slice s+1 are issued
  // before the MFMAs of slice s, so their latency hides behind the matrix cores inside the
  // workgroup (each thread carries 8 + 8 operands)
  constexpr int PER = KS * 64 / 256;
  double el[PER], eu[PER];
  // thread t stages L(r0 + t % 64, kb + k0 + t / 64 + 4 u) and U(kb + k0 + t % KS, c0 + t / KS + 8 u)
  const double *lsrc = interior ? &b.at(r0 + tid % 64, g.kb + tid / 64) : nullptr;
  const double *usrc = interior ? &b.at(g.kb + tid % KS, c0 + tid / KS) : nullptr;
  auto fetch = [&](int k0) {
    if (interior) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        el[u] = lsrc[(size_t)(k0 + 4 * u) * cs];
        eu[u] = usrc[(size_t)k0 + (size_t)(256 / KS * u) * cs];
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = tid + u * 256;
      const int r = t % 64, k = t / 64;
      el[u] = (k0 + k < g.klen && r0 + r < g.re) ? b.get(r0 + r, g.kb + k0 + k) : 0.0;
      const int k2 = t % KS, c = t / KS;
      eu[u] = (k0 + k2 < g.klen && c0 + c < g.ce) ? b.get(g.kb + k0 + k2, c0 + c) : 0.0;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < g.klen; k0 += KS) {
    if (k0) __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = tid + u * 256;
      Ls[t / 64][t % 64] = el[u];
      Us[t % KS][t / KS] = eu[u];
    }
    __syncthreads();
    if (k0 + KS < g.klen) fetch(k0 + KS);
    mfma_tile_64<KS>(Us, Ls, p, acc);
  }
  if (interior) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        double *dst = &b.at(r0 + p.row(c), c0 + p.col(a, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cold[a][c][r] -= acc[a][c][r];
          if (LOOKAHEAD) dst[(size_t)(4 * r) * cs] = cold[a][c][r];  // (the next panel solves read these strips at once)
          else __builtin_nontemporal_store(cold[a][c][r], &dst[(size_t)(4 * r) * cs]);
        }
      }
  } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = r0 + p.row(c), j = c0 + p.col(a, r);
          cold[a][c][r] -= acc[a][c][r];
          if (i < g.re && j < g.ce && b.in_band(i, j)) b.at(i, j) = cold[a][c][r];
        }
  }
  if (!LOOKAHEAD) return;
  if (tx != 0 || ty != 0 || r0 >= g.npiv) return;
  // next diagonal block: rows/columns r0 .. r0 + jbn - 1, values still in registers
  const int jbn = min(NB, g.npiv - r0);
  __syncthreads();  // all waves are done reading Us / Ls
  double(*D)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
  double(*lcol)[NB] = reinterpret_cast<double(*)[NB]>(dsm + NB * LDP);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tr = p.row(c), tcn = p.col(a, r);
        D[tr][tcn] = (tr < jbn && tcn < jbn) ? cold[a][c][r] : (tr == tcn ? 1.0 : 0.0);
      }
  __syncthreads();
  diag_block_factor(b, r0, jbn, D, lcol, singular, next_invL, next_invU);
}

__global__ __launch_bounds__(256) void gemm_update_kernel(Band b, Region g, int *__restrict__ singular,
                                                          double *__restrict__ next_invL,
                                                          double *__restrict__ next_invU) {
  int tx = blockIdx.x, ty = blockIdx.y;
  if (g.lshape) {
    tx = (int)blockIdx.x < g.ntile_rows ? (int)blockIdx.x : 0;
    ty = (int)blockIdx.x < g.ntile_rows ? 0 : (int)blockIdx.x - g.ntile_rows + 1;
  }
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  update_tile<true>(b, g, tx, ty, singular, next_invL, next_invU, dsm);
}

// The same pass without tile (0,0) and without the look-ahead code: 168 registers instead of 172,
// which is the step from two to three wavefronts per SIMD.  Large windows run this kernel on the
// main stream and tile (0,0) with its look-ahead as a one-workgroup launch of gemm_update_kernel on a
// helper stream beside it (factor_loop).
__global__ __launch_bounds__(256) void gemm_update_bulk_kernel(Band b, Region g, int *__restrict__ singular) {
  if (g.lshape == 0 && blockIdx.x == 0 && blockIdx.y == 0) return;  // (lshape 2: the whole window, no tile is somebody else's)
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  update_tile<false>(b, g, (int)blockIdx.x, (int)blockIdx.y, singular, nullptr, nullptr, dsm);
}

// The whole partial factorisation of one SMALL dense front by ONE workgroup (the multifrontal tree
// has thousands of them per level; a launch per block step and front would serialise them).  Same
// device code as the kernels above, tiles taken one after the other; the front lives in HBM / L2.
__device__ __forceinline__ void front_factor_by_workgroup(const Band &b, int npiv, double *invs,
                                                          int *__restrict__ singular, double *dsm) {
  double(*D)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
  double(*lcol)[NB] = reinterpret_cast<double(*)[NB]>(dsm + NB * LDP);
  const int n = b.n;
  for (int j0 = 0; j0 < npiv; j0 += NB) {
    const int jb = min(NB, npiv - j0);
    double *invL = invs + (size_t)(j0 / NB) * (2 * NB * NB), *invU = invL + NB * NB;
    const int tr = threadIdx.x & 63, tc = threadIdx.x >> 6;
    for (int c = tc; c < NB; c += 4)
      D[tr][c] = (tr < jb && c < jb) ? b.get(j0 + tr, j0 + c) : (tr == c ? 1.0 : 0.0);
    __syncthreads();
    diag_block_factor(b, j0, jb, D, lcol, singular, invL, invU);
    __syncthreads();  // factors and inverses of the block are visible to the whole workgroup
    const int rest = n - (j0 + jb);
    if (rest <= 0) break;
    const int ntile = (rest + 63) / 64;
    for (int tile = 0; tile < 2 * ntile; ++tile) {
      trsm_tile(b, j0, jb, rest, rest, invL, invU, tile, dsm);
      __syncthreads();
    }
    Region g{j0 + jb, n, j0 + jb, n, j0, jb, 0, ntile, npiv};
    for (int t = 0; t < ntile * ntile; ++t) {
      update_tile<false>(b, g, t % ntile, t / ntile, singular, nullptr, nullptr, dsm);
      __syncthreads();
    }
  }
}


// ---- complex fronts (two planes, Band::zoff): the same blocked algorithm in complex arithmetic ------------------
// A complex product on the fp64 matrix cores is four real ones: (Lr Ur - Li Ui) + i (Lr Ui + Li Ur).  The tile
// kernels below are the real ones with every operand staged twice (two planes) and four MFMAs where the real kernel
// issues one; indices, tiles, look-ahead and the symmetric mode (U = D L^T, no conjugation: complex SYMMETRIC
// matrices) are unchanged.  Inverse blocks: [inv(L11) re | im | inv(U11) re | im], NB x NB column-major each.
constexpr int kInvBlockZ = 4 * NB * NB;
constexpr size_t kDiagLdsZ = (size_t)(2 * NB * LDP + 4 * NB) * sizeof(double);  // D re, D im, multiplier columns
constexpr size_t kTrsmLdsZ = (size_t)4 * 32 * LDP * sizeof(double);
constexpr int KSZ = 16;  // K slice of the complex update (four LDS tiles per slice)
constexpr size_t kUpdateLdsZ = (size_t)4 * KSZ * LDP * sizeof(double);

__device__ __forceinline__ void crecip(double pr, double pi, double &qr, double &qi) {  // 1 / (pr + i pi), Smith
  if (fabs(pr) >= fabs(pi)) {
    const double r = pi / pr, d = pr + pi * r;
    qr = 1.0 / d;
    qi = -r / d;
  } else {
    const double r = pr / pi, d = pr * r + pi;
    qr = r / d;
    qi = -1.0 / d;
  }
}

// the same values without a branch (both quotients are formed and one is selected)
__device__ __forceinline__ void crecip_straight(double pr, double pi, double &qr, double &qi) {
  const bool wide = fabs(pr) >= fabs(pi);
  const double num = wide ? pi : pr, den = wide ? pr : pi;
  const double r = num / den, d = den + num * r, u = 1.0 / d, v = r / d;
  qr = wide ? u : v;
  qi = wide ? -v : -u;
}
// conj(p) / |p|^2: ONE division on the chain of a pivot instead of Smith's two in a row (the reciprocal of the pivot is
// the longest dependent stretch of a complex panel column: 68 -> 5x us per 64 x 64 block).  |p|^2 must neither overflow
// nor lose its bits: outside 1e-280 .. 1e280 (pivots beyond 1e+-140) Smith's form takes over — a uniform branch.
__device__ __forceinline__ void crecip_short(double pr, double pi, double &qr, double &qi) {
  const double m2 = pr * pr + pi * pi;
  if (m2 > 1e-280 && m2 < 1e280) {
    const double inv = 1.0 / m2;
    qr = pr * inv;
    qi = -pi * inv;
  } else {
    crecip_straight(pr, pi, qr, qi);
  }
}

// diag_block_factor in complex arithmetic: same register layout (lane = row, wave w owns the columns c = w mod 4),
// same pivot loop, two registers per entry
template <bool PIV>
__device__ __forceinline__ void diag_block_factor_zt(const Band &b, int j0, int jb, double (*Dr)[LDP], double (*Di)[LDP],
                                                     double (*lcr)[NB], double (*lci)[NB], int *__restrict__ singular,
                                                     double *__restrict__ invL, double *__restrict__ invU) {
  const int tid = threadIdx.x;
  const int tr = tid & 63, tc = tid >> 6;
  constexpr int NS = NB / 4;
  bool done = false;  // PIV: implicit row interchanges, as in diag_block_factor_t (moduli instead of absolute values)
  int mypos = tr;
  const double rs = (PIV && b.rscale && tr < jb) ? b.rscale[j0 + tr] : 1.0;
  {
    double ar[NS], ai[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      ar[u] = Dr[tr][4 * u + tc];
      ai[u] = Di[tr][4 * u + tc];
    }
#pragma unroll 1
    for (int g = 0; g < NS; ++g) {
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const int k = 4 * g + kw;
        if (tc == kw) {
          int prow = k;
          if (PIV) {
            double best = done ? -1.0 : (ar[0] * ar[0] + ai[0] * ai[0]) * rs * rs;
            int brow = tr;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
              const double ob = __shfl_xor(best, off, 64);
              const int orow = __shfl_xor(brow, off, 64);
              if (ob > best || (ob == best && orow < brow)) { best = ob; brow = orow; }
            }
            const unsigned long long unused = __ballot(!done);
            const int nat = ((unused >> k) & 1ull) ? k : (int)__ffsll((long long)unused) - 1;
            const double nr = readlane_f64(ar[0], nat), ni = readlane_f64(ai[0], nat), ns = readlane_f64(rs, nat);
            prow = ((nr * nr + ni * ni) * ns * ns >= kPivotThreshold * kPivotThreshold * best) ? nat : brow;
            prow = __builtin_amdgcn_readfirstlane(prow);
          }
          const double pr = readlane_f64(ar[0], prow), pi = readlane_f64(ai[0], prow);
          if (pr == 0.0 && pi == 0.0) {
            if (tr == 0) atomicOr(singular, 1);
          } else if (PIV ? (!done && tr != prow) : (tr > k)) {
            double qr, qi;
            crecip(pr, pi, qr, qi);
            const double t = ar[0] * qr - ai[0] * qi;
            ai[0] = ar[0] * qi + ai[0] * qr;
            ar[0] = t;
          }
          lcr[k & 1][tr] = ar[0];
          lci[k & 1][tr] = ai[0];
          if (PIV && tr == 0) reinterpret_cast<int &>(Dr[k][NB]) = prow;
        }
        __syncthreads();
        int prow = k;
        if (PIV) {
          prow = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int &>(Dr[k][NB]));
          if (tr == prow) { done = true; mypos = k; }
        }
        const bool below = PIV ? !done : (tr > k);
        const double lr = below ? lcr[k & 1][tr] : 0.0, li = below ? lci[k & 1][tr] : 0.0;
        if (tc > kw) {
          const double ur = readlane_f64(ar[0], prow), ui = readlane_f64(ai[0], prow);
          ar[0] -= lr * ur - li * ui;
          ai[0] -= lr * ui + li * ur;
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          if (g + 4 * qd < NS) {
#pragma unroll
            for (int j = (qd == 0 ? 1 : 4 * qd); j < 4 * qd + 4; ++j) {
              const double ur = readlane_f64(ar[j], prow), ui = readlane_f64(ai[j], prow);
              ar[j] -= lr * ur - li * ui;
              ai[j] -= lr * ui + li * ur;
            }
          }
        }
      }
      Dr[tr][4 * g + tc] = ar[0];
      Di[tr][4 * g + tc] = ai[0];
#pragma unroll
      for (int j = 0; j + 1 < NS; ++j) {
        ar[j] = ar[j + 1];
        ai[j] = ai[j + 1];
      }
      ar[NS - 1] = 0.0;
      ai[NS - 1] = 0.0;
    }
  }
  __syncthreads();
  if (PIV) {  // rows into pivot order
    double vr[NS], vi[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      vr[u] = Dr[tr][4 * u + tc];
      vi[u] = Di[tr][4 * u + tc];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      Dr[mypos][4 * u + tc] = vr[u];
      Di[mypos][4 * u + tc] = vi[u];
    }
    __syncthreads();
  }
  for (int c = tc; c < jb; c += 4)
    if (tr < jb) {
      double *dst = &b.at(j0 + tr, j0 + c);
      dst[0] = Dr[tr][c];
      dst[b.zoff] = Di[tr][c];
    }
  double *dinvr = &lcr[0][0], *dinvi = &lci[0][0];
  if (tid < NB) {
    const double dr = Dr[tid][tid], di = Di[tid][tid];
    double qr = 0.0, qi = 0.0;
    if (dr != 0.0 || di != 0.0) crecip(dr, di, qr, qi);
    dinvr[tid] = qr;
    dinvi[tid] = qi;
  }
  __syncthreads();
  double xr[NS], xi[NS], yr[NS], yi[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    xr[u] = yr[u] = (tr == 4 * u + tc) ? 1.0 : 0.0;
    xi[u] = yi[u] = 0.0;
  }
#pragma unroll 1
  for (int s = 0; s < NB; ++s) {
    {
      const int k = NB - 1 - s, ks = k >> 2;
      const double dkr = dinvr[k], dki = dinvi[k];
      const double ur = tr < k ? Dr[tr][k] : 0.0, ui = tr < k ? Di[tr][k] : 0.0;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (4 * qd + 3 >= ks) {
#pragma unroll
          for (int u = 4 * qd; u < 4 * qd + 4; ++u) {
            const double rr = readlane_f64(xr[u], k), ri = readlane_f64(xi[u], k);
            const double kr = rr * dkr - ri * dki, ki = rr * dki + ri * dkr;
            xr[u] = (tr == k) ? kr : xr[u] - (ur * kr - ui * ki);
            xi[u] = (tr == k) ? ki : xi[u] - (ur * ki + ui * kr);
          }
        }
      }
    }
    {
      const int k = s, ks = k >> 2;
      const double lr = tr > k ? Dr[tr][k] : 0.0, li = tr > k ? Di[tr][k] : 0.0;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (4 * qd <= ks) {
#pragma unroll
          for (int u = 4 * qd; u < 4 * qd + 4; ++u) {
            const double rr = readlane_f64(yr[u], k), ri = readlane_f64(yi[u], k);
            yr[u] -= lr * rr - li * ri;
            yi[u] -= lr * ri + li * rr;
          }
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int c = 4 * u + tc;
    invU[tr + c * NB] = xr[u];
    invU[NB * NB + tr + c * NB] = xi[u];
    const int cl = PIV ? reinterpret_cast<const int &>(Dr[c][NB]) : c;  // inv(L11) P_b
    invL[tr + cl * NB] = yr[u];
    invL[NB * NB + tr + cl * NB] = yi[u];
  }
}

// diag_block_factor_blocked in complex arithmetic (two tiles, two registers per entry, four real matrix-core products
// per complex block product): the same panels, stages and barriers
struct C4 {
  double4v r, i;
};
__device__ __forceinline__ C4 c4_zero() { return C4{{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}; }
// acc += Ablk * T (complex), Ablk = the 16 x 16 block at (r0, c0) of the tiles, masked as in block_times
__device__ __forceinline__ C4 block_times_z(const double (*Dr)[LDP], const double (*Di)[LDP], int r0, int c0, int mask, C4 T, C4 acc) {
  const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = 4 * r + kq;
    double ar = Dr[r0 + i][c0 + k], ai = Di[r0 + i][c0 + k];
    if (mask == 1) { ar = i > k ? ar : (i == k ? 1.0 : 0.0); ai = i > k ? ai : 0.0; }
    if (mask == 2) { ar = i <= k ? ar : 0.0; ai = i <= k ? ai : 0.0; }
    acc.r = mfma16(ar, T.r[r], acc.r);
    acc.r = mfma16(-ai, T.i[r], acc.r);
    acc.i = mfma16(ar, T.i[r], acc.i);
    acc.i = mfma16(ai, T.r[r], acc.i);
  }
  return acc;
}
__device__ __forceinline__ C4 block_as_b_z(const double (*Dr)[LDP], const double (*Di)[LDP], int r0, int c0, int mask) {
  const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
  C4 T;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = 4 * r + kq;
    double vr = Dr[r0 + k][c0 + j], vi = Di[r0 + k][c0 + j];
    if (mask == 1) { vr = k > j ? vr : (k == j ? 1.0 : 0.0); vi = k > j ? vi : 0.0; }
    if (mask == 2) { vr = k <= j ? vr : 0.0; vi = k <= j ? vi : 0.0; }
    T.r[r] = vr;
    T.i[r] = vi;
  }
  return T;
}

__device__ __forceinline__ void diag_block_factor_blocked_z(const Band &b, int j0, int jb, double (*Dr)[LDP], double (*Di)[LDP],
                                                            int *__restrict__ singular, double *__restrict__ invL,
                                                            double *__restrict__ invU) {
  constexpr int PB = 16, NP = NB / PB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll 1
  for (int p = 0; p < NP; ++p) {
    const int c0 = p * PB;
    if (wave == 0) {
      double ar[PB], ai[PB];
#pragma unroll
      for (int c = 0; c < PB; ++c) { ar[c] = Dr[lane][c0 + c]; ai[c] = Di[lane][c0 + c]; }
      bool zero_pivot = false;
#pragma unroll
      for (int kk = 0; kk < PB; ++kk) {
        const int k = c0 + kk;
        // straight-line code, as in the real form: a zero pivot is reported after the panel and leaves its column as it is
        const double pr = readlane_f64(ar[kk], k), pi = readlane_f64(ai[kk], k);
        const bool zero = pr == 0.0 && pi == 0.0;
        zero_pivot |= zero;
        double qr, qi;
        crecip_short(zero ? 1.0 : pr, pi, qr, qi);
        const bool below = lane > k && !zero;
        const double tr_ = ar[kk] * qr - ai[kk] * qi, ti_ = ar[kk] * qi + ai[kk] * qr;
        ar[kk] = below ? tr_ : ar[kk];
        ai[kk] = below ? ti_ : ai[kk];
        const double lr = lane > k ? ar[kk] : 0.0, li = lane > k ? ai[kk] : 0.0;
#pragma unroll
        for (int c = kk + 1; c < PB; ++c) {
          const double ur = readlane_f64(ar[c], k), ui = readlane_f64(ai[c], k);
          ar[c] -= lr * ur - li * ui;
          ai[c] -= lr * ui + li * ur;
        }
      }
      if (zero_pivot && lane == 0) atomicOr(singular, 1);
#pragma unroll
      for (int c = 0; c < PB; ++c) { Dr[lane][c0 + c] = ar[c]; Di[lane][c0 + c] = ai[c]; }
    }
    __syncthreads();
    if (p + 1 == NP) break;
    if (wave == 1) {
      const int col = c0 + PB + lane;
      if (col < NB) {
        double ur[PB], ui[PB];
#pragma unroll
        for (int r = 0; r < PB; ++r) { ur[r] = Dr[c0 + r][col]; ui[r] = Di[c0 + r][col]; }
#pragma unroll
        for (int k = 0; k + 1 < PB; ++k)
#pragma unroll
          for (int r = k + 1; r < PB; ++r) {
            const double lr = Dr[c0 + r][c0 + k], li = Di[c0 + r][c0 + k];
            ur[r] -= lr * ur[k] - li * ui[k];
            ui[r] -= lr * ui[k] + li * ur[k];
          }
#pragma unroll
        for (int r = 1; r < PB; ++r) { Dr[c0 + r][col] = ur[r]; Di[c0 + r][col] = ui[r]; }
      }
    }
    __syncthreads();
    const int nbk = NP - 1 - p;
    for (int t = wave; t < nbk * nbk; t += 4) {
      const int r0 = (p + 1 + t / nbk) * PB, q0 = (p + 1 + t % nbk) * PB;
      const C4 acc = block_times_z(Dr, Di, r0, c0, 0, block_as_b_z(Dr, Di, c0, q0, 0), c4_zero());
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Dr[r0 + (lane >> 4) + 4 * r][q0 + (lane & 15)] -= acc.r[r];
        Di[r0 + (lane >> 4) + 4 * r][q0 + (lane & 15)] -= acc.i[r];
      }
    }
    __syncthreads();
  }
  for (int c = wave; c < jb; c += 4)
    if (lane < jb && b.in_band(j0 + lane, j0 + c)) {
      double *dst = &b.at(j0 + lane, j0 + c);
      dst[0] = Dr[lane][c];
      dst[b.zoff] = Di[lane][c];
    }
  __syncthreads();
  {
    const int d0 = wave * PB, half = lane >> 4, r = lane & 15;
    double zr[PB], zi[PB], cr[PB], ci[PB];
#pragma unroll
    for (int c = 0; c < PB; ++c) {
      const int col = half == 0 ? c : PB - 1 - c;
      zr[c] = (col == r) ? 1.0 : 0.0;
      zi[c] = 0.0;
      cr[c] = half < 2 ? Dr[d0 + r][d0 + col] : 0.0;
      ci[c] = half < 2 ? Di[d0 + r][d0 + col] : 0.0;
    }
    double qr = 1.0, qi = 0.0;  // 1 / U(r, r) in the U rows
    if (half == 1) {
      const double dr = Dr[d0 + r][d0 + r], di = Di[d0 + r][d0 + r];
      if (dr != 0.0 || di != 0.0) crecip(dr, di, qr, qi);
      else qr = qi = 0.0;
    }
#pragma unroll
    for (int s = 0; s < PB; ++s) {
      const int ulane = 16 + PB - 1 - s;
      const double dkr = readlane_f64(qr, ulane), dki = readlane_f64(qi, ulane);
      const bool below = half == 0 && r > s, above = half == 1 && r < PB - 1 - s, isk = half == 1 && r == PB - 1 - s;
      const double fr = (below || above) ? cr[s] : 0.0, fi = (below || above) ? ci[s] : 0.0;
#pragma unroll
      for (int c = 0; c <= s; ++c) {
        const double lr = readlane_f64(zr[c], s), li = readlane_f64(zi[c], s);
        const double wr = readlane_f64(zr[c], ulane), wi = readlane_f64(zi[c], ulane);
        const double ur = wr * dkr - wi * dki, ui = wr * dki + wi * dkr;
        const double vr = half == 0 ? lr : ur, vi = half == 0 ? li : ui;
        zr[c] = isk ? ur : zr[c] - (fr * vr - fi * vi);
        zi[c] = isk ? ui : zi[c] - (fr * vi + fi * vr);
      }
    }
    if (half == 0) {
#pragma unroll
      for (int c = 0; c < PB; ++c)
        if (c < r) { Dr[d0 + r][d0 + c] = zr[c]; Di[d0 + r][d0 + c] = zi[c]; }
    } else if (half == 1) {
#pragma unroll
      for (int c = 0; c < PB; ++c)
        if (PB - 1 - c >= r) { Dr[d0 + r][d0 + PB - 1 - c] = zr[c]; Di[d0 + r][d0 + PB - 1 - c] = zi[c]; }
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int st = 0; st < NP - 1; ++st) {
    const int jl = NP - 2 - st, ju = 1 + st, nout = st + 1;
    auto output = [&](int t, int &i, int &j) -> C4 {
      const bool isL = t < nout;
      j = isL ? jl : ju;
      i = isL ? jl + 1 + t : t - nout;
      const C4 Mjj = block_as_b_z(Dr, Di, j * PB, j * PB, isL ? 1 : 2);
      C4 acc = c4_zero();
      if (isL) {
        for (int k = j + 1; k <= i; ++k) {
          const C4 T = block_times_z(Dr, Di, k * PB, j * PB, 0, Mjj, c4_zero());
          acc = block_times_z(Dr, Di, i * PB, k * PB, k == i ? 1 : 0, T, acc);
        }
      } else {
        for (int k = i; k < j; ++k) {
          const C4 T = block_times_z(Dr, Di, k * PB, j * PB, 0, Mjj, c4_zero());
          acc = block_times_z(Dr, Di, i * PB, k * PB, k == i ? 2 : 0, T, acc);
        }
      }
      return acc;
    };
    const bool has0 = wave < 2 * nout, has1 = wave + 4 < 2 * nout;
    C4 X0 = c4_zero(), X1 = c4_zero();
    int i0 = 0, q0 = 0, i1 = 0, q1 = 0;
    if (has0) X0 = output(wave, i0, q0);
    if (has1) X1 = output(wave + 4, i1, q1);
    __syncthreads();
    if (has0)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Dr[i0 * PB + (lane >> 4) + 4 * r][q0 * PB + (lane & 15)] = -X0.r[r];
        Di[i0 * PB + (lane >> 4) + 4 * r][q0 * PB + (lane & 15)] = -X0.i[r];
      }
    if (has1)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Dr[i1 * PB + (lane >> 4) + 4 * r][q1 * PB + (lane & 15)] = -X1.r[r];
        Di[i1 * PB + (lane >> 4) + 4 * r][q1 * PB + (lane & 15)] = -X1.i[r];
      }
    __syncthreads();
  }
  for (int c = wave; c < NB; c += 4) {
    const double vr = Dr[lane][c], vi = Di[lane][c];
    invL[lane + c * NB] = lane > c ? vr : (lane == c ? 1.0 : 0.0);
    invL[NB * NB + lane + c * NB] = lane > c ? vi : 0.0;
    invU[lane + c * NB] = lane <= c ? vr : 0.0;
    invU[NB * NB + lane + c * NB] = lane <= c ? vi : 0.0;
  }
}

__device__ __forceinline__ void diag_block_factor_z(const Band &b, int j0, int jb, double (*Dr)[LDP], double (*Di)[LDP],
                                                    double (*lcr)[NB], double (*lci)[NB], int *__restrict__ singular,
                                                    double *__restrict__ invL, double *__restrict__ invU) {
  if (b.piv == 1) diag_block_factor_zt<true>(b, j0, jb, Dr, Di, lcr, lci, singular, invL, invU);
  else if (b.piv == 2) diag_block_factor_zt<false>(b, j0, jb, Dr, Di, lcr, lci, singular, invL, invU);
  else diag_block_factor_blocked_z(b, j0, jb, Dr, Di, singular, invL, invU);
}

// the diagonal block (identity-padded beyond jb) into the two LDS tiles
__device__ __forceinline__ void load_diag_z(const Band &b, int j0, int jb, double (*Dr)[LDP], double (*Di)[LDP]) {
  const int tr = threadIdx.x & 63, tc = threadIdx.x >> 6;
  for (int c = tc; c < NB; c += 4) {
    const bool in = tr < jb && c < jb;
    const double *src = in ? &b.at(j0 + tr, j0 + c) : nullptr;
    Dr[tr][c] = in ? src[0] : (tr == c ? 1.0 : 0.0);
    Di[tr][c] = in ? src[b.zoff] : 0.0;
  }
}

__global__ __launch_bounds__(256) void diag_lu_kernel_z(Band b, int j0, int jb, int *__restrict__ singular,
                                                        double *__restrict__ invL, double *__restrict__ invU) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double(*Dr)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
  double(*Di)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + NB * LDP);
  double(*lcr)[NB] = reinterpret_cast<double(*)[NB]>(dsm + 2 * NB * LDP);
  double(*lci)[NB] = reinterpret_cast<double(*)[NB]>(dsm + 2 * NB * LDP + 2 * NB);
  load_diag_z(b, j0, jb, Dr, Di);
  __syncthreads();
  diag_block_factor_z(b, j0, jb, Dr, Di, lcr, lci, singular, invL, invU);
}

// acc += C * R for complex tiles in planes (Cs: indexed by the output column, Rs: by the output row)
template <int KD>
__device__ __forceinline__ void mfma_tile_64_z(const double (*Cr)[LDP], const double (*Ci)[LDP], const double (*Rr)[LDP],
                                               const double (*Ri)[LDP], const TilePos &p, double4v (&accr)[2][2],
                                               double4v (&acci)[2][2]) {
#pragma unroll 2
  for (int k0 = 0; k0 < KD; k0 += 4) {
    double afr[2], afi[2], bfr[2], bfi[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      afr[a] = Cr[k0 + p.lk][p.qc + a * 16 + p.li];
      afi[a] = Ci[k0 + p.lk][p.qc + a * 16 + p.li];
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bfr[c] = Rr[k0 + p.lk][p.qr + c * 16 + p.li];
      bfi[c] = Ri[k0 + p.lk][p.qr + c * 16 + p.li];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        accr[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[a], bfr[c], accr[a][c], 0, 0, 0);
        accr[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-afi[a], bfi[c], accr[a][c], 0, 0, 0);
        acci[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[a], bfi[c], acci[a][c], 0, 0, 0);
        acci[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(afi[a], bfr[c], acci[a][c], 0, 0, 0);
      }
  }
}

// panel solves of a complex dense front (trsm_tile): L21 tile <- A21 tile * inv(U11), U12 tile <- inv(L11) * A12 tile
__device__ __forceinline__ void trsm_tile_z(const Band &b, int j0, int jb, int nrows_below, int ncols_right,
                                            const double *__restrict__ invL, const double *__restrict__ invU, int tile,
                                            double *dsm) {
  constexpr int KH = 32;
  double(*Csr)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
  double(*Csi)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + KH * LDP);
  double(*Rsr)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + 2 * KH * LDP);
  double(*Rsi)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + 3 * KH * LDP);
  const int tid = threadIdx.x;
  const int ntile_l = (nrows_below + 63) / 64;
  const bool is_l = tile < ntile_l;
  if (b.sym && !is_l) return;
  const int rend = j0 + jb + nrows_below, cend = j0 + jb + ncols_right;
  const int lo = tid % 64, hi = tid / 64;
  const size_t z = b.zoff;
  int r0, c0;
  double rvr[16], rvi[16], cvr[16], cvi[16];
  if (is_l) {
    r0 = j0 + jb + tile * 64, c0 = j0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = hi + 4 * u;
      const bool in = k < jb && r0 + lo < rend;
      const double *src = &b.at(in ? r0 + lo : j0, in ? j0 + k : j0);
      rvr[u] = in ? src[0] : 0.0;
      rvi[u] = in ? src[z] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      cvr[u] = invU[tid + 256 * u];
      cvi[u] = invU[NB * NB + tid + 256 * u];
    }
  } else {
    r0 = j0, c0 = j0 + jb + (tile - ntile_l) * 64;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      rvr[u] = invL[tid + 256 * u];
      rvi[u] = invL[NB * NB + tid + 256 * u];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = hi + 4 * u;
      const bool in = lo < jb && c0 + c < cend;
      const double *src = &b.at(in ? j0 + lo : j0, in ? c0 + c : j0);
      cvr[u] = in ? src[0] : 0.0;
      cvi[u] = in ? src[z] : 0.0;
    }
  }
  const TilePos p;
  double4v accr[2][2], acci[2][2];
  zero_acc(accr);
  zero_acc(acci);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      Rsr[hi + 4 * u][lo] = rvr[8 * h + u];
      Rsi[hi + 4 * u][lo] = rvi[8 * h + u];
    }
    if (lo / KH == h) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        Csr[lo - KH * h][hi + 4 * u] = cvr[u];
        Csi[lo - KH * h][hi + 4 * u] = cvi[u];
      }
    }
    __syncthreads();
    mfma_tile_64_z<KH>(Csr, Csi, Rsr, Rsi, p, accr, acci);
  }
  const int rlim = is_l ? rend : j0 + jb, clim = is_l ? j0 + jb : cend;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = r0 + p.row(c), j = c0 + p.col(a, r);
        if (i < rlim && j < clim) {
          double *dst = &b.at(i, j);
          dst[0] = accr[a][c][r];
          dst[z] = acci[a][c][r];
          if (b.sym) {  // U(j, i) = d_j L(i, j), d_j = U11(j, j); no conjugation: complex symmetric
            const double *d = &b.at(j, j);
            double *ut = &b.at(j, i);
            ut[0] = d[0] * accr[a][c][r] - d[z] * acci[a][c][r];
            ut[z] = d[0] * acci[a][c][r] + d[z] * accr[a][c][r];
          }
        }
      }
}

__global__ __launch_bounds__(256) void trsm_gemm_kernel_z(Band b, int j0, int jb, int nrows_below, int ncols_right,
                                                          const double *__restrict__ invL,
                                                          const double *__restrict__ invU) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  trsm_tile_z(b, j0, jb, nrows_below, ncols_right, invL, invU, (int)blockIdx.x, dsm);
}

// trailing update of a complex dense front (update_tile): K slices of KSZ through four LDS tiles
template <bool LOOKAHEAD>
__device__ __forceinline__ void update_tile_z(const Band &b, const Region &g, int tx, int ty,
                                              int *__restrict__ singular, double *__restrict__ next_invL,
                                              double *__restrict__ next_invU, double *dsm) {
  const int r0 = g.rb + tx * 64, c0 = g.cb + ty * 64;
  if (b.sym && c0 > r0) return;
  double(*Usr)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
  double(*Usi)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + KSZ * LDP);
  double(*Lsr)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + 2 * KSZ * LDP);
  double(*Lsi)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + 3 * KSZ * LDP);
  const int tid = threadIdx.x;
  const size_t z = b.zoff;
  const bool interior = r0 + 64 <= g.re && c0 + 64 <= g.ce && g.klen % KSZ == 0;
  const size_t cs = (size_t)(b.ldab - 1);
  const TilePos p;
  double coldr[2][2][4], coldi[2][2][4];
  if (interior) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double *src = &b.at(r0 + p.row(c), c0 + p.col(a, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          coldr[a][c][r] = src[(size_t)(4 * r) * cs];
          coldi[a][c][r] = src[(size_t)(4 * r) * cs + z];
        }
      }
  } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = r0 + p.row(c), j = c0 + p.col(a, r);
          const bool in = i < b.n && j < b.n;
          const double *src = &b.at(in ? i : 0, in ? j : 0);
          coldr[a][c][r] = in ? src[0] : 0.0;
          coldi[a][c][r] = in ? src[z] : 0.0;
        }
  }
  double4v accr[2][2], acci[2][2];
  zero_acc(accr);
  zero_acc(acci);
  constexpr int PER = KSZ * 64 / 256;
  double elr[PER], eli[PER], eur[PER], eui[PER];
  const double *lsrc = interior ? &b.at(r0 + tid % 64, g.kb + tid / 64) : nullptr;
  const double *usrc = interior ? &b.at(g.kb + tid % KSZ, c0 + tid / KSZ) : nullptr;
  auto fetch = [&](int k0) {
    if (interior) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const double *lp = lsrc + (size_t)(k0 + 4 * u) * cs;
        const double *up = usrc + (size_t)k0 + (size_t)(256 / KSZ * u) * cs;
        elr[u] = lp[0];
        eli[u] = lp[z];
        eur[u] = up[0];
        eui[u] = up[z];
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = tid + u * 256;
      const int r = t % 64, k = t / 64;
      const bool lin = k0 + k < g.klen && r0 + r < g.re;
      const double *lp = &b.at(lin ? r0 + r : 0, lin ? g.kb + k0 + k : 0);
      elr[u] = lin ? lp[0] : 0.0;
      eli[u] = lin ? lp[z] : 0.0;
      const int k2 = t % KSZ, c = t / KSZ;
      const bool uin = k0 + k2 < g.klen && c0 + c < g.ce;
      const double *up = &b.at(uin ? g.kb + k0 + k2 : 0, uin ? c0 + c : 0);
      eur[u] = uin ? up[0] : 0.0;
      eui[u] = uin ? up[z] : 0.0;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < g.klen; k0 += KSZ) {
    if (k0) __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = tid + u * 256;
      Lsr[t / 64][t % 64] = elr[u];
      Lsi[t / 64][t % 64] = eli[u];
      Usr[t % KSZ][t / KSZ] = eur[u];
      Usi[t % KSZ][t / KSZ] = eui[u];
    }
    __syncthreads();
    if (k0 + KSZ < g.klen) fetch(k0 + KSZ);
    mfma_tile_64_z<KSZ>(Usr, Usi, Lsr, Lsi, p, accr, acci);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = r0 + p.row(c), j = c0 + p.col(a, r);
        coldr[a][c][r] -= accr[a][c][r];
        coldi[a][c][r] -= acci[a][c][r];
        if (interior || (i < g.re && j < g.ce)) {
          double *dst = &b.at(i, j);
          dst[0] = coldr[a][c][r];
          dst[z] = coldi[a][c][r];
        }
      }
  if (!LOOKAHEAD) return;
  if (tx != 0 || ty != 0 || r0 >= g.npiv) return;
  const int jbn = min(NB, g.npiv - r0);
  __syncthreads();
  double(*Dr)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
  double(*Di)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + NB * LDP);
  double(*lcr)[NB] = reinterpret_cast<double(*)[NB]>(dsm + 2 * NB * LDP);
  double(*lci)[NB] = reinterpret_cast<double(*)[NB]>(dsm + 2 * NB * LDP + 2 * NB);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tr = p.row(c), tcn = p.col(a, r);
        const bool in = tr < jbn && tcn < jbn;
        Dr[tr][tcn] = in ? coldr[a][c][r] : (tr == tcn ? 1.0 : 0.0);
        Di[tr][tcn] = in ? coldi[a][c][r] : 0.0;
      }
  __syncthreads();
  diag_block_factor_z(b, r0, jbn, Dr, Di, lcr, lci, singular, next_invL, next_invU);
}

__global__ __launch_bounds__(256) void gemm_update_kernel_z(Band b, Region g, int *__restrict__ singular,
                                                            double *__restrict__ next_invL,
                                                            double *__restrict__ next_invU) {
  int tx = blockIdx.x, ty = blockIdx.y;
  if (g.lshape) {
    tx = (int)blockIdx.x < g.ntile_rows ? (int)blockIdx.x : 0;
    ty = (int)blockIdx.x < g.ntile_rows ? 0 : (int)blockIdx.x - g.ntile_rows + 1;
  }
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  update_tile_z<true>(b, g, tx, ty, singular, next_invL, next_invU, dsm);
}

__global__ __launch_bounds__(256) void gemm_update_bulk_kernel_z(Band b, Region g, int *__restrict__ singular) {
  if (g.lshape == 0 && blockIdx.x == 0 && blockIdx.y == 0) return;
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  update_tile_z<false>(b, g, (int)blockIdx.x, (int)blockIdx.y, singular, nullptr, nullptr, dsm);
}

// the whole partial factorisation of one small complex front by one workgroup
__device__ __forceinline__ void front_factor_by_workgroup_z(const Band &b, int npiv, double *invs,
                                                            int *__restrict__ singular, double *dsm) {
  double(*Dr)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
  double(*Di)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + NB * LDP);
  double(*lcr)[NB] = reinterpret_cast<double(*)[NB]>(dsm + 2 * NB * LDP);
  double(*lci)[NB] = reinterpret_cast<double(*)[NB]>(dsm + 2 * NB * LDP + 2 * NB);
  const int n = b.n;
  for (int j0 = 0; j0 < npiv; j0 += NB) {
    const int jb = min(NB, npiv - j0);
    double *invL = invs + (size_t)(j0 / NB) * kInvBlockZ, *invU = invL + 2 * NB * NB;
    load_diag_z(b, j0, jb, Dr, Di);
    __syncthreads();
    diag_block_factor_z(b, j0, jb, Dr, Di, lcr, lci, singular, invL, invU);
    __syncthreads();
    const int rest = n - (j0 + jb);
    if (rest <= 0) break;
    const int ntile = (rest + 63) / 64;
    for (int tile = 0; tile < 2 * ntile; ++tile) {
      trsm_tile_z(b, j0, jb, rest, rest, invL, invU, tile, dsm);
      __syncthreads();
    }
    Region g{j0 + jb, n, j0 + jb, n, j0, jb, 0, ntile, npiv};
    for (int t = 0; t < ntile * ntile; ++t) {
      update_tile_z<false>(b, g, t % ntile, t / ntile, singular, nullptr, nullptr, dsm);
      __syncthreads();
    }
  }
}

// ---- blocked solves -----------------------------------------------------------------------------
// The factorisation keeps inv(L11) and inv(U11) of every diagonal block, so a block of the
// triangular solve is a 64 x 64 matrix-vector product instead of a 64-step substitution chain.
// One launch eliminates a super block of SB = 4 blocks (256 unknowns).  `in` holds the right-hand
// side entries not yet eliminated, `out` receives the solved super block; they are different
// arrays, so every workgroup redoes the small in-super-block solve from the same read-only inputs
// and then updates its own 64 rows of `in` outside the super block.
//   MODE 0: L z = c (unit lower, forward)     1: U x = z (backward)
//   MODE 2: U^T z = c (lower, forward)        3: L^T x = z (unit upper, backward)
constexpr int SB = 4;

// Workgroups of SW = 16 wavefronts: these kernels are chains of short latency-bound phases, and the
// wider workgroup shortens every phase (more loads in flight per row).
constexpr int SW = 16;
// ... 8 where 16 right-hand-side columns travel together (the complex fronts' groups of eight): the partial sums of the
// wavefronts, SW x 64 x NR doubles of LDS, would not fit beside the rest otherwise
template <int NR>
constexpr int solve_waves() { return NR >= 16 ? 8 : SW; }

// Sums over the 64 lanes of N values per lane (N a power of two) by recursive halving: at every step a lane keeps one
// half of its values and sends the other to its partner, so N values cost about N shuffles — a butterfly per value costs
// 6 N (round 3: the transposed solves with 16 columns spent most of their instructions there).  Afterwards lane L holds
// in val[0 .. N / 64) the sums of indices L * N / 64 + k (N >= 64), or in val[0] the sum of index L / (64 / N) (N < 64:
// the 64 / N lanes of a group all hold it).
template <int C, int M, int N>
__device__ __forceinline__ void wave_reduce_step(double (&val)[N], int lane) {
  if constexpr (M >= 1) {
    if constexpr (C > 1) {
      constexpr int half = C / 2;
      const bool up = (lane & M) != 0;
#pragma unroll
      for (int j = 0; j < half; ++j) {
        const double keep = up ? val[j + half] : val[j];
        const double send = up ? val[j] : val[j + half];
        val[j] = keep + __shfl_xor(send, M, 64);
      }
      wave_reduce_step<half, M / 2, N>(val, lane);
    } else {
      val[0] += __shfl_xor(val[0], M, 64);
      wave_reduce_step<1, M / 2, N>(val, lane);
    }
  }
}
template <int N>
__device__ __forceinline__ void wave_reduce_scatter(double (&val)[N]) {
  wave_reduce_step<N, 32, N>(val, (int)(threadIdx.x & 63));
}
// the index of val[k] after wave_reduce_scatter<N>, and whether this lane is the one that should store it
template <int N>
__device__ __forceinline__ int wave_reduce_index(int lane, int k) { return N >= 64 ? lane * (N / 64) + k : lane / (64 / (N < 64 ? N : 64)); }
template <int N>
__device__ __forceinline__ bool wave_reduce_owner(int lane) { return N >= 64 ? true : lane % (64 / (N < 64 ? N : 64)) == 0; }

// res[l][r] = sum_{t < nc} M(rb + l, cb + t) * vv[t][r], l < 64, r < NR right-hand sides; M =
// matrix of the triangular system.  Untransposed (MODE 0/1) band storage runs down the rows:
// lane = row, wave q takes t = q mod SWV, partial sums meet in LDS.  Transposed (MODE 2/3) it runs
// along t: lanes along t, wave q takes rows l = q mod SWV, butterfly reduction.  A band entry is
// loaded once for all NR right-hand sides.  Ends with the result visible to the whole workgroup.
// acc[:] += e * v[:] over NR columns; Z: the columns are (re, im) pairs of NR / 2 complex right-hand sides, e = er + i ei
template <int NR, bool Z, bool FUSED = true>
__device__ __forceinline__ void mac_cols(double (&acc)[NR], double er, double ei, const double *v) {
  // FUSED: fused multiply-adds, spelled out — the library is built with -ffp-contract=off (the SpMV keeps the
  // reference's two roundings), and the solves with many columns are bound by the issue of these instructions where v
  // comes from LDS: half as many this way (the step of a large front with 16 columns: 83 -> 59 us).  Not FUSED: the
  // plain expressions, for big_gemv_kernel, whose v arrive through the scalar cache — there the compiler's schedule of
  // the fused form waits for the scalar loads almost twice as often (measured: 192 -> 334 us per launch).
  if (!FUSED) {
    if (!Z) {
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r] += er * v[r];
    } else {
#pragma unroll
      for (int q = 0; q < NR / 2; ++q) {
        acc[2 * q] += er * v[2 * q] - ei * v[2 * q + 1];
        acc[2 * q + 1] += er * v[2 * q + 1] + ei * v[2 * q];
      }
    }
  } else if (!Z) {
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = __builtin_fma(er, v[r], acc[r]);
  } else {
    // (the two terms of an accumulator NR instructions apart: back to back the second waits for the first)
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = __builtin_fma(er, v[r], acc[r]);
#pragma unroll
    for (int q = 0; q < NR / 2; ++q) {
      acc[2 * q] = __builtin_fma(-ei, v[2 * q + 1], acc[2 * q]);
      acc[2 * q + 1] = __builtin_fma(ei, v[2 * q], acc[2 * q + 1]);
    }
  }
}

// the tile updates of the transposed modes with the rows on the lanes too, where the lane sums cost the most (16
// columns: a step of the transposed pass 81 -> 74 us; strided loads, each line shared by the eight wavefronts)
template <int NR>
constexpr bool tile_rows_on_lanes() { return NR >= 16; }

// (Z: a complex dense matrix in two planes, Band::zoff; the transposed modes are then CONJUGATE transposes)
template <int MODE, int NR, bool Z = false>
__device__ __forceinline__ void gemv64(const Band &b, int rb, int cb, int nc, const double (*vv)[NR],
                                       double (*res)[NR], double *part) {
  constexpr int SWV = solve_waves<NR>();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (MODE <= 1 || tile_rows_on_lanes<NR>()) {
    constexpr bool TR = MODE >= 2;  // M(i, c) = conj(F(c, i))
    const int i = rb + lane;
    double acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.0;
    if (i >= 0 && i < b.n) {
      int t = wave;
      for (; t + 7 * SWV < nc; t += 8 * SWV) {  // 8 independent loads in flight per lane
        double e[8], ei[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = cb + t + SWV * u;
          e[u] = TR ? b.get_nt(c, i) : b.get_nt(i, c);
          ei[u] = Z ? (TR ? -__builtin_nontemporal_load(&b.at(c, i) + b.zoff) : __builtin_nontemporal_load(&b.at(i, c) + b.zoff)) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) mac_cols<NR, Z>(acc, e[u], ei[u], &vv[t + SWV * u][0]);
      }
      for (; t < nc; t += SWV) {
        const int c = cb + t;
        const double e = TR ? b.get_nt(c, i) : b.get_nt(i, c);
        const double ei = Z ? (TR ? -__builtin_nontemporal_load(&b.at(c, i) + b.zoff) : __builtin_nontemporal_load(&b.at(i, c) + b.zoff)) : 0.0;
        mac_cols<NR, Z>(acc, e, ei, &vv[t][0]);
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) part[(wave * NR + r) * 64 + lane] = acc[r];  // (lanes side by side: no bank conflicts)
    __syncthreads();
    for (int o = threadIdx.x; o < 64 * NR; o += SWV * 64) {
      const int r = o / 64, l = o % 64;
      double tot = 0.0;
#pragma unroll
      for (int q = 0; q < SWV; ++q) tot += part[(q * NR + r) * 64 + l];
      res[l][r] = tot;
    }
  } else {
    // nc <= SB * NB = 256: at most 4 strips of 64 columns.  A wavefront has 64 / SWV rows; RG of them at a time, all
    // strips unrolled so that the 4 RG loads of a lane are in flight together (16 columns: two rows at a time — eight
    // rows of accumulators are 256 registers, and the compiler spilled 600 - 800 bytes per lane: 450 us per step)
    constexpr int RPW = 64 / SWV;
    constexpr int RG = NR >= 16 ? 2 : RPW;
#pragma unroll 1
    for (int q0 = 0; q0 < RPW; q0 += RG) {
      double acc[RG * NR];
#pragma unroll
      for (int o = 0; o < RG * NR; ++o) acc[o] = 0.0;
      double e[SB][RG], ei[SB][RG];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int t = lane + 64 * u;
#pragma unroll
        for (int q = 0; q < RG; ++q) {
          const int i = rb + wave + SWV * (q0 + q);
          const bool in = t < nc && i >= 0 && i < b.n;
          e[u][q] = in ? b.get_nt(cb + t, i) : 0.0;
          ei[u][q] = (Z && in) ? -__builtin_nontemporal_load(&b.at(cb + t, i) + b.zoff) : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int t = lane + 64 * u;
        if (t < nc) {
#pragma unroll
          for (int q = 0; q < RG; ++q) mac_cols<NR, Z>(*reinterpret_cast<double(*)[NR]>(&acc[q * NR]), e[u][q], ei[u][q], &vv[t][0]);
        }
      }
      wave_reduce_scatter<RG * NR>(acc);
      if (wave_reduce_owner<RG * NR>(lane)) {
#pragma unroll
        for (int k = 0; k < (RG * NR >= 64 ? RG * NR / 64 : 1); ++k) {
          const int idx = wave_reduce_index<RG * NR>(lane, k);
          res[wave + SWV * (q0 + idx / NR)][idx % NR] = acc[k];
        }
      }
    }
  }
  __syncthreads();
}

// ---- the solve inside a super block -----------------------------------------------------------------------------------
// A super block is SB sub-blocks of 64 unknowns; sub-block k needs (1) its couplings with the k sub-blocks solved
// before it, (2) the stored inverse of its diagonal block, (3) its right-hand side.  Round 3 took two of the three
// dependent rounds of global loads per sub-block out of the chain of barriers: the right-hand side of the whole super
// block is read once, into v, before the first sub-block (a sub-block replaces its 64 entries of v by the solution), and
// the entries of the inverse a thread multiplies with are fetched into registers one sub-block AHEAD (ie / iei), so they
// travel while the previous sub-block goes through its barriers.  The couplings are loaded where they are used, eight
// (or two strips) in flight per thread: a sub-block ahead they would take up to 3 x 64 / wavefronts (complex: twice
// that) more registers per thread, which the kernels with 16 columns do not have (the compiler spills them: measured,
// 83 -> 148 us per step).  Partial sums of the wavefronts meet in LDS as part[wave][r][lane] (lanes side by side: no
// bank conflicts), and the reductions write w / v directly.
// Untransposed (MODE 0/1): lane = row, the wavefronts split the columns.  Transposed (MODE 2/3): lanes along the
// columns, the wavefronts split the rows, butterfly reduction.
// Transposed systems inside a super block with 8 or 16 columns: the rows on the lanes as in the untransposed ones (the
// matrix entries then come by strided loads — the few blocks of a super block sit in the L2 —, but the sums need no
// shuffles: with the lanes along the columns the in-super-block solve of 16 complex columns spent ~100 us, three times
// the untransposed one, mostly in lane reductions).  One or two columns keep the lanes along the columns.
template <int NR>
constexpr bool rows_on_lanes() { return NR >= 8; }

template <int MODE, int NR, bool Z>
__device__ __forceinline__ void inverse_load(const double *__restrict__ inv, double *ie, double *iei) {
  constexpr int SWV = solve_waves<NR>(), NE = NB / SWV;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    // MODE <= 1: T(lane, t), t = wave + SWV u;  transposed: T(l, lane) = inv(lane, l), l likewise — or, with the rows on
    // the lanes there too (rows_on_lanes), T(lane, t) = inv(t, lane): strided loads of a block that sits in the L2
    const int t = wave + SWV * u;
    const int idx = (MODE >= 2 && rows_on_lanes<NR>()) ? t + lane * NB : lane + t * NB;
    ie[u] = inv[idx];
    if (Z) iei[u] = MODE <= 1 ? inv[NB * NB + idx] : -inv[NB * NB + idx];
  }
}

// w[l][:] = raw[l][:] - sum_{t < nc} M(rb + l, cb + t) vv[t][:], l < jb (0 beyond)
template <int MODE, int NR, bool Z>
__device__ __forceinline__ void coupling_apply(const Band &b, int rb, int cb, int nc, const double (*vv)[NR],
                                               const double (*raw)[NR], int jb, double (*w)[NR], double *part) {
  constexpr int SWV = solve_waves<NR>(), NE = NB / SWV;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (nc <= 0) {  // (uniform) the first sub-block of a pass
    for (int o = threadIdx.x; o < NB * NR; o += SWV * 64) {
      const int l = o / NR, r = o % NR;
      w[l][r] = l < jb ? raw[l][r] : 0.0;
    }
  } else if (MODE <= 1 || rows_on_lanes<NR>()) {
    constexpr bool TR = MODE >= 2;  // M(i, c) = conj(F(c, i))
    const int i = rb + lane;
    double acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.0;
    if (i >= 0 && i < b.n) {
      int t = wave;
      for (; t + 7 * SWV < nc; t += 8 * SWV) {  // 8 independent loads in flight per lane
        double e[8], ei[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int c = cb + t + SWV * u;
          e[u] = TR ? b.get(c, i) : b.get(i, c);
          ei[u] = Z ? (TR ? -(&b.at(c, i))[b.zoff] : (&b.at(i, c))[b.zoff]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) mac_cols<NR, Z>(acc, e[u], ei[u], &vv[t + SWV * u][0]);
      }
      for (; t < nc; t += SWV) {
        const int c = cb + t;
        const double e = TR ? b.get(c, i) : b.get(i, c);
        const double ei = Z ? (TR ? -(&b.at(c, i))[b.zoff] : (&b.at(i, c))[b.zoff]) : 0.0;
        mac_cols<NR, Z>(acc, e, ei, &vv[t][0]);
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) part[(wave * NR + r) * 64 + lane] = acc[r];
    __syncthreads();
    for (int o = threadIdx.x; o < 64 * NR; o += SWV * 64) {
      const int r = o / 64, l = o % 64;
      double tot = 0.0;
#pragma unroll
      for (int q = 0; q < SWV; ++q) tot += part[(q * NR + r) * 64 + l];
      w[l][r] = l < jb ? raw[l][r] - tot : 0.0;
    }
  } else {
    // nc <= (SB - 1) * 64: at most 3 strips of 64 columns, all in flight together, RG rows of the wavefront at a time
    constexpr int RG = NR >= 16 ? 2 : NE;
#pragma unroll 1
    for (int q0 = 0; q0 < NE; q0 += RG) {
      double acc[RG * NR];
#pragma unroll
      for (int o = 0; o < RG * NR; ++o) acc[o] = 0.0;
      double e[SB - 1][RG], ei[SB - 1][RG];
#pragma unroll
      for (int u = 0; u < SB - 1; ++u) {
        const int t = lane + 64 * u;
#pragma unroll
        for (int q = 0; q < RG; ++q) {
          const int i = rb + wave + SWV * (q0 + q);
          const bool in = t < nc && i >= 0 && i < b.n;
          e[u][q] = in ? b.get(cb + t, i) : 0.0;
          ei[u][q] = (Z && in) ? -(&b.at(cb + t, i))[b.zoff] : 0.0;  // conjugate transpose
        }
      }
#pragma unroll
      for (int u = 0; u < SB - 1; ++u) {
        const int t = lane + 64 * u;
        if (t < nc) {
#pragma unroll
          for (int q = 0; q < RG; ++q) mac_cols<NR, Z>(*reinterpret_cast<double(*)[NR]>(&acc[q * NR]), e[u][q], ei[u][q], &vv[t][0]);
        }
      }
      wave_reduce_scatter<RG * NR>(acc);
      if (wave_reduce_owner<RG * NR>(lane)) {
#pragma unroll
        for (int k = 0; k < (RG * NR >= 64 ? RG * NR / 64 : 1); ++k) {
          const int idx = wave_reduce_index<RG * NR>(lane, k);
          const int l = wave + SWV * (q0 + idx / NR), r = idx % NR;
          w[l][r] = l < jb ? raw[l][r] - acc[k] : 0.0;
        }
      }
    }
  }
  __syncthreads();
}

// The couplings a sub-block ahead as well, where the registers allow (couplings_ahead): K strips of 64 columns, NE
// entries per strip and thread, loaded without a branch (an entry outside the system or the band is read from the first
// word of the array and replaced by zero: a branch per entry makes the compiler wait for every load where it is issued,
// and a branch around the loads makes it sink the multiplications of the step below them).
template <int NR, bool Z>
constexpr bool couplings_ahead() { return NR <= 2; }

template <int MODE, int NR, bool Z, int K>
__device__ __forceinline__ void coupling_load(const Band &b, int rb, int cb, int nc, double *ce, double *cei) {
  constexpr int SWV = solve_waves<NR>(), NE = NB / SWV;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (MODE <= 1) {
    const int i = rb + lane;
    const bool row = i >= 0 && i < b.n;
#pragma unroll
    for (int j = 0; j < K * NE; ++j) {
      const int t = wave + SWV * j;
      const bool in = row && t < nc && b.in_band(i, cb + t);
      const double *p = in ? &b.at(i, cb + t) : b.AB;
      const double re = *p;
      ce[j] = in ? re : 0.0;
      if (Z) {
        const double im = p[b.zoff];
        cei[j] = in ? im : 0.0;
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const int t = lane + 64 * u;
#pragma unroll
      for (int q = 0; q < NE; ++q) {
        const int i = rb + wave + SWV * q;
        const bool in = t < nc && i >= 0 && i < b.n && b.in_band(cb + t, i);
        const double *p = in ? &b.at(cb + t, i) : b.AB;
        const double re = *p;
        ce[u * NE + q] = in ? re : 0.0;
        if (Z) {
          const double im = p[b.zoff];
          cei[u * NE + q] = in ? -im : 0.0;  // conjugate transpose
        }
      }
    }
  }
}

// coupling_apply with the entries in registers; `prefetch` runs between this thread's multiplications and the barrier
template <int MODE, int NR, bool Z, int K, class Prefetch>
__device__ __forceinline__ void coupling_apply_regs(const double *ce, const double *cei, int nc, const double (*vv)[NR],
                                                    const double (*raw)[NR], int jb, double (*w)[NR], double *part,
                                                    Prefetch prefetch) {
  constexpr int SWV = solve_waves<NR>(), NE = NB / SWV;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (K == 0) {
    prefetch();
    for (int o = threadIdx.x; o < NB * NR; o += SWV * 64) {
      const int l = o / NR, r = o % NR;
      w[l][r] = l < jb ? raw[l][r] : 0.0;
    }
  } else if (MODE <= 1) {
    double acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.0;
#pragma unroll
    for (int j = 0; j < K * NE; ++j) {
      const int t = min(wave + SWV * j, SB * NB - 1);  // (entries beyond nc are zero; v there is finite)
      mac_cols<NR, Z>(acc, ce[j], Z ? cei[j] : 0.0, &vv[t][0]);
    }
    prefetch();
#pragma unroll
    for (int r = 0; r < NR; ++r) part[(wave * NR + r) * 64 + lane] = acc[r];
    __syncthreads();
    for (int o = threadIdx.x; o < 64 * NR; o += SWV * 64) {
      const int r = o / 64, l = o % 64;
      double tot = 0.0;
#pragma unroll
      for (int q = 0; q < SWV; ++q) tot += part[(q * NR + r) * 64 + l];
      w[l][r] = l < jb ? raw[l][r] - tot : 0.0;
    }
  } else {
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      double acc[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r] = 0.0;
#pragma unroll
      for (int u = 0; u < K; ++u) mac_cols<NR, Z>(acc, ce[u * NE + q], Z ? cei[u * NE + q] : 0.0, &vv[lane + 64 * u][0]);
      const int l = wave + SWV * q;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        double sacc = acc[r];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sacc += __shfl_xor(sacc, m, 64);
        if (lane == 0) w[l][r] = l < jb ? raw[l][r] - sacc : 0.0;
      }
    }
    prefetch();
  }
  __syncthreads();
}

// sub-block K of the super block [j0, j0 + jbs) and, recursively, the ones after it (K is a compile-time constant so that
// the prefetched entries stay in registers)
template <int MODE, int NR, bool Z, int K>
__device__ __forceinline__ void super_sub_block(const Band &b, const double *__restrict__ invs, int j0, int jbs, int nsub,
                                                double (*v)[NR], double (*w)[NR], double *part, double *ce, double *cei,
                                                double *ie, double *iei);

// dst[l][:] = sum_t T(l, t) w[t][:], l < jb, with T = inv(L11), inv(U11), inv(U11)^T, inv(L11)^T (MODE 0..3) in ie / iei
template <int MODE, int NR, bool Z, class Prefetch>
__device__ __forceinline__ void inverse_apply(const double *ie, const double *iei, const double (*w)[NR], int jb,
                                              double (*dst)[NR], double *part, Prefetch prefetch) {
  constexpr int SWV = solve_waves<NR>(), NE = NB / SWV;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (MODE <= 1 || rows_on_lanes<NR>()) {
    double acc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = 0.0;
#pragma unroll
    for (int u = 0; u < NE; ++u) mac_cols<NR, Z>(acc, ie[u], Z ? iei[u] : 0.0, &w[wave + SWV * u][0]);
    prefetch();
#pragma unroll
    for (int r = 0; r < NR; ++r) part[(wave * NR + r) * 64 + lane] = acc[r];
    __syncthreads();
    for (int o = threadIdx.x; o < 64 * NR; o += SWV * 64) {
      const int r = o / 64, l = o % 64;
      double tot = 0.0;
#pragma unroll
      for (int q = 0; q < SWV; ++q) tot += part[(q * NR + r) * 64 + l];
      if (l < jb) dst[l][r] = tot;
    }
  } else {
    constexpr int RG = NR >= 16 ? 2 : NE;  // rows of this wavefront whose products are summed together
#pragma unroll
    for (int u0 = 0; u0 < NE; u0 += RG) {
      double prod[RG * NR];
#pragma unroll
      for (int o = 0; o < RG * NR; ++o) prod[o] = 0.0;
#pragma unroll
      for (int q = 0; q < RG; ++q)
        mac_cols<NR, Z>(*reinterpret_cast<double(*)[NR]>(&prod[q * NR]), ie[u0 + q], Z ? iei[u0 + q] : 0.0, &w[lane][0]);
      wave_reduce_scatter<RG * NR>(prod);
      if (wave_reduce_owner<RG * NR>(lane)) {
#pragma unroll
        for (int k = 0; k < (RG * NR >= 64 ? RG * NR / 64 : 1); ++k) {
          const int idx = wave_reduce_index<RG * NR>(lane, k);
          const int l = wave + SWV * (u0 + idx / NR);
          if (l < jb) dst[l][idx % NR] = prod[k];
        }
      }
    }
    prefetch();
  }
  __syncthreads();
}

template <int MODE, int NR, bool Z, int K>
__device__ __forceinline__ void super_sub_block(const Band &b, const double *__restrict__ invs, int j0, int jbs, int nsub,
                                                double (*v)[NR], double (*w)[NR], double *part, double *ce, double *cei,
                                                double *ie, double *iei) {
  if (K >= nsub) return;  // (uniform over the workgroup)
  constexpr bool fwd = (MODE == 0 || MODE == 2);
  constexpr size_t blk = Z ? (size_t)kInvBlockZ : (size_t)(2 * NB * NB);
  constexpr size_t half = (MODE == 1 || MODE == 2) ? blk / 2 : 0;
  const int sblk = fwd ? K : nsub - 1 - K;
  const int js = j0 + sblk * NB, jb = min(NB, j0 + jbs - js);
  const int cb = fwd ? j0 : js + jb;
  const int nc = fwd ? js - j0 : j0 + jbs - (js + jb);
  // the next sub-block's place, for the loads issued on its behalf (none left: this one's again, all entries masked)
  const bool more = K + 1 < nsub;
  const int sblk2 = more ? (fwd ? sblk + 1 : sblk - 1) : sblk;
  const int js2 = j0 + sblk2 * NB, jb2 = min(NB, j0 + jbs - js2);
  const int cb2 = fwd ? j0 : js2 + jb2;
  const int nc2 = more ? (fwd ? js2 - j0 : j0 + jbs - (js2 + jb2)) : 0;
  // (min: vv + 64 u stays inside v whatever nc is)
  coupling_apply_regs<MODE, NR, Z, K>(ce, cei, nc, v + (cb - j0 <= (SB - K) * NB ? cb - j0 : 0), v + (js - j0), jb, w, part, [&]() {
    coupling_load<MODE, NR, Z, (K + 1 < SB ? K + 1 : 0)>(b, js2, cb2, nc2, ce, cei);
  });
  inverse_apply<MODE, NR, Z>(ie, iei, w, jb, v + (js - j0), part, [&]() {
    inverse_load<MODE, NR, Z>(invs + (size_t)(j0 / NB + sblk2) * blk + half, ie, iei);
  });
  if constexpr (K + 1 < SB)
    super_sub_block<MODE, NR, Z, K + 1>(b, invs, j0, jbs, nsub, v, w, part, ce, cei, ie, iei);
}

// The solve INSIDE the super block [j0, j0 + jbs): v holds its right-hand side on entry (rows beyond jbs zero) and its
// solution on return; w, part: scratch.  All threads of the workgroup; ends with a barrier.
// ie / iei: the entries of the FIRST sub-block's inverse this thread multiplies with, requested by the caller before it
// loads v (first_inverse_load), so that they travel meanwhile.
template <int MODE, int NR, bool Z = false>
__device__ __forceinline__ void first_inverse_load(const double *__restrict__ invs, int j0, int jbs, double *ie, double *iei) {
  constexpr bool fwd = (MODE == 0 || MODE == 2);
  const int nsub = (jbs + NB - 1) / NB;
  constexpr size_t blk = Z ? (size_t)kInvBlockZ : (size_t)(2 * NB * NB);
  constexpr size_t half = (MODE == 1 || MODE == 2) ? blk / 2 : 0;
  inverse_load<MODE, NR, Z>(invs + (size_t)(j0 / NB + (fwd ? 0 : nsub - 1)) * blk + half, ie, iei);
}
template <int MODE, int NR, bool Z = false>
__device__ __forceinline__ void solve_super_block_in_lds(const Band &b, const double *__restrict__ invs, int j0, int jbs,
                                                         double (*v)[NR], double (*w)[NR], double *part, double *ie,
                                                         double *iei) {
  constexpr bool fwd = (MODE == 0 || MODE == 2);
  constexpr int NE = NB / solve_waves<NR>();
  const int nsub = (jbs + NB - 1) / NB;
  constexpr size_t blk = Z ? (size_t)kInvBlockZ : (size_t)(2 * NB * NB);
  constexpr size_t half = (MODE == 1 || MODE == 2) ? blk / 2 : 0;
  if (couplings_ahead<NR, Z>()) {
    double ce[(SB - 1) * NE], cei[(SB - 1) * NE];
    super_sub_block<MODE, NR, Z, 0>(b, invs, j0, jbs, nsub, v, w, part, ce, cei, ie, iei);
  } else
  for (int k = 0; k < nsub; ++k) {
    const int sblk = fwd ? k : nsub - 1 - k;
    const int js = j0 + sblk * NB, jb = min(NB, j0 + jbs - js);
    // couplings with the sub-blocks of this super block that are already solved
    const int cb = fwd ? j0 : js + jb;
    const int nc = fwd ? js - j0 : j0 + jbs - (js + jb);
    coupling_apply<MODE, NR, Z>(b, js, cb, nc, v + (cb - j0), v + (js - j0), jb, w, part);
    // (the last sub-block fetches its own inverse again: a branch here would make the compiler sink the multiplications
    // of inverse_apply below it, with every operand they read from LDS kept in registers meanwhile)
    const int next = k + 1 < nsub ? (fwd ? sblk + 1 : sblk - 1) : sblk;
    inverse_apply<MODE, NR, Z>(ie, iei, w, jb, v + (js - j0), part, [&]() {
      inverse_load<MODE, NR, Z>(invs + (size_t)(j0 / NB + next) * blk + half, ie, iei);
    });
  }
}

// Where a BACKWARD pass may put the solution of a super block besides `out` (round 5: the multifrontal walk wrote the
// pivots' part of x with a launch of its own per level, big_scatter_x_kernel — a dependent kernel costs ~5 us whatever it
// does).  x: the caller's solution array at this system's first unknown (complex: packed pairs, at 2 x that unknown);
// column r of the work matrices is x + r * stride (complex: the real or imaginary parts of right-hand side r / 2).
struct SolutionSink {
  double *x = nullptr;
  size_t stride = 0;
};
template <int NR, bool Z>
__device__ __forceinline__ void sink_store(const SolutionSink &sk, int t, int r, double val) {
  if (Z) sk.x[(size_t)(r >> 1) * sk.stride + 2 * (size_t)t + (r & 1)] = val;
  else sk.x[(size_t)r * sk.stride + t] = val;
}

// NR right-hand sides at once: column r of in/out starts at r * stride
// `tile`: which rows outside the super block this workgroup updates: `tiles` consecutive blocks of 64
// rows starting at block tile * tiles (workgroup 0 also writes `out`).  Every workgroup redoes the
// in-super-block solve first, so large systems take several blocks per workgroup.
template <int MODE, int NR, bool Z = false>
__device__ __forceinline__ void solve_super_tile(const Band &b, const double *__restrict__ invs, int j0, int jbs,
                                                 double *in, double *__restrict__ out, size_t stride, int tile,
                                                 double *dsm, int tiles = 1, SolutionSink sink = SolutionSink()) {
  constexpr int SWV = solve_waves<NR>();
  double(*v)[NR] = reinterpret_cast<double(*)[NR]>(dsm);                          // [SB * NB]
  double(*w)[NR] = reinterpret_cast<double(*)[NR]>(dsm + SB * NB * NR);           // [NB]
  double(*res)[NR] = reinterpret_cast<double(*)[NR]>(dsm + (SB + 1) * NB * NR);   // [NB]
  double *part = dsm + (SB + 2) * NB * NR;                                        // [SWV][NR][64]
  constexpr bool fwd = (MODE == 0 || MODE == 2);
  const int tid = threadIdx.x;
  double ie[NB / SWV], iei[NB / SWV];
  first_inverse_load<MODE, NR, Z>(invs, j0, jbs, ie, iei);
  // v starts as the right-hand side of the super block; every sub-block replaces its 64 entries by the solution
  for (int o = tid; o < SB * NB * NR; o += SWV * 64) {
    const int t = o % (SB * NB), r = o / (SB * NB);
    v[t][r] = t < jbs ? in[(size_t)r * stride + j0 + t] : 0.0;
  }
  __syncthreads();
  solve_super_block_in_lds<MODE, NR, Z>(b, invs, j0, jbs, v, w, part, ie, iei);
  if (tile == 0)
    for (int o = tid; o < jbs * NR; o += SWV * 64) {
      const int t = o % jbs, r = o / jbs;
      out[(size_t)r * stride + j0 + t] = v[t][r];
      if (sink.x) sink_store<NR, Z>(sink, j0 + t, r, v[t][r]);
    }
  // the rows of this workgroup outside the super block, 64 at a time
  for (int q = 0; q < tiles; ++q) {
    const int blk = tile * tiles + q;
    const int rb = fwd ? j0 + jbs + blk * 64 : j0 - (blk + 1) * 64;
    if (fwd ? rb >= b.n : rb + 64 <= 0) break;  // workgroup-uniform
    gemv64<MODE, NR, Z>(b, rb, j0, jbs, v, res, part);
    for (int o = tid; o < 64 * NR; o += SWV * 64) {
      const int l = o % 64, r = o / 64;
      const int i = rb + l;
      const bool ok = fwd ? (i < b.n) : (i >= 0);
      if (ok) in[(size_t)r * stride + i] -= res[l][r];
    }
    __syncthreads();  // res and part are reused by the next block
  }
}

// The same steps as a software pipeline over launches (round 4).  Step k of a pass needs its super block's rows final,
// and in the plain form that means every row update of step k - 1: a launch per step, each the whole chain "solve 256
// unknowns in four dependent sub-blocks, THEN stream the rows" — 26 us per step at config C5 for 6.5 us worth of bytes.
// Here launch k carries two kinds of workgroups that depend on EARLIER launches only:
//   lead  (<= SB per system)   solve super block k from  in + carry  (carry: see below), write it to `out`, and
//                              compute the update of the NEXT super block's rows only, as plain stores into `carry`;
//   bulk  (the rest)           the row updates of step k - 1 beyond that next super block, with the solved super block
//                              k - 1 read from `out` (no redone solve): read-modify-write of `in`.
// Rows of super block m thus receive  carry[m]  from the lead of launch m - 1 (one writer, no read-modify-write) and
// their `in` updates from the bulk groups of launches <= m - 1 (steps <= m - 2): no two launches — and no two groups of
// one launch — touch the same words, every sum keeps its order, and the latency chain of a step (the lead) runs beside
// the bytes of the step before (the bulk) instead of in front of its own.  A pass takes steps + 1 launches.
//   role 0: lead, `tile` = which 64-row block of the next super block (0 also writes `out`); first: no carry yet
//   role 1: bulk of the step whose super block is [j0, j0 + jbs): blocks SB + tile * tiles ... of the rows beyond
template <int MODE, int NR, bool Z = false>
__device__ __forceinline__ void solve_super_pipelined(const Band &b, const double *__restrict__ invs, int j0, int jbs,
                                                      double *in, double *out, double *carry, size_t stride, int role,
                                                      int tile, bool first, double *dsm, int tiles,
                                                      SolutionSink sink = SolutionSink()) {
  constexpr int SWV = solve_waves<NR>();
  double(*v)[NR] = reinterpret_cast<double(*)[NR]>(dsm);                          // [SB * NB]
  double(*w)[NR] = reinterpret_cast<double(*)[NR]>(dsm + SB * NB * NR);           // [NB]
  double(*res)[NR] = reinterpret_cast<double(*)[NR]>(dsm + (SB + 1) * NB * NR);   // [NB]
  double *part = dsm + (SB + 2) * NB * NR;                                        // [SWV][NR][64]
  constexpr bool fwd = (MODE == 0 || MODE == 2);
  const int tid = threadIdx.x;
  if (role == 0) {
    double ie[NB / SWV], iei[NB / SWV];
    first_inverse_load<MODE, NR, Z>(invs, j0, jbs, ie, iei);
    for (int o = tid; o < SB * NB * NR; o += SWV * 64) {
      const int t = o % (SB * NB), r = o / (SB * NB);
      double x = 0.0;
      if (t < jbs) {
        x = in[(size_t)r * stride + j0 + t];
        if (!first) x += carry[(size_t)r * stride + j0 + t];
      }
      v[t][r] = x;
    }
    __syncthreads();
    solve_super_block_in_lds<MODE, NR, Z>(b, invs, j0, jbs, v, w, part, ie, iei);
    if (tile == 0)
      for (int o = tid; o < jbs * NR; o += SWV * 64) {
        const int t = o % jbs, r = o / jbs;
        out[(size_t)r * stride + j0 + t] = v[t][r];
        if (sink.x) sink_store<NR, Z>(sink, j0 + t, r, v[t][r]);
      }
    const int rb = fwd ? j0 + jbs + tile * 64 : j0 - (tile + 1) * 64;
    if (fwd ? rb >= b.n : rb + 64 <= 0) return;  // workgroup-uniform: no such rows (the last super block)
    gemv64<MODE, NR, Z>(b, rb, j0, jbs, v, res, part);
    for (int o = tid; o < 64 * NR; o += SWV * 64) {
      const int l = o % 64, r = o / 64;
      const int i = rb + l;
      const bool ok = fwd ? (i < b.n) : (i >= 0);
      if (ok) carry[(size_t)r * stride + i] = -res[l][r];
    }
  } else {
    for (int o = tid; o < SB * NB * NR; o += SWV * 64) {
      const int t = o % (SB * NB), r = o / (SB * NB);
      v[t][r] = t < jbs ? out[(size_t)r * stride + j0 + t] : 0.0;
    }
    __syncthreads();
    for (int q = 0; q < tiles; ++q) {
      const int blk = SB + tile * tiles + q;
      const int rb = fwd ? j0 + jbs + blk * 64 : j0 - (blk + 1) * 64;
      if (fwd ? rb >= b.n : rb + 64 <= 0) break;  // workgroup-uniform
      gemv64<MODE, NR, Z>(b, rb, j0, jbs, v, res, part);
      for (int o = tid; o < 64 * NR; o += SWV * 64) {
        const int l = o % 64, r = o / 64;
        const int i = rb + l;
        const bool ok = fwd ? (i < b.n) : (i >= 0);
        if (ok) in[(size_t)r * stride + i] -= res[l][r];
      }
      __syncthreads();  // res and part are reused by the next block
    }
  }
}

template <int MODE, int NR>
__global__ __launch_bounds__(solve_waves<NR>() * 64) void solve_super_kernel(Band b, const double *__restrict__ invs, int j0,
                                                              int jbs, double *in, double *__restrict__ out,
                                                              size_t stride) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  solve_super_tile<MODE, NR>(b, invs, j0, jbs, in, out, stride, (int)blockIdx.x, dsm);
}


// ---- host drivers shared by the band path and the multifrontal fronts ---------------------------
inline size_t inverse_block_elems(int npiv) { return (size_t)((npiv + NB - 1) / NB) * (2 * NB * NB); }

inline void set_factor_attributes() {
  static std::atomic<uint64_t> attr_set{0};  // one bit per device
  if (!first_use_on_this_device(attr_set)) return;
  SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_lu_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileBytes)));
  SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&trsm_gemm_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileBytes)));
  SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_update_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileBytes)));  // uses less
  SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_update_bulk_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileBytes)));
  SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_lu_kernel_z),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDiagLdsZ));
  SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&trsm_gemm_kernel_z),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrsmLdsZ));
  SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_update_kernel_z),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDiagLdsZ));
  SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_update_bulk_kernel_z),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUpdateLdsZ));
  mark_used_on_this_device(attr_set);
}

// Right-looking blocked LU without interchanges of the first npiv columns/rows of the view b
// (npiv = b.n: the whole matrix); the trailing block is left holding the Schur complement.  The
// inverses of the diagonal blocks go to d_invs (inverse_block_elems(npiv) doubles), a zero pivot sets
// *d_singular.  Asynchronous on stream s.
// `helper`: a second stream for the look-ahead tile of large windows (nullptr: everything on s); the
// two streams are joined by events inside, the caller only sees s.
struct LookaheadFork {
  hipStream_t helper = nullptr;
  hipEvent_t ready = nullptr, done = nullptr;
  explicit LookaheadFork(hipStream_t h) : helper(h) {
    if (!helper) return;
    SPL_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    SPL_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  }
  ~LookaheadFork() {
    if (ready) (void)hipEventDestroy(ready);
    if (done) (void)hipEventDestroy(done);
  }
};

template <bool Z = false>
inline void factor_loop(const Band &b, int npiv, double *d_invs, int *d_singular, hipStream_t s,
                        hipStream_t helper = nullptr) {
  set_factor_attributes();
  LookaheadFork fork(helper);
  const int n = b.n, kl = b.kl, ku = b.ku;
  const size_t gemm_lds = Z ? kDiagLdsZ : kTileBytes + 2 * NB * sizeof(double);
  constexpr size_t blk = Z ? (size_t)kInvBlockZ : (size_t)(2 * NB * NB), half = blk / 2;  // inv(L11) | inv(U11)
  auto slot = [&](int j) { return d_invs + (size_t)(j / NB) * blk; };
  auto below_of = [&](int jend) { return std::max(0, std::min(n, jend + kl) - jend); };  // rows with in-band entries
  auto right_of = [&](int jend) { return std::max(0, std::min(n, jend + ku) - jend); };
  auto diag = [&](int j0, int jb) {
    if (Z)
      hipLaunchKernelGGL(diag_lu_kernel_z, dim3(1), dim3(256), kDiagLdsZ, s, b, j0, jb, d_singular, slot(j0),
                         slot(j0) + half);
    else
      hipLaunchKernelGGL(diag_lu_kernel, dim3(1), dim3(256), kTileBytes + 2 * NB * sizeof(double), s, b, j0, jb,
                         d_singular, slot(j0), slot(j0) + half);
  };
  auto trsm = [&](int j0, int jb) {
    const int below = below_of(j0 + jb), right = right_of(j0 + jb);
    const int tiles = (below + 63) / 64 + (right + 63) / 64;
    if (tiles <= 0) return;
    if (Z)
      hipLaunchKernelGGL(trsm_gemm_kernel_z, dim3((unsigned)tiles), dim3(256), kTrsmLdsZ, s, b, j0, jb, below, right,
                         slot(j0), slot(j0) + half);
    else
      hipLaunchKernelGGL(trsm_gemm_kernel, dim3((unsigned)tiles), dim3(256), kTrsmLds, s, b, j0, jb, below,
                         right, slot(j0), slot(j0) + half);
  };
  // trailing update of rows/cols from `origin` with the K range [kb, kb+klen); factors the diagonal
  // block at `origin` on the way (look-ahead) when it is a pivot block.  Returns whether it did.
  auto update = [&](int origin, int kb, int klen, int kend, bool lshape) {
    // rows/columns reached by the blocks in the K range end at the reach of its LAST block
    const int re = std::min(n, kend + kl), ce = std::min(n, kend + ku);
    if (re <= origin || ce <= origin) return false;
    const int ntr = (re - origin + 63) / 64, ntc = (ce - origin + 63) / 64;
    Region g{origin, re, origin, ce, kb, klen, lshape ? 1 : 0, ntr, npiv};
    if (!lshape && fork.helper && ntr >= kSplitTiles && ntc >= kSplitTiles) {
      // tile (0,0) and the next diagonal block on the helper stream, the rest of the window at three
      // wavefronts per SIMD on this one; both start after the panel solves and meet again after
      SPL_HIP(hipEventRecord(fork.ready, s));
      SPL_HIP(hipStreamWaitEvent(fork.helper, fork.ready, 0));
      if (Z) {
        hipLaunchKernelGGL(gemm_update_kernel_z, dim3(1, 1), dim3(256), gemm_lds, fork.helper, b, g, d_singular,
                           slot(origin), slot(origin) + half);
        SPL_HIP(hipEventRecord(fork.done, fork.helper));
        hipLaunchKernelGGL(gemm_update_bulk_kernel_z, dim3((unsigned)ntr, (unsigned)ntc), dim3(256), kUpdateLdsZ, s, b,
                           g, d_singular);
      } else {
        hipLaunchKernelGGL(gemm_update_kernel, dim3(1, 1), dim3(256), gemm_lds, fork.helper, b, g, d_singular,
                           slot(origin), slot(origin) + half);
        SPL_HIP(hipEventRecord(fork.done, fork.helper));
        hipLaunchKernelGGL(gemm_update_bulk_kernel, dim3((unsigned)ntr, (unsigned)ntc), dim3(256), gemm_lds, s, b, g,
                           d_singular);
      }
      SPL_HIP(hipStreamWaitEvent(s, fork.done, 0));
      return origin < npiv;
    }
    const dim3 grid = lshape ? dim3((unsigned)(ntr + ntc - 1)) : dim3((unsigned)ntr, (unsigned)ntc);
    if (Z)
      hipLaunchKernelGGL(gemm_update_kernel_z, grid, dim3(256), gemm_lds, s, b, g, d_singular, slot(origin),
                         slot(origin) + half);
    else
      hipLaunchKernelGGL(gemm_update_kernel, grid, dim3(256), gemm_lds, s, b, g, d_singular, slot(origin),
                         slot(origin) + half);
    return origin < npiv;
  };
  // Look-ahead over pairs (round 4, dense views with a helper stream): the rest of the window used to be updated on s
  // (one workgroup with the next diagonal block on the helper), and the next pair's panel solves waited behind it:
  // 104 us of chain + the bulk pass per 128 pivots.  The next pair (c, d) only needs ITS two block columns and rows.
  // So after the solves of pair (a, b): on s an L-shaped pass brings block column / row c up to date (K = a, b) and
  // factors its diagonal block; the helper stream updates everything beyond d (K = a, b, the whole window at three
  // wavefronts per SIMD); block column / row d gets a, b together with c in the L-shaped pass of the next pair
  // (K = a, b, c: one contiguous range).  The chain of pair (c, d) then runs beside the bulk pass of pair (a, b); an
  // L-shaped pass waits for the bulk pass before it (they write the same entries), bulk passes follow each other on
  // the helper.  Only where the bulk pass outweighs the extra L-shaped pass: windows of kAheadTiles tiles a side.
  const bool dense = kl >= n - 1 && ku >= n - 1;
  // SPL_LU_LOOKAHEAD: 0 = off, k > 0 = from k tiles a side (tests: small fronts through this code); read at every call
  const char *ahead_env = getenv("SPL_LU_LOOKAHEAD");
  const bool ahead_off = ahead_env && atoi(ahead_env) == 0;
  const int ahead_tiles = ahead_env && atoi(ahead_env) > 0 ? atoi(ahead_env) : kAheadTiles;
  bool bulk_in_flight = false;  // a bulk pass on the helper that s has not waited for yet
  int owed_kb = -1;             // block column / row j0 + NB still lacks the panels from here on (-1: nothing owed)
  auto join_bulk = [&] {
    if (!bulk_in_flight) return;
    SPL_HIP(hipStreamWaitEvent(s, fork.done, 0));
    bulk_in_flight = false;
  };
  bool diag_done = false;  // the previous update already factored this diagonal block
  int j0 = 0;
  while (j0 < npiv) {
    const int jb = std::min(NB, npiv - j0);
    if (!diag_done) diag(j0, jb);
    trsm(j0, jb);
    if (j0 + 2 * NB <= npiv && kl >= NB && ku >= NB) {
      // pair of block steps: the first one only updates the panels of the second (L-shape), then
      // both update the rest of the window in one pass with K = 2 NB
      const int j1 = j0 + NB, j2 = j1 + NB;
      const int kb1 = owed_kb >= 0 ? owed_kb : j0;
      owed_kb = -1;
      if (!update(j1, kb1, j1 - kb1, j0 + NB, true)) diag(j1, NB);
      trsm(j1, NB);
      const int beyond = n - (j2 + 2 * NB);  // side of the window beyond the next pair
      if (dense && fork.helper && !ahead_off && j2 + 2 * NB <= npiv && beyond >= ahead_tiles * 64) {
        SPL_HIP(hipEventRecord(fork.ready, s));  // the panels of a, b
        join_bulk();                             // (the pass before wrote what the L-shaped pass adds to)
        diag_done = update(j2, j0, 2 * NB, j2, true);
        SPL_HIP(hipStreamWaitEvent(fork.helper, fork.ready, 0));
        const int origin = j2 + 2 * NB, nt = (n - origin + 63) / 64;
        Region g{origin, n, origin, n, j0, 2 * NB, 2, nt, npiv};
        if (Z)
          hipLaunchKernelGGL(gemm_update_bulk_kernel_z, dim3((unsigned)nt, (unsigned)nt), dim3(256), kUpdateLdsZ, fork.helper,
                             b, g, d_singular);
        else
          hipLaunchKernelGGL(gemm_update_bulk_kernel, dim3((unsigned)nt, (unsigned)nt), dim3(256), gemm_lds, fork.helper, b, g,
                             d_singular);
        SPL_HIP(hipEventRecord(fork.done, fork.helper));
        bulk_in_flight = true;
        owed_kb = j0;
      } else {
        join_bulk();
        diag_done = update(j2, j0, 2 * NB, j2, false);
      }
      j0 += 2 * NB;
    } else {
      join_bulk();
      diag_done = update(j0 + jb, j0, jb, j0 + jb, false);
      j0 += jb;
    }
  }
  join_bulk();
}

// one triangular pass over the first npiv unknowns of the view (npiv = b.n: the whole system); in a
// forward pass the rows of the view beyond npiv still receive their updates (the boundary rows of a
// multifrontal front)
template <int MODE, int NR>
inline void solve_pass(const Band &b, const double *d_invs, int bw, double *in, double *out, size_t stride,
                       hipStream_t s, int npiv = -1) {
  constexpr bool fwd = (MODE == 0 || MODE == 2);
  constexpr size_t lds = (size_t)((SB + 2) * NB + solve_waves<NR>() * 64) * NR * sizeof(double);
  static std::atomic<uint64_t> attr_set{0};  // one mask per instantiation, one bit per device
  if (first_use_on_this_device(attr_set)) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&solve_super_kernel<MODE, NR>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    mark_used_on_this_device(attr_set);
  }
  if (npiv < 0) npiv = b.n;
  const int n = b.n, step = SB * NB, nsup = (npiv + step - 1) / step;
  for (int k = 0; k < nsup; ++k) {
    const int j0 = (fwd ? k : nsup - 1 - k) * step, jbs = std::min(step, npiv - j0);
    const int rows = fwd ? std::max(0, std::min(n, j0 + jbs + bw) - (j0 + jbs)) : std::min(j0, bw);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(solve_super_kernel<MODE, NR>), dim3((unsigned)std::max(1, (rows + 63) / 64)),
                       dim3(solve_waves<NR>() * 64), lds, s, b, d_invs, j0, jbs, in, out, stride);
  }
}

template <int NR>
inline void solve_group(int sys, const Band &b, const double *d_invs, double *d_c, double *d_z, size_t stride,
                        hipStream_t s) {
  if (sys == 0) {
    solve_pass<0, NR>(b, d_invs, b.kl, d_c, d_z, stride, s);  // L forward: c -> z
    solve_pass<1, NR>(b, d_invs, b.ku, d_z, d_c, stride, s);  // U backward: z -> c
  } else {
    solve_pass<2, NR>(b, d_invs, b.ku, d_c, d_z, stride, s);  // U^T forward: c -> z
    solve_pass<3, NR>(b, d_invs, b.kl, d_z, d_c, stride, s);  // L^T backward: z -> c
  }
}


}  // namespace
}  // namespace spl

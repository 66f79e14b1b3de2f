// mf_chain.hpp — the pivot blocks of the large fronts in the triangular solves as a chain of MATRIX-VECTOR products
// (round 5).  Included into multifrontal.hip behind big_front(); everything lives in its anonymous namespace.
//
// Why.  A pass through the pivot block of a large front (np pivots) is a recurrence over blocks of pivots,
//     z_k = inv(L_kk) (c_k - sum_{j < k} L_kj z_j),
// and its length in DEPENDENT kernels is what a solve costs at the FEAST sizes (profiles/r05_solve_100_kernel_trace.txt:
// 464 of the 931 kernels of a solve at 100^3 are steps of this recurrence, 15.5 us each: the lead workgroup of a step
// pulls 0.75 MB through one CU — 256 pivots in four dependent sub-blocks, then the couplings of the next 256 rows).  The
// recurrence can be cut differently: with
//     T_k = inv(L_kk)            (S x S, S = 512: explicit, built once per factorisation from the stored 64 x 64 inverses)
//     M_k = T_k L_{k,k-1}        (S x S)
// a step is  z_k = T_k a_k - M_k z_{k-1},  a_k = c_k - sum_{j <= k-2} L_kj z_j:  ONE product of an S x 2S matrix with
// a vector whose two halves are both final when the launch starts (a_k: the bulk updates of the launches before; z_{k-1}:
// the lead of the launch before) — no chain inside the step, every row on a wavefront of its own, S / 32 workgroups
// pulling 128 KB each instead of one pulling 768 KB, and half as many steps.  The same for U from the last block up:
// x_k = inv(U_kk) a_k - (inv(U_kk) U_{k,k+1}) x_{k+1}.  T_k and M_k take the place of L_kk and L_{k,k-1} in the bytes a walk
// reads (the rows of C_k = [T_k | M_k] are stored row-major so that a wavefront streams a row), at the cost of one more
// copy of those blocks in memory (np x 2S doubles per large front and factor).
//
// Numerics: an explicit inverse of a 512 x 512 triangular block of a front factored without interchanges (or with
// interchanges inside its 64 x 64 blocks) — the entries are checked when they are built: a chain with an entry beyond
// chain::kLimit (or not finite) is dropped and the walk keeps its substitution steps (solve_super_pipelined).  The solves
// are followed by the residual check of umfpack_*_solve either way.
//
// The transposed systems (U^H z = c, L^H x = z) have chains of their own, from the same construction on G = U^H, L^H
// (chain_build_kernel<Z, true>), built by the first solve of A^T x = b: as much memory again, only for callers who solve both.
#pragma once

namespace chain {

constexpr int kRows = 32;          // rows of a super block per lead workgroup (16 wavefronts x 2 rows)
// ... on the levels whose fronts all have at most kWidePivots pivots: 16 wavefronts x 8 rows (complex: x 4)
__host__ __device__ constexpr int rows_wide(bool z) { return z ? 64 : 128; }
constexpr int kWidePivots = 256;
constexpr double kLimit = 1e8;     // largest entry of a chain matrix that is still trusted

struct View {
  double *buf = nullptr;       // chain matrices of all large fronts, real parts (imaginary parts: + plane)
  const int64_t *off = nullptr;  // per front: first double of its L chain (the U chain np * ld further); -1: none
  size_t plane = 0;
  int span = 0;
};

// columns of the T part of a front's chain rows, and the length of a row
__host__ __device__ __forceinline__ int tw_of(int np, int S) { return np >= S ? S : ((np + 63) & ~63); }
__host__ __device__ __forceinline__ int ld_of(int np, int S) { return tw_of(np, S) + (np > S ? S : 0); }

}  // namespace chain

// ---- build: [T_k | M_k] = inv(L_kk) [I | L_{k,k-1}] by blocked substitution over the 64 x 64 sub-blocks of block k
// (the stored inverses of the diagonal blocks carry the interchanges of a pivoted block, so this is the operator the
// substitution steps apply), on the fp64 matrix cores.  A workgroup (4 wavefronts) owns 64 columns of the right-hand
// side [I | L_{k,k-1}]; sub-block s of its solution needs the sub-blocks solved before it, which it reads back from
// the chain buffer (written by this workgroup: visible after a barrier).  Wavefront w owns rows 16 w .. 16 w + 15 of
// every sub-block: four 16 x 16 accumulators (v_mfma_f64_16x16x4: lane l supplies A[l % 16][k0 + l / 16] and
// B[k0 + l / 16][l % 16], holds C[l / 16 + 4 r][l % 16] in element r).
// items: (front, block k, factor) triples; prefix: column tiles before each item.
template <bool Z, bool TR = false>
__global__ __launch_bounds__(256) void chain_build_kernel(const int *__restrict__ item_f, const int *__restrict__ item_k,
                                                          const int64_t *__restrict__ prefix, int count, TreeView t,
                                                          const double *__restrict__ invs, chain::View cv,
                                                          int *__restrict__ bad) {
  __shared__ double Xs[Z ? 2 : 1][NB][LDP];
  const int64_t flat = (int64_t)blockIdx.x;
  const int it = item_of_tile(prefix, count, flat);
  const int f = item_f[it], up = item_k[it] >> 30, k = item_k[it] & 0x3fffffff;
  const int tile = (int)(flat - prefix[it]);
  const int S = cv.span, np = t.np[f], ldp = t.ldp[f];
  const int j0 = k * S, jbs = min(S, np - j0), nsub = (jbs + 63) / 64;
  const int twf = chain::tw_of(np, S), ldc = chain::ld_of(np, S);
  const int tk = (jbs + 63) / 64;  // column tiles of the T part of this block
  const bool mpart = tile >= tk;
  const int cm = tile - tk;
  const int colbase = mpart ? twf + 64 * cm : 64 * tile;
  const int jn0 = (up ? j0 + S : j0 - S) + 64 * cm;  // M part: first column (in P) of this tile's right-hand side
  const double *P = t.arena + (int64_t)t.zm * t.poff[f];
  const size_t pz = Z ? (size_t)ldp * (size_t)np : 0;
  double *C = cv.buf + cv.off[f] + (up ? (int64_t)np * ldc : 0) + (int64_t)k * S * ldc;
  constexpr size_t iblk = Z ? (size_t)kInvBlockZ : (size_t)(2 * NB * NB);
  // TR: the chains of the TRANSPOSED systems (U^H z = c forward, L^H x = z backward): the same substitution with
  // G = U^H (lower) resp. L^H (unit upper) — entries G(i, j) = conj(F(j, i)), diagonal blocks' inverses transposed
  const double *inv = invs + (Z ? 2 : 1) * t.ioff[f] + (size_t)(j0 / NB) * iblk + ((up != 0) != TR ? iblk / 2 : 0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
  // T part: column tile ct of the identity — sub-blocks before (L) / after (U) its diagonal one are zero
  const int s_first = mpart ? (up ? nsub - 1 : 0) : tile;
  const int nstep = up ? s_first + 1 : nsub - s_first;
  bool wrong = false;
#pragma unroll 1
  for (int q = 0; q < nstep; ++q) {
    const int s = up ? s_first - q : s_first + q;
    const int r0 = j0 + 64 * s;
    double4v acc[4], acci[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wave + kq + 4 * r, col = nt * 16 + li;
        double re = 0.0, im = 0.0;
        if (!mpart) {
          re = (q == 0 && row == col && 64 * s + row < jbs) ? 1.0 : 0.0;
        } else {
          const int gr = r0 + row, gc = jn0 + col;
          if (gr < np && gc < np) {
            const size_t at = TR ? (size_t)gc + (size_t)gr * ldp : (size_t)gr + (size_t)gc * ldp;
            re = P[at];
            if (Z) im = TR ? -P[at + pz] : P[at + pz];
          }
        }
        acc[nt][r] = re;
        acci[nt][r] = im;
      }
    }
    // minus the couplings with the sub-blocks solved before
#pragma unroll 1
    for (int p = 0; p < q; ++p) {
      const int tt = up ? s_first - p : s_first + p;
      const int c0 = j0 + 64 * tt;
#pragma unroll 4
      for (int k0 = 0; k0 < 64; k0 += 4) {
        const int gr = r0 + 16 * wave + li, gc = c0 + k0 + kq;
        const bool in = gr < np && gc < np;
        const double *pa = in ? P + (TR ? (size_t)gc + (size_t)gr * ldp : (size_t)gr + (size_t)gc * ldp) : P;
        double a = *pa, ai = Z ? pa[pz] : 0.0;
        a = in ? -a : 0.0;
        ai = in ? (TR ? ai : -ai) : 0.0;  // (minus the coupling; TR: of its conjugate)
        const int xr = 64 * tt + k0 + kq;
        const bool xin = xr < jbs;
        const double *px = C + (size_t)(xin ? xr : 0) * ldc + colbase + li;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          double b = px[nt * 16], bi = Z ? px[nt * 16 + cv.plane] : 0.0;
          b = xin ? b : 0.0;
          bi = xin ? bi : 0.0;
          acc[nt] = mfma16(a, b, acc[nt]);
          if (Z) {
            acc[nt] = mfma16(-ai, bi, acc[nt]);
            acci[nt] = mfma16(a, bi, acci[nt]);
            acci[nt] = mfma16(ai, b, acci[nt]);
          }
        }
      }
    }
    // X_s = (stored inverse of the diagonal block) x acc: the accumulators of the four wavefronts meet in LDS
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Xs[0][16 * wave + kq + 4 * r][nt * 16 + li] = acc[nt][r];
        if (Z) Xs[Z ? 1 : 0][16 * wave + kq + 4 * r][nt * 16 + li] = acci[nt][r];
      }
    __syncthreads();
    const int jb = min(NB, jbs - 64 * s);
    const double *ig = inv + (size_t)s * iblk;
    double4v x[4], xi[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      x[nt] = (double4v){0.0, 0.0, 0.0, 0.0};
      xi[nt] = (double4v){0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll 4
    for (int k0 = 0; k0 < 64; k0 += 4) {
      const int kc = k0 + kq;
      const int ia = TR ? kc + (16 * wave + li) * NB : (16 * wave + li) + kc * NB;
      double a = ig[ia], ai = Z ? ig[NB * NB + ia] : 0.0;
      a = kc < jb ? a : 0.0;
      ai = kc < jb ? (TR ? -ai : ai) : 0.0;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const double b = Xs[0][kc][nt * 16 + li], bi = Z ? Xs[Z ? 1 : 0][kc][nt * 16 + li] : 0.0;
        x[nt] = mfma16(a, b, x[nt]);
        if (Z) {
          x[nt] = mfma16(-ai, bi, x[nt]);
          xi[nt] = mfma16(a, bi, xi[nt]);
          xi[nt] = mfma16(ai, b, xi[nt]);
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 64 * s + 16 * wave + kq + 4 * r;
        if (row < jbs) {
          double *dst = C + (size_t)row * ldc + colbase + nt * 16 + li;
          dst[0] = x[nt][r];
          wrong |= !(fabs(x[nt][r]) <= chain::kLimit);
          if (Z) {
            dst[cv.plane] = xi[nt][r];
            wrong |= !(fabs(xi[nt][r]) <= chain::kLimit);
          }
        }
      }
    __syncthreads();  // X_s is visible to the workgroup (one CU, one L1), Xs may be rewritten
  }
  if (wrong) atomicOr(bad, 1);
}

// ---- solve: one launch of the pass MODE (0: L z = c forward, 1: U x = z backward) over the pivot blocks of the listed
// fronts: for every front the LEAD groups of block `launch` of its pass — 32 rows of z_k = C_k [a_k ; -z_{k-1}] per
// workgroup, two rows per wavefront, lanes along the row — and the BULK groups of the block before: the updates of the
// rows beyond the next block with the panel itself (gemv64).  Forward: W -> Z, backward: Z -> W (as big_super_pipe_kernel).
typedef double double2v __attribute__((ext_vector_type(2)));

// chunks [q0, q0 + QN) of 128 columns of this wavefront's RW rows: requested (chain_load), then multiplied with u
// (chain_mac; u in LDS: real parts, imaginary parts 2 S further)
template <bool Z, int RW, int QN>
struct ChainRegs {
  double2v cr[RW][QN], ci[Z ? RW : 1][QN];
  bool need[QN];
};
struct ChainRow {
  const double *base;  // first entry of the wavefront's first row
  int rmax;            // rows of the wavefront beyond the first that exist
  size_t plane;
  int ldc, tlo, thi, twf;  // row length; the 64 x 64 blocks of the T part that hold anything: columns [tlo, thi)
  bool hasm, active;
};
template <bool Z, int RW, int QN>
__device__ __forceinline__ void chain_load(const ChainRow &w, int q0, ChainRegs<Z, RW, QN> &g) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < QN; ++j) {
    const int c0 = 128 * (q0 + j);
    // the chunks of the row that hold anything: the T part is block triangular (64 x 64 blocks), the M part full
    g.need[j] = w.active && c0 < w.ldc && ((c0 < w.thi && c0 + 128 > w.tlo && c0 < w.twf) || (w.hasm && c0 >= w.twf));
    if (g.need[j]) {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const double *p = w.base + (size_t)min(r, w.rmax) * (size_t)w.ldc + (size_t)(c0 + 2 * lane);
        g.cr[r][j] = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(p));
        if (Z) g.ci[r][j] = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(p + w.plane));
      }
    }
  }
}
template <bool Z, int RW, int QN>
__device__ __forceinline__ void chain_mac(const ChainRegs<Z, RW, QN> &g, int q0, const double *us, int S2, double *acc) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < QN; ++j)
    if (g.need[j]) {
      const int c0 = 128 * (q0 + j) + 2 * lane;
      const double2v ur = *reinterpret_cast<const double2v *>(us + c0);
      const double2v ui = Z ? *reinterpret_cast<const double2v *>(us + S2 + c0) : (double2v){0.0, 0.0};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          if (!Z) {
            acc[r] = __builtin_fma(g.cr[r][j][e], ur[e], acc[r]);
          } else {
            acc[2 * r] = __builtin_fma(g.cr[r][j][e], ur[e], acc[2 * r]);
            acc[2 * r + 1] = __builtin_fma(g.cr[r][j][e], ui[e], acc[2 * r + 1]);
          }
        }
        if (Z) {
#pragma unroll
          for (int r = 0; r < RW; ++r) {
            acc[2 * r] = __builtin_fma(-g.ci[Z ? r : 0][j][e], ui[e], acc[2 * r]);
            acc[2 * r + 1] = __builtin_fma(g.ci[Z ? r : 0][j][e], ur[e], acc[2 * r + 1]);
          }
        }
      }
    }
}

// the lead groups: rows [16 RW group, 16 RW (group + 1)) of block k of the pass, RW rows per wavefront; all threads of
// the workgroup (one barrier).  The entries of the rows are requested BEFORE u goes to LDS: the two round trips of a step
// travel together.  RW = 2: rows of up to 2 S entries in batches of eight chunks (complex: four); RW = 8: the levels
// (complex: 4) whose fronts have at most 256 pivots (thousands of fronts, rows of at most two chunks) — a quarter of
// the workgroups.
template <int NR, bool Z, int RW, int QB = (Z ? 8 : 16) / RW, int PASSES = 1>
__device__ __forceinline__ void chain_lead(const BigFront &b, const chain::View &cv, bool fwd, int k, int S, int group,
                                           const double *in, const double *prev, double *out, const SolutionSink &sink,
                                           double *us) {
  const int n = b.np, K = (n + S - 1) / S;
  const int j0 = k * S, jbs = min(S, n - j0);
  const int twf = chain::tw_of(n, S), ldc = chain::ld_of(n, S);
  const bool hasm = fwd ? k > 0 : k < K - 1;
  const int jn0 = fwd ? j0 - S : j0 + S, jnb = hasm ? min(S, n - jn0) : 0;
  const double *C = cv.buf + cv.off[b.f] + (fwd ? 0 : (int64_t)n * ldc) + (int64_t)k * S * ldc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t fs = (size_t)b.fs;
  const int S2 = 2 * S;
  // this wavefront's rows: RW at a time, PASSES times (QB: chunks per batch — by default 32 double-pairs of entries in
  // flight per lane; the kernel of the levels of small pivot blocks takes fewer at a time to halve its registers)
  const int il0 = (group * (int)(blockDim.x >> 6) + wave) * RW * PASSES;
  ChainRow w;
  w.plane = cv.plane;
  w.ldc = ldc;
  w.twf = twf;
  w.hasm = hasm;
  auto rows_of = [&](int il) {
    w.active = il < jbs;
    w.base = C + (size_t)(w.active ? il : 0) * ldc;
    w.rmax = w.active ? min(RW - 1, jbs - 1 - il) : 0;
    w.tlo = fwd ? 0 : (il & ~63);
    w.thi = fwd ? (il & ~63) + 64 : twf;
  };
  rows_of(il0);
  ChainRegs<Z, RW, QB> g;
  chain_load<Z, RW, QB>(w, 0, g);
  // u = [a_k ; -z of the block before] in LDS
  const int ulen = min(S2, (ldc + 127) & ~127);
  for (int tt = threadIdx.x; tt < ulen; tt += blockDim.x) {
    double re = 0.0, im = 0.0;
    if (tt < twf) {
      if (tt < jbs) {
        re = in[j0 + tt];
        if (Z) im = in[fs + j0 + tt];
      }
    } else if (tt - twf < jnb) {
      re = -prev[jn0 + tt - twf];
      if (Z) im = -prev[fs + jn0 + tt - twf];
    }
    us[tt] = re;
    if (Z) us[S2 + tt] = im;
  }
  __syncthreads();
#pragma unroll 1
  for (int ps = 0; ps < PASSES; ++ps) {
    const int il = il0 + ps * RW;
    if (ps > 0) {
      rows_of(il);
      chain_load<Z, RW, QB>(w, 0, g);
    }
    if (!w.active) return;  // (wavefront-uniform; the rows of the later passes lie further down)
    constexpr int NV = (Z ? 2 : 1) * RW;
    double acc[NV];
#pragma unroll
    for (int o = 0; o < NV; ++o) acc[o] = 0.0;
    chain_mac<Z, RW, QB>(g, 0, us, S2, acc);
#pragma unroll 1
    for (int q0 = QB; 128 * q0 < ldc; q0 += QB) {
      chain_load<Z, RW, QB>(w, q0, g);
      chain_mac<Z, RW, QB>(g, q0, us, S2, acc);
    }
    wave_reduce_scatter<NV>(acc);
    if (wave_reduce_owner<NV>(lane)) {
      const int idx = wave_reduce_index<NV>(lane, 0);
      const int rowi = Z ? idx >> 1 : idx, part = Z ? idx & 1 : 0;
      if (il + rowi < jbs) {
        const int t = j0 + il + rowi;
        out[(size_t)part * fs + t] = acc[0];
        if (sink.x) sink_store<NR, Z>(sink, t, part, acc[0]);
      }
    }
  }
}

// The lead groups with 8 or 16 right-hand-side columns (NR real columns, or NR / 2 complex ones as (re, im) column
// pairs): the same product, a pair of rows per wavefront at a time, u in LDS one segment of 512 columns at a time as
// us[column r][t] (a lane reads its two t of a column as one 16-byte word, lanes side by side: no bank conflicts).
// lead_rows / wavefronts rows per wavefront: one pair on the levels of large fronts, four on the levels of small pivot
// blocks (whose rows fit one segment, loaded once).
constexpr int kChainSeg = 512;
template <int NR, bool Z>
__device__ __forceinline__ void chain_lead_multi(const BigFront &b, const chain::View &cv, bool fwd, int k, int S, int group,
                                                 int lead_rows, const double *in, const double *prev, double *out,
                                                 const SolutionSink &sink, double *us) {
  constexpr int SWV = solve_waves<NR>();
  const int n = b.np, K = (n + S - 1) / S;
  const int j0 = k * S, jbs = min(S, n - j0);
  const int twf = chain::tw_of(n, S), ldc = chain::ld_of(n, S);
  const bool hasm = fwd ? k > 0 : k < K - 1;
  const int jn0 = fwd ? j0 - S : j0 + S, jnb = hasm ? min(S, n - jn0) : 0;
  const double *C = cv.buf + cv.off[b.f] + (fwd ? 0 : (int64_t)n * ldc) + (int64_t)k * S * ldc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t fs = (size_t)b.fs;
  const int rpw = lead_rows / SWV;  // rows per wavefront
  const int nseg = (ldc + kChainSeg - 1) / kChainSeg;
  // u[seg] -> LDS (all threads)
  auto stage = [&](int seg) {
    const int t0 = seg * kChainSeg, ulen = min(kChainSeg, (ldc - t0 + 127) & ~127);  // (whole chunks of the row)
    for (int o = threadIdx.x; o < ulen * NR; o += SWV * 64) {
      const int tt = t0 + o % ulen, r = o / ulen;
      double val = 0.0;
      if (tt < twf) {
        if (tt < jbs) val = in[(size_t)r * fs + j0 + tt];
      } else if (tt - twf < jnb) {
        val = -prev[(size_t)r * fs + jn0 + tt - twf];
      }
      us[r * kChainSeg + (tt - t0)] = val;
    }
  };
  if (nseg == 1) {
    stage(0);
    __syncthreads();
  }
#pragma unroll 1
  for (int pr = 0; pr < rpw; pr += 2) {
    const int il = group * lead_rows + wave * rpw + pr;  // rows il, il + 1 (the same 64 x 64 block row: rpw is even)
    const bool active = il < jbs, two = il + 1 < jbs;
    const int tlo = fwd ? 0 : (il & ~63), thi = fwd ? (il & ~63) + 64 : twf;
    const double *row0 = C + (size_t)(active ? il : 0) * ldc + 2 * lane, *row1 = row0 + (two ? ldc : 0);
    double acc[2 * NR];
#pragma unroll
    for (int o = 0; o < 2 * NR; ++o) acc[o] = 0.0;
#pragma unroll 1
    for (int seg = 0; seg < nseg; ++seg) {
      if (nseg > 1) {
        __syncthreads();  // (the segment before is done with)
        stage(seg);
        __syncthreads();
      }
      if (!active) continue;
#pragma unroll 1
      for (int q0 = 0; q0 < kChainSeg / 128; q0 += 2) {
        double2v c0r[2], c1r[2], c0i[2], c1i[2];
        bool need[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c0 = seg * kChainSeg + 128 * (q0 + j);
          need[j] = c0 < ldc && ((c0 < thi && c0 + 128 > tlo && c0 < twf) || (hasm && c0 >= twf));
          if (need[j]) {
            c0r[j] = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(row0 + c0));
            c1r[j] = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(row1 + c0));
            if (Z) {
              c0i[j] = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(row0 + c0 + cv.plane));
              c1i[j] = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(row1 + c0 + cv.plane));
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (need[j]) {
            const double *up = us + 128 * (q0 + j) + 2 * lane;
            if (!Z) {
#pragma unroll
              for (int r = 0; r < NR; ++r) {
                const double2v u = *reinterpret_cast<const double2v *>(up + r * kChainSeg);
                acc[r] = __builtin_fma(c0r[j][0], u[0], acc[r]);
                acc[NR + r] = __builtin_fma(c1r[j][0], u[0], acc[NR + r]);
                acc[r] = __builtin_fma(c0r[j][1], u[1], acc[r]);
                acc[NR + r] = __builtin_fma(c1r[j][1], u[1], acc[NR + r]);
              }
            } else {
#pragma unroll
              for (int q = 0; q < NR / 2; ++q) {
                const double2v ur = *reinterpret_cast<const double2v *>(up + (2 * q) * kChainSeg);
                const double2v ui = *reinterpret_cast<const double2v *>(up + (2 * q + 1) * kChainSeg);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                  acc[2 * q] = __builtin_fma(c0r[j][e], ur[e], acc[2 * q]);
                  acc[2 * q + 1] = __builtin_fma(c0r[j][e], ui[e], acc[2 * q + 1]);
                  acc[NR + 2 * q] = __builtin_fma(c1r[j][e], ur[e], acc[NR + 2 * q]);
                  acc[NR + 2 * q + 1] = __builtin_fma(c1r[j][e], ui[e], acc[NR + 2 * q + 1]);
                  acc[2 * q] = __builtin_fma(-c0i[j][e], ui[e], acc[2 * q]);
                  acc[2 * q + 1] = __builtin_fma(c0i[j][e], ur[e], acc[2 * q + 1]);
                  acc[NR + 2 * q] = __builtin_fma(-c1i[j][e], ui[e], acc[NR + 2 * q]);
                  acc[NR + 2 * q + 1] = __builtin_fma(c1i[j][e], ur[e], acc[NR + 2 * q + 1]);
                }
              }
            }
          }
      }
    }
    if (!active) continue;
    wave_reduce_scatter<2 * NR>(acc);
    if (wave_reduce_owner<2 * NR>(lane)) {
      const int idx = wave_reduce_index<2 * NR>(lane, 0);
      const int rowi = idx / NR, r = idx % NR;
      if (rowi == 0 || two) {
        const int t = j0 + il + rowi;
        out[(size_t)r * fs + t] = acc[0];
        if (sink.x) sink_store<NR, Z>(sink, t, r, acc[0]);
      }
    }
  }
}

// res-free form of gemv64 for the bulk groups (untransposed systems, one right-hand side): rows [rb, rb + 64) of `in` lose
// sum_{t < nc} F(i, cb + t) vv[t][:].  Lane = row, the wavefronts split the columns; UB loads in flight per lane (a step's
// columns, 512 / 16 wavefronts, in one round trip where gemv64 takes four), the old values of `in` requested beside them.
template <int NR, bool Z, int UB>
__device__ __forceinline__ void chain_bulk_rows(const Band &b, int rb, int cb, int nc, const double (*vv)[NR], double *in,
                                                size_t stride, bool fwd, double *part) {
  constexpr int SWV = solve_waves<NR>();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
  const int i = rb + lane;
  const bool row = i >= 0 && i < b.n;
  double old = 0.0;
  const int ol = tid % 64, orr = tid / 64, oi = rb + ol;
  const bool mine = tid < 64 * NR && (fwd ? oi < b.n : oi >= 0);
  if (mine) old = in[(size_t)orr * stride + oi];
  double acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = 0.0;
  for (int t = wave; t < nc; t += UB * SWV) {
    double e[UB], ei[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int c = t + SWV * u;
      const bool ok = row && c < nc;
      const double *p = ok ? &b.at(i, cb + c) : b.AB;
      const double re = __builtin_nontemporal_load(p);
      const double im = Z ? __builtin_nontemporal_load(p + b.zoff) : 0.0;
      e[u] = ok ? re : 0.0;
      ei[u] = ok ? im : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) mac_cols<NR, Z>(acc, e[u], ei[u], &vv[min(t + SWV * u, nc - 1)][0]);
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) part[(wave * NR + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (mine) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < SWV; ++q) tot += part[(q * NR + orr) * 64 + ol];
    in[(size_t)orr * stride + oi] = old - tot;
  }
  __syncthreads();  // part is reused by the next block
}

template <int MODE, int NR, bool Z = false>
__global__ __launch_bounds__(solve_waves<NR>() * 64) __attribute__((amdgpu_waves_per_eu((NR == 1 && MODE <= 1) ? 8 : 4, 8))) void big_chain_kernel(const int *__restrict__ list,
                                                                           const int64_t *__restrict__ prefix, int count,
                                                                           int launch, TreeView t, chain::View cv,
                                                                           double *work, double *zbuf, int row_blocks,
                                                                           int lead_rows, double *x, size_t xstride) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  constexpr bool fwd = MODE == 0 || MODE == 2;  // (MODE 2, 3: the transposed systems, cv = their chains)
  constexpr int SWV = solve_waves<NR>();
  const BigFront b = big_front<NR>(list, prefix, count, t, work, zbuf);
  const int S = cv.span, n = b.np, K = (n + S - 1) / S;
  int nlead = 0;
  if (launch < K) {
    const int k = fwd ? launch : K - 1 - launch;
    nlead = (min(S, n - k * S) + lead_rows - 1) / lead_rows;
  }
  double *in = fwd ? b.W : b.Z, *out = fwd ? b.Z : b.W;
  if (b.blk < nlead) {
    const SolutionSink sink{(!fwd && x) ? x + (size_t)(Z ? 2 : 1) * (size_t)t.p0[b.f] : nullptr, xstride};
    if constexpr (NR > 2) {
      chain_lead_multi<NR, Z>(b, cv, fwd, fwd ? launch : K - 1 - launch, S, b.blk, lead_rows, in, out, out, sink, dsm);
    } else {
      chain_lead<NR, Z, 2, 4>(b, cv, fwd, fwd ? launch : K - 1 - launch, S, b.blk, in, out, out, sink, dsm);  // (wide levels: big_chain_wide_kernel)
    }
    return;
  }
  // bulk of the block solved by the launch before: rows beyond the NEXT block (that one has its update inside M)
  const int kb = fwd ? launch - 1 : K - launch;
  const int j0 = kb * S, jbs = min(S, n - j0), tile = b.blk - nlead;
  double(*v)[NR] = reinterpret_cast<double(*)[NR]>(dsm);                     // [S]
  double(*res)[NR] = reinterpret_cast<double(*)[NR]>(dsm + (size_t)S * NR);  // [NB] (8 / 16 columns)
  double *part = dsm + (size_t)(S + NB) * NR;                                // [SWV][NR][64]
  const Band band{const_cast<double *>(b.P), n, n, n, b.ldp + 1, 0, 0, b.pz};
  const int tid = threadIdx.x;
  const size_t stride = (size_t)b.fs;
  for (int o = tid; o < S * NR; o += SWV * 64) {
    const int tt = o % S, r = o / S;
    v[tt][r] = tt < jbs ? out[(size_t)r * stride + j0 + tt] : 0.0;
  }
  __syncthreads();
  for (int q = 0; q < row_blocks; ++q) {
    const int blk = tile * row_blocks + q;
    const int rb = fwd ? j0 + 2 * S + blk * 64 : j0 - S - (blk + 1) * 64;
    if (fwd ? rb >= n : rb + 64 <= 0) break;  // workgroup-uniform
    if constexpr (NR <= 2 && MODE <= 1) {
      chain_bulk_rows<NR, Z, 16>(band, rb, j0, jbs, v, in, stride, fwd, part);
    } else {
      // (the transposed products of gemv64 take at most SB * NB = 256 columns at a time with fewer than 16 columns of right-hand sides)
      constexpr int kMost = (MODE >= 2 && !tile_rows_on_lanes<NR>()) ? SB * NB : 1 << 30;
      for (int c0 = 0; c0 < jbs; c0 += kMost) {
        gemv64<MODE, NR, Z>(band, rb, j0 + c0, min(kMost, jbs - c0), v + c0, res, part);
        for (int o = tid; o < 64 * NR; o += SWV * 64) {
          const int l = o % 64, r = o / 64;
          const int i = rb + l;
          const bool ok = fwd ? (i < n) : (i >= 0);
          if (ok) in[(size_t)r * stride + i] -= res[l][r];
        }
        __syncthreads();  // res and part are reused by the next block
      }
    }
  }
}

// The levels whose fronts all have at most chain::kWidePivots pivots (one block per front, no bulk groups; thousands of
// fronts): the lead groups alone, 8 rows per wavefront (complex: 4) in two passes, one chunk of a row in flight per row —
// half the registers of big_chain_kernel, so that TWO workgroups share a CU: these workgroups live 7 us for 40 - 90 KB each,
// and how many of them a CU holds is what the launch takes.
template <int MODE, int NR, bool Z = false>
__global__ __launch_bounds__(solve_waves<NR>() * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void big_chain_wide_kernel(const int *__restrict__ list,
                                                                                   const int64_t *__restrict__ prefix,
                                                                                   int count, TreeView t, chain::View cv,
                                                                                   double *work, double *zbuf, double *x,
                                                                                   size_t xstride) {
  static_assert(NR <= 2, "one right-hand side; the others: chain_lead_multi");
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  constexpr bool fwd = MODE == 0 || MODE == 2;
  const BigFront b = big_front<NR>(list, prefix, count, t, work, zbuf);
  double *in = fwd ? b.W : b.Z, *out = fwd ? b.Z : b.W;
  const SolutionSink sink{(!fwd && x) ? x + (size_t)(Z ? 2 : 1) * (size_t)t.p0[b.f] : nullptr, xstride};
  chain_lead<NR, Z, (Z ? 2 : 4), 1, 2>(b, cv, fwd, 0, cv.span, b.blk, in, out, out, sink, dsm);
}

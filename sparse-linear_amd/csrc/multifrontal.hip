// multifrontal.hip — multifrontal LU without row interchanges on the nested-dissection tree of
// mf_symbolic.hpp, and its solves.  Serves umfpack_di_numeric / umfpack_di_solve for matrices whose
// band profile is too expensive (2-D / 3-D meshes): the work drops from O(n * band^2) to the
// O(n^2) (3-D) / O(n^1.5) (2-D) of nested dissection, the storage from n * band to the fronts.
//
// Every tree node has a dense column-major frontal matrix F = [pivots | boundary]^2.  Numeric
// factorisation, level by level from the leaves; the whole fronts of a level live in one of two
// transient HBM regions (a level and its children are alive together, levels alternate):
//   assemble   : the entries of P A P^T go to the front of their earlier-eliminated index; the Schur
//                complements of the children are added into their parent ("extend-add", children in
//                a fixed order, so the result is deterministic);
//   compact    : the children are done: their factor panels — P = the pivot columns (fs x np:
//                L11\U11, L21) and U = the pivot rows of the other columns (np x nb: U12) — move to
//                the factor arena, which is all that stays resident (a quarter of the whole fronts);
//   factor     : the first np columns/rows of F are eliminated by the blocked fp64-MFMA kernels of
//                dense_lu_kernels.hpp on a dense view (factor_loop with a pivot limit) — large fronts,
//                spread over side streams — or by one workgroup per front running the same device
//                code — the many small ones; the trailing nb x nb block is then the front's Schur
//                complement.
// A solve walks the tree up (L, or U^T) and down (U, or L^T) on the panels with a per-front work
// matrix (8 right-hand sides together); children hand their boundary part to the parent in the same
// fixed order.  No interchanges: used under the same rule as the band path (diagonal dominance, or
// a speculation that every solve checks — umfpack.hip).
#include <stdio.h>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "dense_lu_kernels.hpp"
#include "mf_symbolic.hpp"

namespace spl {

namespace {

struct DeviceTree {  // depends on the tree only (cached in mf::Tree::device_cache); foff / cboff: per factorisation
  int device = 0;
  DBuf<int> p0, np, nb, ld, parent, front_of, bidx, rel, depth, ldp, ldu;
  DBuf<int64_t> bptr, ioff, woff, roff, poff, uoff;
  std::shared_ptr<void> level_plan;  // mf::LevelPlan (below): the lists and grids of the levels, built by the first factorisation
};

struct TreeView {  // raw pointers for kernels
  const int *p0, *np, *nb, *ld, *parent, *front_of, *bidx, *rel, *depth, *ldp, *ldu;
  const int64_t *bptr, *foff, *ioff, *woff, *roff, *poff, *uoff;
  double *region[2];  // whole fronts of the even / odd tree levels (transient)
  double *arena;      // factor panels (resident)
  const int64_t *cboff;  // >= 0: the Schur complement of this front was saved to `cut` (nb x nb, ld nb)
  double *cut;
  int sym;  // 1: A == A^T, the fronts are factored as L D L^T (Band::sym)
  // zm = 2: COMPLEX fronts (native `zi` factorisation, round 3).  Every array of doubles holds two planes per object —
  // real parts, then imaginary parts — at twice the real offset: a front at region + 2 foff[f] with its planes
  // fplane(f) apart, its P panel at arena + 2 poff[f] (planes ldp np apart), U at arena + 2 uoff[f] (ldu nb), a saved
  // Schur complement at cut + 2 cboff[f] (nb nb), its inverse blocks at invs + 2 ioff[f] (kInvBlockZ per block).
  // Indices, leading dimensions and the tree are those of a real matrix with the same pattern.  zm = 1: real.
  int zm;
  int piv;  // 1: threshold pivoting inside the diagonal blocks (Band::piv)
  const double *rscale;  // ... with these row scales (per unknown of the tree, new ordering; nullptr: none)

  __device__ __forceinline__ double *front(int f) const { return region[depth[f] & 1] + (int64_t)zm * foff[f]; }
  __device__ __forceinline__ int64_t fplane(int f) const {  // doubles of one plane of the whole front (make_plan)
    const int64_t fs = np[f] + nb[f];
    return ((int64_t)ld[f] * (fs > 1 ? fs : 1) + 15) / 16 * 16;
  }
};

template <typename T>
void upload_vec(DBuf<T> &d, const std::vector<T> &h, hipStream_t s) {
  d.alloc(h.size());
  if (!h.empty()) SPL_HIP(hipMemcpyAsync(d.get(), h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

// local position of global (new) index g in front f: pivots first, then the sorted boundary
__device__ __forceinline__ int local_pos(const TreeView &t, int f, int g) {
  const int p0 = t.p0[f], np = t.np[f];
  if (g < p0 + np) return g - p0;
  const int *b = t.bidx + t.bptr[f];
  int lo = 0, hi = t.nb[f] - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (b[mid] < g) lo = mid + 1; else hi = mid;
  }
  return np + lo;
}

__device__ __forceinline__ int item_of_tile(const int64_t *__restrict__ prefix, int count, int64_t flat);

// Entries of A -> the fronts listed (those of one tree level): 8 lanes per pivot g of a front take
// column g of P A P^T from the CSC arrays of A (the rows at or below the diagonal) and row g from
// the CSR arrays (the columns right of it) — every entry lands in the front of its
// earlier-eliminated index exactly once, and a level only reads the columns and rows of its own
// pivots.  Flat grid: prefix = groups of 32 pivots before each front.
// Z: the arrays are those of the real embedding E of a complex matrix (umfpack_zi.hip: interleaved unknowns, block
// (r, j) = [[re, -im], [im, re]]) and perm / inv its expanded ordering; column 2j of E holds column j of the complex
// matrix as (re, im) pairs, row 2j of E holds row j as (re, -im) pairs.
template <bool Z>
__global__ __launch_bounds__(256) void assemble_kernel(const int *__restrict__ list,
                                                       const int64_t *__restrict__ prefix, int count, TreeView t,
                                                       const int *__restrict__ perm, const int *__restrict__ inv,
                                                       const int *__restrict__ Ap, const int *__restrict__ Ai,
                                                       const double *__restrict__ Ax, const int *__restrict__ Rp,
                                                       const int *__restrict__ Rj, const double *__restrict__ Rx) {
  const int64_t flat = prefix[0] + blockIdx.x;
  const int fi = item_of_tile(prefix, count, flat);
  const int f = list[fi];
  const int lp = (int)(flat - prefix[fi]) * 32 + (int)(threadIdx.x >> 3), part = threadIdx.x & 7;
  if (lp >= t.np[f]) return;
  const int g = t.p0[f] + lp;
  double *F = t.front(f);
  const int64_t ld = t.ld[f];
  if (Z) {
    const int j2 = perm[2 * g];  // = 2 j
    const int64_t z = t.fplane(f);
    for (int p = Ap[j2] + 2 * part; p < Ap[j2 + 1]; p += 16) {
      const int gi = inv[Ai[p]] >> 1;
      if (gi >= g) {
        double *dst = F + (int64_t)local_pos(t, f, gi) + (int64_t)lp * ld;
        dst[0] = Ax[p];
        dst[z] = Ax[p + 1];
      }
    }
    for (int p = Rp[j2] + 2 * part; p < Rp[j2 + 1]; p += 16) {
      const int gk = inv[Rj[p]] >> 1;
      if (gk > g) {
        double *dst = F + (int64_t)lp + (int64_t)local_pos(t, f, gk) * ld;
        dst[0] = Rx[p];
        dst[z] = -Rx[p + 1];
      }
    }
    return;
  }
  const int j = perm[g];
  for (int p = Ap[j] + part; p < Ap[j + 1]; p += 8) {
    const int gi = inv[Ai[p]];
    if (gi >= g) F[(int64_t)local_pos(t, f, gi) + (int64_t)lp * ld] = Ax[p];
  }
  for (int p = Rp[j] + part; p < Rp[j + 1]; p += 8) {
    const int gk = inv[Rj[p]];
    if (gk > g) F[(int64_t)lp + (int64_t)local_pos(t, f, gk) * ld] = Rx[p];
  }
}

// rel[roff[c] + k] = position of the k-th boundary index of front c inside its parent's front
__global__ __launch_bounds__(256) void rel_kernel(int nfronts, TreeView t, int *__restrict__ rel) {
  const int c = blockIdx.x;
  const int p = t.parent[c];
  if (p < 0) return;
  const int *b = t.bidx + t.bptr[c];
  for (int k = threadIdx.x; k < t.nb[c]; k += blockDim.x) rel[t.roff[c] + k] = local_pos(t, p, b[k]);
}

// Tile-to-item lookup of the flat grids below: items [0, count) own the tiles
// [prefix[i], prefix[i+1]); returns the item of tile `flat` (prefix[0] <= flat < prefix[count]).
// Round 5: a level of the lower tree lists thousands of fronts, and a plain bisection is 10 - 12 DEPENDENT loads before a
// workgroup knows which front it works for — a third of the lifetime of the short workgroups of the solves.  Here the 64
// lanes of a wavefront probe 64 evenly spaced items at once and a ballot picks the bracket: two round trips for up to
// 4 096 items (every wavefront of the workgroup searches for itself: no barrier).  Called by all lanes, at kernel entry.
__device__ __forceinline__ int item_of_tile(const int64_t *__restrict__ prefix, int count, int64_t flat) {
  int lo = 0, hi = count;  // the answer is the largest i in [lo, hi) with prefix[i] <= flat
  if (count > 16) {
    const int lane = threadIdx.x & 63;
    while (hi - lo > 4) {
      const int step = (hi - lo + 63) >> 6;
      const int idx = lo + lane * step;
      const bool ok = idx < hi && prefix[idx] <= flat;  // (monotone over the lanes; lane 0 holds)
      const int k = __popcll(__ballot(ok)) - 1;
      lo += (k > 0 ? k : 0) * step;
      hi = lo + step < hi ? lo + step : hi;
    }
  }
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= flat) lo = mid; else hi = mid;
  }
  return lo;
}

constexpr int kTileCols = 16;  // a workgroup moves 64 rows x 16 columns: 4 columns per thread

// parent += Schur complement of the listed children.  Flat 1-D grid over the 64 x 16 tiles of all
// children (prefix = tiles before each child): no workgroup is launched for nothing, whatever the
// mix of sizes.  Lanes run down the rows: contiguous in the child, nearly so in the parent.
template <bool Z>
__global__ __launch_bounds__(256) void extend_add_kernel(const int *__restrict__ children,
                                                         const int64_t *__restrict__ prefix, int count,
                                                         TreeView t) {
  const int64_t flat = prefix[0] + blockIdx.x;
  const int ci = item_of_tile(prefix, count, flat);
  const int c = children[ci];
  const int nb = t.nb[c];
  const int ntr = (nb + 63) >> 6;
  const int64_t tile = flat - prefix[ci];
  const int r = (int)(tile % ntr) * 64 + (threadIdx.x & 63);
  const int cc0 = (int)(tile / ntr) * kTileCols + (threadIdx.x >> 6) * 4;
  if (r >= nb) return;
  const int p = t.parent[c], npc = t.np[c];
  const int *rel = t.rel + t.roff[c];
  const int64_t saved = t.cboff[c];
  double *dst = t.front(p) + rel[r];
  const int64_t ldp = t.ld[p];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int cc = cc0 + u;
    if (cc >= nb) break;
    if (t.sym && cc > r) break;  // symmetric fronts: the child's lower triangle only (what lies above its diagonal tiles is stale)
    const double *src = saved >= 0 ? t.cut + (Z ? 2 : 1) * saved + (int64_t)r + (int64_t)cc * nb
                                   : t.front(c) + (int64_t)(npc + r) + (int64_t)(npc + cc) * t.ld[c];
    const double v = src[0];
    dst[(int64_t)rel[cc] * ldp] += v;
    if (Z) {
      const int64_t zc = saved >= 0 ? (int64_t)nb * nb : t.fplane(c), zp = t.fplane(p);
      const double vi = src[zc];
      dst[(int64_t)rel[cc] * ldp + zp] += vi;
      if (t.sym && cc != r && (rel[cc] >> 6) == (rel[r] >> 6)) {
        double *m = t.front(p) + (int64_t)rel[cc] + (int64_t)rel[r] * ldp;
        m[0] += v;
        m[zp] += vi;
      }
      continue;
    }
    // ... mirrored into the parent's upper triangle where the parent reads it: the 64 x 64 blocks on its diagonal
    // (diagonal blocks are factored, and diagonal tiles of the trailing update computed, whole; every other tile above
    // the diagonal is written by the triangular solves, U12 = D L21^T, before anything reads it).  rel is increasing,
    // so (rel[cc], rel[r]) lies above the diagonal and is written by this thread only.
    if (t.sym && cc != r && (rel[cc] >> 6) == (rel[r] >> 6)) t.front(p)[(int64_t)rel[cc] + (int64_t)rel[r] * ldp] += v;
  }
}

// Schur complement of a finished front -> the cut buffer (its parent is assembled much later)
template <bool Z>
__global__ __launch_bounds__(256) void save_cb_kernel(int f, TreeView t) {
  const int np = t.np[f], nb = t.nb[f];
  const int ntr = (nb + 63) >> 6;
  const int r = (int)(blockIdx.x % ntr) * 64 + (threadIdx.x & 63), cc = (int)(blockIdx.x / ntr) * 4 + (threadIdx.x >> 6);
  if (r >= nb || cc >= nb) return;
  const double *src = t.front(f) + (int64_t)(np + r) + (int64_t)(np + cc) * t.ld[f];
  double *dst = t.cut + (Z ? 2 : 1) * t.cboff[f] + (int64_t)r + (int64_t)cc * nb;
  dst[0] = src[0];
  if (Z) dst[(int64_t)nb * nb] = src[t.fplane(f)];
}

// factor panels of the listed (finished) fronts -> arena, flat grid over the 64 x 16 tiles of the
// panels: which = 0: P = columns [0, np) (fs rows); which = 1: U = rows [0, np) of the other columns
template <bool Z>
__global__ __launch_bounds__(256) void compact_kernel(const int *__restrict__ list,
                                                      const int64_t *__restrict__ prefix, int count, TreeView t,
                                                      int which) {
  const int64_t flat = prefix[0] + blockIdx.x;
  const int fi = item_of_tile(prefix, count, flat);
  const int f = list[fi];
  const int np = t.np[f], nb = t.nb[f], fs = np + nb;
  const int rows = which == 0 ? fs : np, cols = which == 0 ? np : nb;
  const int ntr = (rows + 63) >> 6;
  const int64_t tile = flat - prefix[fi];
  const int i = (int)(tile % ntr) * 64 + (threadIdx.x & 63);
  const int j0 = (int)(tile / ntr) * kTileCols + (threadIdx.x >> 6) * 4;
  if (i >= rows) return;
  const double *src = t.front(f) + i;
  const int64_t ld = t.ld[f];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = j0 + u;
    if (j >= cols) break;
    constexpr int ZM = Z ? 2 : 1;
    const double *from = which == 0 ? src + (int64_t)j * ld : src + (int64_t)(np + j) * ld;
    double *to = which == 0 ? t.arena + ZM * t.poff[f] + (int64_t)i + (int64_t)j * t.ldp[f]
                            : t.arena + ZM * t.uoff[f] + (int64_t)i + (int64_t)j * t.ldu[f];
    to[0] = from[0];
    if (Z) to[which == 0 ? (int64_t)t.ldp[f] * np : (int64_t)t.ldu[f] * nb] = from[t.fplane(f)];
  }
}

// small fronts of a tree level: one workgroup per front runs the whole partial factorisation
template <bool Z>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void front_factor_kernel(const int *__restrict__ list, TreeView t,
                                                           double *__restrict__ invs, int *__restrict__ singular) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  const int f = list[blockIdx.x];
  const int np = t.np[f];
  if (np == 0) return;
  const int fs = np + t.nb[f];
  if (Z) {
    const Band b{t.front(f), fs, fs, fs, t.ld[f] + 1, 0, t.sym, (size_t)t.fplane(f), t.piv, t.rscale ? t.rscale + t.p0[f] : nullptr};
    front_factor_by_workgroup_z(b, np, invs + 2 * t.ioff[f], singular, dsm);
  } else {
    const Band b{t.front(f), fs, fs, fs, t.ld[f] + 1, 0, t.sym, 0, t.piv, t.rscale ? t.rscale + t.p0[f] : nullptr};
    front_factor_by_workgroup(b, np, invs + t.ioff[f], singular, dsm);
  }
}
constexpr size_t kFrontLdsZ = kDiagLdsZ > kTrsmLdsZ ? kDiagLdsZ : kTrsmLdsZ;

// Medium fronts of a tree level, all together, one block step at a time: a front of a few
// thousand rows is a chain of launches that each fill a fraction of the chip, and a level holds tens
// to hundreds of them.  Step s of every listed front that still has pivots left goes into ONE
// launch per phase (flat grids: prefix = tiles before each front at this step, computed on the host):
// the diagonal blocks first (mid_diag_kernel, step 0 only), then per step the panel solves and the
// K = 64 update whose tile (0,0) factors the next diagonal block on the way (the look-ahead).
__device__ __forceinline__ Band mid_front(const TreeView &t, int f) {
  const int fs = t.np[f] + t.nb[f];
  return Band{t.front(f), fs, fs, fs, t.ld[f] + 1, 0, t.sym, t.zm == 2 ? (size_t)t.fplane(f) : 0, t.piv,
              t.rscale ? t.rscale + t.p0[f] : nullptr};
}
__device__ __forceinline__ double *mid_slot(const TreeView &t, double *invs, int f, int j0) {
  return invs + (int64_t)t.zm * t.ioff[f] + (int64_t)(j0 / NB) * (t.zm == 2 ? kInvBlockZ : 2 * NB * NB);
}

template <bool Z>
__global__ __launch_bounds__(256) void mid_diag_kernel(const int *__restrict__ list, TreeView t,
                                                       double *__restrict__ invs, int *__restrict__ singular) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  const int f = list[blockIdx.x];
  const Band b = mid_front(t, f);
  const int jb = min(NB, t.np[f]);
  if (Z) {
    double(*Dr)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
    double(*Di)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm + NB * LDP);
    double(*lcr)[NB] = reinterpret_cast<double(*)[NB]>(dsm + 2 * NB * LDP);
    double(*lci)[NB] = reinterpret_cast<double(*)[NB]>(dsm + 2 * NB * LDP + 2 * NB);
    load_diag_z(b, 0, jb, Dr, Di);
    __syncthreads();
    double *slot = mid_slot(t, invs, f, 0);
    diag_block_factor_z(b, 0, jb, Dr, Di, lcr, lci, singular, slot, slot + 2 * NB * NB);
    return;
  }
  double(*D)[LDP] = reinterpret_cast<double(*)[LDP]>(dsm);
  double(*lcol)[NB] = reinterpret_cast<double(*)[NB]>(dsm + NB * LDP);
  const int tr = threadIdx.x & 63, tc = threadIdx.x >> 6;
  for (int c = tc; c < NB; c += 4) D[tr][c] = (tr < jb && c < jb) ? b.get(tr, c) : (tr == c ? 1.0 : 0.0);
  __syncthreads();
  double *slot = mid_slot(t, invs, f, 0);
  diag_block_factor(b, 0, jb, D, lcol, singular, slot, slot + NB * NB);
}

template <bool Z>
__global__ __launch_bounds__(256) void mid_trsm_kernel(const int *__restrict__ list,
                                                       const int64_t *__restrict__ prefix, int count, int step,
                                                       TreeView t, double *__restrict__ invs) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  const int64_t flat = prefix[0] + blockIdx.x;
  const int fi = item_of_tile(prefix, count, flat);
  const int f = list[fi];
  const Band b = mid_front(t, f);
  const int j0 = step * NB, jb = min(NB, t.np[f] - j0), rest = b.n - (j0 + jb);
  const double *slot = mid_slot(t, invs, f, j0);
  if (Z) trsm_tile_z(b, j0, jb, rest, rest, slot, slot + 2 * NB * NB, (int)(flat - prefix[fi]), dsm);
  else trsm_tile(b, j0, jb, rest, rest, slot, slot + NB * NB, (int)(flat - prefix[fi]), dsm);
}

template <bool Z>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(Z ? 1 : 3, 8))) void mid_update_kernel(const int *__restrict__ list,
                                                         const int64_t *__restrict__ prefix, int count, int step,
                                                         TreeView t, double *__restrict__ invs,
                                                         int *__restrict__ singular, int paired) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  // The first `count` workgroups take tile (0,0) of one front each — the tile that goes on to factor the next diagonal
  // block, 30 - 60 us of mostly one wavefront: started last, behind a front's other tiles, the thousand of them of a
  // level were the tail of every launch.  The other workgroups take the tiles in order and skip that one.
  int fi;
  int64_t flat;
  if ((int)blockIdx.x < count) {
    fi = (int)blockIdx.x;
    if (prefix[fi + 1] == prefix[fi]) return;  // this front is done with its pivots
    flat = prefix[fi];
  } else {
    flat = prefix[0] + ((int64_t)blockIdx.x - count);
    fi = item_of_tile(prefix, count, flat);
    if (flat == prefix[fi]) return;
  }
  const int f = list[fi];
  const Band b = mid_front(t, f);
  const int np = t.np[f], j0 = step * NB, jb = min(NB, np - j0), origin = j0 + jb;
  const int ntile = (b.n - origin + 63) / 64;
  const int tile = (int)(flat - prefix[fi]);
  // Steps in pairs (round 4; `paired`): the counters showed these passes moving 86 GB per 100^3 factorisation at K = 64
  // (8 flop per byte of window traffic).  An even step of a front that has another pivot block after it only brings
  // that block's column and row up to date (L-shaped pass) and factors its diagonal block; the odd step that follows
  // updates the rest of the window with both panels at once, K = 128: every second pass over the window is gone.
  // A front whose next pivot block is missing or short takes the whole window at once, as before.
  int kb = j0, klen = jb, lshape = 0, tx = tile % ntile, ty = tile / ntile;
  // (only full blocks pair: the 64-wide tiles of an L-shaped pass over a shorter second block would reach into the window
  // the odd step updates with both panels)
  if (paired) {
    if (step & 1) {
      if (jb == NB) {
        kb = j0 - NB;
        klen = 2 * NB;
      }
    } else if (np >= origin + NB) {
      lshape = 1;
      tx = tile < ntile ? tile : 0;
      ty = tile < ntile ? 0 : tile - ntile + 1;
    }
  }
  const Region g{origin, b.n, origin, b.n, kb, klen, lshape, ntile, np};
  double *next = mid_slot(t, invs, f, origin);  // only written when origin is a pivot block
  if (Z) update_tile_z<true>(b, g, tx, ty, singular, next, next + 2 * NB * NB, dsm);
  else update_tile<true>(b, g, tx, ty, singular, next, next + NB * NB, dsm);
}

constexpr int kSmallFront = 128;   // fronts up to this size are factored by one workgroup each
constexpr int kMidFront = 4096;    // ... up to this size in lockstep with the others of their level
constexpr int kStreams = 8;        // larger fronts of a level are spread over this many streams

// ---- solves ---------------------------------------------------------------------------------------
// NR right-hand sides travel through the tree together: every front has a work matrix W (fs x NR,
// column-major), column r of the right-hand sides / solution is c + r * stride.
constexpr int kSolveThreads = 1024;  // one workgroup per front, 16 wavefronts for the coupling loops ...
template <int NR>
constexpr int solve_threads() { return (NR >= 16 || NR <= 2) ? 512 : kSolveThreads; }  // ... 8 with 16 columns (256 registers per thread) and with one or two (levels of thousands of small fronts: four workgroups per CU instead of two)
constexpr int kSolveRowBlocks = 4;   // blocks of 64 rows per workgroup in the lockstep solve steps
// fronts above this size are solved by many workgroups, in lockstep (round 5: 256 -> 128 — with the pivot blocks as chains of
// matrix-vector products the many-workgroup path is the faster one for all but the leaves: 100^3 11.4 -> 10.3 ms per solve,
// 64^3 4.9 -> 4.1, 40^3 2.45 -> 2.06, complex 100^3 18.6 -> 17.6; 64 gives the same)
constexpr int kBigSolve = 128;

// The panels of a front as the solves see them.  M is the triangular system a front contributes:
// M(i, j) = F(i, j), or F(j, i) for the transposed systems, where F(i, j) lives in P for j < np and in
// U for j >= np (then i < np).
struct Panels {
  const double *P, *U;
  int np, nb, fs, ldp, ldu;
  size_t pz, uz;  // complex fronts: the imaginary planes of P and U lie this many doubles further (0: real)
  // column j of F, rows from 0 (valid rows: fs for j < np, np otherwise)
  __device__ __forceinline__ const double *col(int j) const {
    return j < np ? P + (size_t)j * ldp : U + (size_t)(j - np) * ldu;
  }
  __device__ __forceinline__ int col_stride(int j) const { return j < np ? ldp : ldu; }
  __device__ __forceinline__ size_t zcol(int j) const { return j < np ? pz : uz; }
};

__device__ __forceinline__ Panels panels_of(const TreeView &t, int f) {
  const int np = t.np[f], nb = t.nb[f];
  if (t.zm == 2)
    return Panels{t.arena + 2 * t.poff[f], t.arena + 2 * t.uoff[f], np, nb, np + nb, t.ldp[f], t.ldu[f],
                  (size_t)t.ldp[f] * (size_t)np, (size_t)t.ldu[f] * (size_t)nb};
  return Panels{t.arena + t.poff[f], t.arena + t.uoff[f], np, nb, np + nb, t.ldp[f], t.ldu[f], 0, 0};
}

// v = T w for the 64 x 64 inverse diagonal block (column-major), T = inv or inv^T, NR columns; the
// first 256 threads of the workgroup, 4 per row
// (Z: the inverse in two planes NB * NB apart; TRANS is then the CONJUGATE transpose)
template <bool TRANS, int NR, bool Z = false>
__device__ __forceinline__ void apply_inverse_block(const double *__restrict__ inv, const double (*w)[NR],
                                                    double (*v)[NR]) {
  if (threadIdx.x >= 256) return;
  const int l = threadIdx.x >> 2, q = threadIdx.x & 3;
  double acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = 0.0;
#pragma unroll 4
  for (int u = 0; u < NB / 4; ++u) {
    const int tt = q + 4 * u;
    const int at = TRANS ? tt + l * NB : l + tt * NB;
    const double e = inv[at];
    const double ei = Z ? (TRANS ? -inv[NB * NB + at] : inv[NB * NB + at]) : 0.0;
    mac_cols<NR, Z>(acc, e, ei, &w[tt][0]);
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    acc[r] += __shfl_xor(acc[r], 1, 64);
    acc[r] += __shfl_xor(acc[r], 2, 64);
    if (q == 0) v[l][r] = acc[r];
  }
}

// W[i][:] -= sum_{tt < jb} M(i, c0 + tt) v[tt][:] for i in [ilo, ihi), by all threads of the workgroup;
// a matrix entry is loaded once for all NR columns.  The jb columns c0 .. c0+jb-1 lie on one side of
// np.  Untransposed the panels run down i (thread = row, 8 loads in flight); transposed they run
// along tt (a wavefront per row, lanes along tt, butterfly sums).  W is fs x NR column-major.
template <bool TRANS, int NR, bool Z = false>
__device__ __forceinline__ void couple_block(const Panels &fr, int ilo, int ihi, int c0, int jb,
                                             const double (*v)[NR], double *W) {
  const int fs = fr.fs;
  if (!TRANS) {
    // A wavefront takes 16 rows, its four quarters a quarter of the columns each (tt = g, g + 4, ...): a front of this
    // class has at most 256 rows, so with a thread per row three quarters of the workgroup idled while every row walked
    // its 64 columns eight loads at a time (round 3: two rounds of loads instead of eight, and all 16 wavefronts
    // multiply; the quarters meet by two shuffles per column of W)
    const double *base = fr.col(c0);
    const size_t cs = (size_t)fr.col_stride(c0), zp = fr.zcol(c0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4, rl = lane & 15;
    const int nu = (jb - g + 3) / 4;  // columns of this quarter
    for (int ib = ilo + wave * 16; ib < ihi; ib += nw * 16) {
      const int i = ib + rl;
      double acc[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r] = 0.0;
      if (i < ihi) {
        const double *row = base + i + (size_t)g * cs;
        int u = 0;
        for (; u + 8 <= nu; u += 8) {
          double e[8], ei[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            e[q] = row[(size_t)(4 * (u + q)) * cs];
            ei[q] = Z ? row[(size_t)(4 * (u + q)) * cs + zp] : 0.0;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) mac_cols<NR, Z>(acc, e[q], ei[q], &v[g + 4 * (u + q)][0]);
        }
        for (; u < nu; ++u) {
          const double e = row[(size_t)(4 * u) * cs];
          const double ei = Z ? row[(size_t)(4 * u) * cs + zp] : 0.0;
          mac_cols<NR, Z>(acc, e, ei, &v[g + 4 * u][0]);
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        acc[r] += __shfl_xor(acc[r], 16, 64);
        acc[r] += __shfl_xor(acc[r], 32, 64);
        if (g == 0 && i < ihi) W[(size_t)r * fs + i] -= acc[r];
      }
    }
  } else if (NR >= 8) {
    // M(i, c0 + tt) = F(c0 + tt, i) (Z: its conjugate), with the rows i on the lanes as above: a lane walks rows
    // c0 + g, c0 + g + 4, ... of ITS column of F (the four quarters of a row read neighbouring doubles; a line is used
    // up over four trips), and the sums need two shuffles per column of W instead of a reduction over the 64 lanes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int g = lane >> 4, rl = lane & 15;
    const int nu = (jb - g + 3) / 4;
    for (int ib = ilo + wave * 16; ib < ihi; ib += nw * 16) {
      const int i = ib + rl;
      double acc[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[r] = 0.0;
      if (i < ihi) {
        const double *colp = fr.col(i) + c0 + g;
        const size_t zp = fr.zcol(i);
        int u = 0;
        for (; u + 8 <= nu; u += 8) {
          double e[8], ei[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            e[q] = colp[4 * (u + q)];
            ei[q] = Z ? -colp[4 * (u + q) + zp] : 0.0;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) mac_cols<NR, Z>(acc, e[q], ei[q], &v[g + 4 * (u + q)][0]);
        }
        for (; u < nu; ++u) {
          const double e = colp[4 * u];
          const double ei = Z ? -colp[4 * u + zp] : 0.0;
          mac_cols<NR, Z>(acc, e, ei, &v[g + 4 * u][0]);
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        acc[r] += __shfl_xor(acc[r], 16, 64);
        acc[r] += __shfl_xor(acc[r], 32, 64);
        if (g == 0 && i < ihi) W[(size_t)r * fs + i] -= acc[r];
      }
    }
  } else {
    // M(i, c0 + tt) = F(c0 + tt, i) (Z: its conjugate): rows c0 .. of column i of F, contiguous
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int RW = NR <= 2 ? 8 : 2;  // rows per wavefront and trip (one real or one complex right-hand side: more loads in flight)
    for (int i0 = ilo + wave * RW; i0 < ihi; i0 += nw * RW) {
      double e[RW], ei[RW];
#pragma unroll
      for (int u = 0; u < RW; ++u) {
        const bool in = i0 + u < ihi && lane < jb;
        const double *src = in ? fr.col(i0 + u) + c0 + lane : nullptr;
        e[u] = in ? src[0] : 0.0;
        ei[u] = (Z && in) ? -src[fr.zcol(i0 + u)] : 0.0;
      }
      double part[RW * NR];
#pragma unroll
      for (int o = 0; o < RW * NR; ++o) part[o] = 0.0;
      if (lane < jb) {
#pragma unroll
        for (int u = 0; u < RW; ++u) mac_cols<NR, Z>(*reinterpret_cast<double(*)[NR]>(&part[u * NR]), e[u], ei[u], &v[lane][0]);
      }
      wave_reduce_scatter<RW * NR>(part);  // (recursive halving: about one shuffle per sum instead of six)
      if (wave_reduce_owner<RW * NR>(lane)) {
        const int idx = wave_reduce_index<RW * NR>(lane, 0), u = idx / NR, r = idx % NR;  // (RW * NR <= 64)
        if (i0 + u < ihi) W[(size_t)r * fs + i0 + u] -= part[0];
      }
    }
  }
}

// W = [rhs at the pivots | 0] of front blockIdx.x: all fronts in one launch before the walk (every front has its own W;
// round 3: a launch per level cost 27 x 36 us per walk at 80^3 — each level waited for its largest front), gridDim.y
// workgroups per front
// (Z: column r of W is the real (r even) or imaginary part of the packed complex right-hand side r / 2)
template <int NR, bool Z = false>
__global__ __launch_bounds__(256) void solve_init_kernel(TreeView t, const double *__restrict__ c, size_t stride,
                                                         double *__restrict__ work) {
  const int f = blockIdx.x;
  const int np = t.np[f], fs = np + t.nb[f];
  double *W = work + (size_t)t.woff[f] * NR;
  for (int o = blockIdx.y * blockDim.x + threadIdx.x; o < fs * NR; o += gridDim.y * blockDim.x) {
    const int i = o % fs, r = o / fs;
    if (Z) W[o] = i < np ? c[(size_t)(r >> 1) * stride + 2 * (size_t)(t.p0[f] + i) + (r & 1)] : 0.0;
    else W[o] = i < np ? c[(size_t)r * stride + t.p0[f] + i] : 0.0;
  }
}

// W(parent)[rel] += boundary part of W(child), for the listed children: one launch per child slot (two children may add
// into the same row of their parent, so the slots take turns).  Round 5 measured ONE launch for both slots — a workgroup
// per parent and range of its rows, child 1 behind a barrier after child 0, same bits —: slower, 15.42 against 15.17 ms
// per solve at 100^3 and 10.84 against 10.55 at complex 64^3 (same box, alternating runs).
template <int NR>
__global__ __launch_bounds__(256) void solve_gather_kernel(const int *__restrict__ children, TreeView t,
                                                                double *__restrict__ work) {
  const int c = children[blockIdx.x];
  const int p = t.parent[c], npc = t.np[c], nbc = t.nb[c];
  const int fsc = npc + nbc, fsp = t.np[p] + t.nb[p];
  const int *rel = t.rel + t.roff[c];
  const double *Wc = work + (size_t)t.woff[c] * NR;
  double *Wp = work + (size_t)t.woff[p] * NR;
  for (int o = blockIdx.y * blockDim.x + threadIdx.x; o < nbc * NR; o += gridDim.y * blockDim.x) {
    const int k = o % nbc, r = o / nbc;
    Wp[(size_t)r * fsp + rel[k]] += Wc[(size_t)r * fsc + npc + k];
  }
}

// forward elimination inside a front: y = M11^-1 W[0:np) block by block (stored inverses of the diagonal
// blocks), every later entry of W loses its coupling with the block just solved.  One workgroup; w, v: its LDS.
template <bool TRANS, int NR, bool Z = false>
__device__ __forceinline__ void front_forward(const TreeView &t, int f, const double *__restrict__ invs,
                                              double *__restrict__ work, double (*w)[NR], double (*v)[NR]) {
  const Panels fr = panels_of(t, f);
  const int np = fr.np, fs = fr.fs;
  double *W = work + (size_t)t.woff[f] * NR;
  for (int j0 = 0; j0 < np; j0 += NB) {
    const int jb = min(NB, np - j0);
    for (int o = threadIdx.x; o < NB * NR; o += blockDim.x) {
      const int l = o % NB, r = o / NB;
      w[l][r] = l < jb ? W[(size_t)r * fs + j0 + l] : 0.0;
    }
    __syncthreads();
    // forward: L (unit lower) or U^T -> inverse of L11, or of U11 transposed
    constexpr size_t blk = Z ? (size_t)kInvBlockZ : (size_t)(2 * NB * NB);
    const double *inv = invs + (Z ? 2 : 1) * t.ioff[f] + (size_t)(j0 / NB) * blk + (TRANS ? blk / 2 : 0);
    apply_inverse_block<TRANS, NR, Z>(inv, w, v);
    __syncthreads();
    for (int o = threadIdx.x; o < jb * NR; o += blockDim.x) {
      const int l = o % jb, r = o / jb;
      W[(size_t)r * fs + j0 + l] = v[l][r];
    }
    couple_block<TRANS, NR, Z>(fr, j0 + jb, fs, j0, jb, v, W);
    __syncthreads();
  }
}

template <bool TRANS, int NR, bool Z = false>
__global__ __launch_bounds__(solve_threads<NR>()) void solve_forward_kernel(const int *__restrict__ list, TreeView t,
                                                                      const double *__restrict__ invs,
                                                                      double *__restrict__ work) {
  __shared__ double w[NB][NR], v[NB][NR];
  front_forward<TRANS, NR, Z>(t, list[blockIdx.x], invs, work, w, v);
}

// back substitution inside a front: x_piv = M11^-1 (y - M12 x_bnd), x_bnd read from the solution of
// the ancestors; writes the pivots' part of the solution
template <bool TRANS, int NR, bool Z = false>
__device__ __forceinline__ void front_backward(const TreeView &t, int f, const double *__restrict__ invs,
                                               double *__restrict__ work, double *__restrict__ x, size_t stride,
                                               double (*w)[NR], double (*v)[NR]) {
  const Panels fr = panels_of(t, f);
  const int np = fr.np, nb = fr.nb, p0 = t.p0[f], fs = fr.fs;
  double *W = work + (size_t)t.woff[f] * NR;
  const int *b = t.bidx + t.bptr[f];
  for (int k0 = 0; k0 < nb; k0 += NB) {  // boundary columns in strips of 64, through LDS
    const int kb = min(NB, nb - k0);
    for (int o = threadIdx.x; o < NB * NR; o += blockDim.x) {
      const int l = o % NB, r = o / NB;
      if (Z) v[l][r] = l < kb ? x[(size_t)(r >> 1) * stride + 2 * (size_t)b[k0 + l] + (r & 1)] : 0.0;
      else v[l][r] = l < kb ? x[(size_t)r * stride + b[k0 + l]] : 0.0;
    }
    __syncthreads();
    couple_block<TRANS, NR, Z>(fr, 0, np, np + k0, kb, v, W);
    __syncthreads();
  }
  const int nblk = (np + NB - 1) / NB;
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int j0 = blk * NB, jb = min(NB, np - j0);
    for (int o = threadIdx.x; o < NB * NR; o += blockDim.x) {
      const int l = o % NB, r = o / NB;
      w[l][r] = l < jb ? W[(size_t)r * fs + j0 + l] : 0.0;
    }
    __syncthreads();
    // backward: U or L^T -> inverse of U11, or of L11 transposed
    constexpr size_t ib = Z ? (size_t)kInvBlockZ : (size_t)(2 * NB * NB);
    const double *inv = invs + (Z ? 2 : 1) * t.ioff[f] + (size_t)blk * ib + (TRANS ? 0 : ib / 2);
    apply_inverse_block<TRANS, NR, Z>(inv, w, v);
    __syncthreads();
    for (int o = threadIdx.x; o < jb * NR; o += blockDim.x) {
      const int l = o % jb, r = o / jb;
      if (Z) x[(size_t)(r >> 1) * stride + 2 * (size_t)(p0 + j0 + l) + (r & 1)] = v[l][r];
      else x[(size_t)r * stride + p0 + j0 + l] = v[l][r];
    }
    couple_block<TRANS, NR, Z>(fr, 0, j0, j0, jb, v, W);
    __syncthreads();
  }
}

template <bool TRANS, int NR, bool Z = false>
__global__ __launch_bounds__(solve_threads<NR>()) void solve_backward_kernel(const int *__restrict__ list, TreeView t,
                                                                       const double *__restrict__ invs,
                                                                       double *__restrict__ work,
                                                                       double *__restrict__ x, size_t stride) {
  __shared__ double w[NB][NR], v[NB][NR];
  front_backward<TRANS, NR, Z>(t, list[blockIdx.x], invs, work, x, stride, w, v);
}

// ---- large fronts: the same steps spread over many workgroups, all large fronts of a level in
// lockstep.  Every kernel below runs on a flat grid over the workgroups of all listed fronts
// (prefix = workgroups before each front, computed on the host when the factors are built): one
// launch per step and level however many fronts the level holds.
struct BigFront {
  int f, np, nb, fs, ldp, ldu, blk, item;  // blk: index of this workgroup inside its front; item: of the front in the list
  const double *P, *U;
  double *W, *Z;
  size_t pz, uz;  // complex fronts: offsets of the imaginary planes of P and U (0: real)
};
template <int NR>
__device__ __forceinline__ BigFront big_front(const int *__restrict__ list, const int64_t *__restrict__ prefix,
                                              int count, const TreeView &t, double *work, double *zbuf) {
  const int64_t flat = prefix[0] + blockIdx.x;
  const int fi = item_of_tile(prefix, count, flat);
  const int f = list[fi];
  BigFront b;
  b.f = f;
  b.item = fi;
  b.np = t.np[f];
  b.nb = t.nb[f];
  b.fs = b.np + b.nb;
  b.ldp = t.ldp[f];
  b.ldu = t.ldu[f];
  b.blk = (int)(flat - prefix[fi]);
  b.P = t.arena + (int64_t)t.zm * t.poff[f];
  b.U = t.arena + (int64_t)t.zm * t.uoff[f];
  b.pz = t.zm == 2 ? (size_t)b.ldp * (size_t)b.np : 0;
  b.uz = t.zm == 2 ? (size_t)b.ldu * (size_t)b.nb : 0;
  b.W = work + (size_t)t.woff[f] * NR;
  b.Z = zbuf + (size_t)t.woff[f] * NR;
  return b;
}

#include "mf_chain.hpp"

// one super-block step (256 pivots) of the triangular pass MODE through the pivot columns of every
// listed front that has a step `step`: forward passes take W -> Z, backward passes Z -> W
template <int MODE, int NR, bool Z = false>
__global__ __launch_bounds__(solve_waves<NR>() * 64) void big_super_kernel(const int *__restrict__ list,
                                                            const int64_t *__restrict__ prefix, int count, int step,
                                                            TreeView t, const double *__restrict__ invs,
                                                            double *work, double *zbuf, int row_blocks, int pivots_only,
                                                            double *x, size_t xstride) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  constexpr bool fwd = (MODE == 0 || MODE == 2);
  const BigFront b = big_front<NR>(list, prefix, count, t, work, zbuf);
  const int span = SB * NB, nsup = (b.np + span - 1) / span;
  const int j0 = (fwd ? step : nsup - 1 - step) * span, jbs = min(span, b.np - j0);
  // untransposed forward: the boundary rows of the front get their updates inside the pass (rounds 1 - 3), or —
  // pivots_only, round 4 — from big_gemv_chunk_kernel<.., FWD> once the pivots are solved (see there)
  const int n = (MODE == 0 && !pivots_only) ? b.fs : b.np;
  const Band band{const_cast<double *>(b.P), n, n, n, b.ldp + 1, 0, 0, b.pz};
  // backward passes: the solved pivots go straight to the solution as well (x: nullptr in the forward passes)
  const SolutionSink sink{(!fwd && x) ? x + (size_t)(Z ? 2 : 1) * (size_t)t.p0[b.f] : nullptr, xstride};
  solve_super_tile<MODE, NR, Z>(band, invs + (Z ? 2 : 1) * t.ioff[b.f], j0, jbs, fwd ? b.W : b.Z, fwd ? b.Z : b.W,
                                (size_t)b.fs, b.blk, dsm, row_blocks, sink);
}

// The same pass as a software pipeline over launches (solve_super_pipelined, dense_lu_kernels.hpp): launch k carries,
// for every listed front, the lead groups of its step k (solve super block k, hand the next super block its update
// through `cbuf`) and the bulk groups of its step k - 1 (all other row updates, with the solved super block read back).
// Pivot block only (n = np): the boundary rows have their own streaming product.  prefix: groups before each front in
// THIS launch (lead groups first).
template <int MODE, int NR, bool Z = false>
__global__ __launch_bounds__(solve_waves<NR>() * 64) void big_super_pipe_kernel(const int *__restrict__ list,
                                                            const int64_t *__restrict__ prefix, int count, int launch,
                                                            TreeView t, const double *__restrict__ invs,
                                                            double *work, double *zbuf, double *cbuf, int row_blocks,
                                                            double *x, size_t xstride) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  constexpr bool fwd = (MODE == 0 || MODE == 2);
  const BigFront b = big_front<NR>(list, prefix, count, t, work, zbuf);
  constexpr int span = SB * NB;
  const int n = b.np, nsup = (n + span - 1) / span;
  int nlead = 0;
  if (launch < nsup) {
    const int j0 = (fwd ? launch : nsup - 1 - launch) * span, jbs = min(span, n - j0);
    const int beyond = fwd ? n - (j0 + jbs) : j0;
    nlead = max(1, min(SB, (beyond + 63) / 64));
  }
  const int role = b.blk < nlead ? 0 : 1;
  const int step = role == 0 ? launch : launch - 1;  // (a bulk group exists only where step launch - 1 does: host prefix)
  const int j0 = (fwd ? step : nsup - 1 - step) * span, jbs = min(span, n - j0);
  const Band band{const_cast<double *>(b.P), n, n, n, b.ldp + 1, 0, 0, b.pz};
  double *carry = cbuf + (size_t)t.woff[b.f] * NR;
  const SolutionSink sink{(!fwd && x) ? x + (size_t)(Z ? 2 : 1) * (size_t)t.p0[b.f] : nullptr, xstride};
  solve_super_pipelined<MODE, NR, Z>(band, invs + (Z ? 2 : 1) * t.ioff[b.f], j0, jbs, fwd ? b.W : b.Z, fwd ? b.Z : b.W, carry,
                                     (size_t)b.fs, role, role == 0 ? b.blk : b.blk - nlead, step == 0, dsm, row_blocks, sink);
}

// Z[i][:] -= sum_k M(i, np + k) W[np + k][:], i < np: the boundary's part in the back substitution of a large front.
// Untransposed M(i, np + k) = U(i, k); transposed = F(np + k, i) in P (Z: its conjugate).
// A level of the upper tree has few large fronts and a front np / 64 blocks of rows: with one workgroup per block (the
// kernel of rounds 1 - 2) 40 - 150 CUs were busy, each with 64 x nb x NR multiply-adds — for 16 columns that is the
// issue rate of ONE CU's vector pipes per block (195 us per launch at 80^3, 1 TB/s).  Here a workgroup takes 64 rows x
// kGemvChunk boundary columns (round 3); the entries of x it needs go through LDS once.  Fronts with one chunk subtract
// their sums from Z directly; the chunks of larger ones write partial sums to scratch ([chunk][r][i] behind the front's
// offset) and big_gemv_reduce_kernel subtracts them in chunk order: the result does not depend on the schedule.
// Untransposed: lane = row, gemv_waves wavefronts split the chunk's columns, x k-major in LDS (a multiply-add reads its
// NR values as broadcast 16-byte words, so the fused form of mac_cols pays), partial sums meet in LDS.
// Transposed: lanes along the chunk's columns (contiguous in P), the wavefronts split the rows, x r-major in LDS (lanes
// read side by side), lane sums by recursive halving.
template <int NR>
constexpr int gemv_waves() { return NR >= 16 ? 8 : 16; }  // (the partial sums: GW x 64 x NR doubles of LDS)
constexpr int kGemvChunk = 512;
// FWD (round 4): the same product for the FORWARD pass of the untransposed systems, the boundary rows' share of it:
// W[np + i][:] -= sum_t L21(i, t) Z[t][:], i < nb, with L21 = rows np .. fs of the pivot columns (in P) and Z the solved
// pivots.  Rounds 1 - 3 did this inside the super-block steps (a step updated ALL rows below it): every workgroup of a
// step first redoes the in-super-block solve — a chain of barriers without a byte of HBM traffic — and only then
// streams its rows, so the part of the forward pass that carries the bytes ran at 2.1 - 2.2 TB/s on the top levels of
// config C5 where the backward pass, whose boundary product always was a kernel of its own, reaches 3.2 - 3.9.  Now the
// steps stay inside the pivot block and the nb x np panel below it is streamed once, by this kernel, with every CU on it.
template <int NR, bool Z = false, bool FWD = false>
// (up to eight columns: at most 64 registers, so that two workgroups of 16 wavefronts share a CU — the complex backward
// product took 66, the products with eight columns 66 - 68)
__global__ __launch_bounds__(gemv_waves<NR>() * 64) __attribute__((amdgpu_waves_per_eu(NR <= 8 ? 8 : 2, 8)))
void big_gemv_chunk_kernel(const int *__restrict__ list,
                                                                               const int64_t *__restrict__ prefix, int count,
                                                                               TreeView t, double *work, double *zbuf,
                                                                               const int64_t *__restrict__ pofs,
                                                                               double *__restrict__ scratch,
                                                                               const double *__restrict__ x = nullptr, size_t xstride = 0) {
  constexpr int GW = gemv_waves<NR>();
  extern __shared__ __attribute__((aligned(16))) double gsm[];
  double(*xs)[NR] = reinterpret_cast<double(*)[NR]>(gsm);  // [kGemvChunk]
  double *part = gsm;  // [GW][NR][64], in the place of xs once every wavefront is done with it (two workgroups per CU)
  const BigFront b = big_front<NR>(list, prefix, count, t, work, zbuf);
  // the product is (np x nb) from U, or — FWD — (nb x np) from the rows of P below the pivot block: `np` / `nb` below
  // are the rows / columns of the product
  const int np = FWD ? b.nb : b.np, nb = FWD ? b.np : b.nb, fs = b.fs, ldu = FWD ? b.ldp : b.ldu;
  const int nch = (nb + kGemvChunk - 1) / kGemvChunk;
  const int rb = b.blk / nch, ch = b.blk - rb * nch;
  // FWD: the rows start np rows down the pivot columns; the blocks of 64 are laid from the line boundary at or above
  // that row (`skew` rows of the first block belong to the pivot block and are masked), so every block is whole lines
  const int skew = FWD ? (b.np & 15) : 0;
  const int i0 = rb * 64 - skew, k0 = ch * kGemvChunk, kn = min(kGemvChunk, nb - k0);
  // backward pass (x given): the boundary's part of the solution straight from x through the front's boundary indices
  // (round 4 gathered it into W with a launch of its own per level, big_gather_x_kernel)
  const double *xb = FWD ? b.Z + k0 : b.W + b.np + k0;
  const int *bi = t.bidx + t.bptr[b.f] + k0;
  for (int o = threadIdx.x; o < kn * NR; o += GW * 64) {
    const int kk = o % kn, r = o / kn;
    if (!FWD && x) xs[kk][r] = Z ? x[(size_t)(r >> 1) * xstride + 2 * (size_t)bi[kk] + (r & 1)] : x[(size_t)r * xstride + bi[kk]];
    else xs[kk][r] = xb[(size_t)r * fs + kk];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = i0 + lane;
  double acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = 0.0;
  if (i >= 0 && i < np) {
    const double *row = (FWD ? b.P + (size_t)b.np : b.U) + (ptrdiff_t)i + (size_t)k0 * ldu;
    const size_t zo = FWD ? b.pz : b.uz;
    int k = wave;
    for (; k + 7 * GW < kn; k += 8 * GW) {
      double e[8], ei[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        e[u] = __builtin_nontemporal_load(&row[(size_t)(k + GW * u) * ldu]);  // (a panel is read once per walk)
        ei[u] = Z ? __builtin_nontemporal_load(&row[(size_t)(k + GW * u) * ldu + zo]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) mac_cols<NR, Z>(acc, e[u], ei[u], &xs[k + GW * u][0]);
    }
    for (; k < kn; k += GW) {
      const double e = __builtin_nontemporal_load(&row[(size_t)k * ldu]);
      const double ei = Z ? __builtin_nontemporal_load(&row[(size_t)k * ldu + zo]) : 0.0;
      mac_cols<NR, Z>(acc, e, ei, &xs[k][0]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NR; ++r) part[(wave * NR + r) * 64 + lane] = acc[r];
  __syncthreads();
  double *out = nch > 1 ? scratch + (size_t)pofs[b.item] * NR + (size_t)ch * NR * np : nullptr;
  for (int o = threadIdx.x; o < 64 * NR; o += GW * 64) {
    const int r = o / 64, l = o % 64;
    if (i0 + l < 0 || i0 + l >= np) continue;
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < GW; ++q) tot += part[(q * NR + r) * 64 + l];
    if (out) out[(size_t)r * np + i0 + l] = tot;
    else if (FWD) b.W[(size_t)r * fs + b.np + i0 + l] -= tot;
    else b.Z[(size_t)r * fs + i0 + l] -= tot;
  }
}

template <int NR, bool Z = false>
__global__ __launch_bounds__(gemv_waves<NR>() * 64) void big_gemv_chunk_t_kernel(const int *__restrict__ list,
                                                                                 const int64_t *__restrict__ prefix, int count,
                                                                                 TreeView t, double *work, double *zbuf,
                                                                                 const int64_t *__restrict__ pofs,
                                                                                 double *__restrict__ scratch,
                                                                                 const double *__restrict__ x = nullptr, size_t xstride = 0) {
  constexpr int GW = gemv_waves<NR>();
  extern __shared__ __attribute__((aligned(16))) double gsm[];
  double(*xs)[kGemvChunk] = reinterpret_cast<double(*)[kGemvChunk]>(gsm);  // [NR]
  const BigFront b = big_front<NR>(list, prefix, count, t, work, zbuf);
  const int np = b.np, nb = b.nb, fs = b.fs, ldp = b.ldp;
  const int nch = (nb + kGemvChunk - 1) / kGemvChunk;
  const int rb = b.blk / nch, ch = b.blk - rb * nch;
  const int i0 = rb * 64, k0 = ch * kGemvChunk, kn = min(kGemvChunk, nb - k0);
  const double *xb = b.W + b.np + k0;
  const int *bi = t.bidx + t.bptr[b.f] + k0;
  for (int o = threadIdx.x; o < kGemvChunk * NR; o += GW * 64) {
    const int kk = o % kGemvChunk, r = o / kGemvChunk;
    double xv = 0.0;  // (zeros past the chunk's end: the loads below are clamped, not masked)
    if (kk < kn) xv = x ? (Z ? x[(size_t)(r >> 1) * xstride + 2 * (size_t)bi[kk] + (r & 1)] : x[(size_t)r * xstride + bi[kk]]) : xb[(size_t)r * fs + kk];
    xs[r][kk] = xv;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *out = nch > 1 ? scratch + (size_t)pofs[b.item] * NR + (size_t)ch * NR * np : nullptr;
  const double *P = b.P + (size_t)(np + k0);
  constexpr int RW = NR <= 2 ? 4 : 2;  // rows per wavefront and trip
  for (int rr = wave * RW; rr < 64; rr += GW * RW) {
    double acc[RW * NR];
#pragma unroll
    for (int o = 0; o < RW * NR; ++o) acc[o] = 0.0;
    const double *col[RW];
#pragma unroll
    for (int u = 0; u < RW; ++u) col[u] = P + (size_t)min(i0 + rr + u, np - 1) * ldp;  // (rows past np: read again, not stored)
#pragma unroll 4
    for (int kk = lane; kk < kGemvChunk; kk += 64) {
      if (kk - lane >= kn) break;  // (uniform)
      const int kc = min(kk, kn - 1);
      double e[RW], ei[RW];
#pragma unroll
      for (int u = 0; u < RW; ++u) {
        e[u] = __builtin_nontemporal_load(&col[u][kc]);
        ei[u] = Z ? -__builtin_nontemporal_load(&col[u][kc + b.pz]) : 0.0;  // conjugate transpose
      }
      double xv[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) xv[r] = xs[r][kk];
#pragma unroll
      for (int u = 0; u < RW; ++u) mac_cols<NR, Z>(*reinterpret_cast<double(*)[NR]>(&acc[u * NR]), e[u], ei[u], xv);
    }
    wave_reduce_scatter<RW * NR>(acc);
    if (wave_reduce_owner<RW * NR>(lane)) {
      const int idx = wave_reduce_index<RW * NR>(lane, 0), u = idx / NR, r = idx % NR;  // (RW * NR <= 64)
      const int i = i0 + rr + u;
      if (i < np) {
        if (out) out[(size_t)r * np + i] = acc[0];
        else b.Z[(size_t)r * fs + i] -= acc[0];
      }
    }
  }
}

template <int NR, bool FWD = false>
__global__ __launch_bounds__(256) void big_gemv_reduce_kernel(const int *__restrict__ list, const int64_t *__restrict__ prefix,
                                                              int count, TreeView t, double *work, double *zbuf,
                                                              const int64_t *__restrict__ pofs,
                                                              const double *__restrict__ scratch) {
  const BigFront b = big_front<NR>(list, prefix, count, t, work, zbuf);
  const int rows = FWD ? b.nb : b.np, cols = FWD ? b.np : b.nb;
  const int i = b.blk * 256 + (int)threadIdx.x;
  if (i >= rows) return;
  const int nch = (cols + kGemvChunk - 1) / kGemvChunk;
  const double *in = scratch + (size_t)pofs[b.item] * NR;
  double tot[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) tot[r] = 0.0;
  for (int c = 0; c < nch; ++c)  // (NR loads in flight per thread; chunk order: the sums do not depend on the schedule)
#pragma unroll
    for (int r = 0; r < NR; ++r) tot[r] += in[((size_t)c * NR + r) * rows + i];
  double *dst = FWD ? b.W + b.np : b.Z;
#pragma unroll
  for (int r = 0; r < NR; ++r) dst[(size_t)r * b.fs + i] -= tot[r];
}

// transposed forward elimination, boundary part: W[np + k][:] -= sum_t U(t, k) Z[t][:]  (U^T y); one
// wavefront per boundary index, lanes along t
template <int NR, bool Z = false>
__global__ __launch_bounds__(256) void big_boundary_t_kernel(const int *__restrict__ list,
                                                             const int64_t *__restrict__ prefix, int count,
                                                             TreeView t, double *work, double *zbuf) {
  const BigFront b = big_front<NR>(list, prefix, count, t, work, zbuf);
  const int lane = threadIdx.x & 63;
  const int k = b.blk * 4 + (int)(threadIdx.x >> 6);
  if (k >= b.nb) return;
  const double *col = b.U + (size_t)k * b.ldu;
  double acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = 0.0;
  for (int tt = lane; tt < b.np; tt += 64) {
    const double e = col[tt];
    const double ei = Z ? -col[tt + b.uz] : 0.0;  // conjugate transpose
    double zv[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) zv[r] = b.Z[(size_t)r * b.fs + tt];
    mac_cols<NR, Z>(acc, e, ei, zv);
  }
  wave_reduce_scatter<NR>(acc);
  if (wave_reduce_owner<NR>(lane)) b.W[(size_t)wave_reduce_index<NR>(lane, 0) * b.fs + b.np + k] -= acc[0];
}

}  // namespace

namespace mf {

// Everything about the levels that depends on the tree and the size limits only — which fronts of a depth go to which
// kernel, the prefix sums that turn a level into flat grids (factorisation and solves alike) —: built by the first
// factorisation of a tree on a device and shared by all that follow (round 4; a refactorisation used to rebuild and
// upload it: 1.8 ms of 134 at 100^3, 0.8 of 7 ms at 32^3, some 440 small copies).  Read-only once built.
struct LevelPlan {
  int small_limit = 0, mid_limit = 0, mid_paired = 0;  // what it was built for (with big_solve below)
  std::vector<DBuf<int>> level_lists;                // fronts of each depth
  std::vector<DBuf<int>> small_lists;                // ... those factored by one workgroup each
  std::vector<int> small_counts;
  std::vector<DBuf<int>> solve_lists;                // ... those solved by one workgroup each
  std::vector<int> solve_counts;
  std::vector<DBuf<int>> child_lists[2];             // children (by slot) of the fronts of each depth
  std::vector<int> child_counts[2];
  std::vector<int> child_maxnb[2];                   // ... and the largest boundary among them
  std::vector<std::vector<int>> h_small, h_child[2];  // host copies (ascending ids) of small_lists / child_lists
  // tiles (64 x 16) before each item of child_lists (Schur complements) and of level_lists (P and
  // U panels): the flat grids of extend-add and compaction
  std::vector<std::vector<int64_t>> h_ctile[2], h_ptile, h_utile, h_atile;  // h_atile: groups of 32 pivots (assembly)
  std::vector<DBuf<int64_t>> ctile[2], ptile, utile, atile;
  int big_solve = 0;  // fronts above this size are solved by many workgroups (kBigSolve; SPL_MF_BIGSOLVE)
  int64_t gemv_scratch = 0;  // doubles per right-hand-side column the chunked boundary product of a level needs at most
  // those fronts, per depth, and the flat grids of their lockstep solve kernels: segments of
  // count + 1 prefix sums (workgroups before each front), in this order: untransposed forward
  // steps [0, steps), transposed forward steps, backward steps, then the boundary kernel of the
  // transposed forward pass, gather, gemv, scatter
  struct BigLevel {
    int count = 0, steps = 0;
    int row_blocks = 1;  // blocks of 64 rows per workgroup in the super-block steps (levels that fill the chip: more)
    DBuf<int> list;
    std::vector<int64_t> h;
    DBuf<int64_t> d;
    // kind 0 fwd, 1 fwd^T, 2 bwd, 3 boundary^T, 4 gather, 5 gemv (transposed), 6 scatter, 7 gemv in chunks, 8 its
    // reduction, 9 offsets of the fronts in its scratch (doubles per column); 10, 11, 12: the same three for the
    // forward boundary product (nb rows x np columns)
    // 13, 14: the pipelined passes over the pivot block (forward, backward): launches [0, steps]
    size_t seg(int kind, int k = 0) const {
      size_t which;
      if (kind < 3) which = (size_t)kind * (size_t)steps + (size_t)k;
      else if (kind < 13) which = (size_t)3 * steps + (size_t)(kind - 3);
      else which = (size_t)3 * steps + 10 + (size_t)(kind - 13) * (size_t)(steps + 1) + (size_t)k;
      return which * (size_t)(count + 1);
    }
    unsigned total(int kind, int k = 0) const { return (unsigned)h[seg(kind, k) + (size_t)count]; }
    const int64_t *prefix(int kind, int k = 0) const { return d.get() + seg(kind, k); }
  };
  std::vector<BigLevel> big;
  // the lockstep medium fronts of each depth (whole level) and the prefix arrays of their steps:
  // [step][phase 0 = panel solves, 1 = update][count + 1]
  struct Mid {
    int count = 0, steps = 0;
    DBuf<int> list;
    std::vector<int64_t> h;
    DBuf<int64_t> d;
  };
  std::vector<Mid> mid;
};

struct Factors {
  std::shared_ptr<const Tree> tree;
  std::shared_ptr<DeviceTree> D;     // the tree's own arrays and `rel`: shared by every factorisation of this tree
  DBuf<int64_t> d_foff, d_cboff;     // where this factorisation's memory plan puts the fronts
  TreeView view;
  DBuf<double> arena, invs;  // factor panels, inverses of the diagonal blocks

  std::shared_ptr<LevelPlan> lp;  // lists and grids of the levels (shared with the other factorisations of this tree)
  using BigLevel = LevelPlan::BigLevel;
  int singular = 0;
  int zm = 1;         // 2: complex fronts in two planes (TreeView::zm)
  // the pivot blocks of the large fronts as chains of matrix-vector products (mf_chain.hpp): built by the first
  // untransposed solve with one right-hand side, under `once`; ok = false: none (switched off, no memory, or an entry
  // beyond chain::kLimit) — the walk then keeps its substitution steps
  struct Chain {
    std::once_flag once;
    std::atomic<bool> ok{false};  // (written once, under `once`; read by any thread: the solves, mf_chain_info)
    int span = 0;
    int64_t elems = 0;      // doubles of one plane
    DBuf<double> buf;
    DBuf<int64_t> off;      // per front (see chain::View)
    // the flat grids of the chain launches: per depth, [pass 0 = forward, 1 = backward][launch 0 .. steps][count + 1]
    struct Level {
      int steps = 0, count = 0, row_blocks = 1;
      // rows of a block per lead workgroup: chain::kRows, or rows_wide() on the levels of small pivot blocks; [1]: with 16
      // right-hand-side columns (workgroups of 8 wavefronts): half of that, and 64
      int lead_rows[2] = {0, 0};
      size_t base = 0;
    };
    std::vector<Level> levels;
    std::vector<int64_t> h;
    DBuf<int64_t> d;
    // (pass: 0 forward, 1 backward; + 2 for the grids of 16 columns)
    size_t at(int depth, int pass, int launch) const {
      const Level &L = levels[(size_t)depth];
      return L.base + ((size_t)pass * (size_t)(L.steps + 1) + (size_t)launch) * (size_t)(L.count + 1);
    }
    double build_ms = 0.0;
  };
  mutable Chain chain, chain_t;  // A x = b; A^T x = b (A^H for complex fronts)
  // independent large fronts of a level run on these (factorisation): one set per device and host
  // thread, created on first use and never destroyed (objects come and go by the thousand in a
  // contour integration; work of different objects on the same stream is merely ordered)
  hipStream_t *side = nullptr;
  int nside = 0;
  void make_streams() {
    // (round 3: one set per device AND host thread — factorisations issued by different threads, the contour points of
    // a FEAST iteration, then overlap on the device instead of queueing behind each other on shared streams)
    static std::mutex mu;
    static std::map<std::pair<int, std::thread::id>, std::unique_ptr<hipStream_t[]>> sets;
    int device = 0;
    SPL_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> lk(mu);
    std::unique_ptr<hipStream_t[]> &set = sets[std::make_pair(device, std::this_thread::get_id())];
    if (!set) {
      // kStreams for the fronts, kStreams beside them for the look-ahead tiles of their large windows, one for the
      // assembly work that runs beside the factorisation of a level (mf_factor_t)
      std::unique_ptr<hipStream_t[]> fresh(new hipStream_t[2 * kStreams + 1]);
      for (int i = 0; i < 2 * kStreams + 1; ++i) {  // (from the pool the analysis filled beside its own work, or new)
        fresh[i] = pooled_stream_take(device);
        if (!fresh[i]) throw DeviceError{SPL_ERROR_internal};
      }
      set = std::move(fresh);
    }
    side = set.get();
    nside = kStreams;
  }
};

}  // namespace mf

void mf_free(mf::Factors *F) { delete F; }

int mf_singular(const mf::Factors *F) { return F->singular; }

void mf_chain_info(const mf::Factors *F, double out[3]) {
  out[0] = out[1] = out[2] = 0.0;
  for (const mf::Factors::Chain *c : {&F->chain, &F->chain_t})  // (both sets, where both systems have been solved)
    if (c->ok.load(std::memory_order_acquire)) {
      out[0] += (double)F->zm * (double)c->elems * 8.0;
      out[1] += c->build_ms;
      out[2] = (double)c->span;
    }
}

namespace {

// Transient memory plan of the numeric factorisation.  With cut = 0 the tree is processed level by
// level as a whole: the whole fronts of a level and of its children are alive together.  With
// cut = K > 0 the subtrees hanging below depth K are processed one after the other (each needs
// only its own share of every level), the Schur complement of each subtree root waits in the
// `cut` buffer, and the top K levels come last.  The smallest K that fits the budget is used.
struct Plan {
  int cut = 0;
  std::vector<int64_t> foff;         // offset of every whole front inside its region (per segment and level)
  std::vector<int64_t> cboff;        // offset of the saved Schur complement of a subtree root, or -1
  std::vector<int> first;            // smallest id of the subtree of every front
  std::vector<int> roots;            // the fronts at depth `cut` (ascending ids); empty for cut = 0
  int64_t region_elems[2] = {0, 0}, cut_elems = 0;
  int64_t transient_elems() const { return region_elems[0] + region_elems[1] + cut_elems; }
};

Plan make_plan(const mf::Tree &T, int cut) {
  Plan P;
  P.cut = cut;
  const int nf = T.nfronts, nd = T.maxdepth + 1;
  P.foff.assign((size_t)nf, 0);
  P.cboff.assign((size_t)nf, -1);
  P.first.resize((size_t)nf);
  for (int f = 0; f < nf; ++f) P.first[(size_t)f] = f;
  for (int f = 0; f < nf; ++f)
    if (T.parent[(size_t)f] >= 0)
      P.first[(size_t)T.parent[(size_t)f]] = std::min(P.first[(size_t)T.parent[(size_t)f]], P.first[(size_t)f]);
  auto front_elems = [&](int f) {
    return ((int64_t)T.ld[(size_t)f] * std::max(T.fs(f), 1) + 15) / 16 * 16;
  };
  // lays out the fronts with ids in [lo, hi] and depths in [dtop, nd): returns nothing, grows the regions
  auto layout = [&](int lo, int hi, int dtop) {
    for (int d = dtop; d < nd; ++d) {
      const std::vector<int> &L = T.by_depth[(size_t)d];
      int64_t off = 0;
      for (auto it = std::lower_bound(L.begin(), L.end(), lo); it != L.end() && *it <= hi; ++it) {
        P.foff[(size_t)*it] = off;
        off += front_elems(*it);
      }
      P.region_elems[d & 1] = std::max(P.region_elems[d & 1], off);
    }
  };
  if (cut <= 0 || cut >= nd) {
    P.cut = 0;
    layout(0, nf - 1, 0);
    return P;
  }
  P.roots = T.by_depth[(size_t)cut];
  for (int r : P.roots) {
    layout(P.first[(size_t)r], r, cut);
    P.cboff[(size_t)r] = P.cut_elems;
    P.cut_elems += ((int64_t)T.nb[(size_t)r] * T.nb[(size_t)r] + 15) / 16 * 16;
  }
  // the top: depths < cut (their lists hold nothing else)
  for (int d = 0; d < cut; ++d) {
    int64_t off = 0;
    for (int f : T.by_depth[(size_t)d]) {
      P.foff[(size_t)f] = off;
      off += front_elems(f);
    }
    P.region_elems[d & 1] = std::max(P.region_elems[d & 1], off);
  }
  return P;
}

}  // namespace

size_t mf_device_bytes(const mf::Tree &T, int zm) {  // resident part; the transient part is planned in mf_factor
  return ((size_t)zm * ((size_t)T.panel_elems + (size_t)T.inv_elems) + 3 * (size_t)zm * (size_t)T.work_elems) * sizeof(double) +
         ((size_t)T.bidx.size() + (size_t)T.rel_elems + 14 * (size_t)T.nfronts + (size_t)T.n) * sizeof(int64_t);
}

// numeric factorisation of P A P^T; d_Ap/d_Ai/d_Ax: CSC arrays of A on the device, d_Rp/d_Rj/d_Rx: its CSR
// arrays, d_perm: new -> old, d_inv: old -> new
namespace {
template <bool Z>
mf::Factors *mf_factor_t(std::shared_ptr<const mf::Tree> tree, const int *d_Ap, const int *d_Ai, const double *d_Ax,
                         const int *d_Rp, const int *d_Rj, const double *d_Rx, const int *d_perm, const int *d_inv,
                         hipStream_t s, bool symmetric, bool pivot, const double *d_rscale) {
  constexpr int ZM = Z ? 2 : 1;
  if (symmetric) pivot = false;  // L D L^T needs symmetric interchanges
  const mf::Tree &T = *tree;
  const bool timing = getenv("SPL_MF_TIMING") != nullptr;  // phase times on stderr (diagnostic)
  auto clock_now = [] { return std::chrono::steady_clock::now(); };
  auto t_start = clock_now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto now = clock_now();
    fprintf(stderr, "[mf_factor] %-52s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_start).count());
    t_start = now;
  };
  std::unique_ptr<mf::Factors> Fp(new mf::Factors());
  mf::Factors &F = *Fp;
  F.tree = tree;
  F.zm = ZM;
  const int big_solve = getenv("SPL_MF_BIGSOLVE") ? std::max(64, atoi(getenv("SPL_MF_BIGSOLVE"))) : kBigSolve;
  const int nd = T.maxdepth + 1, nf = T.nfronts;
  const int small_limit = getenv("SPL_MF_SMALL") ? std::max(64, atoi(getenv("SPL_MF_SMALL"))) : kSmallFront;  // tuning knob
  // medium fronts (kSmallFront < size <= mid_limit) go through the lockstep kernels; SPL_MF_MID=0
  // sends them down the per-front pipeline instead (ablation)
  const int mid_limit = (getenv("SPL_MF_MID") && atoi(getenv("SPL_MF_MID")) == 0) ? small_limit
                        : std::max(getenv("SPL_MF_MIDMAX") ? atoi(getenv("SPL_MF_MIDMAX")) : kMidFront, small_limit);
  const bool mid_paired = !(getenv("SPL_MF_MIDPAIR") && atoi(getenv("SPL_MF_MIDPAIR")) == 0);  // 0: K = 64 every step (ablation)
  // ---- memory plan: the smallest cut depth whose transient part fits next to the resident part
  Plan plan;
  {
    const size_t free_b = device_free_bytes();
    const double budget = (double)free_b - (double)free_b / 16 - (double)mf_device_bytes(T, ZM);
    const char *force = getenv("SPL_MF_CUT");  // tests: force a cut depth
    bool ok = false;
    for (int cut = force ? std::max(0, std::min(atoi(force), nd - 1)) : 0; cut < std::max(nd, 1) && cut <= 10; ++cut) {
      plan = make_plan(T, cut);
      if ((double)plan.transient_elems() * sizeof(double) * ZM <= budget || force) { ok = true; break; }
    }
    if (!ok) throw DeviceError{SPL_ERROR_out_of_memory};
    if (timing)
      fprintf(stderr, "[mf_factor] %spanels %.1f GB, transient %.1f GB at cut depth %d (free %.1f GB)\n",
              Z ? "complex fronts: " : "", ZM * T.panel_elems * 8e-9, ZM * plan.transient_elems() * 8e-9, plan.cut,
              free_b * 1e-9);
  }
  // the device copy of the tree and the relative indices: built once per tree and device, then shared
  {
    std::lock_guard<std::mutex> lk(T.device_cache_mu);
    int dev = 0;
    SPL_HIP(hipGetDevice(&dev));
    std::shared_ptr<DeviceTree> cached = std::static_pointer_cast<DeviceTree>(T.device_cache);
    if (!cached || cached->device != dev) {
      std::shared_ptr<DeviceTree> fresh = std::make_shared<DeviceTree>();
      DeviceTree &N = *fresh;
      N.device = dev;
      upload_vec(N.p0, T.p0, s);
      upload_vec(N.np, T.np, s);
      upload_vec(N.nb, T.nb, s);
      upload_vec(N.ld, T.ld, s);
      upload_vec(N.parent, T.parent, s);
      upload_vec(N.front_of, T.front_of, s);
      upload_vec(N.bidx, T.bidx, s);
      upload_vec(N.bptr, T.bptr, s);
      upload_vec(N.ioff, T.ioff, s);
      upload_vec(N.woff, T.woff, s);
      upload_vec(N.roff, T.roff, s);
      upload_vec(N.depth, T.depth, s);
      upload_vec(N.ldp, T.ldp, s);
      upload_vec(N.ldu, T.ldu, s);
      upload_vec(N.poff, T.poff, s);
      upload_vec(N.uoff, T.uoff, s);
      N.rel.alloc((size_t)T.rel_elems);
      const TreeView v{N.p0.get(),   N.np.get(),   N.nb.get(),   N.ld.get(),   N.parent.get(), N.front_of.get(),
                       N.bidx.get(), N.rel.get(),  N.depth.get(), N.ldp.get(), N.ldu.get(),    N.bptr.get(),
                       nullptr,      N.ioff.get(), N.woff.get(), N.roff.get(), N.poff.get(),   N.uoff.get(),
                       {nullptr, nullptr}, nullptr, nullptr, nullptr, 0, 1, 0, nullptr};
      if (nf > 0) hipLaunchKernelGGL(rel_kernel, dim3((unsigned)nf), dim3(256), 0, s, nf, v, N.rel.get());
      SPL_HIP(hipStreamSynchronize(s));  // the host vectors of the tree may go away with it
      SPL_HIP(hipGetLastError());
      T.device_cache = fresh;
      cached = fresh;
    }
    F.D = cached;
  }
  DeviceTree &D = *F.D;
  upload_vec(F.d_foff, plan.foff, s);
  upload_vec(F.d_cboff, plan.cboff, s);
  lap("tree uploads (queued)");
  F.arena.alloc((size_t)ZM * (size_t)T.panel_elems);
  F.invs.alloc((size_t)ZM * (size_t)T.inv_elems);
  DBuf<double> region0((size_t)ZM * (size_t)plan.region_elems[0]), region1((size_t)ZM * (size_t)plan.region_elems[1]),
      cutbuf((size_t)ZM * (size_t)plan.cut_elems);  // transient
  lap("hipMalloc");
  F.view = TreeView{D.p0.get(),    D.np.get(),   D.nb.get(),   D.ld.get(),   D.parent.get(), D.front_of.get(),
                    D.bidx.get(),  D.rel.get(),  D.depth.get(), D.ldp.get(), D.ldu.get(),    D.bptr.get(),
                    F.d_foff.get(), D.ioff.get(), D.woff.get(), D.roff.get(), D.poff.get(),  D.uoff.get(),
                    {region0.get(), region1.get()}, F.arena.get(), F.d_cboff.get(), cutbuf.get(), symmetric ? 1 : 0, ZM,
                    pivot ? 1 : diag_form_without_interchanges(), pivot ? d_rscale : nullptr};
  std::vector<std::vector<int>> staged;  // host copies must outlive the asynchronous uploads
  // the medium fronts among the fronts [b0, b1) of a level's list, the number of block steps of the longest and the
  // prefix arrays of all steps: [step][phase 0 = panel solves, 1 = update][count + 1]
  auto mid_fronts_of = [&](const std::vector<int> &fronts, int b0, int b1, std::vector<int> &mid, int &steps,
                           std::vector<int64_t> &pre) {
    mid.clear();
    steps = 0;
    for (int i = b0; i < b1; ++i) {
      const int f = fronts[(size_t)i];
      if (T.np[(size_t)f] == 0 || T.fs(f) <= small_limit || T.fs(f) > mid_limit) continue;
      mid.push_back(f);
      steps = std::max(steps, (T.np[(size_t)f] + NB - 1) / NB);
    }
    const int count = (int)mid.size();
    pre.assign((size_t)steps * 2 * (size_t)(count + 1), 0);
    for (int st = 0; st < steps; ++st) {
      int64_t *pt = pre.data() + ((size_t)st * 2) * (size_t)(count + 1), *pu = pt + (count + 1);
      for (int k = 0; k < count; ++k) {
        const int f = mid[(size_t)k], np = T.np[(size_t)f], j0 = st * NB;
        int64_t tt = 0, tu = 0;
        if (j0 < np) {
          const int jb = std::min(NB, np - j0), rest = T.fs(f) - (j0 + jb);
          const int64_t nt = (rest + 63) / 64;
          tt = 2 * nt;
          tu = nt * nt;
          if (mid_paired && !(st & 1) && np >= j0 + jb + NB) tu = 2 * nt - 1;  // (the L-shaped pass of an even step)
        }
        pt[k + 1] = pt[k] + tt;
        pu[k + 1] = pu[k] + tu;
      }
    }
  };
  // the lists and grids of the levels: from the tree's cache, or built now (and kept there)
  std::unique_lock<std::mutex> plan_lock(T.device_cache_mu);
  {
    std::shared_ptr<mf::LevelPlan> have = std::static_pointer_cast<mf::LevelPlan>(D.level_plan);
    if (have && have->small_limit == small_limit && have->big_solve == big_solve && have->mid_limit == mid_limit &&
        have->mid_paired == (mid_paired ? 1 : 0))
      F.lp = have;
  }
  const bool build_plan = !F.lp;
  if (build_plan) {
  F.lp = std::make_shared<mf::LevelPlan>();
  F.lp->small_limit = small_limit;
  F.lp->big_solve = big_solve;
  F.lp->mid_limit = mid_limit;
  F.lp->mid_paired = mid_paired ? 1 : 0;
  F.lp->level_lists.resize((size_t)nd);
  F.lp->small_lists.resize((size_t)nd);
  F.lp->small_counts.assign((size_t)nd, 0);
  F.lp->solve_lists.resize((size_t)nd);
  F.lp->big.resize((size_t)nd);
  F.lp->solve_counts.assign((size_t)nd, 0);
  F.lp->h_small.assign((size_t)nd, std::vector<int>());
  for (int sl = 0; sl < 2; ++sl) {
    F.lp->child_lists[sl].resize((size_t)nd);
    F.lp->child_counts[sl].assign((size_t)nd, 0);
    F.lp->child_maxnb[sl].assign((size_t)nd, 0);
    F.lp->h_child[sl].assign((size_t)nd, std::vector<int>());
  }
  // the flat grids of the lockstep solve kernels for the large fronts `large` of one depth
  auto build_big = [&](mf::Factors::BigLevel &B, std::vector<int> large) {
    B.count = (int)large.size();
    constexpr int span = SB * NB;
    for (int f : large) B.steps = std::max(B.steps, (T.np[(size_t)f] + span - 1) / span);
    if (B.count > 0) {
      // every workgroup of a step redoes the in-super-block solve: where the first step alone
      // would launch thousands of workgroups, each takes kSolveRowBlocks blocks of rows instead
      int64_t first_step = 0;
      for (int f : large) first_step += (T.fs(f) + 63) / 64;
      B.row_blocks = first_step >= 4096 ? kSolveRowBlocks : 1;
      const int rbk = B.row_blocks;
      B.h.assign((size_t)(3 * B.steps + 10 + 2 * (B.steps + 1)) * (size_t)(B.count + 1), 0);
      auto fill = [&](int kind, int k, auto groups_of) {
        int64_t *pre = B.h.data() + B.seg(kind, k);
        for (int i = 0; i < B.count; ++i) pre[i + 1] = pre[i] + groups_of(large[(size_t)i]);
      };
      for (int k = 0; k < B.steps; ++k) {
        auto fwd_rows = [&](int f, int n) -> int64_t {  // workgroups of forward step k of a pass over n rows
          const int np = T.np[(size_t)f], j0 = k * span;
          if (j0 >= np) return 0;
          const int jbs = std::min(span, np - j0);
          return std::max(1, ((n - (j0 + jbs) + 63) / 64 + rbk - 1) / rbk);
        };
        fill(0, k, [&](int f) { return fwd_rows(f, T.fs(f)); });
        fill(1, k, [&](int f) { return fwd_rows(f, T.np[(size_t)f]); });
        fill(2, k, [&](int f) -> int64_t {
          const int np = T.np[(size_t)f], nsup = (np + span - 1) / span;
          if (k >= nsup) return 0;
          return std::max(1, (((nsup - 1 - k) * span + 63) / 64 + rbk - 1) / rbk);
        });
      }
      for (int k = 0; k <= B.steps; ++k) {  // the pipelined passes: lead groups of step k + bulk groups of step k - 1
        auto groups = [&](int f, bool forward) -> int64_t {
          const int np = T.np[(size_t)f], nsup = (np + span - 1) / span;
          auto beyond = [&](int st) {  // rows of the pivot block still to be updated by step st
            const int j0 = (forward ? st : nsup - 1 - st) * span, jbs = std::min(span, np - j0);
            return forward ? np - (j0 + jbs) : j0;
          };
          int64_t g = 0;
          if (k < nsup) g += std::max(1, std::min(SB, (beyond(k) + 63) / 64));
          if (k >= 1 && k - 1 < nsup) {
            const int rest = beyond(k - 1) - span;
            if (rest > 0) g += ((rest + 63) / 64 + rbk - 1) / rbk;
          }
          return g;
        };
        fill(13, k, [&](int f) { return groups(f, true); });
        fill(14, k, [&](int f) { return groups(f, false); });
      }
      fill(3, 0, [&](int f) -> int64_t { return (T.nb[(size_t)f] + 3) / 4; });
      fill(4, 0, [&](int f) -> int64_t { return (T.nb[(size_t)f] + 255) / 256; });
      fill(5, 0, [&](int f) -> int64_t { return T.nb[(size_t)f] > 0 ? (T.np[(size_t)f] + 63) / 64 : 0; });
      fill(6, 0, [&](int f) -> int64_t { return (T.np[(size_t)f] + 255) / 256; });
      auto chunks = [&](int f) -> int64_t { return (T.nb[(size_t)f] + kGemvChunk - 1) / kGemvChunk; };
      fill(7, 0, [&](int f) -> int64_t { return (int64_t)((T.np[(size_t)f] + 63) / 64) * chunks(f); });
      fill(8, 0, [&](int f) -> int64_t { return chunks(f) > 1 ? (T.np[(size_t)f] + 255) / 256 : 0; });
      fill(9, 0, [&](int f) -> int64_t { return chunks(f) > 1 ? chunks(f) * T.np[(size_t)f] : 0; });
      F.lp->gemv_scratch = std::max(F.lp->gemv_scratch, B.h[B.seg(9) + (size_t)B.count]);
      auto fchunks = [&](int f) -> int64_t { return (T.np[(size_t)f] + kGemvChunk - 1) / kGemvChunk; };
      fill(10, 0, [&](int f) -> int64_t { return (int64_t)((T.nb[(size_t)f] + (T.np[(size_t)f] & 15) + 63) / 64) * fchunks(f); });
      fill(11, 0, [&](int f) -> int64_t { return fchunks(f) > 1 ? (T.nb[(size_t)f] + 255) / 256 : 0; });
      fill(12, 0, [&](int f) -> int64_t { return fchunks(f) > 1 ? fchunks(f) * T.nb[(size_t)f] : 0; });
      F.lp->gemv_scratch = std::max(F.lp->gemv_scratch, B.h[B.seg(12) + (size_t)B.count]);
      staged.push_back(std::move(large));
      upload_vec(B.list, staged.back(), s);
      upload_vec(B.d, B.h, s);
    }
      };
  staged.reserve((size_t)nd);
  for (int d = 0; d < nd; ++d) {
    upload_vec(F.lp->level_lists[(size_t)d], T.by_depth[(size_t)d], s);
    for (int f : T.by_depth[(size_t)d])
      if (T.np[(size_t)f] > 0 && T.fs(f) <= small_limit) F.lp->h_small[(size_t)d].push_back(f);
    F.lp->small_counts[(size_t)d] = (int)F.lp->h_small[(size_t)d].size();
    upload_vec(F.lp->small_lists[(size_t)d], F.lp->h_small[(size_t)d], s);
    std::vector<int> one_wg;
    for (int f : T.by_depth[(size_t)d])
      if (T.fs(f) <= F.lp->big_solve) one_wg.push_back(f);
    F.lp->solve_counts[(size_t)d] = (int)one_wg.size();
    staged.push_back(std::move(one_wg));
    upload_vec(F.lp->solve_lists[(size_t)d], staged.back(), s);
    {
      std::vector<int> large;
      for (int f : T.by_depth[(size_t)d])
        if (T.fs(f) > F.lp->big_solve && T.np[(size_t)f] > 0) large.push_back(f);
      build_big(F.lp->big[(size_t)d], std::move(large));
    }
    if (d + 1 < nd) {
      for (int c : T.by_depth[(size_t)d + 1]) F.lp->h_child[T.slot[(size_t)c]][(size_t)d].push_back(c);
      for (int sl = 0; sl < 2; ++sl) {
        F.lp->child_counts[sl][(size_t)d] = (int)F.lp->h_child[sl][(size_t)d].size();
        for (int c : F.lp->h_child[sl][(size_t)d]) F.lp->child_maxnb[sl][(size_t)d] = std::max(F.lp->child_maxnb[sl][(size_t)d], T.nb[(size_t)c]);
        upload_vec(F.lp->child_lists[sl][(size_t)d], F.lp->h_child[sl][(size_t)d], s);
      }
    }
  }
  {
    auto tiles = [](int64_t rows, int64_t cols) { return ((rows + 63) / 64) * ((cols + kTileCols - 1) / kTileCols); };
    F.lp->h_ptile.assign((size_t)nd, std::vector<int64_t>());
    F.lp->h_utile.assign((size_t)nd, std::vector<int64_t>());
    F.lp->h_atile.assign((size_t)nd, std::vector<int64_t>());
    F.lp->ptile.resize((size_t)nd);
    F.lp->utile.resize((size_t)nd);
    F.lp->atile.resize((size_t)nd);
    for (int sl = 0; sl < 2; ++sl) {
      F.lp->h_ctile[sl].assign((size_t)nd, std::vector<int64_t>());
      F.lp->ctile[sl].resize((size_t)nd);
    }
    for (int d = 0; d < nd; ++d) {
      std::vector<int64_t> &pp = F.lp->h_ptile[(size_t)d], &uu = F.lp->h_utile[(size_t)d], &aa = F.lp->h_atile[(size_t)d];
      pp.assign(1, 0);
      uu.assign(1, 0);
      aa.assign(1, 0);
      for (int f : T.by_depth[(size_t)d]) {
        pp.push_back(pp.back() + tiles(T.fs(f), T.np[(size_t)f]));
        uu.push_back(uu.back() + tiles(T.np[(size_t)f], T.nb[(size_t)f]));
        aa.push_back(aa.back() + (T.np[(size_t)f] + 31) / 32);
      }
      upload_vec(F.lp->ptile[(size_t)d], pp, s);
      upload_vec(F.lp->utile[(size_t)d], uu, s);
      upload_vec(F.lp->atile[(size_t)d], aa, s);
      for (int sl = 0; sl < 2; ++sl) {
        std::vector<int64_t> &cc = F.lp->h_ctile[sl][(size_t)d];
        cc.assign(1, 0);
        for (int c : F.lp->h_child[sl][(size_t)d]) cc.push_back(cc.back() + tiles(T.nb[(size_t)c], T.nb[(size_t)c]));
        upload_vec(F.lp->ctile[sl][(size_t)d], cc, s);
      }
    }
  }
  // the lockstep medium fronts of every (whole) level and the prefix arrays of their steps
  F.lp->mid.resize((size_t)nd);
  for (int d = 0; d < nd; ++d) {
    mf::LevelPlan::Mid &M = F.lp->mid[(size_t)d];
    std::vector<int> mid;
    mid_fronts_of(T.by_depth[(size_t)d], 0, (int)T.by_depth[(size_t)d].size(), mid, M.steps, M.h);
    M.count = (int)mid.size();
    if (M.count == 0) continue;
    upload_vec(M.list, mid, s);
    upload_vec(M.d, M.h, s);
    staged.push_back(std::move(mid));
  }
  SPL_HIP(hipStreamSynchronize(s));
  staged.clear();
  D.level_plan = F.lp;
  }  // build_plan
  plan_lock.unlock();
  lap(build_plan ? "level lists (built)" : "level lists (from the tree's cache)");
  DBuf<int> singular(1);
  SPL_HIP(hipMemsetAsync(singular.get(), 0, sizeof(int), s));
  set_factor_attributes();
  static std::atomic<uint64_t> attr_set{0};  // one bit per device
  if (first_use_on_this_device(attr_set)) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&front_factor_kernel<Z>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(Z ? kFrontLdsZ : 2 * kTileBytes)));
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mid_trsm_kernel<Z>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(Z ? kTrsmLdsZ : 2 * kTileBytes)));
    if (Z) {
      SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mid_diag_kernel<Z>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDiagLdsZ));
      SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mid_update_kernel<Z>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDiagLdsZ));
    }
    mark_used_on_this_device(attr_set);
  }
  F.make_streams();
  hipStream_t *side = F.side;
  auto region_of = [&](int d) { return (d & 1) ? region1.get() : region0.get(); };
  std::vector<DBuf<int>> mid_lists;          // (subtrees of a cut tree: their lists are made on the way) alive until the factorisation has run
  std::vector<DBuf<int64_t>> mid_prefixes;
  std::vector<std::vector<int64_t>> staged64;
  // [begin, end) of the ids lo..hi inside an ascending list
  auto range_of = [](const std::vector<int> &L, int lo, int hi, int &begin, int &end) {
    begin = (int)(std::lower_bound(L.begin(), L.end(), lo) - L.begin());
    end = (int)(std::upper_bound(L.begin(), L.end(), hi) - L.begin());
  };
  auto compact_fronts = [&](int d, int lo, int hi, hipStream_t q) {  // factor panels of the fronts lo..hi of level d -> arena
    int b0, b1;
    range_of(T.by_depth[(size_t)d], lo, hi, b0, b1);
    if (b1 == b0) return;
    const int64_t tp = F.lp->h_ptile[(size_t)d][(size_t)b1] - F.lp->h_ptile[(size_t)d][(size_t)b0];
    const int64_t tu = F.lp->h_utile[(size_t)d][(size_t)b1] - F.lp->h_utile[(size_t)d][(size_t)b0];
    if (tp > 0)
      hipLaunchKernelGGL(compact_kernel<Z>, dim3((unsigned)tp), dim3(256), 0, q, F.lp->level_lists[(size_t)d].get() + b0,
                         F.lp->ptile[(size_t)d].get() + b0, b1 - b0, F.view, 0);
    if (tu > 0)
      hipLaunchKernelGGL(compact_kernel<Z>, dim3((unsigned)tu), dim3(256), 0, q, F.lp->level_lists[(size_t)d].get() + b0,
                         F.lp->utile[(size_t)d].get() + b0, b1 - b0, F.view, 1);
  };
  // Assembly beside the factorisation (round 4): the fronts of a level live in the region the level two below it
  // used, which is free as soon as the level between them has taken its children's Schur complements.  So while
  // level d is factored, stream `aux` moves the panels of level d + 1 to the arena, then zeroes that region and
  // scatters the entries of A of level d - 1 into it: of the three parts of an assembly only the extend-add stays on
  // the chain of the levels (at 100^3: 12 of 27 ms).  SPL_MF_OVERLAP=0: everything on the main stream, as before.
  const bool overlap = !(getenv("SPL_MF_OVERLAP") && atoi(getenv("SPL_MF_OVERLAP")) == 0);
  hipStream_t aux = side[2 * kStreams];
  struct Events {
    hipEvent_t taken = nullptr, prepared = nullptr;  // extend-add of a level queued; region of the next level ready
    ~Events() {
      if (taken) (void)hipEventDestroy(taken);
      if (prepared) (void)hipEventDestroy(prepared);
    }
  } ev;
  SPL_HIP(hipEventCreateWithFlags(&ev.taken, hipEventDisableTiming));
  SPL_HIP(hipEventCreateWithFlags(&ev.prepared, hipEventDisableTiming));
  // the fronts lo..hi of level d start from zero in their region and receive their entries of A
  auto prepare_level = [&](int d, int lo, int hi, hipStream_t q) -> int64_t {
    int b0, b1;
    range_of(T.by_depth[(size_t)d], lo, hi, b0, b1);
    if (b1 == b0) return 0;
    int64_t extent = 0;
    for (int i = b0; i < b1; ++i) {
      const int f = T.by_depth[(size_t)d][(size_t)i];
      extent = std::max(extent, plan.foff[(size_t)f] + ((int64_t)T.ld[(size_t)f] * std::max(T.fs(f), 1) + 15) / 16 * 16);
    }
    SPL_HIP(hipMemsetAsync(region_of(d), 0, (size_t)ZM * (size_t)extent * sizeof(double), q));
    const int64_t groups = F.lp->h_atile[(size_t)d][(size_t)b1] - F.lp->h_atile[(size_t)d][(size_t)b0];
    if (groups > 0)
      hipLaunchKernelGGL(assemble_kernel<Z>, dim3((unsigned)groups), dim3(256), 0, q, F.lp->level_lists[(size_t)d].get() + b0,
                         F.lp->atile[(size_t)d].get() + b0, b1 - b0, F.view, d_perm, d_inv, d_Ap, d_Ai, d_Ax, d_Rp, d_Rj, d_Rx);
    return extent;
  };
  // levels dbot .. dtop (bottom-up) of the fronts with ids lo..hi.  children_saved: the children of
  // level plan.cut - 1 are subtree roots, already compacted, their Schur complements in the cut buffer
  auto process = [&](int lo, int hi, int dtop, int dbot, bool children_saved) {
    int prepared_level = -1;  // the level whose region stream `aux` has been told to prepare
    int64_t prepared_extent = 0;
    for (int d = dbot; d >= dtop; --d) {
      int b0, b1;
      range_of(T.by_depth[(size_t)d], lo, hi, b0, b1);
      if (b1 == b0) continue;
      // this level's fronts start from zero in their region, receive their entries of A ...
      int64_t extent = prepared_extent;
      if (prepared_level == d) SPL_HIP(hipStreamWaitEvent(s, ev.prepared, 0));
      else extent = prepare_level(d, lo, hi, s);
      // ... and the Schur complements of the children, one child slot after the other (two children
      // of a parent may touch the same entry: a fixed order keeps the sums reproducible)
      if (d + 1 < nd) {
        for (int sl = 0; sl < 2; ++sl) {
          int c0, c1;
          range_of(F.lp->h_child[sl][(size_t)d], lo, hi, c0, c1);
          if (c1 == c0) continue;
          const int64_t tc = F.lp->h_ctile[sl][(size_t)d][(size_t)c1] - F.lp->h_ctile[sl][(size_t)d][(size_t)c0];
          if (tc > 0)
            hipLaunchKernelGGL(extend_add_kernel<Z>, dim3((unsigned)tc), dim3(256), 0, s,
                               F.lp->child_lists[sl][(size_t)d].get() + c0, F.lp->ctile[sl][(size_t)d].get() + c0, c1 - c0,
                               F.view);
        }
        // the children are done with: keep their panels, their region is free again (subtree
        // roots were compacted when their subtree finished)
        if (overlap) {
          SPL_HIP(hipEventRecord(ev.taken, s));  // (everything level d + 1 queued on s lies before it as well)
          SPL_HIP(hipStreamWaitEvent(aux, ev.taken, 0));
        }
        if (!(children_saved && d + 1 == plan.cut)) compact_fronts(d + 1, lo, hi, overlap ? aux : s);
      }
      if (overlap && d + 1 < nd && d - 1 >= dtop) {
        prepared_extent = prepare_level(d - 1, lo, hi, aux);
        SPL_HIP(hipEventRecord(ev.prepared, aux));
        prepared_level = d - 1;
      }
      // small fronts: one launch, one workgroup each; large fronts: the multi-launch blocked
      // factorisation, independent fronts spread over side streams
      SPL_HIP(hipStreamSynchronize(s));
      if (timing) {
        double ea = 0.0, cp = 0.0;  // entries the extend-add moves, entries of the children's panels
        if (d + 1 < nd)
          for (int sl = 0; sl < 2; ++sl) {
            int c0, c1;
            range_of(F.lp->h_child[sl][(size_t)d], lo, hi, c0, c1);
            for (int i = c0; i < c1; ++i) {
              const int c = F.lp->h_child[sl][(size_t)d][(size_t)i];
              const double q = T.nb[(size_t)c], pp = T.np[(size_t)c];
              ea += symmetric ? q * (q + 1) / 2 : q * q;
              cp += pp * (pp + 2 * q);
            }
          }
        char what[96];
        snprintf(what, sizeof what, "level %d: assembly (zero %.0f, extend-add %.0f, panels %.0f MB)", d, ZM * extent * 8e-6,
                 ZM * ea * 24e-6, ZM * cp * 16e-6);
        lap(what);
      }
      int s0, s1;
      range_of(F.lp->h_small[(size_t)d], lo, hi, s0, s1);
      if (s1 > s0)
        hipLaunchKernelGGL(front_factor_kernel<Z>, dim3((unsigned)(s1 - s0)), dim3(256), Z ? kFrontLdsZ : 2 * kTileBytes, s,
                           F.lp->small_lists[(size_t)d].get() + s0, F.view, F.invs.get(), singular.get());
      // medium fronts: lockstep over block steps, one flat launch per phase and step
      {
        int count = 0, steps = 0;
        const int *dl = nullptr;
        const int64_t *dp = nullptr;
        const std::vector<int64_t> *hpp = nullptr;
        if (b0 == 0 && b1 == (int)T.by_depth[(size_t)d].size()) {  // the whole level: from the plan
          const mf::LevelPlan::Mid &M = F.lp->mid[(size_t)d];
          count = M.count;
          steps = M.steps;
          dl = M.list.get();
          dp = M.d.get();
          hpp = &M.h;
        } else {  // a subtree of a cut tree
          std::vector<int> mid;
          std::vector<int64_t> pre;
          mid_fronts_of(T.by_depth[(size_t)d], b0, b1, mid, steps, pre);
          count = (int)mid.size();
          if (count > 0) {
            mid_lists.emplace_back();
            mid_prefixes.emplace_back();
            upload_vec(mid_lists.back(), mid, s);
            upload_vec(mid_prefixes.back(), pre, s);
            staged.push_back(std::move(mid));
            staged64.push_back(std::move(pre));
            dl = mid_lists.back().get();
            dp = mid_prefixes.back().get();
            hpp = &staged64.back();
          }
        }
        if (count > 0) {
          const std::vector<int64_t> &hp = *hpp;
          hipLaunchKernelGGL(mid_diag_kernel<Z>, dim3((unsigned)count), dim3(256),
                             Z ? kDiagLdsZ : kTileBytes + 2 * NB * sizeof(double), s, dl, F.view, F.invs.get(), singular.get());
          for (int st = 0; st < steps; ++st) {
            const size_t base = ((size_t)st * 2) * (size_t)(count + 1);
            const int64_t nt = hp[base + (size_t)count], nu = hp[base + (size_t)(count + 1) + (size_t)count];
            if (nt > 0)
              hipLaunchKernelGGL(mid_trsm_kernel<Z>, dim3((unsigned)nt), dim3(256), Z ? kTrsmLdsZ : kTrsmLds, s, dl,
                                 dp + base, count, st, F.view, F.invs.get());
            if (nu > 0)
              hipLaunchKernelGGL(mid_update_kernel<Z>, dim3((unsigned)(nu + count)), dim3(256),
                                 Z ? kDiagLdsZ : kTileBytes + 2 * NB * sizeof(double), s, dl,
                                 dp + base + (size_t)(count + 1), count, st, F.view, F.invs.get(), singular.get(),
                                 mid_paired ? 1 : 0);
          }
        }
      }
      int turn = 0;
      for (int i = b0; i < b1; ++i) {
        const int f = T.by_depth[(size_t)d][(size_t)i];
        if (T.np[(size_t)f] == 0 || T.fs(f) <= mid_limit) continue;
        const size_t plane = Z ? (size_t)(((int64_t)T.ld[(size_t)f] * std::max(T.fs(f), 1) + 15) / 16 * 16) : 0;
        const Band b = dense_view(region_of(d) + (size_t)ZM * (size_t)plan.foff[(size_t)f], T.fs(f), T.ld[(size_t)f],
                                  symmetric ? 1 : 0, plane, pivot ? 1 : diag_form_without_interchanges(),
                                  pivot && d_rscale ? d_rscale + T.p0[(size_t)f] : nullptr);
        const int lane = turn++ % kStreams;
        factor_loop<Z>(b, T.np[(size_t)f], F.invs.get() + (size_t)ZM * (size_t)T.ioff[(size_t)f], singular.get(), side[lane],
                       side[kStreams + lane]);
      }
      if (turn > 0)
        for (int i = 0; i < kStreams && i < turn; ++i) SPL_HIP(hipStreamSynchronize(side[i]));
      if (timing) {
        int big = 0;
        double fl = 0.0;
        for (int i = b0; i < b1; ++i) {
          const int f = T.by_depth[(size_t)d][(size_t)i];
          big = std::max(big, T.fs(f));
          const double p = T.np[(size_t)f], q = T.nb[(size_t)f];
          fl += 2.0 / 3.0 * p * p * p + 2.0 * p * p * q + 2.0 * p * q * q;
        }
        char what[96];
        snprintf(what, sizeof what, "level %d: %d fronts, max %d, %.3g LU flops", d, b1 - b0, big, fl * ZM * ZM);
        lap(what);
      }
    }
    if (overlap) SPL_HIP(hipStreamSynchronize(aux));  // the panels of the last level but one are in the arena
  };
  if (plan.cut == 0) {
    process(0, nf - 1, 0, nd - 1, false);
    compact_fronts(0, 0, nf - 1, s);
  } else {
    for (int r : plan.roots) {  // one subtree after the other; its root's Schur complement is saved
      process(plan.first[(size_t)r], r, plan.cut, nd - 1, false);
      compact_fronts(plan.cut, r, r, s);
      const int nb = T.nb[(size_t)r];
      const int64_t ntile = (int64_t)((nb + 63) / 64) * ((nb + 3) / 4);
      if (ntile > 0) hipLaunchKernelGGL(save_cb_kernel<Z>, dim3((unsigned)ntile), dim3(256), 0, s, r, F.view);
      SPL_HIP(hipStreamSynchronize(s));
    }
    process(0, nf - 1, 0, plan.cut - 1, true);  // the top of the tree
    compact_fronts(0, 0, nf - 1, s);
  }
  SPL_HIP(hipStreamSynchronize(s));
  lap("levels");
  SPL_HIP(hipMemcpyAsync(&F.singular, singular.get(), sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  SPL_HIP(hipGetLastError());
  F.view.region[0] = F.view.region[1] = nullptr;  // the transient buffers are released here
  F.view.cut = nullptr;
  return Fp.release();
}
}  // namespace

// zfront: the arrays are those of the real embedding of a complex matrix and `tree` is the tree of the COMPLEX pattern
// (half the unknowns): fronts, panels and inverses are complex, in two planes (TreeView::zm)
mf::Factors *mf_factor(std::shared_ptr<const mf::Tree> tree, const int *d_Ap, const int *d_Ai, const double *d_Ax,
                       const int *d_Rp, const int *d_Rj, const double *d_Rx, const int *d_perm, const int *d_inv,
                       hipStream_t s, bool symmetric, bool zfront, bool pivot, const double *d_rscale) {
  return zfront ? mf_factor_t<true>(tree, d_Ap, d_Ai, d_Ax, d_Rp, d_Rj, d_Rx, d_perm, d_inv, s, symmetric, pivot, d_rscale)
                : mf_factor_t<false>(tree, d_Ap, d_Ai, d_Ax, d_Rp, d_Rj, d_Rx, d_perm, d_inv, s, symmetric, pivot, d_rscale);
}

// NR columns (c + r * stride) through the tree: up with L (or U^T), down with U (or L^T).  Per level:
// the fronts of up to big_solve rows by one workgroup each, the larger ones in lockstep — one flat
// launch per super-block step over all of them (Factors::BigLevel).
template <int MODE, int NR, bool Z = false>
static void launch_big_super(const mf::Factors &F, const mf::Factors::BigLevel &B, int kind, int step, double *work,
                             double *zbuf, hipStream_t s, int pivots_only = 0, double *x = nullptr, size_t xstride = 0) {
  constexpr size_t lds = (size_t)((SB + 2) * NB + solve_waves<NR>() * 64) * NR * sizeof(double);
  static std::atomic<uint64_t> attr_set{0};  // one mask per instantiation, one bit per device
  if (first_use_on_this_device(attr_set)) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&big_super_kernel<MODE, NR, Z>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    mark_used_on_this_device(attr_set);
  }
  const unsigned groups = B.total(kind, step);
  if (groups > 0)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(big_super_kernel<MODE, NR, Z>), dim3(groups), dim3(solve_waves<NR>() * 64), lds, s, B.list.get(),
                       B.prefix(kind, step), B.count, step, F.view, F.invs.get(), work, zbuf, B.row_blocks, pivots_only, x, xstride);
}

template <int MODE, int NR, bool Z = false>
static void launch_big_pipe(const mf::Factors &F, const mf::Factors::BigLevel &B, int kind, double *work, double *zbuf,
                            double *cbuf, hipStream_t s, double *x = nullptr, size_t xstride = 0) {
  constexpr size_t lds = (size_t)((SB + 2) * NB + solve_waves<NR>() * 64) * NR * sizeof(double);
  static std::atomic<uint64_t> attr_set{0};  // one mask per instantiation, one bit per device
  if (first_use_on_this_device(attr_set)) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&big_super_pipe_kernel<MODE, NR, Z>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    mark_used_on_this_device(attr_set);
  }
  for (int k = 0; k <= B.steps; ++k) {
    const unsigned groups = B.total(kind, k);
    if (groups > 0)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(big_super_pipe_kernel<MODE, NR, Z>), dim3(groups), dim3(solve_waves<NR>() * 64), lds, s,
                         B.list.get(), B.prefix(kind, k), B.count, k, F.view, F.invs.get(), work, zbuf, cbuf, B.row_blocks, x, xstride);
  }
}

// ---- chains (mf_chain.hpp): built once per factorisation, by its first untransposed solve with one right-hand side
template <bool Z, bool TR>
static void build_chain_t(const mf::Factors &F, hipStream_t s) {
  mf::Factors::Chain &Cn = TR ? F.chain_t : F.chain;
  const mf::Tree &T = *F.tree;
  int S = 512;
  if (const char *e = getenv("SPL_MF_CHAIN")) {  // 0: none (ablation); 256: shorter blocks
    const int v = atoi(e);
    if (v <= 0) return;
    if (v == 256 || v == 512 || v == 1024) S = v;
  }
  const auto t0 = std::chrono::steady_clock::now();
  const int nd = T.maxdepth + 1;
  std::vector<int64_t> off((size_t)T.nfronts, -1);
  std::vector<int> item_f, item_k;
  std::vector<int64_t> pre{0};
  int64_t elems = 0;
  Cn.levels.assign((size_t)nd, mf::Factors::Chain::Level());
  Cn.h.clear();
  for (int d = 0; d < nd; ++d) {
    std::vector<int> large;  // (the order of LevelPlan::BigLevel::list)
    for (int f : T.by_depth[(size_t)d])
      if (T.fs(f) > F.lp->big_solve && T.np[(size_t)f] > 0) large.push_back(f);
    mf::Factors::Chain::Level &L = Cn.levels[(size_t)d];
    L.count = (int)large.size();
    if (L.count == 0) continue;
    for (int f : large) L.steps = std::max(L.steps, (T.np[(size_t)f] + S - 1) / S);
    int maxnp = 0;
    for (int f : large) maxnp = std::max(maxnp, T.np[(size_t)f]);
    L.lead_rows[0] = maxnp <= chain::kWidePivots ? chain::rows_wide(Z) : chain::kRows;
    L.lead_rows[1] = maxnp <= chain::kWidePivots ? 64 : chain::kRows / 2;
    L.base = Cn.h.size();
    Cn.h.resize(Cn.h.size() + (size_t)4 * (size_t)(L.steps + 1) * (size_t)(L.count + 1), 0);
    // blocks of 64 rows per bulk workgroup: ONE (the substitution steps take four on the levels that fill the chip, because
    // every workgroup of theirs redoes the solve inside the super block first; a bulk group of a chain only loads the
    // solved block — and the root front of config C5 has a hundred groups of four blocks per launch for 256 CUs)
    const int rbk = getenv("SPL_MF_CHAIN_RB") ? std::max(1, atoi(getenv("SPL_MF_CHAIN_RB"))) : 1;
    L.row_blocks = rbk;
    for (int pc = 0; pc < 4; ++pc)
      for (int l = 0; l <= L.steps; ++l) {
        const int pass = pc & 1, lead_rows = L.lead_rows[pc >> 1];
        int64_t *p = Cn.h.data() + Cn.at(d, pc, l);
        for (int i = 0; i < L.count; ++i) {
          const int np = T.np[(size_t)large[(size_t)i]], K = (np + S - 1) / S;
          int64_t g = 0;
          if (l < K) {  // lead groups of block l of the pass
            const int k = pass == 0 ? l : K - 1 - l;
            g += (std::min(S, np - k * S) + lead_rows - 1) / lead_rows;
          }
          if (l >= 1 && l - 1 < K) {  // bulk groups of the block before: the rows beyond the next block
            const int kb = pass == 0 ? l - 1 : K - l;
            const int rest = pass == 0 ? np - (kb + 2) * S : (kb - 1) * S;
            if (rest > 0) g += ((rest + 63) / 64 + rbk - 1) / rbk;
          }
          p[i + 1] = p[i] + g;
        }
      }
    for (int f : large) {
      const int np = T.np[(size_t)f], K = (np + S - 1) / S;
      off[(size_t)f] = elems;
      elems += (int64_t)2 * np * chain::ld_of(np, S);
      for (int up = 0; up < 2; ++up)
        for (int k = 0; k < K; ++k) {
          const int jbs = std::min(S, np - k * S);
          int tiles = (jbs + 63) / 64;
          if (up == 0 && k > 0) tiles += S / 64;
          if (up == 1 && k < K - 1) tiles += (std::min(S, np - (k + 1) * S) + 63) / 64;
          item_f.push_back(f);
          item_k.push_back(k | (up << 30));
          pre.push_back(pre.back() + tiles);
        }
    }
  }
  if (item_f.empty()) return;
  try {
    constexpr int ZM = Z ? 2 : 1;
    Cn.buf.alloc((size_t)ZM * (size_t)elems + 256);
    SPL_HIP(hipMemsetAsync(Cn.buf.get(), 0, ((size_t)ZM * (size_t)elems + 256) * sizeof(double), s));
    upload_vec(Cn.off, off, s);
    upload_vec(Cn.d, Cn.h, s);
    DBuf<int> d_f, d_k, bad(1);
    DBuf<int64_t> d_pre;
    upload_vec(d_f, item_f, s);
    upload_vec(d_k, item_k, s);
    upload_vec(d_pre, pre, s);
    SPL_HIP(hipMemsetAsync(bad.get(), 0, sizeof(int), s));
    const chain::View cv{Cn.buf.get(), Cn.off.get(), (size_t)elems, S};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(chain_build_kernel<Z, TR>), dim3((unsigned)pre.back()), dim3(256), 0, s, d_f.get(), d_k.get(), d_pre.get(),
                       (int)item_f.size(), F.view, F.invs.get(), cv, bad.get());
    int hbad = 0;
    SPL_HIP(hipMemcpyAsync(&hbad, bad.get(), sizeof(int), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    SPL_HIP(hipGetLastError());
    if (hbad) {  // an entry too large to trust (or not finite): keep the substitution steps
      Cn.buf.release();
      return;
    }
  } catch (const DeviceError &) {  // no memory for the chains: the walk does without
    Cn.buf.release();
    (void)hipGetLastError();
    return;
  }
  Cn.span = S;
  Cn.elems = elems;
  Cn.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  Cn.ok.store(true, std::memory_order_release);
  if (getenv("SPL_MF_TIMING"))
    fprintf(stderr, "[mf_solve] chains%s: span %d, %zu blocks, %.2f GB, built in %.2f ms\n", TR ? " (transposed systems)" : "", S, item_f.size(),
            (Z ? 2 : 1) * elems * 8e-9, Cn.build_ms);
}

template <int MODE, int NR, bool Z = false>
static void launch_big_chain(const mf::Factors &F, int depth, double *work, double *zbuf, hipStream_t s, double *x = nullptr,
                             size_t xstride = 0) {
  const mf::Factors::Chain &Cn = MODE >= 2 ? F.chain_t : F.chain;
  const mf::Factors::Chain::Level &L = Cn.levels[(size_t)depth];
  const mf::Factors::BigLevel &B = F.lp->big[(size_t)depth];
  constexpr int cls = NR >= 16 ? 1 : 0;  // (workgroups of 8 wavefronts: grids of their own)
  // LDS: the bulk groups' v, res and partial sums; levels without bulk groups (one block per front): the lead groups' u
  const size_t lds_bulk = (size_t)(Cn.span + NB + solve_waves<NR>() * 64) * NR * sizeof(double);
  const size_t lds_lead = NR <= 2 ? (size_t)2 * Cn.span * NR * sizeof(double) : (size_t)kChainSeg * NR * sizeof(double);
  const size_t lds = L.steps > 1 ? std::max(lds_bulk, lds_lead) : lds_lead;
  static std::atomic<uint64_t> attr_set{0};  // one mask per instantiation, one bit per device
  if (first_use_on_this_device(attr_set)) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&big_chain_kernel<MODE, NR, Z>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)std::min<size_t>(160 * 1024, (size_t)(1024 + NB + solve_waves<NR>() * 64) * NR * sizeof(double))));
    mark_used_on_this_device(attr_set);
  }
  const chain::View cv{Cn.buf.get(), Cn.off.get(), (size_t)Cn.elems, Cn.span};
  if constexpr (NR <= 2) {
    if (L.lead_rows[0] == chain::rows_wide(Z)) {  // a level of small pivot blocks: one launch of lead groups, two workgroups per CU
      const size_t at = Cn.at(depth, MODE & 1, 0);
      const unsigned groups = (unsigned)Cn.h[at + (size_t)L.count];
      if (groups > 0)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(big_chain_wide_kernel<MODE, NR, Z>), dim3(groups), dim3(solve_waves<NR>() * 64), lds_lead, s,
                           B.list.get(), Cn.d.get() + at, B.count, F.view, cv, work, zbuf, x, xstride);
      return;
    }
  }
  for (int l = 0; l <= L.steps; ++l) {
    const size_t at = Cn.at(depth, (MODE & 1) + 2 * cls, l);
    const unsigned groups = (unsigned)Cn.h[at + (size_t)L.count];
    if (groups > 0)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(big_chain_kernel<MODE, NR, Z>), dim3(groups), dim3(solve_waves<NR>() * 64), lds, s,
                         B.list.get(), Cn.d.get() + at, B.count, l, F.view, cv, work, zbuf, L.row_blocks, L.lead_rows[cls], x, xstride);
  }
}

// the solve lists of one depth
struct SolveLevel {
  const int *solve_list;
  int solve_count;
  const int *child_list[2];  // the children (by slot) of the fronts of this depth, and the largest boundary among them
  int child_count[2], child_maxnb;
  const mf::Factors::BigLevel *big;
  int depth;
};
static SolveLevel whole_level(const mf::LevelPlan &lp, int d, int nd) {
  SolveLevel L;
  L.depth = d;
  L.solve_list = lp.solve_lists[(size_t)d].get();
  L.solve_count = lp.solve_counts[(size_t)d];
  L.child_maxnb = d + 1 < nd ? std::max(lp.child_maxnb[0][(size_t)d], lp.child_maxnb[1][(size_t)d]) : 0;
  for (int sl = 0; sl < 2; ++sl) {
    L.child_list[sl] = d + 1 < nd ? lp.child_lists[sl][(size_t)d].get() : nullptr;
    L.child_count[sl] = d + 1 < nd ? lp.child_counts[sl][(size_t)d] : 0;
  }
  L.big = &lp.big[(size_t)d];
  return L;
}
// NR columns through the tree, up with L (or U^T) and down with U (or L^T), level by level on the caller's stream.
// What bounds a walk at the FEAST sizes (kernel trace of a 100^3 solve, profiles/r05_solve_100_kernel_trace.txt): 931
// kernels of 5 - 40 us in 16 ms, the GPU "busy" 97 % of the time, launch gaps 0.5 ms in all — the time is the fixed cost
// and the inner latency chain of each kernel, not the gaps between them, so a captured graph buys nothing.  Two other
// shapes of the walk were built and measured in round 5, both bit-identical to this one and both slower
// (profiles/r05_solve_walk_experiments.txt): a whole subtree per workgroup below a cut depth (one launch per direction
// for the deep levels; a front of 1 000 rows takes a single workgroup 100 us: 19.9 ms at depth 14, 15.4 at 18 against
// 15.5), and the subtrees of the top nodes side by side on streams of their own (2 branches 16.1 ms, 4 branches 18.3:
// kernels of different streams do not overlap their fixed costs here, the dispatcher takes them one at a time).
template <bool TRANS, int NR, bool Z = false>
static void solve_columns_on_tree(const mf::Factors &F, double *c, size_t stride, double *work, double *zbuf,
                                  double *gscr, double *cbuf, hipStream_t s) {
  const mf::Tree &T = *F.tree;
  const int nd = T.maxdepth + 1;
  double *invs = F.invs.get();
  constexpr int FWD = TRANS ? 2 : 0, BWD = TRANS ? 3 : 1;
  const bool timing = getenv("SPL_MF_TIMING") != nullptr;  // per-level times on stderr (diagnostic; the walk is then not split)
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *dir, int d) {
    if (!timing) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(now - t_last).count();
    // bytes of factor panels this phase reads: forward L11 / 2 + L21 (transposed: U11 / 2, U12 in the boundary kernel),
    // backward U11 / 2 + U12 — half the pivot block and one coupling panel either way
    const bool large = dir[0] == 'u' ? dir[3] == 'l' : dir[5] == 'l';
    double bytes = 0.0;
    int maxnp = 0;
    for (int f : T.by_depth[(size_t)d]) {
      const bool is_big = T.fs(f) > F.lp->big_solve && T.np[(size_t)f] > 0;
      if (is_big != large) continue;
      const double np = T.np[(size_t)f], nb = T.fs(f) - T.np[(size_t)f];
      bytes += (np * np * 0.5 + np * nb) * 8.0 * F.zm;
      maxnp = std::max(maxnp, T.np[(size_t)f]);
    }
    fprintf(stderr, "[mf_solve] %s level %2d: %4d large of %6zu fronts %8.2f ms  %8.1f MB %6.2f TB/s  steps %d max np %d\n", dir, d,
            F.lp->big[(size_t)d].count, T.by_depth[(size_t)d].size(), ms, bytes * 1e-6, ms > 0 ? bytes / ms * 1e-9 : 0.0,
            large ? F.lp->big[(size_t)d].steps : 0, maxnp);
    t_last = now;
  };
  // (split: SPL_MF_SPLIT_FWD=0 restores the steps over all fs rows of rounds 1 - 3; pipe: SPL_MF_PIPE=0 restores one
  // launch per step with the whole chain in it: ablations, read once per walk)
  const char *sf = getenv("SPL_MF_SPLIT_FWD"), *pe = getenv("SPL_MF_PIPE");
  const bool split_fwd = !(sf && sf[0] == '0'), pipe_on = !(pe && pe[0] == '0');
  // the pivot blocks as chains of matrix-vector products (mf_chain.hpp); the transposed systems have a set of their own
  const char *cm = getenv("SPL_MF_CHAIN_MULTI");  // 0: chains for one right-hand side only (ablation)
  // (8 / 16 columns: blocks of at most 512 pivots — the bulk groups' LDS)
  const mf::Factors::Chain &Ch = TRANS ? F.chain_t : F.chain;
  const bool chain_on = (TRANS || split_fwd) && pipe_on && Ch.ok && (NR <= 2 || (Ch.span <= 512 && !(cm && cm[0] == '0')));

  // one level on the way up: children's boundaries into their parents, the one-workgroup fronts, the large ones in lockstep
  auto up_level = [&](const SolveLevel &L, double *scr, hipStream_t q) {
    if (L.child_count[0] + L.child_count[1] > 0) {
      const int64_t most = (int64_t)L.child_maxnb * NR;  // entries of the largest child
      const unsigned share = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (most + 2047) / 2048));
      for (int sl = 0; sl < 2; ++sl)
        if (L.child_count[sl] > 0)
          hipLaunchKernelGGL(solve_gather_kernel<NR>, dim3((unsigned)L.child_count[sl], share), dim3(256), 0, q, L.child_list[sl], F.view, work);
    }
    if (L.solve_count > 0)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(solve_forward_kernel<TRANS, NR, Z>), dim3((unsigned)L.solve_count),
                         dim3(solve_threads<NR>()), 0, q, L.solve_list, F.view, invs, work);
    const mf::Factors::BigLevel &B = *L.big;
    if (B.count > 0) {
      // untransposed: the boundary rows get their updates from a streaming product of their own once the pivots are
      // solved; transposed: U11^T on the pivots, then the boundary with U12^T
      const bool pivots_only = !TRANS && split_fwd;
      // (pipelined steps with one or two columns only: with 8 or 16 the in-super-block solve is instruction-bound, and the
      // bulk groups' reload of the solved super block costs more than the overlap gains: FEAST 80^3 solve stage 2.95 -> 3.05 s)
      const bool pipe = pipe_on && (TRANS || pivots_only) && NR <= 2;
      if (chain_on && (TRANS || pivots_only)) launch_big_chain<FWD, NR, Z>(F, L.depth, work, zbuf, q);
      else if (pipe) launch_big_pipe<FWD, NR, Z>(F, B, 13, work, zbuf, cbuf, q);
      else
        for (int k = 0; k < B.steps; ++k) launch_big_super<FWD, NR, Z>(F, B, (TRANS || pivots_only) ? 1 : 0, k, work, zbuf, q, pivots_only ? 1 : 0);
      if (pivots_only && B.total(10) > 0) {
        constexpr size_t lds = (size_t)std::max(kGemvChunk, gemv_waves<NR>() * 64) * NR * sizeof(double);
        static std::atomic<uint64_t> attr_set{0};  // one mask per instantiation, one bit per device
        if (first_use_on_this_device(attr_set)) {
          SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&big_gemv_chunk_kernel<NR, Z, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          mark_used_on_this_device(attr_set);
        }
        hipLaunchKernelGGL(HIP_KERNEL_NAME(big_gemv_chunk_kernel<NR, Z, true>), dim3(B.total(10)), dim3(gemv_waves<NR>() * 64), lds, q,
                           B.list.get(), B.prefix(10), B.count, F.view, work, zbuf, B.prefix(12), scr);
        if (B.total(11) > 0)
          hipLaunchKernelGGL(HIP_KERNEL_NAME(big_gemv_reduce_kernel<NR, true>), dim3(B.total(11)), dim3(256), 0, q, B.list.get(),
                             B.prefix(11), B.count, F.view, work, zbuf, B.prefix(12), scr);
      }
      if (TRANS && B.total(3) > 0)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(big_boundary_t_kernel<NR, Z>), dim3(B.total(3)), dim3(256), 0, q, B.list.get(),
                           B.prefix(3), B.count, F.view, work, zbuf);
    }
  };
  // ... and on the way down
  auto down_level = [&](const SolveLevel &L, double *scr, hipStream_t q) {
    if (L.solve_count > 0)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(solve_backward_kernel<TRANS, NR, Z>), dim3((unsigned)L.solve_count),
                         dim3(solve_threads<NR>()), 0, q, L.solve_list, F.view, invs, work, c, stride);
    const mf::Factors::BigLevel &B = *L.big;
    if (B.count > 0) {
      if (B.total(4) > 0) {  // (fronts with a boundary; its part of the solution is read from x inside the product)
        {
          constexpr size_t lds = (size_t)(TRANS ? kGemvChunk : std::max(kGemvChunk, gemv_waves<NR>() * 64)) * NR * sizeof(double);
          static std::atomic<uint64_t> attr_set{0};  // one mask per instantiation, one bit per device
          if (first_use_on_this_device(attr_set)) {
            if (TRANS)
              SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&big_gemv_chunk_t_kernel<NR, Z>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            else
              SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&big_gemv_chunk_kernel<NR, Z>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            mark_used_on_this_device(attr_set);
          }
          if (TRANS)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(big_gemv_chunk_t_kernel<NR, Z>), dim3(B.total(7)), dim3(gemv_waves<NR>() * 64), lds,
                               q, B.list.get(), B.prefix(7), B.count, F.view, work, zbuf, B.prefix(9), scr, (const double *)c, stride);
          else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(big_gemv_chunk_kernel<NR, Z>), dim3(B.total(7)), dim3(gemv_waves<NR>() * 64), lds, q,
                               B.list.get(), B.prefix(7), B.count, F.view, work, zbuf, B.prefix(9), scr, (const double *)c, stride);
          if (B.total(8) > 0)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(big_gemv_reduce_kernel<NR>), dim3(B.total(8)), dim3(256), 0, q, B.list.get(),
                               B.prefix(8), B.count, F.view, work, zbuf, B.prefix(9), scr);
        }
      }
      // the pivot block alone; columns of Z / W are fs apart
      // (the solved pivots go to x from inside these steps: round 4 had a scatter launch per level behind them)
      if (chain_on) launch_big_chain<BWD, NR, Z>(F, L.depth, work, zbuf, q, c, stride);
      else if (pipe_on && NR <= 2) launch_big_pipe<BWD, NR, Z>(F, B, 14, work, zbuf, cbuf, q, c, stride);
      else
        for (int k = 0; k < B.steps; ++k) launch_big_super<BWD, NR, Z>(F, B, 2, k, work, zbuf, q, 0, c, stride);
    }
  };

  if (timing) (void)hipStreamSynchronize(s);
  t_last = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(HIP_KERNEL_NAME(solve_init_kernel<NR, Z>), dim3((unsigned)T.nfronts, 8), dim3(256), 0, s, F.view, c, stride,
                     work);
  for (int d = nd - 1; d >= 0; --d) {
    const SolveLevel L = whole_level(*F.lp, d, nd);
    if (!timing) { up_level(L, gscr, s); continue; }
    // (timed: the one-workgroup fronts and the large ones apart, as the level table of profiles/ wants them)
    SolveLevel small = L, large = L;
    static const mf::Factors::BigLevel none;
    small.big = &none;
    large.solve_count = 0;
    large.child_count[0] = large.child_count[1] = 0;
    up_level(small, gscr, s);
    lap("up small  ", d);
    up_level(large, gscr, s);
    lap("up large  ", d);
  }
  for (int d = 0; d < nd; ++d) {
    const SolveLevel L = whole_level(*F.lp, d, nd);
    if (!timing) { down_level(L, gscr, s); continue; }
    SolveLevel small = L, large = L;
    static const mf::Factors::BigLevel none;
    small.big = &none;
    large.solve_count = 0;
    down_level(small, gscr, s);
    lap("down small", d);
    down_level(large, gscr, s);
    lap("down large", d);
  }
}

// columns of c (device, new ordering, column r at d_c + r * stride) <- (P A P^T)^-1 c or its transpose;
// k > 1: the caller allocates a multiple of kSolveGroup columns (zero-padded), taken 8 at a time
void mf_solve(const mf::Factors *Fp, int sys, double *d_c, int k, size_t stride, hipStream_t s) {
  const mf::Factors &F = *Fp;
  const mf::Tree &T = *F.tree;
  if (T.n == 0 || k == 0) return;
  const bool z = F.zm == 2;
  // complex fronts: the k columns are packed complex vectors of T.n entries (stride doubles apart); a complex right-hand
  // side is two real columns of the work matrices: one at a time, or eight together (= kSolveGroup columns of the caller)
  constexpr int kGroupZ = 8;
  const int nr = z ? (k == 1 ? 2 : 2 * kGroupZ) : (k == 1 ? 1 : kSolveGroup);
  const size_t elems = ((size_t)T.work_elems * 3 + (size_t)F.lp->gemv_scratch) * (size_t)nr;  // work, z and carry matrices of all fronts, scratch
  // every walk of this call on stream q, with the work matrices at `base`
  auto run = [&](double *base, hipStream_t q) {
    double *wk = base, *zb = base + (size_t)T.work_elems * nr, *gs = zb + (size_t)T.work_elems * nr;
    double *cb = gs + (size_t)F.lp->gemv_scratch * nr;
    if (z) {
      if (k == 1) {
        if (sys == 0) solve_columns_on_tree<false, 2, true>(F, d_c, stride, wk, zb, gs, cb, q);
        else solve_columns_on_tree<true, 2, true>(F, d_c, stride, wk, zb, gs, cb, q);
      } else {
        for (int c0 = 0; c0 < k; c0 += kGroupZ) {  // (k is a multiple of kSolveGroup = 8 here: zero-padded by the caller)
          double *c = d_c + (size_t)c0 * stride;
          if (sys == 0) solve_columns_on_tree<false, 2 * kGroupZ, true>(F, c, stride, wk, zb, gs, cb, q);
          else solve_columns_on_tree<true, 2 * kGroupZ, true>(F, c, stride, wk, zb, gs, cb, q);
        }
      }
    } else if (k == 1) {
      if (sys == 0) solve_columns_on_tree<false, 1>(F, d_c, stride, wk, zb, gs, cb, q);
      else solve_columns_on_tree<true, 1>(F, d_c, stride, wk, zb, gs, cb, q);
    } else {
      for (int c0 = 0; c0 < k; c0 += kSolveGroup) {
        double *c = d_c + (size_t)c0 * stride;
        if (sys == 0) solve_columns_on_tree<false, kSolveGroup>(F, c, stride, wk, zb, gs, cb, q);
        else solve_columns_on_tree<true, kSolveGroup>(F, c, stride, wk, zb, gs, cb, q);
      }
    }
  };
  if (sys == 0) std::call_once(F.chain.once, [&] { z ? build_chain_t<true, false>(F, s) : build_chain_t<false, false>(F, s); });
  else std::call_once(F.chain_t.once, [&] { z ? build_chain_t<true, true>(F, s) : build_chain_t<false, true>(F, s); });
  DBuf<double> both(elems);
  run(both.get(), s);
  SPL_HIP(hipStreamSynchronize(s));  // the work matrices are freed on return
}

}  // namespace spl

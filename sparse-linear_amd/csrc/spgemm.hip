// spgemm.hip — C = A * B on CSC operands (reference: mm, Sparse.hs:691-702, with
// the dense sparse accumulator of Data/Vector/Sparse/ScatterGather.hs).
//
// Reference semantics kept exactly:
//   * column j of C has the UNION pattern of { A[:,k] : B[k,j] stored }; numerical
//     cancellation keeps a stored zero (the mask is set regardless of the value);
//   * row indices ascend inside a column; pointers are the exclusive prefix sum;
//   * C[i,j] = fold (\acc k -> acc + A[i,k]*B[k,j]) 0 over k ascending, each multiply and
//     add separately rounded.  Every path below delivers the contributions of one output
//     entry in ascending-k order, so VALUES are bit-identical to the reference order.
//
// MI355X design (expand - sort - compress in LDS): the reference's O(nrows) dense
// accumulator per column is replaced by work proportional to the column's number of
// intermediate products P_j (an upper bound of its nnz), by which columns are binned:
//   bin S  P <= 256    one wavefront per column   (4 columns per workgroup, no barrier)
//   bin M  P <= 2048   one workgroup per column (<= 256 entries in the column of B), 8 per CU
//   bin X  P <= 4096   one workgroup per column, 1-3 workgroups per CU
//   bin L  more        one workgroup per column, dense accumulator in HBM from a small pool
//                      (the reference's own data structure, only for the heavy columns)
// ESC step for bins S/M/X: (1) the column's B entries and the extents of the A columns they
// select are staged in LDS and prefix-summed, so product t of the column is found by a binary
// search — ALL products of the column are gathered from HBM at once (no dependent chain per
// k); (2) the keys (row, t) — t = position in k-then-row order breaks ties, which keeps equal
// rows in ascending-k order — are sorted in LDS: the products of one B entry are a column of
// A, already ascending, so the keys go through a merge tree over these runs (merge-path): packed
// (row << bits | t) in 32 bits when the rows fit, else the row with t as a 16-bit payload;
// (3) each run of equal rows is summed left to right by its first element and written, already
// in ascending row order.  The products a*b are NOT kept in LDS: the kernel is
// latency-bound, LDS per column sets the workgroups per CU, and a run head recomputes its
// products from operands the expansion has just pulled into L2.
// One pass when 24 B x products fits in half of the free HBM (every column is written at its
// upper-bound slot, one copy compacts); otherwise symbolic (count distinct) + scan + numeric.
// HBM/latency-bound integer + fp64 work; no MFMA (no dense contraction).
#include <atomic>
#include <chrono>
#include <type_traits>
#include "common.hpp"

namespace spl {

namespace {

constexpr int kSmallProducts = 256;
constexpr int kMediumProducts = 2048, kMediumB = 256;
constexpr int kLargeProducts = 4096, kLargeB = 2048;
// Columns the ordered form's own kernel takes (the others are computed beforehand by the listed kernels): at most
// `cap` products and `nbcap` entries of B.  The two sizes fix its LDS image — workspace + two (values, keys)
// buffers.  Two shapes are compiled: 1536 / 96 (40 640 bytes: FOUR workgroups per CU; the kernel lives on
// overlapping the latencies of its resident workgroups — C4: 13.8 ms against 19.5 ms at three) and 2048 / 128
// (53 440 bytes, three per CU) for products whose columns sit between 1537 and 2048 products, which the small shape
// would send to the listed kernels (scale 20, edge factor 44: 0.119 s against 0.044 s).  The host picks per product.
constexpr int kOrdCapSmall = 1536, kOrdPNbSmall = 96;
constexpr int kOrdPBuckets = 512;
constexpr int kOrdCapLarge = 2048, kOrdPNbLarge = 128;
// Round 5: the image is a workspace + a RING of entry slots whose two ends hold the column being sorted and the pending
// one (OrdPipeLds).  With 2 CAP slots the ends are round 4's two full buffers.  A NARROW ring (2 400 slots = 32 576 bytes,
// five workgroups per CU: columns of C4 have 1 024 products on average) was built and measured (SPL_SPGEMM_RING=narrow):
// 49 ms against 13.4 — when the two columns do not fit together (9 % of C4's pairs) the pending one must be finished
// BEFORE the column in hand is sorted, i.e. while its ticket is held unpublished, and every such delay ages the look-back
// of a thousand later columns: the chain tips over into a convoy (1.9 M sleeps against a hundred per million).  Even
// with overflow ruled out (SPL_SPGEMM_ORD_CAP=1200) five workgroups per CU at 96 registers were slower than four at 109
// (14.6 against 13.7 ms): the kernel no longer lives on occupancy.  Kept as an ablation, not used.
constexpr int kOrdRingSmall = 2 * kOrdCapSmall, kOrdRingSmallNarrow = 2400, kOrdRingLarge = 2 * kOrdCapLarge;
constexpr int kOrdAdmitSmall = kOrdCapSmall, kOrdAdmitLarge = kOrdCapLarge;
static_assert(kOrdAdmitSmall <= 2 * kOrdRingSmallNarrow / 3, "the merge-tree fallback's second key buffer must fit the narrow ring");
constexpr int kMaxPool = 2048;  // dense-accumulator slots = resident workgroups of spgemm_dense_kernel: eight per CU (it lives on latency overlap)

inline unsigned blocks_for(int64_t n, int per_block) {
  int64_t b = (n + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : b);
}

struct Csc {
  const int *p;
  const int *i;
  const double *x;
};

// bins: 0 empty, 1 S, 2 M, 3 X, 4 L
__device__ inline int bin_of(int64_t np, int nb) {
  if (np == 0) return 0;
  // (a column of B may be long and still have few products — most of the columns of A it selects empty: the
  // one-wavefront kernel stages at most kSmallProducts entries of B, such a column goes to a bin that holds it)
  if (np <= kSmallProducts && nb <= kSmallProducts) return 1;
  if (np <= kMediumProducts && nb <= kMediumB) return 2;
  if (np <= kLargeProducts && nb <= kLargeB) return 3;
  return 4;
}

__global__ __launch_bounds__(256) void products_kernel(Csc A, Csc B, int64_t ncolsB,
                                                       int64_t *__restrict__ nprod,
                                                       int64_t *__restrict__ medium_list,
                                                       int64_t *__restrict__ xlarge_list,
                                                       int64_t *__restrict__ dense_list,
                                                       int *__restrict__ list_counts, int ordered, int ord_cap,
                                                       int ord_nb, int x_heavy) {
  // 8 lanes per column of B: the extents of the selected columns of A are independent loads
  const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  const int64_t j = g < ncolsB ? g : ncolsB - 1;
  int64_t n = 0;
  const int qs = B.p[j], qe = B.p[j + 1];
  for (int q = qs + sub; q < qe; q += 8) {
    const int k = B.i[q];
    n += A.p[k + 1] - A.p[k];
  }
  n += __shfl_xor(n, 1, 64);
  n += __shfl_xor(n, 2, 64);
  n += __shfl_xor(n, 4, 64);
  // bin lists: one global atomic per workgroup and bin (a million columns of one bin would
  // otherwise serialise on one address), positions inside the workgroup from LDS counters
  // (bin X is listed in two halves of one array: columns with at most kMediumB entries of B from the front — they run
  // with the small staging area, four workgroups per CU instead of two —, the others from the back: slot 3)
  __shared__ int local_count[4], base[4];
  if (threadIdx.x < 4) local_count[threadIdx.x] = 0;
  __syncthreads();
  const bool owner = g < ncolsB && sub == 0;
  int bin = 0, pos = 0;
  bool xback = false;
  if (owner) {
    nprod[j] = n;
    bin = bin_of(n, qe - qs);
    if (x_heavy && bin == 3) bin = 4;  // columns of 2049 .. 4096 products with the heavy ones (row-range kernel)
    // ordered form: the columns its own kernel handles (<= ord_cap products, <= ord_nb entries of B) are not listed;
    // every other column must be, also a light one with a long column of B (bin S has no list: it goes with M)
    if (ordered && n > 0) bin = (n <= ord_cap && qe - qs <= ord_nb) ? 1 : (bin < 2 ? 2 : bin);
    xback = bin == 3 && qe - qs > kMediumB;
    if (bin >= 2) pos = atomicAdd(&local_count[xback ? 3 : bin - 2], 1);
  }
  __syncthreads();
  if (threadIdx.x < 4 && local_count[threadIdx.x] > 0)
    base[threadIdx.x] = atomicAdd(&list_counts[threadIdx.x], local_count[threadIdx.x]);
  __syncthreads();
  if (owner && bin >= 2) {
    if (xback) {
      xlarge_list[ncolsB - 1 - (base[3] + pos)] = j;
    } else {
      int64_t *list = bin == 2 ? medium_list : bin == 3 ? xlarge_list : dense_list;
      list[base[bin - 2] + pos] = j;
    }
  }
}

// The bin lists of the ORDERED form from the per-column product counts the first pass left behind: no second walk
// over the columns of B and the extents of A (round 4 ran products_kernel twice: 0.23 ms of C4's 16.7).
__global__ __launch_bounds__(256) void relist_ordered_kernel(Csc B, int64_t ncolsB, const int64_t *__restrict__ nprod,
                                                             int64_t *__restrict__ medium_list, int64_t *__restrict__ xlarge_list,
                                                             int64_t *__restrict__ dense_list, int *__restrict__ list_counts,
                                                             int ord_cap, int ord_nb, int x_heavy) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ int local_count[4], base[4];
  if (threadIdx.x < 4) local_count[threadIdx.x] = 0;
  __syncthreads();
  int bin = 0, pos = 0;
  bool xback = false;
  if (j < ncolsB) {
    const int64_t n = nprod[j];
    const int nb = B.p[j + 1] - B.p[j];
    bin = bin_of(n, nb);
    if (x_heavy && bin == 3) bin = 4;
    if (n > 0) bin = (n <= ord_cap && nb <= ord_nb) ? 1 : (bin < 2 ? 2 : bin);  // (as products_kernel with ordered = 1)
    xback = bin == 3 && nb > kMediumB;
    if (bin >= 2) pos = atomicAdd(&local_count[xback ? 3 : bin - 2], 1);
  }
  __syncthreads();
  if (threadIdx.x < 4 && local_count[threadIdx.x] > 0)
    base[threadIdx.x] = atomicAdd(&list_counts[threadIdx.x], local_count[threadIdx.x]);
  __syncthreads();
  if (bin >= 2) {
    if (xback) {
      xlarge_list[ncolsB - 1 - (base[3] + pos)] = j;
    } else {
      int64_t *list = bin == 2 ? medium_list : bin == 3 ? xlarge_list : dense_list;
      list[base[bin - 2] + pos] = j;
    }
  }
}

template <int NT>
__device__ inline void group_sync() {
  if (NT == 64) __builtin_amdgcn_wave_barrier();  // LDS operations of one wavefront are in order
  else __syncthreads();
}

// LDS footprint of one group (NT threads cooperating on one column)
constexpr int ilog2_ceil(int v) { return v <= 1 ? 0 : 1 + ilog2_ceil((v + 1) / 2); }

// Numeric sort keys.  KEY32: (row << tbits) | t packed in 32 bits (possible when nrows < 2^(31 - tbits));
// otherwise the key is the row alone and t (position in k-then-row order, < CAP <= 4096) travels with
// it as a 16-bit payload — the merge is stable, so equal rows keep their ascending-t order either
// way.  The products themselves are never kept in LDS (see the fold).
template <int CAP, int NBCAP, bool NUMERIC, bool KEY32 = false>
struct EscLds {
  static constexpr bool kSplit = NUMERIC && !KEY32;
  static constexpr size_t key_bytes = CAP * sizeof(int);
  static constexpr size_t key2_bytes = CAP * sizeof(int);  // second buffer of the merge tree
  static constexpr size_t tpos_bytes = kSplit ? 2 * CAP * sizeof(unsigned short) : 0;  // both buffers
  static constexpr size_t kb_bytes = NUMERIC ? NBCAP * sizeof(double) : 0;
  static constexpr size_t start_bytes = NBCAP * sizeof(int);
  static constexpr size_t off_bytes = (NBCAP + 8) * sizeof(int);
  static constexpr size_t scratch_bytes = 16 * sizeof(int);
  static constexpr size_t total =
      kb_bytes + key_bytes + key2_bytes + tpos_bytes + start_bytes + off_bytes + scratch_bytes;
};

// expand - sort - compress for ONE column j with np products and nb entries in B[:,j];
// `tid` in [0, NT) is the thread's index inside the group, `lds` the group's LDS region.
template <int NT, int CAP, int NBCAP, bool NUMERIC, bool KEY32 = false>
__device__ inline void esc_column(const Csc &A, const Csc &B, int64_t j, int np, unsigned char *lds, int tid,
                                  int *__restrict__ counts, const int64_t *__restrict__ Cp,
                                  int *__restrict__ Ci, double *__restrict__ Cx) {
  typedef EscLds<CAP, NBCAP, NUMERIC, KEY32> L;
  constexpr int TB = ilog2_ceil(CAP);  // bits of the tie-break t
  constexpr bool SPLIT = L::kSplit;  // key = row, t carried as a 16-bit payload
  // 8-byte values first (alignment), then the 4-byte and 2-byte arrays
  double *kb = reinterpret_cast<double *>(lds);
  int *key32 = reinterpret_cast<int *>(lds + L::kb_bytes);
  int *key32b = key32 + CAP;
  int *kstart = key32b + CAP;
  int *koff = kstart + NBCAP;
  int *scratch = koff + NBCAP + 8;
  unsigned short *tpa = reinterpret_cast<unsigned short *>(scratch + 16), *tpb = tpa + CAP;  // SPLIT only
  const int lane = tid & 63;
  const int qs = B.p[j];
  const int nb = B.p[j + 1] - qs;

  // (1) stage the column of B and the extents of the selected columns of A
  for (int q = tid; q < nb; q += NT) {
    const int k = B.i[qs + q];
    const int s = A.p[k];
    kstart[q] = s;
    koff[q] = A.p[k + 1] - s;
    if (NUMERIC) kb[q] = B.x[qs + q];
  }
  group_sync<NT>();
  // exclusive prefix sum of the extents: per-thread chunk, then a serial pass over the NT chunk sums
  {
    const int chunk = (nb + NT - 1) / NT;
    const int lo = tid * chunk, hi = min(nb, lo + chunk);
    int sum = 0;
    for (int q = lo; q < hi; ++q) sum += koff[q];
    // wave-level inclusive scan of the chunk sums, then across waves through scratch
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (NT > 64) {
      if (lane == 63) scratch[tid >> 6] = incl;
      __syncthreads();
      int woff = 0;
      for (int wv = 0; wv < (tid >> 6); ++wv) woff += scratch[wv];
      incl += woff;
      __syncthreads();
    }
    int run = incl - sum;
    for (int q = lo; q < hi; ++q) {
      const int l = koff[q];
      koff[q] = run;
      run += l;
    }
    if (tid == NT - 1) koff[nb] = run;  // == np
  }
  group_sync<NT>();

  // (2) expand: every product of the column is fetched independently; a thread first locates all
  // of its products, then issues all of its loads, so that their latencies overlap
  {
    constexpr int PER = CAP / NT;
    int pp[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = tid + u * NT;
      pp[u] = -1;
      if (t < np) {
        int lo = 0, hi = nb - 1;  // largest q with koff[q] <= t
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (koff[mid] <= t) lo = mid; else hi = mid - 1;
        }
        pp[u] = kstart[lo] + (t - koff[lo]);
      }
    }
    int rows[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) rows[u] = pp[u] >= 0 ? A.i[pp[u]] : 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = tid + u * NT;
      if (pp[u] >= 0) {
        if (SPLIT) {
          key32[t] = rows[u];
          tpa[t] = (unsigned short)t;  // ties: ascending t = ascending k (kept by the stable merge)
        } else if (NUMERIC) {
          key32[t] = (rows[u] << TB) | t;
        } else {
          key32[t] = rows[u];
        }
      }
    }
  }
  group_sync<NT>();

  // (3) sort.  The products of one B entry are a column of A, already ascending by row, and the
  // expansion laid these nb runs out one after the other, so the keys are sorted by a merge tree:
  // ceil(log2 nb) levels of pairwise merges, every thread producing E consecutive outputs of a
  // level (merge-path search for its start, then a sequential merge; ties take the left run first,
  // which is the one with the smaller t).  That is ~5 LDS accesses per product and level instead of
  // the 2 x 66 of a bitonic network over 2048 keys.  E is odd so that the output chunks of a
  // wavefront fall into different LDS banks.
  const int *skeys = key32;              // where the sorted keys end up
  const unsigned short *stp = tpa;       // ... and their t (SPLIT)
  {
    constexpr int E = CAP / NT + 1;
    int *src = key32, *dst = key32b;
    unsigned short *tsrc = tpa, *tdst = tpb;
    for (int width = 1; width < nb; width <<= 1) {
      for (int c0 = tid * E; c0 < np; c0 += NT * E) {
        int pos = c0;
        const int end = min(np, c0 + E);
        while (pos < end) {
          int lq = 0, hq = nb - 1;  // run holding output position pos: largest q with koff[q] <= pos
          while (lq < hq) {
            const int mid = (lq + hq + 1) >> 1;
            if (koff[mid] <= pos) lq = mid; else hq = mid - 1;
          }
          const int g0 = lq & ~(2 * width - 1);
          const int a0 = koff[g0], a1 = koff[min(nb, g0 + width)], b1 = koff[min(nb, g0 + 2 * width)];
          const int d = pos - a0;  // outputs of this pair before pos
          int lo = max(0, d - (b1 - a1)), hi = min(d, a1 - a0);
          while (lo < hi) {  // merge path: how many of the first d outputs come from the left run
            const int mid = (lo + hi) >> 1;
            if (src[a0 + mid] <= src[a1 + d - mid - 1]) lo = mid + 1; else hi = mid;
          }
          int ia = a0 + lo, ib = a1 + d - lo;
          const int stop = min(end, b1);
          int ka = ia < a1 ? src[ia] : 0x7fffffff, kb2 = ib < b1 ? src[ib] : 0x7fffffff;
          for (; pos < stop; ++pos) {
            if (ia < a1 && (ib >= b1 || ka <= kb2)) {
              dst[pos] = ka;
              if (SPLIT) tdst[pos] = tsrc[ia];
              ++ia;
              ka = ia < a1 ? src[ia] : 0x7fffffff;
            } else {
              dst[pos] = kb2;
              if (SPLIT) tdst[pos] = tsrc[ib];
              ++ib;
              kb2 = ib < b1 ? src[ib] : 0x7fffffff;
            }
          }
        }
      }
      group_sync<NT>();
      int *tmp = src;
      src = dst;
      dst = tmp;
      unsigned short *ttmp = tsrc;
      tsrc = tdst;
      tdst = ttmp;
    }
    skeys = src;
    stp = tsrc;
  }

  // (4) compress: run heads in ascending row order; numeric heads fold their run left to right
  const int64_t base = NUMERIC ? Cp[j] : 0;
  int running = 0;
  for (int t0 = 0; t0 < np; t0 += NT) {
    const int t = t0 + tid;
    int row = 0;
    bool head = false;
    if (t < np) {
      constexpr bool PACKED = NUMERIC && !SPLIT;
      row = PACKED ? (skeys[t] >> TB) : skeys[t];
      const int prev = t > 0 ? (PACKED ? (skeys[t - 1] >> TB) : skeys[t - 1]) : -1;
      head = (t == 0) || (row != prev);
    }
    const unsigned long long m = __ballot(head);
    int off = running + __popcll(m & ((1ull << lane) - 1ull));
    int total = __popcll(m);
    if (NT > 64) {
      if (lane == 0) scratch[tid >> 6] = total;
      __syncthreads();
      total = 0;
      for (int wv = 0; wv < NT / 64; ++wv) {
        if (wv < (tid >> 6)) off += scratch[wv];
        total += scratch[wv];
      }
      __syncthreads();
    }
    if (NUMERIC && head) {
      double acc = 0.0;  // SG.reset 0
      // the products are not kept in LDS (the kernel is latency-bound and the 8 bytes per product
      // would halve the workgroups per CU); a run head recomputes a * b of its run from the
      // operands, which the expansion has just pulled into L2
      for (int u = t; u < np; ++u) {
        const int tt = SPLIT ? (int)stp[u] : (skeys[u] & ((1 << TB) - 1));
        if ((SPLIT ? skeys[u] : (skeys[u] >> TB)) != row) break;
        int lo = 0, hi = nb - 1;  // largest q with koff[q] <= tt
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (koff[mid] <= tt) lo = mid; else hi = mid - 1;
        }
        acc = acc + A.x[kstart[lo] + (tt - koff[lo])] * kb[lo];  // c + a * b
      }
      Ci[base + off] = row;
      Cx[base + off] = acc;
    }
    running += total;
  }
  if (counts && tid == 0) counts[j] = running;
}

// bin S: one wavefront per column of B, four columns per workgroup
template <bool NUMERIC, bool KEY32>
__global__ __launch_bounds__(256) void spgemm_wave_kernel(Csc A, Csc B, int64_t ncolsB,
                                                          const int64_t *__restrict__ nprod,
                                                          int *__restrict__ counts,
                                                          const int64_t *__restrict__ Cp,
                                                          int *__restrict__ Ci, double *__restrict__ Cx) {
  typedef EscLds<kSmallProducts, kSmallProducts, NUMERIC, KEY32> L;
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[4][(L::total + 15) / 16 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 4 + wave;
  if (j >= ncolsB) return;
  const int64_t np = nprod[j];
  if (np > kSmallProducts) return;
  if (np == 0) {
    if (counts && lane == 0) counts[j] = 0;
    return;
  }
  if (B.p[j + 1] - B.p[j] > kSmallProducts) return;  // listed in another bin (bin_of)
  esc_column<64, kSmallProducts, kSmallProducts, NUMERIC, KEY32>(A, B, j, (int)np, lds_all[wave], lane, counts,
                                                                Cp, Ci, Cx);
}

// bins M and X: one workgroup per listed column
template <int CAP, int NBCAP, bool NUMERIC, bool KEY32>
__global__ __launch_bounds__(256) void spgemm_block_kernel(Csc A, Csc B, const int64_t *__restrict__ list,
                                                           const int64_t *__restrict__ nprod,
                                                           int *__restrict__ counts,
                                                           const int64_t *__restrict__ Cp,
                                                           int *__restrict__ Ci, double *__restrict__ Cx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int64_t j = list[blockIdx.x];
  esc_column<256, CAP, NBCAP, NUMERIC, KEY32>(A, B, j, (int)nprod[j], smem, (int)threadIdx.x, counts, Cp, Ci, Cx);
}

// bin L: persistent workgroups, each owning one dense accumulator of the pool
template <bool NUMERIC>
__global__ __launch_bounds__(256) void spgemm_dense_kernel(Csc A, Csc B, int64_t nrowsA,
                                                           const int64_t *__restrict__ list, int nlist,
                                                           unsigned char *__restrict__ pool_flags,
                                                           double *__restrict__ pool_vals,
                                                           int *__restrict__ counts,
                                                           const int64_t *__restrict__ Cp,
                                                           int *__restrict__ Ci, double *__restrict__ Cx,
                                                           unsigned long long *__restrict__ stamps = nullptr) {
  __shared__ int wave_counts[4];
  __shared__ int64_t running;
  unsigned long long t_acc = 0, t_gather = 0, t_max = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t npad = (nrowsA + 15) & ~(int64_t)15;  // stride of a flag slot: whole 16-byte vectors (see the gather)
  unsigned char *flags = pool_flags + (size_t)blockIdx.x * (size_t)npad;
  double *w = NUMERIC ? pool_vals + (size_t)blockIdx.x * (size_t)nrowsA : nullptr;
  for (int li = blockIdx.x; li < nlist; li += gridDim.x) {
    const int64_t j = list[li];
    if (threadIdx.x == 0) running = 0;
    const unsigned long long t0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    const int qs = B.p[j], qe = B.p[j + 1];
    for (int q = qs; q < qe; ++q) {
      const int k = B.i[q];
      const double b = NUMERIC ? B.x[q] : 0.0;
      const int ps = A.p[k], pe = A.p[k + 1];
      // The rows of one column of A are distinct, so the read-modify-writes of one pass over it are independent:
      // four per thread are in flight at a time (loads of the indices, then of the accumulators, then the stores).
      // A thread that did them one after the other (each store before the next load, as the compiler must assume
      // they alias) ran a hub column of 10^8 products alone for most of a second.
      for (int p0 = ps + (int)threadIdx.x; p0 < pe; p0 += 4 * 256) {
        int r[4];
        double a[4], acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = p0 + u * 256;
          r[u] = p < pe ? A.i[p] : -1;
          a[u] = (NUMERIC && p < pe) ? A.x[p] : 0.0;
        }
        if (NUMERIC) {
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[u] = r[u] >= 0 ? w[r[u]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (r[u] >= 0) {
            flags[r[u]] = 1;
            if (NUMERIC) w[r[u]] = acc[u] + a[u] * b;
          }
      }
      __syncthreads();
    }
    // gather in row order (ScatterGather.hs:97-147), clearing the accumulator as we go
    const unsigned long long t1 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    // A step covers 256 x 64 rows: every thread reads the flags of 64 consecutive rows as four 16-byte vectors (a flag
    // is 0 or 1, so the population count of a word is its number of used rows) and the vector of the next step is
    // requested before this one is consumed.  (Row by row — 256 rows, three barriers and a dependent load per
    // step — the sweep of all nrowsA flags cost every dense column 2 ms, ten times its accumulation, whatever
    // the number of rows it had touched: profiles/r02_spgemm_dense_stamps.txt.)
    const int64_t base = NUMERIC ? Cp[j] : 0;
    uint4 *f4 = reinterpret_cast<uint4 *>(flags);
    const int64_t nvec = npad >> 4;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    constexpr int VPT = 4;  // vectors per thread and step: 64 consecutive rows per thread, 16 384 rows per step
    uint4 cur[VPT], nxt[VPT];
#pragma unroll
    for (int e = 0; e < VPT; ++e) {
      const int64_t v = (int64_t)threadIdx.x * VPT + e;
      cur[e] = v < nvec ? f4[v] : zero4;
    }
    for (int64_t v0 = 0; v0 < nvec; v0 += 256 * VPT) {
      const int64_t vb = v0 + (int64_t)threadIdx.x * VPT;
      int c = 0;
#pragma unroll
      for (int e = 0; e < VPT; ++e) {
        const int64_t vn = vb + 256 * VPT + e;
        nxt[e] = vn < nvec ? f4[vn] : zero4;
        c += __popc(cur[e].x) + __popc(cur[e].y) + __popc(cur[e].z) + __popc(cur[e].w);
      }
      int incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
      }
      if (lane == 63) wave_counts[wave] = incl;
      __syncthreads();
      int64_t off = running + (incl - c);
      for (int ww = 0; ww < wave; ++ww) off += wave_counts[ww];
      if (c) {
#pragma unroll
        for (int e = 0; e < VPT; ++e) {
          const unsigned words[4] = {cur[e].x, cur[e].y, cur[e].z, cur[e].w};
          if ((words[0] | words[1] | words[2] | words[3]) == 0u) continue;
          const int64_t v = vb + e;
          if (NUMERIC) {
            const double *wrow = w + (v << 4);
            double got[16];  // all the accumulators of the used rows first (independent loads), then the stores
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int bidx = 0; bidx < 4; ++bidx)
                got[4 * q + bidx] = ((words[q] >> (8 * bidx)) & 1u) ? wrow[4 * q + bidx] : 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int bidx = 0; bidx < 4; ++bidx)
                if ((words[q] >> (8 * bidx)) & 1u) {
                  const int64_t r = (v << 4) + 4 * q + bidx;
                  Ci[base + off] = (int)r;
                  Cx[base + off] = got[4 * q + bidx];
                  w[r] = 0.0;
                  ++off;
                }
          }
          f4[v] = zero4;
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) running += wave_counts[0] + wave_counts[1] + wave_counts[2] + wave_counts[3];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < VPT; ++e) cur[e] = nxt[e];
    }
    if (counts && threadIdx.x == 0) counts[j] = (int)running;
    __syncthreads();
    if (stamps) {
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      t_acc += t1 - t0;
      t_gather += t2 - t1;
      t_max = t2 - t0 > t_max ? t2 - t0 : t_max;
    }
  }
  if (stamps && threadIdx.x == 0) {  // SPL_SPGEMM_STAMPS=1: ticks of accumulation / gather summed over columns, the longest column
    atomicAdd(stamps, t_acc);
    atomicAdd(stamps + 1, t_gather);
    atomicMax(stamps + 2, t_max);
  }
}

// bin L, row-range form (round 3).  The dense accumulator above is the reference's own structure: 9 bytes of HBM
// per row of A and column in flight, random read-modify-writes per product and a sweep of all nrowsA flags per column
// — 409 019 such columns cost 1.2 of the 1.44 s of a scale-20 product with skewed quadrants.  Here a heavy column
// is cut into RANGES of rows with at most kRngCap products each, and every range goes through expand - sort -
// compress in LDS like a column of its own:
//   (1) the column of B and the extents of the columns of A it selects are staged and prefix-summed (as in esc_column);
//   (2) all products are enumerated once to count them over 4096 buckets of rows; consecutive buckets are grouped
//       into ranges (a serial scan by one thread: 4096 additions);
//   (3) all products are enumerated again and dealt to their ranges in a scratch area in HBM (8 bytes per product:
//       the key (row - first row of the range) << 11 | q and the position of the entry of A; q, the position of
//       (k, b) in the column of B, is the tie-break: a column of A has a row at most once, so (row, q) is unique and
//       ascending q is ascending k) — every load of (2) and (3) is independent of every other, a thread keeps eight
//       in flight;
//   (4) a range at a time: its keys are read back, counted into monotone buckets of rows, placed at bucket start +
//       arrival rank and then at bucket start + number of smaller keys of the bucket (the ordered kernel's rank
//       sort), run heads fold their run left to right — c + a * b in ascending k, separately rounded multiplies and
//       adds (Sparse.hs:699): the same bits as every other path — and write row and sum.  Ranges ascend, so the
//       column comes out in ascending row order with no further pass.
// (A first version walked a cursor per entry of B through the columns of A range by range: chains of dependent
// loads, 0.52 s where the dense accumulators take 0.41 s on scale 19.)
// Not taken (the column is appended to `fallback` and goes to the dense kernel): more than kRngNb entries in the
// column of B, more than kRngMaxProducts products, a single bucket of rows with more than kRngCap products (hub
// rows), or more than kRngMaxRanges ranges.
constexpr int kRngNb = 2048, kRngCap = 2048, kRngBuckets = 4096, kRngTB = 11, kRngSortBuckets = 2048;
constexpr int kRngMaxRanges = 1024, kRngMaxProducts = 1 << 19;
constexpr int kRngThreads = 512;  // 8 wavefronts per workgroup, two workgroups per CU: the kernel lives on loads in flight
constexpr size_t kRngLdsBytes =
    (size_t)(2 * kRngNb + 8 + kRngBuckets + 8 + 3 * kRngMaxRanges + 16 + 2 * kRngCap + kRngSortBuckets + 8) * 4;

template <bool NUMERIC>
__global__ __launch_bounds__(kRngThreads) void spgemm_range_kernel(Csc A, Csc B, int64_t nrowsA, const int64_t *__restrict__ list,
                                                           int nlist, const int64_t *__restrict__ nprod, int *__restrict__ counts,
                                                           const int64_t *__restrict__ Cp, int *__restrict__ Ci,
                                                           double *__restrict__ Cx, int64_t *__restrict__ fallback,
                                                           int *__restrict__ nfallback, unsigned *__restrict__ scratch_key,
                                                           double *__restrict__ scratch_val) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rng_lds[];  // kRngLdsBytes: 72 KB, two workgroups per CU
  int *cur = reinterpret_cast<int *>(rng_lds), *off = cur + kRngNb;
  int *hist = off + kRngNb + 8;                 // products per bucket of rows, then (in place) the range of every bucket
  int *rstart = hist + kRngBuckets + 8;         // first product of range r in the scratch area (kRngMaxRanges + 8)
  int *rlo = rstart + kRngMaxRanges + 8;        // first row of range r (kRngMaxRanges + 8)
  int *rfill = rlo + kRngMaxRanges + 8;         // products dealt to range r so far
  unsigned *keys = reinterpret_cast<unsigned *>(rfill + kRngMaxRanges);
  int *ppos = reinterpret_cast<int *>(keys + kRngCap);  // where the product of a sorted key sits in the range's scratch segment
  int *bcount = ppos + kRngCap;                 // kRngSortBuckets + 8
  constexpr int NT = kRngThreads, NW = NT / 64, PT = kRngCap / NT;  // threads, wavefronts, products per thread and batch
  __shared__ int wsum[2 * NW], sh_nr, sh_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned *gkey = scratch_key + (size_t)blockIdx.x * kRngMaxProducts;
  double *gval = NUMERIC ? scratch_val + (size_t)blockIdx.x * kRngMaxProducts : nullptr;
  int shift = 0;
  while (((int64_t)kRngBuckets << shift) < nrowsA) ++shift;
  // exclusive prefix sum of arr[0 .. n) in place, arr[n] = total; PER entries per thread (n <= NT * PER)
  auto scan_arr = [&](int *arr, int n, auto per_tag) {
    constexpr int PER = decltype(per_tag)::value;
    int v[PER], sum = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) { const int q = tid * PER + u; v[u] = q < n ? arr[q] : 0; sum += v[u]; }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int u = 0; u < PER; ++u) { const int q = tid * PER + u; if (q < n) arr[q] = run; run += v[u]; }
    if (tid == NT - 1) arr[n] = run;
    __syncthreads();
  };
  typedef std::integral_constant<int, kRngNb / NT> PerNb;            // scans of kRngNb = kRngSortBuckets entries
  typedef std::integral_constant<int, kRngBuckets / NT> PerBuckets;  // scan of the bucket counts
  auto find_q = [&](int t, int nb) {  // largest q with off[q] <= t
    int lo = 0, hi = nb - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[mid] <= t) lo = mid; else hi = mid - 1; }
    return lo;
  };
  for (int li = blockIdx.x; li < nlist; li += gridDim.x) {
    const int64_t j = list[li];
    const int qs = B.p[j], nb = B.p[j + 1] - qs;
    if (nb > kRngNb || nprod[j] > (int64_t)kRngMaxProducts) {  // (the 64-bit count: the prefix sums below are 32-bit)
      if (tid == 0) fallback[atomicAdd(nfallback, 1)] = j;
      continue;
    }
    // (1) the column of B: where each selected column of A starts, how long it is
    for (int q = tid; q < nb; q += NT) {
      const int k = B.i[qs + q];
      const int s0 = A.p[k];
      cur[q] = s0;
      off[q] = A.p[k + 1] - s0;
    }
    for (int b = tid; b < kRngBuckets; b += NT) hist[b] = 0;
    for (int r = tid; r < kRngMaxRanges; r += NT) rfill[r] = 0;
    __syncthreads();
    scan_arr(off, nb, PerNb());
    const int np = off[nb];
    if (np > kRngMaxProducts) {  // (uniform: every thread reads the same total)
      if (tid == 0) fallback[atomicAdd(nfallback, 1)] = j;
      __syncthreads();
      continue;
    }
    // (2) products per bucket of rows.  A thread takes PT CONSECUTIVE products of the k-then-row enumeration: one
    // search for the first, a walk for the others, PT independent loads in flight
    auto locate8 = [&](int tfirst, int (&pp)[PT], int (&qq)[PT]) {
      int q = find_q(tfirst < np ? tfirst : np - 1, nb);
      int nxt = off[q + 1];
#pragma unroll
      for (int u = 0; u < PT; ++u) {
        int t = tfirst + u;
        t = t < np ? t : np - 1;
        while (t >= nxt) { ++q; nxt = off[q + 1]; }  // (empty columns of A are skipped)
        qq[u] = q;
        pp[u] = cur[q] + (t - off[q]);
      }
    };
    for (int t0 = 0; t0 < np; t0 += NT * PT) {
      int pp[PT], qq[PT], rr[PT];
      locate8(t0 + tid * PT, pp, qq);
#pragma unroll
      for (int u = 0; u < PT; ++u) rr[u] = A.i[pp[u]];
#pragma unroll
      for (int u = 0; u < PT; ++u)
        if (t0 + tid * PT + u < np) atomicAdd(&hist[rr[u] >> shift], 1);
    }
    __syncthreads();
    // (3a) ranges: consecutive buckets while their products fit kRngCap.  Prefix sums of the bucket counts (all
    // threads), then one thread finds the end of every range by bisection (a dozen ranges, twelve steps each); a
    // bucket with more than kRngCap products cannot be taken
    {
      int over = 0;
      for (int b = tid; b < kRngBuckets; b += NT) over |= hist[b] > kRngCap ? 1 : 0;
      if (tid == 0) sh_bad = 0;
      __syncthreads();
      if (over) sh_bad = 1;
      __syncthreads();
    }
    if (sh_bad) {
      if (tid == 0) fallback[atomicAdd(nfallback, 1)] = j;
      __syncthreads();
      continue;
    }
    scan_arr(hist, kRngBuckets, PerBuckets());  // hist[b] = products in front of bucket b, hist[kRngBuckets] = np
    if (tid == 0) {
      int nr = 0, start = 0, bad = 0;
      while (start < kRngBuckets) {
        if (nr >= kRngMaxRanges) { bad = 1; break; }
        const int p0 = hist[start];
        int lo = start + 1, hi = kRngBuckets;  // largest e with hist[e] - p0 <= kRngCap (e = start + 1 always fits)
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (hist[mid] - p0 <= kRngCap) lo = mid; else hi = mid - 1; }
        rstart[nr] = p0;
        rlo[nr] = start;  // bucket for now, row below
        ++nr;
        start = lo;
      }
      rstart[nr] = np;
      rlo[nr] = kRngBuckets;
      sh_nr = nr;
      sh_bad = bad;
    }
    __syncthreads();
    if (sh_bad) {
      if (tid == 0) fallback[atomicAdd(nfallback, 1)] = j;
      __syncthreads();
      continue;
    }
    for (int b = tid; b < kRngBuckets; b += NT) {  // hist[b] <- range of bucket b: largest r with rlo[r] <= b
      int lo = 0, hi = sh_nr - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (rlo[mid] <= b) lo = mid; else hi = mid - 1; }
      hist[b] = lo;
    }
    __syncthreads();
    for (int r = tid; r <= sh_nr; r += NT) {
      const int64_t row = (int64_t)rlo[r] << shift;
      rlo[r] = (int)(row < nrowsA ? row : nrowsA);
    }
    __syncthreads();
    const int nr = sh_nr;
    // (3b) every product to its range in the scratch area
    for (int t0 = 0; t0 < np; t0 += NT * PT) {
      int pp[PT], qq[PT], rr[PT];
      locate8(t0 + tid * PT, pp, qq);
#pragma unroll
      for (int u = 0; u < PT; ++u) rr[u] = A.i[pp[u]];
      double prod[PT];
      if (NUMERIC) {  // the product itself travels with the key: a * b, rounded once, here (the fold only adds)
        double av[PT], bv[PT];
#pragma unroll
        for (int u = 0; u < PT; ++u) { av[u] = A.x[pp[u]]; bv[u] = B.x[qs + qq[u]]; }
#pragma unroll
        for (int u = 0; u < PT; ++u) prod[u] = av[u] * bv[u];
      }
#pragma unroll
      for (int u = 0; u < PT; ++u)
        if (t0 + tid * PT + u < np) {
          const int rid = hist[rr[u] >> shift];
          const int slot = rstart[rid] + atomicAdd(&rfill[rid], 1);
          gkey[slot] = ((unsigned)(rr[u] - rlo[rid]) << kRngTB) | (unsigned)qq[u];
          if (NUMERIC) gval[slot] = prod[u];
        }
    }
    __threadfence_block();
    __syncthreads();
    const int64_t base = NUMERIC ? Cp[j] : 0;
    int running = 0;
    // (4) a range at a time
    for (int r = 0; r < nr; ++r) {
      const int s0 = rstart[r], npr = rstart[r + 1] - s0;  // <= kRngCap
      if (npr == 0) continue;  // (uniform)
      const int lo_row = rlo[r];
      int bshift = 0;
      {
        const int64_t span = (int64_t)rlo[r + 1] - lo_row;
        while ((span >> bshift) >= kRngSortBuckets) ++bshift;
      }
      for (int b = tid; b < kRngSortBuckets; b += NT) bcount[b] = 0;
      __syncthreads();
      unsigned mykey[PT];
      int mypp[PT], mybkt[PT], myarr[PT];
#pragma unroll
      for (int u = 0; u < PT; ++u) {
        const int t = u * NT + tid;
        const int tc = t < npr ? t : npr - 1;
        mykey[u] = gkey[s0 + tc];
        mypp[u] = tc;
      }
#pragma unroll
      for (int u = 0; u < PT; ++u) {
        mybkt[u] = -1;
        if (u * NT + tid < npr) {
          mybkt[u] = (int)((mykey[u] >> kRngTB) >> bshift);
          myarr[u] = atomicAdd(&bcount[mybkt[u]], 1);
        }
      }
      __syncthreads();
      scan_arr(bcount, kRngSortBuckets, PerNb());
#pragma unroll
      for (int u = 0; u < PT; ++u)
        if (mybkt[u] >= 0) keys[bcount[mybkt[u]] + myarr[u]] = mykey[u];
      __syncthreads();
      int mypos[PT];
#pragma unroll
      for (int u = 0; u < PT; ++u) {
        mypos[u] = -1;
        if (mybkt[u] >= 0) {
          const int b0 = bcount[mybkt[u]], b1 = bcount[mybkt[u] + 1];
          int c = 0;
          for (int i = b0; i < b1; ++i) c += keys[i] < mykey[u] ? 1 : 0;
          mypos[u] = b0 + c;
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < PT; ++u)
        if (mypos[u] >= 0) { keys[mypos[u]] = mykey[u]; ppos[mypos[u]] = mypp[u]; }
      __syncthreads();
      // compress: run heads in ascending row order; a head folds its run left to right (ascending q = ascending k)
      for (int t0 = 0; t0 < npr; t0 += NT) {
        const int t = t0 + tid;
        bool head = false;
        unsigned rowl = 0;
        if (t < npr) {
          rowl = keys[t] >> kRngTB;
          head = t == 0 || (keys[t - 1] >> kRngTB) != rowl;
        }
        const unsigned long long m = __ballot(head);
        int offp = running + __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[NW + wave] = __popcll(m);
        __syncthreads();
        int total = 0;
        for (int w = 0; w < NW; ++w) { if (w < wave) offp += wsum[NW + w]; total += wsum[NW + w]; }
        __syncthreads();
        if (NUMERIC && head) {
          double acc = 0.0;  // SG.reset 0
          for (int u = t; u < npr; ++u) {
            const unsigned ku = keys[u];
            if ((ku >> kRngTB) != rowl) break;
            acc = acc + gval[s0 + ppos[u]];  // c + a * b, the product as rounded in (3)
          }
          Ci[base + offp] = (int)rowl + lo_row;
          Cx[base + offp] = acc;
        }
        running += total;
      }
    }
    if (counts && tid == 0) counts[j] = running;
    __syncthreads();
  }
}

// single-pass mode: column j was written at its upper-bound slot (offset = products before j);
// one wavefront per column moves it to its final place
__global__ __launch_bounds__(256) void compact_columns_kernel(int64_t ncols, const int64_t *__restrict__ slot,
                                                              const int64_t *__restrict__ Cp,
                                                              const int *__restrict__ Ti, const double *__restrict__ Tx,
                                                              int *__restrict__ Ci, double *__restrict__ Cx) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= ncols) return;
  const int64_t src = slot[j], dst = Cp[j], len = Cp[j + 1] - dst;
  for (int64_t t = lane; t < len; t += 64) {
    Ci[dst + t] = Ti[src + t];
    Cx[dst + t] = Tx[src + t];
  }
}


// ---- ordered single-pass form -------------------------------------------------------------------------
// The single-pass form above writes every column at its upper-bound slot and compacts afterwards: one more
// read and write of the whole result (15 % of C4 although only 0.05 % of its products merge).  Here every
// column is written straight to its final place: the columns are handed out IN ORDER to persistent
// workgroups (a ticket counter), a column's length is known once its keys are sorted, and its offset is the
// sum of the lengths before it, obtained by a decoupled look-back over per-column status words
// (flag + value in one 64-bit word; a column publishes its length at once, looks back until it meets a
// predecessor that already knows its own inclusive prefix, then publishes its own).  A workgroup only ever
// waits for columns with smaller tickets, all of which are held by resident workgroups.
//
// The rocprofv3 counters of the compacting form (profiles/r02_spgemm_c4_before_*) show its main kernel to
// be bound by instruction issue, not by memory: 8 450 vector + 5 100 scalar instructions per wavefront
// and column of 1 024 products, most of them in the five merge levels and in the binary search by which a
// run head finds the operands of every product again.  Here
//   * a thread expands a CONSECUTIVE run of products (one binary search, then a walk), a*b is computed
//     then and kept in LDS, so the fold is a plain sum over LDS in sorted order;
//   * the keys (row << 11 | t) are sorted by ONE counting pass over buckets of row ranges (CAP/2
//     buckets, rank inside a bucket from the returning LDS atomic, exclusive scan, scatter) followed by an
//     insertion sort inside each bucket (a thread per bucket; about two keys per bucket when the rows are
//     spread evenly).  A bucket that outgrows kOrdBucketLimit (rows crowded into a narrow range) sends
//     the column through the merge tree instead — same result, the old cost.
// Order of the sums is unchanged: equal rows are adjacent in ascending t = ascending k, folded left to
// right from 0 with separately rounded multiplies and adds (Sparse.hs:699) — bit-identical values.
// Columns beyond the LDS budget of this kernel (more than 2 048 products or 256 entries in the column of
// B) are computed beforehand by the kernels above into scratch slots; their owner here only copies them.
constexpr int kOrdTB = 11;                         // tie-break bits of the packed key: t < 2048
constexpr int kOrdWaveCap = 256, kOrdWaveNb = 64;  // column handled by one wavefront (4 per workgroup)
constexpr int kOrdBucketLimit = 24;
constexpr int kOrdMaxRowBits = 32 - kOrdTB;        // rows must fit the packed (unsigned) 32-bit key

template <int CAP, int NBCAP>
struct OrdLds {
  static constexpr int kBuckets = CAP / 2;
  static constexpr size_t kb_bytes = NBCAP * sizeof(double);
  static constexpr size_t val_bytes = CAP * sizeof(double);
  static constexpr size_t key_bytes = CAP * sizeof(unsigned);
  static constexpr size_t start_bytes = NBCAP * sizeof(int);
  static constexpr size_t off_bytes = (NBCAP + 8) * sizeof(int);
  static constexpr size_t hist_bytes = (kBuckets + 8) * sizeof(int);
  static constexpr size_t scratch_bytes = 16 * sizeof(int);
  static constexpr size_t total = kb_bytes + val_bytes + 2 * key_bytes + start_bytes + off_bytes + hist_bytes + scratch_bytes;
};

// status word of the look-back chain: bits 62-63 = 0 nothing yet, 1 the column's own length, 2 inclusive prefix
constexpr unsigned long long kOrdFlagAgg = 1ull << 62, kOrdFlagPrefix = 2ull << 62, kOrdValueMask = (1ull << 62) - 1ull;

// The chain.  publish: the column's own length, as soon as it is known (column 0 publishes its inclusive prefix
// at once).  look-back: exclusive prefix of column j (sum of the lengths of all columns before it), then the
// column's inclusive prefix is published.  Both are called by ONE whole wavefront; results are wave-uniform.
__device__ inline void ord_publish(unsigned long long *__restrict__ status, int64_t j, int64_t count) {
  if ((threadIdx.x & 63) == 0)
    __hip_atomic_store(status + j, (j == 0 ? kOrdFlagPrefix : kOrdFlagAgg) | (unsigned long long)count, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// Look-back: kLookBlocks x 64 status words per dependent round trip (independent loads, nearest block first), evaluated
// block by block.  Measured on C4 (SPL_SPGEMM_STAMPS=1, round 5): the pipelined path needs ONE round for 99.6 % of its
// look-backs and sleeps for an unpublished predecessor about a hundred times in a million columns — the chain costs one
// memory round trip per column (3 200 cycles under this kernel's load), not a walk and not a wait; eight blocks per
// round were slower than two (the surplus loads queue in front of other wavefronts' gathers).
constexpr int kLookBlocks = 2;

__device__ inline int64_t ord_lookback(unsigned long long *__restrict__ status, int64_t j, int64_t count, bool polled = true,
                                       unsigned long long *diag = nullptr) {
  const int lane = threadIdx.x & 63;
  if (j == 0) return 0;
  const unsigned long long t_in = diag ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned long long rounds = 0, polls = 0, t_poll = 0;
  int64_t sum = 0;
  int64_t p = j - 1;  // nearest predecessor not yet accounted for
  auto poll_nearest = [&]() {  // one lane polls, asleep in between (64 lanes polling from a thousand waiting wavefronts
    if (lane == 0) {           // take issue slots and L2 requests from the wavefronts they are waiting for)
      while ((__hip_atomic_load(status + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 62) == 0) __builtin_amdgcn_s_sleep(32);
    }
    __builtin_amdgcn_wave_barrier();
  };
  // `polled`: the paths that look back right after publishing wait for the nearest predecessor first — it took its
  // ticket just before this column and is usually the last to publish.  The pipelined path looks back a whole column
  // of work later: everything is there, and the poll would only be one more dependent round trip.
  if (polled) poll_nearest();
  for (;;) {
    unsigned long long w[kLookBlocks];
#pragma unroll
    for (int k = 0; k < kLookBlocks; ++k) {
      const int64_t q = p - 64 * k - lane;
      w[k] = q >= 0 ? __hip_atomic_load(status + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
    long long part = 0;
    int consumed = 0;
    bool done = false, stalled = false;
    ++rounds;
#pragma unroll
    for (int k = 0; k < kLookBlocks; ++k) {
      if (done || stalled) break;  // (wave-uniform)
      const int64_t q = p - 64 * k - lane;
      const unsigned flag = (unsigned)(w[k] >> 62);
      const unsigned long long not_ready = __ballot(q >= 0 && flag == 0);
      const unsigned long long is_prefix = __ballot(q >= 0 && flag == 2);
      // lanes are ordered by distance: use everything up to the nearest prefix, provided nothing nearer is missing
      const int first_missing = not_ready ? __builtin_ctzll(not_ready) : 64;
      const int first_prefix = is_prefix ? __builtin_ctzll(is_prefix) : 64;
      const int upto = first_prefix < first_missing ? first_prefix + 1 : first_missing;  // lanes [0, upto) are usable
      if (lane < upto && q >= 0) part += (long long)(w[k] & kOrdValueMask);
      consumed += upto;
      if (first_prefix < first_missing) done = true;  // met a predecessor that knows everything before it
      else if (first_missing < 64) stalled = true;    // a column in this block has not published yet
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
    sum += part;
    if (done) break;
    p -= consumed;
    if (p < 0) break;  // ran past column 0 (cannot happen: column 0 publishes a prefix)
    if (stalled) {
      const unsigned long long tp = diag ? __builtin_amdgcn_s_memtime() : 0ull;
      poll_nearest();
      if (diag) { t_poll += __builtin_amdgcn_s_memtime() - tp; ++polls; }
    }
  }
  if (diag) {  // SPL_SPGEMM_STAMPS=1: ticks inside the look-back, of them asleep waiting for a column that had not published, rounds, waits
    diag[0] += __builtin_amdgcn_s_memtime() - t_in;
    diag[1] += t_poll;
    diag[2] += rounds;
    diag[3] += polls;
  }
  if (lane == 0)
    __hip_atomic_store(status + j, kOrdFlagPrefix | (unsigned long long)(sum + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return sum;
}

__device__ inline int64_t ord_chain(unsigned long long *__restrict__ status, int64_t j, int64_t count) {
  ord_publish(status, j, count);
  return ord_lookback(status, j, count);
}

// merge tree over the nb ascending runs laid out by the expansion (packed keys only): the fallback sort
template <int NT, int CAP>
__device__ inline unsigned *ord_merge_tree(unsigned *key, unsigned *key2, const int *koff, int nb, int np, int tid) {
  constexpr int E = CAP / NT + 1;
  unsigned *src = key, *dst = key2;
  for (int width = 1; width < nb; width <<= 1) {
    for (int c0 = tid * E; c0 < np; c0 += NT * E) {
      int pos = c0;
      const int end = min(np, c0 + E);
      while (pos < end) {
        int lq = 0, hq = nb - 1;
        while (lq < hq) {
          const int mid = (lq + hq + 1) >> 1;
          if (koff[mid] <= pos) lq = mid; else hq = mid - 1;
        }
        const int g0 = lq & ~(2 * width - 1);
        const int a0 = koff[g0], a1 = koff[min(nb, g0 + width)], b1 = koff[min(nb, g0 + 2 * width)];
        const int d = pos - a0;
        int lo = max(0, d - (b1 - a1)), hi = min(d, a1 - a0);
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (src[a0 + mid] <= src[a1 + d - mid - 1]) lo = mid + 1; else hi = mid;
        }
        int ia = a0 + lo, ib = a1 + d - lo;
        const int stop = min(end, b1);
        unsigned ka = ia < a1 ? src[ia] : 0xffffffffu, kb2 = ib < b1 ? src[ib] : 0xffffffffu;
        for (; pos < stop; ++pos) {
          if (ia < a1 && (ib >= b1 || ka <= kb2)) {
            dst[pos] = ka;
            ++ia;
            ka = ia < a1 ? src[ia] : 0xffffffffu;
          } else {
            dst[pos] = kb2;
            ++ib;
            kb2 = ib < b1 ? src[ib] : 0xffffffffu;
          }
        }
      }
    }
    group_sync<NT>();
    unsigned *tmp = src;
    src = dst;
    dst = tmp;
  }
  return src;
}

// One column of C, written at its final place.  NT threads cooperate (64: one wavefront, wave-level
// synchronisation only; 256: the workgroup); `chain_wave`: this thread belongs to the wavefront that
// walks the look-back chain (NT = 256: wavefront 0; the result travels through `scratch`).
template <int NT, int CAP, int NBCAP>
__device__ inline void ord_column(const Csc &A, const Csc &B, int64_t j, int np, int bucket_shift, unsigned char *lds, int tid,
                                  unsigned long long *__restrict__ status, int64_t *__restrict__ Cp, int *__restrict__ Ci,
                                  double *__restrict__ Cx, unsigned long long *__restrict__ stamps) {
  typedef OrdLds<CAP, NBCAP> L;
  // SPL_SPGEMM_STAMPS=1 (diagnostic): cycles of thread 0 per phase, summed over columns
  // (accumulated in registers of thread 0 and flushed once per workgroup: an atomic per phase would serialise
  // a million columns on six addresses and distort what it measures)
  unsigned long long t_prev = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  auto stamp = [&](int phase) {
    if (stamps && tid == 0) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      stamps[phase] += now - t_prev;
      t_prev = now;
    }
  };
  constexpr int NBK = L::kBuckets;
  double *kb = reinterpret_cast<double *>(lds);
  double *vals = kb + NBCAP;
  unsigned *key = reinterpret_cast<unsigned *>(lds + L::kb_bytes + L::val_bytes);
  unsigned *key2 = key + CAP;
  int *kstart = reinterpret_cast<int *>(key2 + CAP);
  int *koff = kstart + NBCAP;
  int *hist = koff + NBCAP + 8;
  int *scratch = hist + NBK + 8;
  const int lane = tid & 63;
  const int qs = B.p[j];
  const int nb = B.p[j + 1] - qs;

  // (1) stage the column of B and the extents of the selected columns of A; clear the histogram
  for (int q = tid; q < nb; q += NT) {
    const int k = B.i[qs + q];
    const int s = A.p[k];
    kstart[q] = s;
    koff[q] = A.p[k + 1] - s;
    kb[q] = B.x[qs + q];
  }
  for (int b = tid; b < NBK + 1; b += NT) hist[b] = 0;
  if (tid == 0) scratch[8] = 0;  // overflow flag
  group_sync<NT>();
  {  // exclusive prefix sum of the extents
    const int chunk = (nb + NT - 1) / NT;
    const int lo = tid * chunk, hi = min(nb, lo + chunk);
    int sum = 0;
    for (int q = lo; q < hi; ++q) sum += koff[q];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (NT > 64) {
      if (lane == 63) scratch[tid >> 6] = incl;
      __syncthreads();
      int woff = 0;
      for (int wv = 0; wv < (tid >> 6); ++wv) woff += scratch[wv];
      incl += woff;
      __syncthreads();
    }
    int run = incl - sum;
    for (int q = lo; q < hi; ++q) {
      const int l = koff[q];
      koff[q] = run;
      run += l;
    }
    if (tid == NT - 1) koff[nb] = run;  // == np
  }
  group_sync<NT>();
  stamp(0);

  // (2) expand a consecutive run of products per thread: one search, then a walk; all loads issued together
  constexpr int PER = CAP / NT;
  const int per = (np + NT - 1) / NT;  // <= PER
  unsigned myk[PER];
  int myrank[PER];
  {
    const int t0 = tid * per;
    int q = 0;
    if (t0 < np) {
      int lo = 0, hi = nb - 1;  // largest q with koff[q] <= t0
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (koff[mid] <= t0) lo = mid; else hi = mid - 1;
      }
      q = lo;
    }
    int pp[PER], qq[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = t0 + u;
      pp[u] = -1;
      qq[u] = 0;
      if (u < per && t < np) {
        while (t >= koff[q + 1]) ++q;  // runs may be empty: skip them
        pp[u] = kstart[q] + (t - koff[q]);
        qq[u] = q;
      }
    }
    int rows[PER];
    double av[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      rows[u] = pp[u] >= 0 ? A.i[pp[u]] : 0;
      av[u] = pp[u] >= 0 ? A.x[pp[u]] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      myk[u] = 0xffffffffu;
      myrank[u] = 0;
      if (pp[u] >= 0) {
        const int t = t0 + u;
        myk[u] = ((unsigned)rows[u] << kOrdTB) | (unsigned)t;
        key[t] = myk[u];
        vals[t] = av[u] * kb[qq[u]];  // a * b, rounded once (Sparse.hs:699)
        myrank[u] = atomicAdd(&hist[rows[u] >> bucket_shift], 1);
      }
    }
  }
  group_sync<NT>();
  stamp(1);

  // (3) exclusive scan of the histogram (NBK buckets, NBK / NT consecutive ones per thread)
  {
    constexpr int BPT = (NBK + NT - 1) / NT;
    const int b0 = tid * BPT;
    int cnt[BPT], sum = 0, big = 0;
#pragma unroll
    for (int u = 0; u < BPT; ++u) {
      cnt[u] = b0 + u < NBK ? hist[b0 + u] : 0;
      sum += cnt[u];
      big |= cnt[u] > kOrdBucketLimit;
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (NT > 64) {
      if (lane == 63) scratch[tid >> 6] = incl;
      __syncthreads();
      int woff = 0;
      for (int wv = 0; wv < (tid >> 6); ++wv) woff += scratch[wv];
      incl += woff;
      __syncthreads();
    }
    int run = incl - sum;
#pragma unroll
    for (int u = 0; u < BPT; ++u)
      if (b0 + u < NBK) { hist[b0 + u] = run; run += cnt[u]; }
    if (tid == NT - 1) hist[NBK] = np;
    if (big) scratch[8] = 1;
  }
  group_sync<NT>();
  stamp(2);
  const bool overflow = scratch[8] != 0;
  const unsigned *skeys;
  if (!overflow) {
    // (4) scatter to the bucket ranges, then sort inside each bucket (a thread per bucket)
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (myk[u] != 0xffffffffu) key2[hist[(myk[u] >> kOrdTB) >> bucket_shift] + myrank[u]] = myk[u];
    group_sync<NT>();
    for (int b = tid; b < NBK; b += NT) {
      const int s = hist[b], e = hist[b + 1];
      for (int i = s + 1; i < e; ++i) {
        const unsigned k = key2[i];
        int h = i - 1;
        while (h >= s && key2[h] > k) { key2[h + 1] = key2[h]; --h; }
        key2[h + 1] = k;
      }
    }
    group_sync<NT>();
    skeys = key2;
  } else {
    skeys = ord_merge_tree<NT, CAP>(key, key2, koff, nb, np, tid);
  }
  stamp(3);

  // (5) run heads: count, chain, write
  int count = 0;
  for (int t0 = 0; t0 < np; t0 += NT) {
    const int t = t0 + tid;
    const bool head = t < np && (t == 0 || (skeys[t] >> kOrdTB) != (skeys[t - 1] >> kOrdTB));
    count += __popcll(__ballot(head));
  }
  if (NT > 64) {  // per-wavefront counts -> workgroup total
    if (lane == 0) scratch[tid >> 6] = count;
    __syncthreads();
    count = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    __syncthreads();
  }
  int64_t base;
  if (NT > 64) {
    if ((tid >> 6) == 0) {
      const int64_t e = ord_chain(status, j, count);
      if (lane == 0) { scratch[10] = (int)(e & 0xffffffffll); scratch[11] = (int)(e >> 32); }
    }
    __syncthreads();
    base = ((int64_t)scratch[11] << 32) | (int64_t)(unsigned)scratch[10];
  } else {
    base = ord_chain(status, j, count);
  }
  if (tid == 0) Cp[j] = base;
  stamp(4);
  int running = 0;
  for (int t0 = 0; t0 < np; t0 += NT) {
    const int t = t0 + tid;
    int row = 0;
    bool head = false;
    if (t < np) {
      row = (int)(skeys[t] >> kOrdTB);
      head = t == 0 || (int)(skeys[t - 1] >> kOrdTB) != row;
    }
    const unsigned long long m = __ballot(head);
    int off = running + __popcll(m & ((1ull << lane) - 1ull));
    int total = __popcll(m);
    if (NT > 64) {
      if (lane == 0) scratch[tid >> 6] = total;
      __syncthreads();
      total = 0;
      for (int wv = 0; wv < NT / 64; ++wv) {
        if (wv < (tid >> 6)) off += scratch[wv];
        total += scratch[wv];
      }
      __syncthreads();
    }
    if (head) {
      double acc = 0.0;  // SG.reset 0
      for (int u = t; u < np; ++u) {
        const unsigned k = skeys[u];
        if ((int)(k >> kOrdTB) != row) break;
        acc = acc + vals[k & ((1u << kOrdTB) - 1u)];  // c + a * b in ascending k
      }
      Ci[base + off] = row;
      Cx[base + off] = acc;
    }
    running += total;
  }
  group_sync<NT>();  // the LDS image is reused by the next column
  stamp(5);
}

// ---- the workgroup path, pipelined over two columns -----------------------------------------------------------
// ord_column above waits in the look-back right after it has published its length: measured on C4, 25-30 % of
// the kernel's cycles.  Here the workgroup keeps the sorted keys and the products of a column in one of TWO
// result buffers and goes on to expand and sort its next column; the earlier column is finished (look-back,
// fold, write) after that — one whole column of work later, when its predecessors have long published.  A
// workgroup still only ever waits for columns with smaller tickets.

// Round 5: the two result buffers are the two ENDS of one ring of RING entry slots (a product column and a key column
// side by side): the pending column lies at one end with exactly its np entries, the column being sorted goes to the
// other end.  RING = 2 CAP: they always fit.  A narrower ring finishes the pending column first when they do not —
// measured and ruinous, see kOrdRingSmallNarrow above.
template <int CAP, int NBCAP, int RING>
struct OrdPipeLds {  // workspace + the ring
  static_assert(RING >= CAP, "a column must fit the ring by itself");
  static constexpr size_t kb_bytes = NBCAP * sizeof(double);
  static constexpr size_t start_bytes = NBCAP * sizeof(int);
  static constexpr size_t off_bytes = (NBCAP + 8) * sizeof(int);
  static constexpr size_t hist_bytes = (kOrdPBuckets + 8) * sizeof(int);
  static constexpr size_t scratch_bytes = 32 * sizeof(int);
  static constexpr size_t work_bytes = kb_bytes + start_bytes + off_bytes + hist_bytes + scratch_bytes;
  static constexpr size_t vals_bytes = (size_t)RING * sizeof(double), keys_bytes = (size_t)RING * sizeof(unsigned);
  static constexpr size_t total = work_bytes + vals_bytes + keys_bytes;
};

struct OrdPending {
  int64_t j = 0;
  int np = 0, count = 0;
  int before = 0;       // distinct rows in the parts of the sorted column that belong to earlier wavefronts
  int end = 0;          // which end of the ring holds the column (0: slots [0, np), 1: slots [RING - np, RING))
  bool valid = false;
};

// The sorted column is dealt to the four wavefronts in contiguous parts of whole 64-entry chunks: wavefront w counts,
// folds and writes positions [w L, (w + 1) L).  ordp_count leaves every wavefront the number of distinct rows before
// its part, so the write loop of ordp_finish needs no workgroup barrier (round 4 crossed two per 256 entries: ~10 per
// column) and every wavefront writes one contiguous stretch of Ci / Cx.
__device__ inline int ordp_part(int np) { return (((np + 3) >> 2) + 63) & ~63; }

// finish a column whose sorted keys and products wait in LDS: look-back, fold, write (scratch: 16 ints of its own)
__device__ inline void ordp_finish(const OrdPending &P, const unsigned *skeys, const double *vals, int *scratch, int tid,
                                   unsigned long long *__restrict__ status, int64_t *__restrict__ Cp,
                                   int *__restrict__ Ci, double *__restrict__ Cx, unsigned long long *diag = nullptr) {
  const int lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    const int64_t e = ord_lookback(status, P.j, P.count, false, diag);
    if (lane == 0) { scratch[10] = (int)(e & 0xffffffffll); scratch[11] = (int)(e >> 32); }
  }
  __syncthreads();
  const int64_t base = ((int64_t)scratch[11] << 32) | (int64_t)(unsigned)scratch[10];
  if (tid == 0) Cp[P.j] = base;
  const int np = P.np, L = ordp_part(np);
  const int t1 = min(np, (wave + 1) * L);
  int64_t out = base + P.before;
  for (int t0 = wave * L; t0 < t1; t0 += 64) {
    const int t = t0 + lane;
    int row = 0;
    bool head = false;
    if (t < t1) {
      row = (int)(skeys[t] >> kOrdTB);
      head = t == 0 || (int)(skeys[t - 1] >> kOrdTB) != row;
    }
    const unsigned long long m = __ballot(head);
    if (head) {
      const int64_t o = out + __popcll(m & ((1ull << lane) - 1ull));
      double acc = 0.0;  // SG.reset 0
      for (int u = t; u < np; ++u) {  // (a run may reach into the next wavefront's part: its entries are no heads there)
        const unsigned k = skeys[u];
        if ((int)(k >> kOrdTB) != row) break;
        acc = acc + vals[k & ((1u << kOrdTB) - 1u)];  // c + a * b in ascending k
      }
      Ci[o] = row;
      Cx[o] = acc;
    }
    out += __popcll(m);
  }
}

// expand + sort column j into (keys, vals); returns the number of distinct rows, or -1 when a bucket overflowed:
// then `keys` holds the expanded keys in their original order (nb ascending runs) for the merge tree
// one entry (k, b) of B as the ordered kernel wants it: where A[:, k] starts, how long it is, and b — 16 bytes,
// one load, no dependent chain B.p -> B.i -> A.p in front of the gathers (ord_bmeta_kernel)
struct __attribute__((aligned(16))) OrdBMeta {
  int start, len;
  double b;
};

__global__ __launch_bounds__(256) void ord_bmeta_kernel(Csc A, Csc B, int64_t nnzB, OrdBMeta *__restrict__ out) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nnzB) return;
  const int k = B.i[q];
  const int s0 = A.p[k];
  OrdBMeta m;
  m.start = s0;
  m.len = A.p[k + 1] - s0;
  m.b = B.x[q];
  out[q] = m;
}

// PER = CAP / 256 slots per thread for the products it expands, ranks and scatters; the loops over them are fully
// unrolled (the slots live in registers).  A column of 1 024 products needs four of the six slots of the small shape:
// every unrolled loop leaves at the first slot beyond per = ceil(np / 256) — a wave-uniform scalar branch — instead of
// running the surplus slots predicated off (half again the instructions of such a column).  (One instantiation per
// value of `per` does the same and was measured SLOWER, 20.3 against 13.7 ms: 56 KB of code for sixteen wavefronts in
// different phases against a 64 KB instruction cache.)
template <int CAP, int NBCAP, int RING>
__device__ inline int ordp_sort(const Csc &A, const OrdBMeta *__restrict__ bmeta, int qs, int nb, int np, int bucket_shift,
                                unsigned char *work, unsigned *keys, double *vals, int tid,
                                unsigned long long *stamps = nullptr) {
  unsigned long long t_prev = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  auto stamp = [&](int phase) {  // SPL_SPGEMM_STAMPS=1 (diagnostic): cycles of thread 0 per phase
    if (stamps) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      stamps[phase] += now - t_prev;
      t_prev = now;
    }
  };
  constexpr int NT = 256, NBK = kOrdPBuckets, PER = CAP / NT;
  typedef OrdPipeLds<CAP, NBCAP, RING> PL;
  double *kb = reinterpret_cast<double *>(work);
  int *kstart = reinterpret_cast<int *>(work + PL::kb_bytes);
  int *koff = kstart + NBCAP;
  int *hist = koff + NBCAP + 8;
  int *scratch = hist + NBK + 8;
  const int lane = tid & 63;
  for (int q = tid; q < nb; q += NT) {
    const OrdBMeta m = bmeta[(int64_t)qs + q];
    kstart[q] = m.start;
    koff[q] = m.len;
    kb[q] = m.b;
  }
  for (int b = tid; b < NBK + 1; b += NT) hist[b] = 0;
  if (tid == 0) scratch[8] = 0;
  __syncthreads();
  {  // exclusive prefix sum of the extents (nb <= NBCAP <= 128: the first two wavefronts hold one entry per lane)
    const int v = tid < nb ? koff[tid] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) scratch[tid >> 6] = incl;
    __syncthreads();
    if ((tid >> 6) == 1) incl += scratch[0];
    if (tid < nb) koff[tid] = incl - v;
    if (tid == 0) koff[nb] = np;
    __syncthreads();
  }
  stamp(1);
  const int per = __builtin_amdgcn_readfirstlane((np + NT - 1) / NT);  // wave-uniform, in a scalar register
  unsigned myk[PER];
  int myrank[PER];
  {
    const int t0 = tid * per;
    int q = 0;
    if (t0 < np) {
      int lo = 0, hi = nb - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (koff[mid] <= t0) lo = mid; else hi = mid - 1;
      }
      q = lo;
    }
    int pp[PER], qq[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      if (u >= per) break;  // (wave-uniform)
      const int t = t0 + u;
      pp[u] = -1;
      qq[u] = 0;
      if (u < per && t < np) {
        while (t >= koff[q + 1]) ++q;
        pp[u] = kstart[q] + (t - koff[q]);
        qq[u] = q;
      }
    }
    int rows[PER];
    double av[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      if (u >= per) break;
      rows[u] = pp[u] >= 0 ? A.i[pp[u]] : 0;
      av[u] = pp[u] >= 0 ? A.x[pp[u]] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      if (u >= per) break;
      myk[u] = 0xffffffffu;
      myrank[u] = 0;
      if (pp[u] >= 0) {
        const int t = t0 + u;
        myk[u] = ((unsigned)rows[u] << kOrdTB) | (unsigned)t;
        vals[t] = av[u] * kb[qq[u]];  // a * b, rounded once (Sparse.hs:699)
        myrank[u] = atomicAdd(&hist[rows[u] >> bucket_shift], 1);
      }
    }
  }
  __syncthreads();
  stamp(2);
  {  // exclusive scan of the histogram: two consecutive buckets per thread
    const int b0 = tid * 2;
    const int c0 = hist[b0], c1 = hist[b0 + 1];
    const int sum = c0 + c1;
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) scratch[tid >> 6] = incl;
    __syncthreads();
    int woff = 0;
    for (int wv = 0; wv < (tid >> 6); ++wv) woff += scratch[wv];
    __syncthreads();
    const int run = incl + woff - sum;
    hist[b0] = run;
    hist[b0 + 1] = run + c0;
    if (tid == NT - 1) hist[NBK] = np;
    if (c0 > kOrdBucketLimit || c1 > kOrdBucketLimit) scratch[8] = 1;
  }
  __syncthreads();
  stamp(3);
  if (scratch[8] != 0) {  // a crowded bucket: hand the keys over in run order, the caller merges them
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      if (u >= per) break;
      if (myk[u] != 0xffffffffu) keys[myk[u] & ((1u << kOrdTB) - 1u)] = myk[u];
    }
    __syncthreads();
    return -1;
  }
  // Every key goes to its bucket in arrival order; then each thread RANKS its own keys (still in registers)
  // inside their buckets — the number of smaller keys there; keys are unique, so the ranks are a permutation —
  // and stores them at bucket start + rank.  The reads of a bucket are independent LDS loads (no
  // data-dependent chain of load, compare, move as in an insertion sort: that chain, in the slowest lane,
  // was 30 % of the kernel on C4).
  int bs[PER], be[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    if (u >= per) break;
    bs[u] = 0;
    be[u] = 0;
    if (myk[u] != 0xffffffffu) {
      const int b = (int)((myk[u] >> kOrdTB) >> bucket_shift);
      bs[u] = hist[b];
      be[u] = hist[b + 1];
      keys[bs[u] + myrank[u]] = myk[u];
    }
  }
  __syncthreads();
  int rank[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    if (u >= per) break;
    rank[u] = 0;
    for (int i = bs[u]; i < be[u]; i += 4) {  // four independent loads per wait; the index is clamped, the test masks
      const unsigned k0 = keys[i], k1 = keys[min(i + 1, np - 1)], k2 = keys[min(i + 2, np - 1)], k3 = keys[min(i + 3, np - 1)];
      rank[u] += (k0 < myk[u] ? 1 : 0) + ((i + 1 < be[u] && k1 < myk[u]) ? 1 : 0) + ((i + 2 < be[u] && k2 < myk[u]) ? 1 : 0) +
                 ((i + 3 < be[u] && k3 < myk[u]) ? 1 : 0);
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    if (u >= per) break;
    if (myk[u] != 0xffffffffu) keys[bs[u] + rank[u]] = myk[u];
  }
  __syncthreads();
  stamp(4);
  return 0;
}

// distinct rows of a sorted column (workgroup-wide); *before = those in the parts of the wavefronts before this one
__device__ inline int ordp_count(const unsigned *skeys, int np, int *scratch, int tid, int *before) {
  const int lane = tid & 63, wave = tid >> 6, L = ordp_part(np);
  const int t1 = min(np, (wave + 1) * L);
  int count = 0;
  for (int t0 = wave * L; t0 < t1; t0 += 64) {
    const int t = t0 + lane;
    const bool head = t < t1 && (t == 0 || (skeys[t] >> kOrdTB) != (skeys[t - 1] >> kOrdTB));
    count += __popcll(__ballot(head));
  }
  if (lane == 0) scratch[4 + wave] = count;
  __syncthreads();
  const int c0 = scratch[4], c1 = scratch[5], c2 = scratch[6], c3 = scratch[7];
  __syncthreads();
  *before = wave == 0 ? 0 : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
  return c0 + c1 + c2 + c3;
}

// column classes of the ordered form
__global__ __launch_bounds__(256) void ord_classify_kernel(Csc B, int64_t ncolsB, const int64_t *__restrict__ nprod,
                                                           unsigned char *__restrict__ cls, int64_t *__restrict__ heavy_prod,
                                                           int ord_cap, int ord_nb) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ncolsB) return;
  const int64_t np = nprod[j];
  const int nb = B.p[j + 1] - B.p[j];
  const int c = np == 0 ? 0 : (np <= kOrdWaveCap && nb <= kOrdWaveNb) ? 1 : (np <= ord_cap && nb <= ord_nb) ? 2 : 3;
  cls[j] = (unsigned char)c;
  heavy_prod[j] = c == 3 ? np : 0;
}

// Tasks of the ordered kernel, in column order: an aligned tile of four light columns (classes 0 / 1: a
// wavefront each, in parallel) is ONE task; every column of any other tile is a task of its own.  A
// workgroup must never hold several chain elements that it processes one after the other: the length of
// its last column would only be known after the earlier ones were WRITTEN, which waits for the columns
// before them — the whole product would serialise along that chain.
__global__ __launch_bounds__(256) void ord_task_count_kernel(int64_t ncolsB, const unsigned char *__restrict__ cls,
                                                             int *__restrict__ ntasks) {
  const int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t j0 = tile * 4;
  if (j0 >= ncolsB) return;
  bool light = true;
  int n = 0;
  for (int w = 0; w < 4 && j0 + w < ncolsB; ++w) { light = light && cls[j0 + w] <= 1; ++n; }
  ntasks[tile] = light ? 1 : n;
}
// A task record is everything the kernel needs to start on a column, in ONE 16-byte load after the ticket:
// x = the column (or ~first column of a tile of light columns), y = its products, z = its first entry in B,
// w = entries of B | class << 16.  (Before: ticket -> task -> class, products, B.p — three dependent round trips.)
__global__ __launch_bounds__(256) void ord_task_fill_kernel(Csc B, int64_t ncolsB, const unsigned char *__restrict__ cls,
                                                            const int64_t *__restrict__ nprod,
                                                            const int64_t *__restrict__ task_off,
                                                            int4 *__restrict__ tasks) {
  const int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t j0 = tile * 4;
  if (j0 >= ncolsB) return;
  const int64_t o = task_off[tile];
  bool light = true;
  int n = 0;
  for (int w = 0; w < 4 && j0 + w < ncolsB; ++w) { light = light && cls[j0 + w] <= 1; ++n; }
  if (light) { tasks[o] = make_int4((int)~j0, 0, 0, 0); return; }  // a tile of light columns: one task
  for (int w = 0; w < n; ++w) {
    const int64_t j = j0 + w;
    const int c = cls[j];
    const int qs = B.p[j], nb = B.p[j + 1] - qs;
    // classes 1 / 2 have at most 2048 products and 128 entries of B; the others do not use y / w's low half
    tasks[o + w] = make_int4((int)j, c == 1 || c == 2 ? (int)nprod[j] : 0, qs, (nb & 0xffff) | (c << 16));
  }
}

template <int CAP, int NBCAP, int RING, int WGS>
__global__ __launch_bounds__(256, WGS) void spgemm_ordered_kernel(Csc A, Csc B, int64_t ncolsB, int bucket_shift_wave,
                                                             int bucket_shift_group, const int64_t *__restrict__ nprod,
                                                             const unsigned char *__restrict__ cls,
                                                             const int64_t *__restrict__ heavy_slot,
                                                             const int *__restrict__ heavy_count,
                                                             const int *__restrict__ Ti, const double *__restrict__ Tx,
                                                             unsigned long long *__restrict__ status,
                                                             unsigned long long *__restrict__ ticket,
                                                             const int4 *__restrict__ tasks, int64_t ntasks,
                                                             const OrdBMeta *__restrict__ bmeta,
                                                             int64_t *__restrict__ Cp, int *__restrict__ Ci,
                                                             double *__restrict__ Cx, unsigned long long *__restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ long long s_tile;
  __shared__ long long s_base;
  __shared__ int4 s_rec;
  typedef OrdLds<kOrdWaveCap, kOrdWaveNb> LW;
  typedef OrdPipeLds<CAP, NBCAP, RING> PL;
  constexpr size_t wave_bytes = (LW::total + 15) / 16 * 16;
  static_assert(4 * wave_bytes <= PL::total, "the four one-wavefront images of a tile of light columns lie over the ring");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long acc_stamps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long *const local_stamps = stamps ? acc_stamps : nullptr;
  // pipelined workgroup path: workspace, then the ring (value column, key column)
  unsigned char *work = smem;
  double *const ring_vals = reinterpret_cast<double *>(smem + PL::work_bytes);
  unsigned *const ring_keys = reinterpret_cast<unsigned *>(smem + PL::work_bytes + PL::vals_bytes);
  auto pvals = [&](int end, int np) { return end ? ring_vals + (RING - np) : ring_vals; };
  auto pkeys = [&](int end, int np) { return end ? ring_keys + (RING - np) : ring_keys; };
  int *scratch_a = reinterpret_cast<int *>(smem + PL::work_bytes) - 32;  // the workspace's scratch: [0,16) sort, [16,32) finish
  int *scratch_b = scratch_a + 16;
  OrdPending pend;
  auto finish_pending = [&]() {
    if (pend.valid) {
      ordp_finish(pend, pkeys(pend.end, pend.np), pvals(pend.end, pend.np), scratch_b, tid, status, Cp, Ci, Cx,
                  (stamps && tid == 0) ? acc_stamps + 8 : nullptr);
      pend.valid = false;
      __syncthreads();
    }
  };
  unsigned long long t_mark = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  auto mark = [&](int phase) {
    if (stamps) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      acc_stamps[phase] += now - t_mark;
      t_mark = now;
    }
  };
  // Thread 0 asks for the next ticket and its task record once the column in hand is sorted (two dependent round
  // trips that travel while the column is counted and the pending one finished), so the top of the loop finds both
  // in registers.  Taking the ticket EARLIER — when the column starts — was measured and is 45 % slower (20.2 against
  // 13.9 ms on C4): a ticket held for a whole column before its column is even started ages every later column's
  // look-back.  A workgroup only ever waits for columns with smaller tickets, all held by resident workgroups.
  long long nxt_ticket = 0;
  int4 nxt_rec = make_int4(0, 0, 0, 0);
  auto request_next = [&]() {
    if (tid == 0) {
      nxt_ticket = (long long)atomicAdd(ticket, 1ull);
      nxt_rec = tasks[nxt_ticket < ntasks ? nxt_ticket : ntasks - 1];
    }
  };
  request_next();
  for (;;) {
    mark(7);  // (whatever is not covered below)
    if (tid == 0) { s_tile = nxt_ticket; s_rec = nxt_rec; }
    __syncthreads();
    const int64_t tk = s_tile;
    const int4 rec = s_rec;
    __syncthreads();
    if (tk >= ntasks) break;
    mark(0);
    if (rec.x < 0) {  // four light columns: one wavefront each, no workgroup barrier inside
      finish_pending();  // (their LDS image overlaps the ring)
      const int64_t j = (int64_t)~rec.x + wave;
      const int cw = j < ncolsB ? (int)cls[j] : -1;
      if (cw == 0) {
        const int64_t e = ord_chain(status, j, 0);
        if (lane == 0) Cp[j] = e;
      } else if (cw == 1) {
        ord_column<64, kOrdWaveCap, kOrdWaveNb>(A, B, j, (int)nprod[j], bucket_shift_wave, smem + wave * wave_bytes, lane,
                                                status, Cp, Ci, Cx, wave == 0 ? local_stamps : nullptr);
      }
      request_next();
    } else {
      const int64_t j = rec.x;
      const int cw = rec.w >> 16, nbj = rec.w & 0xffff;
      if (cw == 1 || cw == 2) {
        // sort this column into the end of the ring the pending one does not use, publish its length, THEN finish the
        // pending column: its look-back has had a whole column of work to become a formality
        const int np = rec.y;
        if (pend.valid && pend.np + np > RING) finish_pending();  // the two do not fit the ring together
        const int end = pend.valid ? 1 - pend.end : 0;
        unsigned *keys = pkeys(end, np);
        double *vals = pvals(end, np);
        const int st = ordp_sort<CAP, NBCAP, RING>(A, bmeta, rec.z, nbj, np, bucket_shift_group, work, keys, vals, tid,
                                                   tid == 0 ? local_stamps : nullptr);
        request_next();
        if (stamps) t_mark = __builtin_amdgcn_s_memtime();
        if (st < 0) {  // crowded bucket: merge tree over the runs; its second key buffer is the other end of the ring when
          finish_pending();  // two such columns fit, else the tail of the value column behind this column's products
          unsigned *second = 2 * np <= RING ? pkeys(1 - end, np)
                                            : reinterpret_cast<unsigned *>(end ? ring_vals : ring_vals + np);
          const int *koff = reinterpret_cast<const int *>(work + PL::kb_bytes + PL::start_bytes);
          const unsigned *sorted = ord_merge_tree<256, CAP>(keys, second, koff, nbj, np, tid);
          if (sorted != keys) {  // (the merge tree ends with a barrier)
            for (int t = tid; t < np; t += 256) keys[t] = sorted[t];
            __syncthreads();
          }
        }
        int before = 0;
        const int count = ordp_count(keys, np, scratch_a, tid, &before);
        if (wave == 0) ord_publish(status, j, count);
        mark(5);
        finish_pending();
        mark(6);
        pend.j = j; pend.np = np; pend.count = count; pend.before = before; pend.end = end; pend.valid = true;
      } else {  // empty, or computed beforehand into its scratch slot: publish the length, copy
        const int cnt = cw == 3 ? heavy_count[j] : 0;
        if (wave == 0) {
          const int64_t e = ord_chain(status, j, cnt);
          if (lane == 0) { s_base = e; Cp[j] = e; }
        }
        __syncthreads();
        const int64_t base = s_base;
        if (cw == 3) {
          const int64_t src = heavy_slot[j];
          for (int t = tid; t < cnt; t += 256) {
            Ci[base + t] = Ti[src + t];
            Cx[base + t] = Tx[src + t];
          }
        }
        __syncthreads();
        request_next();
      }
    }
  }
  finish_pending();
  if (stamps && tid == 0)
    for (int i = 0; i < 12; ++i) atomicAdd(stamps + i, acc_stamps[i]);
}

// products in the columns where the ordered kernel's workgroup path pays: out[0] with the large shape
// (513 ... 2048 products, <= 128 entries of B), out[1] with the small one (... 1536, <= 96)
__global__ __launch_bounds__(256) void ord_share_kernel(Csc B, int64_t ncolsB, const int64_t *__restrict__ nprod,
                                                        unsigned long long *__restrict__ out) {
  // grid-stride over the columns, one pair of atomics per WORKGROUP: same-address atomics cost ~12 ns apiece on this part
  // (round 4: one pair per wavefront of 64 columns, 32 768 of them = 0.42 ms of C4's 16.7)
  unsigned long long v = 0, w = 0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < ncolsB; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t np = nprod[j];
    const int nb = B.p[j + 1] - B.p[j];
    if (np > 2 * kOrdWaveCap && np <= kOrdAdmitLarge && nb <= kOrdPNbLarge) v += (unsigned long long)np;
    if (np > 2 * kOrdWaveCap && np <= kOrdAdmitSmall && nb <= kOrdPNbSmall) w += (unsigned long long)np;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    v += __shfl_xor(v, d, 64);
    w += __shfl_xor(w, d, 64);
  }
  __shared__ unsigned long long sv[4], sw[4];
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = v; sw[threadIdx.x >> 6] = w; }
  __syncthreads();
  if (threadIdx.x == 0) {
    v = sv[0] + sv[1] + sv[2] + sv[3];
    w = sw[0] + sw[1] + sw[2] + sw[3];
    if (v) atomicAdd(out, v);
    if (w) atomicAdd(out + 1, w);
  }
}

// upper bound of a column's length in the result: exact for the columns computed beforehand, its products otherwise
__global__ __launch_bounds__(256) void ord_bound_kernel(int64_t ncolsB, const unsigned char *__restrict__ cls,
                                                        const int64_t *__restrict__ nprod, const int *__restrict__ heavy_count,
                                                        int64_t *__restrict__ bound) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < ncolsB) bound[j] = cls[j] == 3 ? (int64_t)heavy_count[j] : nprod[j];
}

__global__ void ord_total_kernel(const unsigned long long *__restrict__ status, int64_t ncolsB, int64_t *__restrict__ Cp) {
  if (threadIdx.x == 0 && blockIdx.x == 0) Cp[ncolsB] = (int64_t)(status[ncolsB - 1] & kOrdValueMask);
}

}  // namespace

// C = A B; all pointers are device pointers.  Cp is 64-bit (nnz(C) may exceed 2^31).
void spgemm_device(int64_t nrowsA, int64_t ncolsA, const int *Ap, const int *Ai, const double *Ax,
                   int64_t ncolsB, const int *Bp, const int *Bi, const double *Bx, DBuf<int64_t> &Cp,
                   DBuf<int> &Ci, DBuf<double> &Cx, int64_t *nnzC, int64_t *products, hipStream_t s) {
  (void)ncolsA;
  const bool timing = getenv("SPL_SPGEMM_TIMING") != nullptr;  // host laps on stderr (diagnostic; each lap synchronises)
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[spgemm] %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  Csc A{Ap, Ai, Ax}, B{Bp, Bi, Bx};
  Cp.alloc((size_t)ncolsB + 1);
  *nnzC = 0;
  if (products) *products = 0;
  if (ncolsB == 0) {
    SPL_HIP(hipMemsetAsync(Cp.get(), 0, sizeof(int64_t), s));
    Ci.alloc(0);
    Cx.alloc(0);
    SPL_HIP(hipStreamSynchronize(s));
    return;
  }
  DBuf<int64_t> nprod((size_t)ncolsB), medium_list((size_t)ncolsB), xlarge_list((size_t)ncolsB),
      dense_list((size_t)ncolsB);
  DBuf<int> list_counts(4), counts((size_t)ncolsB);
  SPL_HIP(hipMemsetAsync(list_counts.get(), 0, 4 * sizeof(int), s));
  // Which single-pass form?  The ordered form (every column straight to its final place, no compaction, the
  // pipelined workgroup path) wins where the products sit in columns of about a thousand products — C4: 0.023 s
  // vs 0.030 s — and loses where they sit in light columns (64 products per column: 8.2 ms vs 5.4 ms; 256 per
  // column: 47 ms vs 32 ms; its LDS image admits 12 wavefronts per CU where the plain one-wavefront-per-column
  // kernel runs 32; profiles/r02_spgemm_ordered_phases.txt).  It is taken when at least half of the products
  // belong to columns of 513 ... 2048 products; SPL_SPGEMM_ORDERED=1 / 0 forces / forbids it.  Its packed 32-bit keys need
  // nrows < 2^21 (strictly: the key of row 2^21 - 1 with the last tie-break position is the 0xffffffff "no product" sentinel).
  const char *xh_env = getenv("SPL_SPGEMM_X_AS_HEAVY");
  // Columns of 2049 .. 4096 products (bin X) go with the heavy ones through the row-range kernel: its counting + rank
  // sort beats the merge tree of the workgroup kernels there (scale 20, edge factor 44: 0.0616 -> 0.0527 s, 48: 0.0885 ->
  // 0.0691 s).  SPL_SPGEMM_X_AS_HEAVY=0 keeps them in bin X.
  const int x_heavy = (!(xh_env && xh_env[0] == '0') && nrowsA <= (1LL << 21)) ? 1 : 0;
  const char *ord_env = getenv("SPL_SPGEMM_ORDERED");
  const char *two_pass_env = getenv("SPL_SPGEMM_TWO_PASS"), *split_keys_env = getenv("SPL_SPGEMM_SPLIT_KEYS");
  const bool ordered_possible = !(ord_env && ord_env[0] == '0') && nrowsA < (1LL << kOrdMaxRowBits) &&
                                !(two_pass_env && two_pass_env[0] == '1') && !(split_keys_env && split_keys_env[0] == '1');
  hipLaunchKernelGGL(products_kernel, dim3(blocks_for(ncolsB, 32)), dim3(256), 0, s, A, B, ncolsB, nprod.get(),
                     medium_list.get(), xlarge_list.get(), dense_list.get(), list_counts.get(), 0, 0, 0, x_heavy);
  int hc[4] = {0, 0, 0, 0};
  SPL_HIP(hipMemcpyAsync(hc, list_counts.get(), 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  DBuf<int64_t> pscan((size_t)ncolsB + 1);  // products before column j: its upper-bound output slot
  exclusive_scan_i64(nprod.get(), pscan.get(), ncolsB, s);
  int64_t total_products = 0;
  SPL_HIP(hipMemcpy(&total_products, pscan.get() + ncolsB, sizeof(int64_t), hipMemcpyDeviceToHost));
  if (products) *products = total_products;
  lap("products, scan, totals");
  // Single pass when the upper bound nnz(C) <= products fits comfortably in HBM: the numeric
  // kernels write every column at its slot and report its length, one copy compacts.  This
  // skips the symbolic pass (a second expand + sort of every column).  SPL_SPGEMM_TWO_PASS=1
  // forces the symbolic + numeric form.
  bool single_pass = false;
  {
    const size_t free_b = device_free_bytes();
    const char *force = getenv("SPL_SPGEMM_TWO_PASS");
    single_pass = !(force && force[0] == '1') && total_products > 0 &&
                  (double)total_products * 24.0 < 0.5 * (double)free_b;
  }
  bool ordered = false;
  // shape of the ordered kernel's columns: the small one (four workgroups per CU) unless a good part of the products
  // sits in columns only the large one takes (1537 ... 2048 products, or 97 ... 128 entries of B);
  // SPL_SPGEMM_ORDERED_SHAPE=small|large forces it (tests)
  bool large_shape = false;
  if (ordered_possible && single_pass) {
    DBuf<unsigned long long> share(2);
    SPL_HIP(hipMemsetAsync(share.get(), 0, 2 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(ord_share_kernel, dim3(std::min(blocks_for(ncolsB, 256), 256u)), dim3(256), 0, s, B, ncolsB, nprod.get(), share.get());
    unsigned long long h[2] = {0, 0};
    SPL_HIP(hipMemcpyAsync(h, share.get(), sizeof(h), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    ordered = (ord_env && ord_env[0] == '1') || 2.0 * (double)h[0] >= (double)total_products;
    large_shape = (double)(h[0] - h[1]) > 0.08 * (double)total_products;
    if (const char *e = getenv("SPL_SPGEMM_ORDERED_SHAPE")) large_shape = e[0] == 'l';
  }
  int ord_cap = large_shape ? kOrdAdmitLarge : kOrdAdmitSmall;
  const int ord_nb = large_shape ? kOrdPNbLarge : kOrdPNbSmall;
  if (const char *e = getenv("SPL_SPGEMM_ORD_CAP")) {  // ablation: admit fewer products per column to the ordered kernel's own path
    const int v = atoi(e);
    if (v >= 2 * kOrdWaveCap && v < ord_cap) ord_cap = v;
  }
  lap("form chosen (share)");
  if (ordered) {  // the columns the ordered kernel handles itself leave the bin lists
    SPL_HIP(hipMemsetAsync(list_counts.get(), 0, 4 * sizeof(int), s));
    hipLaunchKernelGGL(relist_ordered_kernel, dim3(blocks_for(ncolsB, 256)), dim3(256), 0, s, B, ncolsB, nprod.get(),
                       medium_list.get(), xlarge_list.get(), dense_list.get(), list_counts.get(), ord_cap, ord_nb, x_heavy);
    SPL_HIP(hipMemcpyAsync(hc, list_counts.get(), 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
  }
  lap("ordered bin lists");
  const int nmedium = hc[0], nxlarge = hc[1], ndense = hc[2], nxback = hc[3];
  const int64_t *xback_list = xlarge_list.get() + (ncolsB - nxback);  // the columns of bin X with a long column of B
  typedef EscLds<kMediumProducts, kMediumB, false> LMs;
  typedef EscLds<kLargeProducts, kLargeB, false> LXs;
  typedef EscLds<kLargeProducts, kMediumB, false> LXsShort;
  typedef EscLds<kLargeProducts, kLargeB, true, false> LXn;
  typedef EscLds<kLargeProducts, kLargeB, true, true> LXn32;
  // 32-bit packed sort keys need row + tie-break bits to fit 31 bits in every bin
  // SPL_SPGEMM_SPLIT_KEYS=1 (tests) forces the row + 16-bit-position keys that large matrices need
  const char *split_env = getenv("SPL_SPGEMM_SPLIT_KEYS");
  const bool allow32 = !(split_env && split_env[0] == '1');
  const bool key32_s = allow32 && nrowsA <= (1LL << (31 - ilog2_ceil(kSmallProducts)));
  const bool key32_m = allow32 && nrowsA <= (1LL << (31 - ilog2_ceil(kMediumProducts)));
  const bool key32_x = allow32 && nrowsA <= (1LL << (31 - ilog2_ceil(kLargeProducts)));
  static std::atomic<uint64_t> attr_set{0};  // one bit per device
  if (first_use_on_this_device(attr_set)) {
    SPL_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void *>(&spgemm_block_kernel<kLargeProducts, kLargeB, true, false>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LXn::total));
    SPL_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void *>(&spgemm_block_kernel<kLargeProducts, kLargeB, true, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LXn32::total));
    mark_used_on_this_device(attr_set);
  }
  // Heavy columns (bin L): the row-range kernel takes those it can keep in LDS (spgemm_range_kernel), the others —
  // appended to a second list — go to the dense accumulators, whose pool is only allocated when that list is not empty.
  int pool = 0;
  DBuf<unsigned char> pool_flags;
  DBuf<double> pool_vals;
  auto ensure_pool = [&](int ncolumns) {
    if (pool > 0) return;
    pool = ncolumns < kMaxPool ? ncolumns : kMaxPool;
    // keep the accumulator pool under ~32 GB (of 288) and under a quarter of what is free right now (the result
    // buffers of a single pass are budgeted against half of it; a busy device gets a smaller pool, not an error)
    const int64_t per_slot = 9 * (nrowsA > 0 ? nrowsA : 1);
    int64_t budget = (int64_t)32e9;
    const int64_t quarter = (int64_t)(device_free_bytes() / 4);
    if (quarter < budget) budget = quarter;
    const int64_t cap = budget / per_slot;
    if (pool > cap) pool = (int)(cap < 1 ? 1 : cap);
    const size_t flag_stride = ((size_t)nrowsA + 15) & ~(size_t)15;  // spgemm_dense_kernel reads flags 16 at a time
    pool_flags.alloc((size_t)pool * flag_stride);
    pool_vals.alloc((size_t)pool * (size_t)nrowsA);
    SPL_HIP(hipMemsetAsync(pool_flags.get(), 0, (size_t)pool * flag_stride, s));
    SPL_HIP(hipMemsetAsync(pool_vals.get(), 0, (size_t)pool * (size_t)nrowsA * sizeof(double), s));
  };
  DBuf<int64_t> fallback_list;
  DBuf<int> fallback_count;
  DBuf<unsigned> range_key;
  DBuf<double> range_val;
  unsigned range_grid = 0;
  const char *range_env = getenv("SPL_SPGEMM_RANGE");
  const bool use_range = ndense > 0 && !(range_env && range_env[0] == '0') && nrowsA <= (1LL << 21);
  const int64_t *dense_run_list = dense_list.get();
  if (use_range) {
    fallback_list.alloc((size_t)ndense);
    fallback_count.alloc(1);
    range_grid = (unsigned)(ndense < 512 ? ndense : 512);  // two resident workgroups per CU
    range_key.alloc((size_t)range_grid * kRngMaxProducts);  // 6 MB of scratch per workgroup: key + product
    range_val.alloc((size_t)range_grid * kRngMaxProducts);
    dense_run_list = fallback_list.get();
    static std::atomic<uint64_t> rng_set{0};
    int dev = 0;
    SPL_HIP(hipGetDevice(&dev));
    if (!(rng_set.load(std::memory_order_acquire) >> (dev & 63) & 1u)) {
      SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spgemm_range_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRngLdsBytes));
      SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spgemm_range_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRngLdsBytes));
      rng_set.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
  }
  // runs the row-range kernel over the heavy columns; returns how many are left for the dense accumulators
  auto run_range = [&](bool numeric, int *cnt, const int64_t *where, int *oi, double *ox) -> int {
    if (!use_range) return ndense;
    SPL_HIP(hipMemsetAsync(fallback_count.get(), 0, sizeof(int), s));
    if (numeric)
      hipLaunchKernelGGL(spgemm_range_kernel<true>, dim3(range_grid), dim3(kRngThreads), kRngLdsBytes, s, A, B, nrowsA, dense_list.get(),
                         ndense, nprod.get(), cnt, where, oi, ox, fallback_list.get(), fallback_count.get(), range_key.get(), range_val.get());
    else
      hipLaunchKernelGGL(spgemm_range_kernel<false>, dim3(range_grid), dim3(kRngThreads), kRngLdsBytes, s, A, B, nrowsA, dense_list.get(),
                         ndense, nprod.get(), cnt, where, oi, ox, fallback_list.get(), fallback_count.get(), range_key.get(), range_val.get());
    int nfb = 0;
    SPL_HIP(hipMemcpyAsync(&nfb, fallback_count.get(), sizeof(int), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    return nfb;
  };

  DBuf<int> Ti;
  DBuf<double> Tx;
  const int64_t *slots = Cp.get();  // where the numeric kernels write column j
  int *out_i = nullptr;
  double *out_x = nullptr;
  int *numeric_counts = nullptr;
  DBuf<unsigned char> cls;
  DBuf<int64_t> heavy_prod, heavy_slot;
  int64_t total_heavy = 0;
  if (ordered) {
    // heavy columns (beyond the ordered kernel's LDS budget) go to scratch slots first, by the kernels above
    cls.alloc((size_t)ncolsB);
    heavy_prod.alloc((size_t)ncolsB);
    heavy_slot.alloc((size_t)ncolsB + 1);
    hipLaunchKernelGGL(ord_classify_kernel, dim3(blocks_for(ncolsB, 256)), dim3(256), 0, s, B, ncolsB, nprod.get(),
                       cls.get(), heavy_prod.get(), ord_cap, ord_nb);
    exclusive_scan_i64(heavy_prod.get(), heavy_slot.get(), ncolsB, s);
    SPL_HIP(hipMemcpyAsync(&total_heavy, heavy_slot.get() + ncolsB, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    Ti.alloc((size_t)total_heavy);
    Tx.alloc((size_t)total_heavy);
    slots = heavy_slot.get();
    out_i = Ti.get();
    out_x = Tx.get();
    numeric_counts = counts.get();
  } else if (single_pass) {
    Ti.alloc((size_t)total_products);
    Tx.alloc((size_t)total_products);
    slots = pscan.get();
    out_i = Ti.get();
    out_x = Tx.get();
    numeric_counts = counts.get();
  } else {
    // ---- symbolic: nnz per column
    hipLaunchKernelGGL((spgemm_wave_kernel<false, false>), dim3(blocks_for(ncolsB, 4)), dim3(256), 0, s, A, B,
                       ncolsB, nprod.get(), counts.get(), (const int64_t *)nullptr, (int *)nullptr,
                       (double *)nullptr);
    lap("symbolic: one wavefront per column");
    if (nmedium > 0)
      hipLaunchKernelGGL((spgemm_block_kernel<kMediumProducts, kMediumB, false, false>), dim3((unsigned)nmedium),
                         dim3(256), LMs::total, s, A, B, medium_list.get(), nprod.get(), counts.get(),
                         (const int64_t *)nullptr, (int *)nullptr, (double *)nullptr);
    if (nxlarge > 0)
      hipLaunchKernelGGL((spgemm_block_kernel<kLargeProducts, kMediumB, false, false>), dim3((unsigned)nxlarge),
                         dim3(256), LXsShort::total, s, A, B, xlarge_list.get(), nprod.get(), counts.get(),
                         (const int64_t *)nullptr, (int *)nullptr, (double *)nullptr);
    if (nxback > 0)
      hipLaunchKernelGGL((spgemm_block_kernel<kLargeProducts, kLargeB, false, false>), dim3((unsigned)nxback),
                         dim3(256), LXs::total, s, A, B, xback_list, nprod.get(), counts.get(),
                         (const int64_t *)nullptr, (int *)nullptr, (double *)nullptr);
    lap("symbolic: workgroup bins");
    if (ndense > 0) {
      const int nfb = run_range(false, counts.get(), (const int64_t *)nullptr, (int *)nullptr, (double *)nullptr);
      lap("symbolic: heavy columns, row ranges");
      if (nfb > 0) {
        ensure_pool(nfb);
        hipLaunchKernelGGL(spgemm_dense_kernel<false>, dim3((unsigned)pool), dim3(256), 0, s, A, B, nrowsA,
                           dense_run_list, nfb, pool_flags.get(), (double *)nullptr, counts.get(),
                           (const int64_t *)nullptr, (int *)nullptr, (double *)nullptr);
      }
    }
    lap("symbolic: heavy columns, dense accumulators");
    exclusive_scan_i32_to_i64(counts.get(), Cp.get(), ncolsB, s);
    int64_t nz = 0;
    SPL_HIP(hipMemcpyAsync(&nz, Cp.get() + ncolsB, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    Ci.alloc((size_t)nz);
    Cx.alloc((size_t)nz);
    *nnzC = nz;
    if (nz == 0) return;
    out_i = Ci.get();
    out_x = Cx.get();
    lap("scan, result buffers");
  }

  // ---- numeric: every path writes its column already sorted by row
#define SPL_NUMERIC_WAVE(K32)                                                                                  \
  hipLaunchKernelGGL((spgemm_wave_kernel<true, K32>), dim3(blocks_for(ncolsB, 4)), dim3(256), 0, s, A, B, ncolsB, \
                     nprod.get(), numeric_counts, slots, out_i, out_x)
#define SPL_NUMERIC_BLOCK(CAP, NBCAP, K32, LIST, COUNT)                                                        \
  hipLaunchKernelGGL((spgemm_block_kernel<CAP, NBCAP, true, K32>), dim3((unsigned)(COUNT)), dim3(256),         \
                     (EscLds<CAP, NBCAP, true, K32>::total), s, A, B, LIST, nprod.get(), numeric_counts,         \
                     slots, out_i, out_x)
  if (!ordered) { if (key32_s) SPL_NUMERIC_WAVE(true); else SPL_NUMERIC_WAVE(false); }
  if (nmedium > 0) {
    if (key32_m) SPL_NUMERIC_BLOCK(kMediumProducts, kMediumB, true, medium_list.get(), nmedium);
    else SPL_NUMERIC_BLOCK(kMediumProducts, kMediumB, false, medium_list.get(), nmedium);
  }
  if (nxlarge > 0) {  // bin X, short columns of B: 36 KB of LDS, four workgroups per CU
    if (key32_x) SPL_NUMERIC_BLOCK(kLargeProducts, kMediumB, true, xlarge_list.get(), nxlarge);
    else SPL_NUMERIC_BLOCK(kLargeProducts, kMediumB, false, xlarge_list.get(), nxlarge);
  }
  if (nxback > 0) {   // bin X, long columns of B (up to kLargeB entries): 64 KB, two per CU
    if (key32_x) SPL_NUMERIC_BLOCK(kLargeProducts, kLargeB, true, xback_list, nxback);
    else SPL_NUMERIC_BLOCK(kLargeProducts, kLargeB, false, xback_list, nxback);
  }
#undef SPL_NUMERIC_WAVE
#undef SPL_NUMERIC_BLOCK
  lap("numeric: wavefront and workgroup bins");
  if (ndense > 0) {
    DBuf<unsigned long long> dstamps;
    if (getenv("SPL_SPGEMM_STAMPS")) {
      dstamps.alloc(4);
      SPL_HIP(hipMemsetAsync(dstamps.get(), 0, 4 * sizeof(unsigned long long), s));
    }
    const int nfb = run_range(true, numeric_counts, slots, out_i, out_x);
    lap("numeric: heavy columns, row ranges");
    if (nfb > 0) {
      ensure_pool(nfb);
      hipLaunchKernelGGL(spgemm_dense_kernel<true>, dim3((unsigned)pool), dim3(256), 0, s, A, B, nrowsA,
                         dense_run_list, nfb, pool_flags.get(), pool_vals.get(), numeric_counts, slots, out_i,
                         out_x, dstamps.get());
    }
    lap("numeric: heavy columns, dense accumulators");
    if (getenv("SPL_SPGEMM_TIMING")) fprintf(stderr, "[spgemm] heavy columns: %d, of them %d through the dense accumulators\n", ndense, nfb);
    if (dstamps.get()) {
      unsigned long long h[4];
      SPL_HIP(hipMemcpy(h, dstamps.get(), sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "[spgemm dense] %d columns on %d workgroups: accumulate %llu ticks, gather %llu ticks (summed), longest "
              "column %llu ticks; medium %d, xlarge %d + %d columns\n", ndense, pool, h[0], h[1], h[2], nmedium, nxlarge, nxback);
    }
  }
  if (ordered) {
    // ---- every column to its final place, in column order
    // capacity of the result: the exact lengths of the columns computed beforehand + the products of the others
    // (an upper bound; trimmed below when many products merged)
    int64_t capacity = total_products;
    if (total_heavy > 0) {
      hipLaunchKernelGGL(ord_bound_kernel, dim3(blocks_for(ncolsB, 256)), dim3(256), 0, s, ncolsB, cls.get(), nprod.get(),
                         counts.get(), heavy_prod.get());
      DBuf<int64_t> bscan((size_t)ncolsB + 1);
      exclusive_scan_i64(heavy_prod.get(), bscan.get(), ncolsB, s);
      SPL_HIP(hipMemcpyAsync(&capacity, bscan.get() + ncolsB, sizeof(int64_t), hipMemcpyDeviceToHost, s));
      SPL_HIP(hipStreamSynchronize(s));
    }
    lap("listed kernels (heavy columns)");
    Ci.alloc((size_t)capacity);
    Cx.alloc((size_t)capacity);
    lap("result buffers");
    DBuf<unsigned long long> status((size_t)ncolsB + 1);
    SPL_HIP(hipMemsetAsync(status.get(), 0, ((size_t)ncolsB + 1) * sizeof(unsigned long long), s));
    unsigned long long *ticket = status.get() + ncolsB;
    int rb = 0;
    while ((1LL << rb) < nrowsA) ++rb;
    const int sh_wave = rb > 7 ? rb - 7 : 0, sh_group = rb > 9 ? rb - 9 : 0;  // 128 / 512 buckets
    typedef OrdLds<kOrdWaveCap, kOrdWaveNb> LW;
    // SPL_SPGEMM_RING=narrow (ablation): the small shape with 2 400 ring slots, five workgroups per CU (see kOrdRingSmallNarrow)
    const char *ring_env = getenv("SPL_SPGEMM_RING");
    const bool narrow_ring = !large_shape && ring_env && ring_env[0] == 'n';
    const size_t pipe_lds = large_shape ? (size_t)OrdPipeLds<kOrdCapLarge, kOrdPNbLarge, kOrdRingLarge>::total
                            : narrow_ring ? (size_t)OrdPipeLds<kOrdCapSmall, kOrdPNbSmall, kOrdRingSmallNarrow>::total
                                          : (size_t)OrdPipeLds<kOrdCapSmall, kOrdPNbSmall, kOrdRingSmall>::total;
    const size_t lds = std::max((LW::total + 15) / 16 * 16 * 4, pipe_lds);
    int cus = 256;
    {
      int dev = 0;
      SPL_HIP(hipGetDevice(&dev));
      SPL_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 8) per_cu = 8;
    const int64_t ntiles = (ncolsB + 3) / 4;
    DBuf<int> tile_tasks((size_t)ntiles);
    DBuf<int64_t> task_off((size_t)ntiles + 1);
    hipLaunchKernelGGL(ord_task_count_kernel, dim3(blocks_for(ntiles, 256)), dim3(256), 0, s, ncolsB, cls.get(),
                       tile_tasks.get());
    exclusive_scan_i32_to_i64(tile_tasks.get(), task_off.get(), ntiles, s);
    int64_t ntasks = 0;
    int nnzB_h = 0;
    SPL_HIP(hipMemcpyAsync(&ntasks, task_off.get() + ntiles, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipMemcpyAsync(&nnzB_h, Bp + ncolsB, sizeof(int), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    DBuf<int4> tasks((size_t)ntasks);
    hipLaunchKernelGGL(ord_task_fill_kernel, dim3(blocks_for(ntiles, 256)), dim3(256), 0, s, B, ncolsB, cls.get(),
                       nprod.get(), task_off.get(), tasks.get());
    DBuf<OrdBMeta> bmeta((size_t)nnzB_h);
    if (nnzB_h > 0)
      hipLaunchKernelGGL(ord_bmeta_kernel, dim3(blocks_for(nnzB_h, 256)), dim3(256), 0, s, A, B, (int64_t)nnzB_h, bmeta.get());
    lap("status, tasks, B metadata");
    int64_t grid = (int64_t)cus * per_cu;
    if (grid > ntasks) grid = ntasks;
    DBuf<unsigned long long> stamps;
    if (const char *ev = getenv("SPL_SPGEMM_STAMPS")) {
      if (ev[0] == '1') {
        stamps.alloc(12);
        SPL_HIP(hipMemsetAsync(stamps.get(), 0, 12 * sizeof(unsigned long long), s));
      }
    }
#define SPL_ORD_LAUNCH(CAPV, NBV, RINGV, WGSV)                                                                          \
  hipLaunchKernelGGL((spgemm_ordered_kernel<CAPV, NBV, RINGV, WGSV>), dim3((unsigned)grid), dim3(256), lds, s, A, B, ncolsB,  \
                     sh_wave, sh_group, nprod.get(), cls.get(), heavy_slot.get(), counts.get(), Ti.get(), Tx.get(),            \
                     status.get(), ticket, tasks.get(), ntasks, bmeta.get(), Cp.get(), Ci.get(), Cx.get(), stamps.get())
    if (large_shape) SPL_ORD_LAUNCH(kOrdCapLarge, kOrdPNbLarge, kOrdRingLarge, 3);
    else if (narrow_ring) SPL_ORD_LAUNCH(kOrdCapSmall, kOrdPNbSmall, kOrdRingSmallNarrow, 5);
    else SPL_ORD_LAUNCH(kOrdCapSmall, kOrdPNbSmall, kOrdRingSmall, 4);
#undef SPL_ORD_LAUNCH
    if (stamps.get()) {
      unsigned long long h[12];
      SPL_HIP(hipMemcpy(h, stamps.get(), sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "[spgemm ordered] look-back inside the finish (wide form only): %llu ticks, of them %llu asleep for a column that had "
              "not published; %llu rounds, %llu waits\n", h[8], h[9], h[10], h[11]);
      // (the wavefront path of light columns adds its own phases to slots 0-5: meaningful for one class at a time)
      fprintf(stderr, "[spgemm ordered] s_memtime ticks of thread 0 per phase, summed over %lld tasks: ticket+record %llu | "
              "stage+scan %llu | expand %llu | histogram scan %llu | scatter+bucket sort %llu | count+publish %llu | "
              "finish of the pending column %llu | rest %llu\n",
              (long long)ntasks, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    lap("ordered kernel");
    hipLaunchKernelGGL(ord_total_kernel, dim3(1), dim3(64), 0, s, status.get(), ncolsB, Cp.get());
    int64_t nz = 0;
    SPL_HIP(hipMemcpyAsync(&nz, Cp.get() + ncolsB, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    SPL_HIP(hipGetLastError());
    *nnzC = nz;
    if ((double)nz < 0.75 * (double)capacity) {  // many products merged: give the surplus back
      DBuf<int> Ci2((size_t)nz);
      DBuf<double> Cx2((size_t)nz);
      if (nz > 0) {
        SPL_HIP(hipMemcpyAsync(Ci2.get(), Ci.get(), (size_t)nz * sizeof(int), hipMemcpyDeviceToDevice, s));
        SPL_HIP(hipMemcpyAsync(Cx2.get(), Cx.get(), (size_t)nz * sizeof(double), hipMemcpyDeviceToDevice, s));
        SPL_HIP(hipStreamSynchronize(s));
      }
      Ci = std::move(Ci2);
      Cx = std::move(Cx2);
    }
    return;
  }
  if (single_pass) {
    exclusive_scan_i32_to_i64(counts.get(), Cp.get(), ncolsB, s);
    int64_t nz = 0;
    SPL_HIP(hipMemcpyAsync(&nz, Cp.get() + ncolsB, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    *nnzC = nz;
    if (nz == total_products) {  // no column compressed: the slots are the final layout
      Ci = std::move(Ti);
      Cx = std::move(Tx);
    } else {
      Ci.alloc((size_t)nz);
      Cx.alloc((size_t)nz);
      if (nz > 0)
        hipLaunchKernelGGL(compact_columns_kernel, dim3(blocks_for(ncolsB, 4)), dim3(256), 0, s, ncolsB, pscan.get(),
                           Cp.get(), Ti.get(), Tx.get(), Ci.get(), Cx.get());
    }
  }
  SPL_HIP(hipGetLastError());
  SPL_HIP(hipStreamSynchronize(s));
}

}  // namespace spl

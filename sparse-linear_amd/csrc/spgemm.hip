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
#include "common.hpp"

namespace spl {

namespace {

constexpr int kSmallProducts = 256;
constexpr int kMediumProducts = 2048, kMediumB = 256;
constexpr int kLargeProducts = 4096, kLargeB = 2048;
constexpr int kMaxPool = 512;

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
  if (np <= kSmallProducts) return 1;
  if (np <= kMediumProducts && nb <= kMediumB) return 2;
  if (np <= kLargeProducts && nb <= kLargeB) return 3;
  return 4;
}

__global__ __launch_bounds__(256) void products_kernel(Csc A, Csc B, int64_t ncolsB,
                                                       int64_t *__restrict__ nprod,
                                                       int64_t *__restrict__ medium_list,
                                                       int64_t *__restrict__ xlarge_list,
                                                       int64_t *__restrict__ dense_list,
                                                       int *__restrict__ list_counts) {
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
  __shared__ int local_count[3], base[3];
  if (threadIdx.x < 3) local_count[threadIdx.x] = 0;
  __syncthreads();
  const bool owner = g < ncolsB && sub == 0;
  int bin = 0, pos = 0;
  if (owner) {
    nprod[j] = n;
    bin = bin_of(n, qe - qs);
    if (bin >= 2) pos = atomicAdd(&local_count[bin - 2], 1);
  }
  __syncthreads();
  if (threadIdx.x < 3 && local_count[threadIdx.x] > 0)
    base[threadIdx.x] = atomicAdd(&list_counts[threadIdx.x], local_count[threadIdx.x]);
  __syncthreads();
  if (owner && bin >= 2) {
    int64_t *list = bin == 2 ? medium_list : bin == 3 ? xlarge_list : dense_list;
    list[base[bin - 2] + pos] = j;
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
                                                           int *__restrict__ Ci, double *__restrict__ Cx) {
  __shared__ int wave_counts[4];
  __shared__ int64_t running;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char *flags = pool_flags + (size_t)blockIdx.x * (size_t)nrowsA;
  double *w = NUMERIC ? pool_vals + (size_t)blockIdx.x * (size_t)nrowsA : nullptr;
  for (int li = blockIdx.x; li < nlist; li += gridDim.x) {
    const int64_t j = list[li];
    if (threadIdx.x == 0) running = 0;
    const int qs = B.p[j], qe = B.p[j + 1];
    for (int q = qs; q < qe; ++q) {
      const int k = B.i[q];
      const double b = NUMERIC ? B.x[q] : 0.0;
      const int ps = A.p[k], pe = A.p[k + 1];
      for (int p = ps + (int)threadIdx.x; p < pe; p += 256) {
        const int r = A.i[p];
        flags[r] = 1;
        if (NUMERIC) w[r] = w[r] + A.x[p] * b;
      }
      __syncthreads();
    }
    // gather in row order (ScatterGather.hs:97-147), clearing the accumulator as we go
    const int64_t base = NUMERIC ? Cp[j] : 0;
    for (int64_t r0 = 0; r0 < nrowsA; r0 += 256) {
      const int64_t r = r0 + threadIdx.x;
      const bool used = r < nrowsA && flags[r] != 0;
      const unsigned long long m = __ballot(used);
      if (lane == 0) wave_counts[wave] = __popcll(m);
      __syncthreads();
      int64_t off = running;
      for (int ww = 0; ww < wave; ++ww) off += wave_counts[ww];
      if (used) {
        if (NUMERIC) {
          off += __popcll(m & ((1ull << lane) - 1ull));
          Ci[base + off] = (int)r;
          Cx[base + off] = w[r];
          w[r] = 0.0;
        }
        flags[r] = 0;
      }
      __syncthreads();
      if (threadIdx.x == 0) running += wave_counts[0] + wave_counts[1] + wave_counts[2] + wave_counts[3];
      __syncthreads();
    }
    if (counts && threadIdx.x == 0) counts[j] = (int)running;
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

}  // namespace

// C = A B; all pointers are device pointers.  Cp is 64-bit (nnz(C) may exceed 2^31).
void spgemm_device(int64_t nrowsA, int64_t ncolsA, const int *Ap, const int *Ai, const double *Ax,
                   int64_t ncolsB, const int *Bp, const int *Bi, const double *Bx, DBuf<int64_t> &Cp,
                   DBuf<int> &Ci, DBuf<double> &Cx, int64_t *nnzC, int64_t *products, hipStream_t s) {
  (void)ncolsA;
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
  DBuf<int> list_counts(3), counts((size_t)ncolsB);
  SPL_HIP(hipMemsetAsync(list_counts.get(), 0, 3 * sizeof(int), s));
  hipLaunchKernelGGL(products_kernel, dim3(blocks_for(ncolsB, 32)), dim3(256), 0, s, A, B, ncolsB, nprod.get(),
                     medium_list.get(), xlarge_list.get(), dense_list.get(), list_counts.get());
  int hc[3] = {0, 0, 0};
  SPL_HIP(hipMemcpyAsync(hc, list_counts.get(), 3 * sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  const int nmedium = hc[0], nxlarge = hc[1], ndense = hc[2];
  DBuf<int64_t> pscan((size_t)ncolsB + 1);  // products before column j: its upper-bound output slot
  exclusive_scan_i64(nprod.get(), pscan.get(), ncolsB, s);
  int64_t total_products = 0;
  SPL_HIP(hipMemcpy(&total_products, pscan.get() + ncolsB, sizeof(int64_t), hipMemcpyDeviceToHost));
  if (products) *products = total_products;
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
  typedef EscLds<kMediumProducts, kMediumB, false> LMs;
  typedef EscLds<kLargeProducts, kLargeB, false> LXs;
  typedef EscLds<kLargeProducts, kLargeB, true, false> LXn;
  typedef EscLds<kLargeProducts, kLargeB, true, true> LXn32;
  // 32-bit packed sort keys need row + tie-break bits to fit 31 bits in every bin
  // SPL_SPGEMM_SPLIT_KEYS=1 (tests) forces the row + 16-bit-position keys that large matrices need
  const char *split_env = getenv("SPL_SPGEMM_SPLIT_KEYS");
  const bool allow32 = !(split_env && split_env[0] == '1');
  const bool key32_s = allow32 && nrowsA <= (1LL << (31 - ilog2_ceil(kSmallProducts)));
  const bool key32_m = allow32 && nrowsA <= (1LL << (31 - ilog2_ceil(kMediumProducts)));
  const bool key32_x = allow32 && nrowsA <= (1LL << (31 - ilog2_ceil(kLargeProducts)));
  static bool attr_set = false;
  if (!attr_set) {
    SPL_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void *>(&spgemm_block_kernel<kLargeProducts, kLargeB, true, false>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LXn::total));
    SPL_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void *>(&spgemm_block_kernel<kLargeProducts, kLargeB, true, true>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)LXn32::total));
    attr_set = true;
  }
  int pool = ndense < kMaxPool ? ndense : kMaxPool;
  {  // keep the accumulator pool under ~8 GB
    const int64_t cap = (int64_t)8e9 / (9 * (nrowsA > 0 ? nrowsA : 1));
    if (pool > cap) pool = (int)(cap < 1 ? 1 : cap);
  }
  DBuf<unsigned char> pool_flags;
  DBuf<double> pool_vals;
  if (ndense > 0) {
    pool_flags.alloc((size_t)pool * (size_t)nrowsA);
    pool_vals.alloc((size_t)pool * (size_t)nrowsA);
    SPL_HIP(hipMemsetAsync(pool_flags.get(), 0, (size_t)pool * (size_t)nrowsA, s));
    SPL_HIP(hipMemsetAsync(pool_vals.get(), 0, (size_t)pool * (size_t)nrowsA * sizeof(double), s));
  }

  DBuf<int> Ti;
  DBuf<double> Tx;
  const int64_t *slots = Cp.get();  // where the numeric kernels write column j
  int *out_i = nullptr;
  double *out_x = nullptr;
  int *numeric_counts = nullptr;
  if (single_pass) {
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
    if (nmedium > 0)
      hipLaunchKernelGGL((spgemm_block_kernel<kMediumProducts, kMediumB, false, false>), dim3((unsigned)nmedium),
                         dim3(256), LMs::total, s, A, B, medium_list.get(), nprod.get(), counts.get(),
                         (const int64_t *)nullptr, (int *)nullptr, (double *)nullptr);
    if (nxlarge > 0)
      hipLaunchKernelGGL((spgemm_block_kernel<kLargeProducts, kLargeB, false, false>), dim3((unsigned)nxlarge),
                         dim3(256), LXs::total, s, A, B, xlarge_list.get(), nprod.get(), counts.get(),
                         (const int64_t *)nullptr, (int *)nullptr, (double *)nullptr);
    if (ndense > 0)
      hipLaunchKernelGGL(spgemm_dense_kernel<false>, dim3((unsigned)pool), dim3(256), 0, s, A, B, nrowsA,
                         dense_list.get(), ndense, pool_flags.get(), (double *)nullptr, counts.get(),
                         (const int64_t *)nullptr, (int *)nullptr, (double *)nullptr);
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
  }

  // ---- numeric: every path writes its column already sorted by row
#define SPL_NUMERIC_WAVE(K32)                                                                                  \
  hipLaunchKernelGGL((spgemm_wave_kernel<true, K32>), dim3(blocks_for(ncolsB, 4)), dim3(256), 0, s, A, B, ncolsB, \
                     nprod.get(), numeric_counts, slots, out_i, out_x)
#define SPL_NUMERIC_BLOCK(CAP, NBCAP, K32, LIST, COUNT)                                                        \
  hipLaunchKernelGGL((spgemm_block_kernel<CAP, NBCAP, true, K32>), dim3((unsigned)(COUNT)), dim3(256),         \
                     (EscLds<CAP, NBCAP, true, K32>::total), s, A, B, LIST.get(), nprod.get(), numeric_counts,   \
                     slots, out_i, out_x)
  if (key32_s) SPL_NUMERIC_WAVE(true); else SPL_NUMERIC_WAVE(false);
  if (nmedium > 0) {
    if (key32_m) SPL_NUMERIC_BLOCK(kMediumProducts, kMediumB, true, medium_list, nmedium);
    else SPL_NUMERIC_BLOCK(kMediumProducts, kMediumB, false, medium_list, nmedium);
  }
  if (nxlarge > 0) {
    if (key32_x) SPL_NUMERIC_BLOCK(kLargeProducts, kLargeB, true, xlarge_list, nxlarge);
    else SPL_NUMERIC_BLOCK(kLargeProducts, kLargeB, false, xlarge_list, nxlarge);
  }
#undef SPL_NUMERIC_WAVE
#undef SPL_NUMERIC_BLOCK
  if (ndense > 0)
    hipLaunchKernelGGL(spgemm_dense_kernel<true>, dim3((unsigned)pool), dim3(256), 0, s, A, B, nrowsA,
                       dense_list.get(), ndense, pool_flags.get(), pool_vals.get(), numeric_counts, slots, out_i,
                       out_x);
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

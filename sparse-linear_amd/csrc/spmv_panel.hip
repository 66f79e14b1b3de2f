// spmv_panel.hip — workgroup-wide, column-sorted panels: the order-free (1e-10) SpMV mode.
//
// Why: on a matrix without column locality (config C2) the column-blocked lockstep kernel
// (spmv_blocked.hip) is bound by the NUMBER of requests its x gathers send from the CU's vector
// L1 to the L2: one 128-byte line per 8-byte gather, 2.0e8 of them per product, whatever the hit
// rate (profiles/r01_spmv_random_blocked_pmc_detail.txt, profiles/r01_l1_gather_probe.txt).
// Lanes of one load instruction that fall into the same line share one request, so the only way
// to send fewer is to put entries with neighbouring columns next to each other.  Inside one
// wavefront's 1 221-row panel a 128-byte line of x (16 columns) meets 0.04 entries; inside a
// panel that fills the whole LDS (19 532 rows, one per workgroup) it meets 0.63, and sorting the
// panel's entries of a column block BY COLUMN makes 64 neighbouring entries span ~100 lines and
// touch ~47 of them: a quarter fewer requests.
//
// Price: the entries of one row no longer arrive in ascending column order at one wavefront — the
// 16 wavefronts of the workgroup each take every 16th chunk of the column-sorted stream and add
// into the panel's y in LDS with ds_add_f64, so the order in which a row's products are summed is
// the order the hardware happens to execute them in.  Every product a*x and every add is still
// separately rounded; only the ORDER of the adds differs from the reference (Sparse.hs:447-451),
// and may differ from run to run.  north_star's contract is 1e-10 relative on values; this mode
// meets it with rounding-level differences (tests/test_gpu_spmv_panel.py), the column-blocked
// kernel stays available as the reference-order (bit-identical) mode.
//
// Image, built once per matrix in HBM:
//   * rows cut into panels of P rows (P*8 bytes of y = one CU's LDS), columns into index blocks
//     of 2^w columns, w <= 17;
//   * segment (panel, index block): its entries sorted by (column, row) as the packed 32-bit key
//     (local_col << 15 | local_row) + the fp64 value — 12 bytes per entry as before — padded to
//     a multiple of 64 entries with (column 0, row P, value 0): row P is a dummy slot of the LDS
//     image that is never written back, so a whole 64-entry chunk never straddles two blocks and
//     no lane needs a validity test;
//   * segc[panel*nib + ib] = first CHUNK (64 entries) of the segment.
// Kernel: one 16-wavefront workgroup per CU, panels in generations like the lockstep kernel;
// a phase covers K consecutive index blocks (K * 2^w * 8 bytes of x: the window the CUs of an XCD
// gather from together), wavefront i takes chunks i, i+16, ... of the phase, keeps U of them in
// registers, and the stream of the next phase is requested before the barrier that ends this one.
#include "common.hpp"
#include <atomic>

namespace spl {

namespace {

constexpr int kRowBits = 15;
constexpr unsigned kRowMask = (1u << kRowBits) - 1u;
constexpr int kPanelWaves = 16;
constexpr int kPanelSlackChunks = 16 * 14 + 16;  // stream loads of a register set may run past the end

inline unsigned blocks_for(int64_t n, int per_block) {
  int64_t b = (n + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : b);
}

// ---- image construction ----------------------------------------------------------------------
template <typename PtrT>
__global__ __launch_bounds__(256) void pnl_count_kernel(int64_t nrows, const PtrT *__restrict__ rowptr,
                                                        const int *__restrict__ colidx, int P, int w,
                                                        int64_t nib, int *__restrict__ segcount) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int64_t base = (r / P) * nib;
  for (PtrT k = rowptr[r]; k < rowptr[r + 1]; ++k) atomicAdd(&segcount[base + (colidx[k] >> w)], 1);
}

__global__ __launch_bounds__(256) void pnl_chunks_kernel(int64_t nseg, const int *__restrict__ segcount,
                                                         int *__restrict__ chunks, int pair) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nseg) {
    const int c = (segcount[i] + 63) >> 6;
    chunks[i] = pair ? ((c + 1) & ~1) : c;  // paired storage: whole pairs of chunks
  }
}

// paired storage: inside every 128-entry pair of chunks (A, B) the entries are stored A0 B0 A1 B1 ...,
// so that one 16-byte load per lane brings lane l the values (A_l, B_l) and one 8-byte load the keys
__global__ __launch_bounds__(128) void pnl_interleave_kernel(int64_t npairs, unsigned *__restrict__ key,
                                                             double *__restrict__ val) {
  const int64_t pr = blockIdx.x;
  if (pr >= npairs) return;
  const int t = threadIdx.x;
  const int64_t base = pr << 7;
  const unsigned k = key[base + t];
  const double v = val[base + t];
  __syncthreads();
  const int dst = ((t & 63) << 1) | (t >> 6);
  key[base + dst] = k;
  val[base + dst] = v;
}

__global__ __launch_bounds__(256) void pnl_entryptr_kernel(int64_t n, const int *__restrict__ segc,
                                                           int64_t *__restrict__ ptr64) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ptr64[i] = (int64_t)segc[i] << 6;
}

__global__ __launch_bounds__(256) void pnl_padseg_kernel(int64_t n, int last, int *__restrict__ segc, int64_t from) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) segc[from + i] = last;
}

template <typename PtrT>
__global__ __launch_bounds__(256) void pnl_fill_kernel(int64_t nrows, const PtrT *__restrict__ rowptr,
                                                       const int *__restrict__ colidx,
                                                       const double *__restrict__ val, int P, int w,
                                                       int64_t nib, const int *__restrict__ segc,
                                                       int *__restrict__ cursor, unsigned *__restrict__ key,
                                                       double *__restrict__ pval) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int64_t base = (r / P) * nib;
  const unsigned lr = (unsigned)(r % P);
  const int wmask = (1 << w) - 1;
  for (PtrT k = rowptr[r]; k < rowptr[r + 1]; ++k) {
    const int c = colidx[k];
    const int64_t seg = base + (c >> w);
    const int64_t pos = ((int64_t)segc[seg] << 6) + atomicAdd(&cursor[seg], 1);
    key[pos] = ((unsigned)(c & wmask) << kRowBits) | lr;
    pval[pos] = val[k];
  }
}

// ---- the kernel ---------------------------------------------------------------------------------
__device__ inline double pnl_gather_issue(const double *p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int N>
__device__ inline void pnl_gather_wait(double &v) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}
__device__ inline void pnl_fold(unsigned id, double prod, double *yp) {
  __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)(yp + (id & kRowMask)), prod);
}

// One phase: gather + fold this wavefront's chunks of the phase [cs, ce) held in (idC, aC) — chunk
// u of wavefront i is chunk cs + i + 16 u of the panel's stream — while its chunks of the next
// phase (which starts at ce) are loaded into (idN, aN).  mid[j] is the first chunk of index block
// ib0 + j + 1 (K - 1 of them): the x block a chunk gathers from follows from its position.
// ABL (timing-only ablations, wrong results; refused unless SPL_ALLOW_ABLATION=1): bit 0 every gather reads
// x[lane] (no L2 requests beyond one line), bit 1 no value loads (a = 1), bit 2 no LDS fold
template <int U, int K, int ABL = 0>
__device__ inline void panel_phase(unsigned (&idC)[U], double (&aC)[U], unsigned (&idN)[U], double (&aN)[U],
                                   int cs, const int (&mid)[K > 1 ? K - 1 : 1], int ce, int64_t ib0, int w,
                                   const unsigned *__restrict__ key, const double *__restrict__ val,
                                   const double *__restrict__ x, double *yp, int wave, int nextlen, double &sink,
                                   int64_t dummy) {
  const int lane = threadIdx.x & 63;
  double xv[U];
  const double *xp[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int c = cs + wave + kPanelWaves * u;
    int64_t ib = ib0;
#pragma unroll
    for (int j = 0; j + 1 < K; ++j) ib += (c >= mid[j]) ? 1 : 0;
    const bool ok = c < ce;  // wave-uniform
    xp[u] = x + (ok ? ((ib << w) + (int64_t)(idC[u] >> kRowBits)) : 0);
    if (ABL & 1) xp[u] = x + lane;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u) xv[u] = pnl_gather_issue(xp[u]);  // gathers first ...
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u) {  // ... then the next phase's stream, left in flight across the barrier
    // a chunk past the next phase's end would be fetched again by the phase it belongs to: read the
    // (L2-resident) dummy chunk behind the stream instead
    const int c = ce + wave + kPanelWaves * u;
    const int64_t e = (c < ce + nextlen ? ((int64_t)c << 6) : dummy) + lane;  // wave-uniform select
    idN[u] = __builtin_nontemporal_load(key + e);
    if (ABL & 2) aN[u] = 1.0;
    else aN[u] = __builtin_nontemporal_load(val + e);
  }
  __builtin_amdgcn_sched_barrier(0);
  constexpr int Y = ((ABL & 2) ? 2 : 3) * U - 1;  // younger than gather u here: U-1-u gathers + 2U stream loads
  pnl_gather_wait<Y>(xv[0]);
  if (U > 1) pnl_gather_wait<Y - 1>(xv[U > 1 ? 1 : 0]);
  if (U > 2) pnl_gather_wait<Y - 2>(xv[U > 2 ? 2 : 0]);
  if (U > 3) pnl_gather_wait<Y - 3>(xv[U > 3 ? 3 : 0]);
  if (U > 4) pnl_gather_wait<Y - 4>(xv[U > 4 ? 4 : 0]);
  if (U > 5) pnl_gather_wait<Y - 5>(xv[U > 5 ? 5 : 0]);
  if (U > 6) pnl_gather_wait<Y - 6>(xv[U > 6 ? 6 : 0]);
  if (U > 7) pnl_gather_wait<Y - 7>(xv[U > 7 ? 7 : 0]);
  if (U > 8) pnl_gather_wait<Y - 8>(xv[U > 8 ? 8 : 0]);
  if (U > 9) pnl_gather_wait<Y - 9>(xv[U > 9 ? 9 : 0]);
  if (U > 10) pnl_gather_wait<Y - 10>(xv[U > 10 ? 10 : 0]);
  if (U > 11) pnl_gather_wait<Y - 11>(xv[U > 11 ? 11 : 0]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (cs + wave + kPanelWaves * u >= ce) break;  // wave-uniform
    if (ABL & 4) sink += aC[u] * xv[u] + (double)(idC[u] & 1u);
    else pnl_fold(idC[u], aC[u] * xv[u], yp);
  }
  for (int c = cs + wave + kPanelWaves * U; c < ce; c += kPanelWaves) {  // tail of an over-long phase
    int64_t ib = ib0;
#pragma unroll
    for (int j = 0; j + 1 < K; ++j) ib += (c >= mid[j]) ? 1 : 0;
    const unsigned id = __builtin_nontemporal_load(key + ((int64_t)c << 6) + lane);
    const double a = __builtin_nontemporal_load(val + ((int64_t)c << 6) + lane);
    pnl_fold(id, a * x[(ib << w) + (int64_t)(id >> kRowBits)], yp);
  }
  __builtin_amdgcn_s_barrier();  // pacing only: no fence, vector memory stays in flight
}

template <int U, int K, int ABL = 0>
__global__ __launch_bounds__(kPanelWaves * 64) void spmv_panel_kernel(
    int64_t nrows, int64_t npanels, int P, int w, int64_t nib, const int *__restrict__ segc,
    const unsigned *__restrict__ key, const double *__restrict__ val, const double *__restrict__ x,
    double *__restrict__ y, int accumulate, unsigned *__restrict__ arrive, int64_t dummy) {
  extern __shared__ __attribute__((aligned(16))) double ylds[];  // P + 1 doubles
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nb = gridDim.x;
  const int64_t ngen = (npanels + nb - 1) / nb;
  const int64_t nph = (nib + K - 1) / K;
  for (int64_t g = 0; g < ngen; ++g) {
    const int64_t p = g * nb + blockIdx.x;
    if (p >= npanels) break;  // only in the last generation: no rendezvous follows
    const int64_t row_base = p * P;
    const int *sp = segc + p * nib;  // segc carries K + 1 trailing copies of its last entry
    const int c0 = sp[0];
    unsigned idA[U], idB[U];
    double aA[U], aB[U];
    double sink = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {  // prologue: this wavefront's first chunks of phase 0
      const int c = c0 + wave + kPanelWaves * u;
      const int64_t k = (c < (K < nib ? sp[K] : sp[nib]) ? ((int64_t)c << 6) : dummy) + lane;
      idA[u] = __builtin_nontemporal_load(key + k);
      aA[u] = __builtin_nontemporal_load(val + k);
    }
    for (int i = threadIdx.x; i <= P; i += kPanelWaves * 64)
      ylds[i] = (accumulate && i < P && row_base + i < nrows) ? y[row_base + i] : 0.0;
    __syncthreads();
    const int cend = sp[nib];
    for (int64_t ph = 0; ph < nph; ph += 2) {
      {
        const int64_t ib0 = ph * K;
        int mid[K > 1 ? K - 1 : 1];
#pragma unroll
        for (int j = 0; j + 1 < K; ++j) { const int t = sp[ib0 + j + 1]; mid[j] = t < cend ? t : cend; }
        const int cs = sp[ib0];
        int ce = sp[ib0 + K]; ce = (ib0 + K < nib) ? ce : cend;
        int cn = sp[ib0 + 2 * K]; cn = (ib0 + 2 * K < nib) ? cn : cend;
        panel_phase<U, K, ABL>(idA, aA, idB, aB, cs, mid, ce, ib0, w, key, val, x, ylds, wave, cn - ce, sink, dummy);
      }
      if (ph + 1 < nph) {
        const int64_t ib0 = (ph + 1) * K;
        int mid[K > 1 ? K - 1 : 1];
#pragma unroll
        for (int j = 0; j + 1 < K; ++j) { const int t = sp[ib0 + j + 1]; mid[j] = t < cend ? t : cend; }
        const int cs = sp[ib0];
        int ce = sp[ib0 + K]; ce = (ib0 + K < nib) ? ce : cend;
        int cn = sp[ib0 + 2 * K]; cn = (ib0 + 2 * K < nib) ? cn : cend;
        panel_phase<U, K, ABL>(idB, aB, idA, aA, cs, mid, ce, ib0, w, key, val, x, ylds, wave, cn - ce, sink, dummy);
      }
    }
    if (ABL && sink == 1.2345e-300) ylds[0] = sink;  // keeps the ablated arithmetic alive
    __syncthreads();  // every wavefront's LDS adds are done (s_barrier above does not wait for lgkmcnt)
    for (int i = threadIdx.x; i < P; i += kPanelWaves * 64)
      if (row_base + i < nrows) y[row_base + i] = ylds[i];
    if (g + 1 < ngen) {  // re-align the CUs between generations (bounded, performance only)
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)((g + 1) * nb);
        const unsigned long long t0 = wall_clock64();  // 100 MHz
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (wall_clock64() - t0 > 20000ull) break;  // 200 us: give up, stay correct
          __builtin_amdgcn_s_sleep(8);
        }
      }
      __syncthreads();
    }
  }
}


// ---- paired form ------------------------------------------------------------------------------------
// Same two-stage schedule on the paired storage: a wavefront's unit of work is a PAIR of chunks, read
// with one 8-byte key load and one 16-byte value load per lane (half the stream instructions for the
// same bytes: the HBM stream of a CU runs closer to its peak with fewer, wider requests in flight —
// tools/probe/tcp_mix_probe.hip S rows), gathered with two instructions (chunk A, chunk B: each a run
// of 64 column-sorted entries as before) and folded with two.  All units here are pairs.
typedef unsigned pnl_u2 __attribute__((ext_vector_type(2)));
typedef double pnl_d2 __attribute__((ext_vector_type(2)));

template <int U, int K>
__device__ inline void panelw_phase(pnl_u2 (&idC)[U], pnl_d2 (&aC)[U], pnl_u2 (&idN)[U], pnl_d2 (&aN)[U], int cs,
                                    const int (&mid)[K > 1 ? K - 1 : 1], int ce, int64_t ib0, int w,
                                    const pnl_u2 *__restrict__ key2, const pnl_d2 *__restrict__ val2,
                                    const double *__restrict__ x, double *yp, int wave, int cn, int64_t dummy) {
  const int lane = threadIdx.x & 63;
  double xa[U], xb[U];
  const double *pa[U], *pb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int c = cs + wave + kPanelWaves * u;
    int64_t ib = ib0;
#pragma unroll
    for (int j = 0; j + 1 < K; ++j) ib += (c >= mid[j]) ? 1 : 0;
    const bool ok = c < ce;  // wave-uniform
    pa[u] = x + (ok ? ((ib << w) + (int64_t)(idC[u].x >> kRowBits)) : 0);
    pb[u] = x + (ok ? ((ib << w) + (int64_t)(idC[u].y >> kRowBits)) : 0);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u) {  // gathers first ...
    xa[u] = pnl_gather_issue(pa[u]);
    xb[u] = pnl_gather_issue(pb[u]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u) {  // ... then the next phase's stream, left in flight across the barrier
    // a unit past the next phase's end [ce, cn) would be fetched again by the phase it belongs to: read
    // the (L2-resident) dummy unit instead, so that U may exceed the mean count without costing HBM bytes
    const int c = ce + wave + kPanelWaves * u;
    const int64_t e = (c < cn ? ((int64_t)c << 6) : dummy) + lane;  // wave-uniform select
    idN[u] = __builtin_nontemporal_load(key2 + e);
    aN[u] = __builtin_nontemporal_load(val2 + e);
  }
  __builtin_amdgcn_sched_barrier(0);
  // younger than gather j (j = 2u for A, 2u + 1 for B) here: 2U-1-j gathers + 2U stream loads
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (u == 0) { pnl_gather_wait<4 * U - 1>(xa[u]); pnl_gather_wait<4 * U - 2>(xb[u]); }
    if (u == 1) { pnl_gather_wait<4 * U - 3>(xa[u]); pnl_gather_wait<4 * U - 4>(xb[u]); }
    if (u == 2) { pnl_gather_wait<(4 * U - 5 > 0 ? 4 * U - 5 : 0)>(xa[u]); pnl_gather_wait<(4 * U - 6 > 0 ? 4 * U - 6 : 0)>(xb[u]); }
    if (u == 3) { pnl_gather_wait<(4 * U - 7 > 0 ? 4 * U - 7 : 0)>(xa[u]); pnl_gather_wait<(4 * U - 8 > 0 ? 4 * U - 8 : 0)>(xb[u]); }
    if (u == 4) { pnl_gather_wait<(4 * U - 9 > 0 ? 4 * U - 9 : 0)>(xa[u]); pnl_gather_wait<(4 * U - 10 > 0 ? 4 * U - 10 : 0)>(xb[u]); }
    if (u == 5) { pnl_gather_wait<(4 * U - 11 > 0 ? 4 * U - 11 : 0)>(xa[u]); pnl_gather_wait<(4 * U - 12 > 0 ? 4 * U - 12 : 0)>(xb[u]); }
    if (u == 6) { pnl_gather_wait<(4 * U - 13 > 0 ? 4 * U - 13 : 0)>(xa[u]); pnl_gather_wait<(4 * U - 14 > 0 ? 4 * U - 14 : 0)>(xb[u]); }
    if (u == 7) { pnl_gather_wait<(4 * U - 15 > 0 ? 4 * U - 15 : 0)>(xa[u]); pnl_gather_wait<(4 * U - 16 > 0 ? 4 * U - 16 : 0)>(xb[u]); }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (cs + wave + kPanelWaves * u >= ce) break;  // wave-uniform
    pnl_fold(idC[u].x, aC[u].x * xa[u], yp);
    pnl_fold(idC[u].y, aC[u].y * xb[u], yp);
  }
  for (int c = cs + wave + kPanelWaves * U; c < ce; c += kPanelWaves) {  // tail of an over-long phase
    int64_t ib = ib0;
#pragma unroll
    for (int j = 0; j + 1 < K; ++j) ib += (c >= mid[j]) ? 1 : 0;
    const pnl_u2 id = __builtin_nontemporal_load(key2 + ((int64_t)c << 6) + lane);
    const pnl_d2 a = __builtin_nontemporal_load(val2 + ((int64_t)c << 6) + lane);
    pnl_fold(id.x, a.x * x[(ib << w) + (int64_t)(id.x >> kRowBits)], yp);
    pnl_fold(id.y, a.y * x[(ib << w) + (int64_t)(id.y >> kRowBits)], yp);
  }
  __builtin_amdgcn_s_barrier();  // pacing only
}

// Column slices (round 3): with `ns` > 1 the index blocks are dealt to ns slices and workgroup b works on slice
// b % ns (workgroups go round-robin to the 8 XCDs, so with ns = 8 a slice is what ONE XCD's L2 has to hold) of
// panel slot b / ns: a panel's row sums are then completed by ns workgroups, which add their parts into y with
// global_atomic_add_f64 (y zeroed by the launcher unless accumulating).  Why: every XCD that works on a panel
// pulls the x lines that panel touches through its own L2 once per generation, so x costs (rows / (panels per
// generation of one XCD * P)) * 8 bytes * ncols of fabric traffic per product — a row block too short to give
// every CU a full-height panel (a rank's block at N = 4, 8) would otherwise pay with short panels (fewer lanes
// per line of x, more passes over x): 640 MB of x for 300 MB of matrix at N = 8.  With slices the panels keep
// the full LDS height and each XCD reads an eighth of x per generation.
template <int U, int K>
__global__ __launch_bounds__(kPanelWaves * 64) void spmv_panelw_kernel(
    int64_t nrows, int64_t npanels, int P, int w, int64_t nib, const int *__restrict__ segc,
    const unsigned *__restrict__ key, const double *__restrict__ val, const double *__restrict__ x,
    double *__restrict__ y, int accumulate, unsigned *__restrict__ arrive, int64_t dummy, int ns) {
  extern __shared__ __attribute__((aligned(16))) double ylds[];  // P + 1 doubles
  const pnl_u2 *key2 = reinterpret_cast<const pnl_u2 *>(key);
  const pnl_d2 *val2 = reinterpret_cast<const pnl_d2 *>(val);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ppg = gridDim.x / ns;          // panels per generation
  const int slice = (int)(blockIdx.x % ns);
  const int64_t slot = blockIdx.x / ns;
  if (slot >= ppg) return;                      // grid not a multiple of ns: the extra workgroups idle
  const int64_t nb = ppg * ns;                  // workgroups that take part in the rendezvous
  const int64_t ngen = (npanels + ppg - 1) / ppg;
  const int64_t ibs = nib * slice / ns, ibe = nib * (slice + 1) / ns;  // this slice's index blocks
  const int64_t nph = (ibe - ibs + K - 1) / K;
  for (int64_t g = 0; g < ngen; ++g) {
    const int64_t p = g * ppg + slot;
    if (p >= npanels) break;
    const int64_t row_base = p * P;
    const int *sp = segc + p * nib;  // in chunks; every boundary is even (whole pairs)
    const int c0 = sp[ibs] >> 1;
    const int cend = sp[ibe] >> 1;
    const int c1 = (ibs + K < ibe ? sp[ibs + K] >> 1 : cend);
    pnl_u2 idA[U], idB[U];
    pnl_d2 aA[U], aB[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + wave + kPanelWaves * u;
      const int64_t k = (c < c1 ? ((int64_t)c << 6) : dummy) + lane;
      idA[u] = __builtin_nontemporal_load(key2 + k);
      aA[u] = __builtin_nontemporal_load(val2 + k);
    }
    for (int i = threadIdx.x; i <= P; i += kPanelWaves * 64)
      ylds[i] = (ns == 1 && accumulate && i < P && row_base + i < nrows) ? y[row_base + i] : 0.0;
    __syncthreads();
    for (int64_t ph = 0; ph < nph; ph += 2) {
      {
        const int64_t ib0 = ibs + ph * K;
        int mid[K > 1 ? K - 1 : 1];
#pragma unroll
        for (int j = 0; j + 1 < K; ++j) { const int t = (ib0 + j + 1 < ibe) ? sp[ib0 + j + 1] >> 1 : cend; mid[j] = t < cend ? t : cend; }
        const int cs = sp[ib0] >> 1;
        const int ce = (ib0 + K < ibe) ? sp[ib0 + K] >> 1 : cend;
        const int cn = (ib0 + 2 * K < ibe) ? sp[ib0 + 2 * K] >> 1 : cend;
        panelw_phase<U, K>(idA, aA, idB, aB, cs, mid, ce, ib0, w, key2, val2, x, ylds, wave, cn, dummy);
      }
      if (ph + 1 < nph) {
        const int64_t ib0 = ibs + (ph + 1) * K;
        int mid[K > 1 ? K - 1 : 1];
#pragma unroll
        for (int j = 0; j + 1 < K; ++j) { const int t = (ib0 + j + 1 < ibe) ? sp[ib0 + j + 1] >> 1 : cend; mid[j] = t < cend ? t : cend; }
        const int cs = sp[ib0] >> 1;
        const int ce = (ib0 + K < ibe) ? sp[ib0 + K] >> 1 : cend;
        const int cn = (ib0 + 2 * K < ibe) ? sp[ib0 + 2 * K] >> 1 : cend;
        panelw_phase<U, K>(idB, aB, idA, aA, cs, mid, ce, ib0, w, key2, val2, x, ylds, wave, cn, dummy);
      }
    }
    __syncthreads();
    if (ns == 1) {
      for (int i = threadIdx.x; i < P; i += kPanelWaves * 64)
        if (row_base + i < nrows) y[row_base + i] = ylds[i];
    } else {
      for (int i = threadIdx.x; i < P; i += kPanelWaves * 64)
        if (row_base + i < nrows) unsafeAtomicAdd(y + row_base + i, ylds[i]);  // global_atomic_add_f64, no return
    }
    if (g + 1 < ngen) {
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)((g + 1) * nb);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (wall_clock64() - t0 > 20000ull) break;
          __builtin_amdgcn_s_sleep(8);
        }
      }
      __syncthreads();
    }
  }
}

// ---- ring form (round 3): loader wavefronts + gather wavefronts ---------------------------------------
// What the probes say (tools/probe/lds_dma_mix_probe.hip, profiles/r03_tcp_mix_lds_dma.txt): when every
// wavefront carries both the HBM stream and the L2 gathers, the two times ADD (vmcnt retires in order, so a
// wavefront's gathers wait behind its own stream loads, and the stream arrives in bursts of a whole phase);
// when a few wavefronts do nothing but keep ~32 KiB of 16-byte stream loads in flight and the others do
// nothing but gather, the mix takes 0.82x the sum.  LDS-DMA for the stream does not help (it adds up like
// the register loads do, and more).  So: NL loader wavefronts stream the paired image into registers (D
// units of 1.5 KiB each in flight per loader) and hand every unit through a 1.5 KiB slot in LDS to one of
// its R = (16 - NL) / NL gather wavefronts, which keeps GD units (2 GD gather instructions) in flight and
// folds with ds_add_f64 as before.  The panel keeps (almost) the whole LDS: the hand-over slots are NL * S
// units (6 KiB for NL = 4, S = 1).
//   slot header {seq, ib}: seq = t + 1 while the loader's t-th unit waits in the slot, 0 = free.  LDS
//   operations of one wavefront execute in program order, so data written before the header is visible
//   to whoever sees the header, and a header cleared after the reads have returned frees the slot.
//   Phase barriers (pacing only, as in the other forms): a gather wavefront crosses barrier p before it
//   gathers its first unit beyond phase p (it has taken that unit out of its slot by then); a loader
//   crosses it once each of its gather wavefronts has been handed a unit beyond phase p (its last R
//   deliveries are all beyond p), or at the end of the stream.  Nobody waits at a barrier for something
//   that only a wavefront behind that barrier can provide.  Every wait is bounded (spin limit -> error
//   word, results then wrong but the grid drains).
constexpr int kRingUnitBytes = 1536;
constexpr unsigned kRingSpinLimit = 1u << 22;

// a wave-uniform int through the scalar cache: a vector load here would sit on vmcnt behind the loader's stream
__device__ inline int ring_sload(const int *p) {
  int v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

// the loader's stream loads, outside the compiler's wait bookkeeping (it drains vmcnt to 0 at every loop
// header): issued here, awaited with a counted vmcnt by ring_stream_wait
__device__ inline void ring_stream_issue(pnl_u2 &k, pnl_d2 &v, const pnl_u2 *kp, const pnl_d2 *vp) {
  asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(k) : "v"(kp) : "memory");
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(vp) : "memory");
}
template <int N>
__device__ inline void ring_stream_wait(pnl_u2 &k, pnl_d2 &v) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(k), "+v"(v) : "n"(N) : "memory");
}

__device__ inline void ring_fail(unsigned *err) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NL, int D, int GD, int K, int S>
__global__ __launch_bounds__(kPanelWaves * 64) void spmv_panelr_kernel(
    int64_t nrows, int64_t npanels, int P, int w, int64_t nib, const int *__restrict__ segc,
    const unsigned *__restrict__ key, const double *__restrict__ val, const double *__restrict__ x,
    double *__restrict__ y, int accumulate, unsigned *__restrict__ arrive, int64_t dummy) {
  extern __shared__ __attribute__((aligned(16))) double ylds[];  // P + 1 doubles, then the slots, then their headers
  constexpr int R = (kPanelWaves - NL) / NL;
  static_assert(NL * (R + 1) == kPanelWaves, "NL must divide 16");
  const pnl_u2 *key2 = reinterpret_cast<const pnl_u2 *>(key);
  const pnl_d2 *val2 = reinterpret_cast<const pnl_d2 *>(val);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // explicit LDS pointers: a generic pointer would turn every access into a flat_ instruction (vmcnt + lgkmcnt)
  typedef __attribute__((address_space(3))) char lds_char;
  typedef __attribute__((address_space(3))) pnl_u2 lds_u2;
  typedef __attribute__((address_space(3))) pnl_d2 lds_d2;
  lds_char *ring = (lds_char *)ylds + ((((size_t)P + 1) * sizeof(double) + 15) & ~(size_t)15);
  volatile lds_u2 *hdr = (volatile lds_u2 *)(ring + NL * S * kRingUnitBytes);
  const int64_t nb = gridDim.x;
  const int64_t ngen = (npanels + nb - 1) / nb;
  const int nph = (int)((nib + K - 1) / K);
  for (int64_t g = 0; g < ngen; ++g) {
    const int64_t p = g * nb + blockIdx.x;
    if (p >= npanels) break;
    const int64_t row_base = p * P;
    const int *sp = segc + p * nib;  // in chunks; every boundary is even (whole pairs)
    const int c0 = sp[0] >> 1;
    const int nu = (sp[nib] >> 1) - c0;  // units (pairs of chunks) of this panel
    for (int i = threadIdx.x; i <= P; i += kPanelWaves * 64)
      ylds[i] = (accumulate && i < P && row_base + i < nrows) ? y[row_base + i] : 0.0;
    if (threadIdx.x < NL * S) { pnl_u2 z = {0u, 0u}; hdr[threadIdx.x] = z; }
    __syncthreads();
    int bar_done = 0;
    if (wave < NL) {
      // ---- loader ----
      const int L = wave;
      const int nt = nu > L ? (nu - L + NL - 1) / NL : 0;
      pnl_u2 kr[D];
      pnl_d2 vr[D];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int m = L + NL * d;
        const int64_t e = (m < nu ? ((int64_t)(c0 + m) << 6) : dummy) + lane;
        ring_stream_issue(kr[d], vr[d], key2 + e, val2 + e);
      }
      int ibc = 0;
      int hist[R];  // phases of the last R units handed over, oldest first
#pragma unroll
      for (int r = 0; r < R; ++r) hist[r] = 0;
      for (int t0 = 0; t0 < nt; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const int t = t0 + d;
          if (t >= nt) break;  // wave-uniform
          const int c = c0 + L + NL * t;
          while (ibc + 1 < (int)nib && c >= (ring_sload(sp + ibc + 1) >> 1)) ++ibc;
          const int sl = L * S + (S > 1 ? t % S : 0);
          unsigned spins = 0;
          while (__builtin_amdgcn_readfirstlane(hdr[sl].x) != 0u) {  // the slot still holds an earlier unit
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kRingSpinLimit) { ring_fail(arrive + 1); break; }
          }
          lds_char *slot = ring + sl * kRingUnitBytes;
          ring_stream_wait<2 * (D - 1)>(kr[d], vr[d]);  // D - 1 younger units stay in flight
          ((lds_u2 *)slot)[lane] = kr[d];
          ((lds_d2 *)(slot + 512))[lane] = vr[d];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) { pnl_u2 h = {(unsigned)(t + 1), (unsigned)ibc}; hdr[sl] = h; }
          {  // refill this register set: the unit D places further down this loader's sequence
            const int m = L + NL * (t + D);
            const int64_t e = (m < nu ? ((int64_t)(c0 + m) << 6) : dummy) + lane;
            ring_stream_issue(kr[d], vr[d], key2 + e, val2 + e);
          }
#pragma unroll
          for (int r = 0; r + 1 < R; ++r) hist[r] = hist[r + 1];
          hist[R - 1] = ibc / K;
          while (bar_done < hist[0]) { __builtin_amdgcn_s_barrier(); ++bar_done; }
        }
      }
    } else {
      // ---- gather wavefront ----
      const int j = wave - NL;
      const int L = j % NL, q = j / NL;
      const int nt = nu > L ? (nu - L + NL - 1) / NL : 0;  // units of my loader; mine are q, q + R, ...
      pnl_u2 id[GD];
      pnl_d2 a[GD];
      double xa[GD], xb[GD];
      bool first = true;
      int t = q;
      int live = 0;  // units issued and not yet folded at loop exit
      while (t < nt) {
        live = 0;
#pragma unroll
        for (int u = 0; u < GD; ++u) {
          if (t >= nt) break;  // wave-uniform
          if (!first) {
            pnl_gather_wait<2 * (GD - 1) + 1>(xa[u]);
            pnl_gather_wait<2 * (GD - 1)>(xb[u]);
            pnl_fold(id[u].x, a[u].x * xa[u], ylds);
            pnl_fold(id[u].y, a[u].y * xb[u], ylds);
          }
          const int sl = L * S + (S > 1 ? t % S : 0);
          unsigned spins = 0;
          pnl_u2 h;
          for (;;) {
            h = hdr[sl];
            if (__builtin_amdgcn_readfirstlane(h.x) == (unsigned)(t + 1)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kRingSpinLimit) { ring_fail(arrive + 1); break; }
          }
          const int ib = __builtin_amdgcn_readfirstlane(h.y);
          const lds_char *slot = ring + sl * kRingUnitBytes;
          id[u] = ((const lds_u2 *)slot)[lane];
          a[u] = ((const lds_d2 *)(slot + 512))[lane];
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(id[u]), "+v"(a[u]) : : "memory");
          if (lane == 0) { pnl_u2 z = {0u, 0u}; hdr[sl] = z; }
          const int ph = ib / K;
          while (bar_done < ph) { __builtin_amdgcn_s_barrier(); ++bar_done; }
          const double *xw = x + ((int64_t)ib << w);
          xa[u] = pnl_gather_issue(xw + (id[u].x >> kRowBits));
          xb[u] = pnl_gather_issue(xw + (id[u].y >> kRowBits));
          t += R;
          ++live;
        }
        if (live == GD) first = false;
        else break;
      }
      // drain: everything still in flight (the last full round's units that were not re-used + the partial round)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < GD; ++u) {
        const bool pending = first ? (u < live) : true;  // after a full round every set holds an unfolded unit
        if (pending) {
          pnl_gather_wait<0>(xa[u]);
          pnl_gather_wait<0>(xb[u]);
          pnl_fold(id[u].x, a[u].x * xa[u], ylds);
          pnl_fold(id[u].y, a[u].y * xb[u], ylds);
        }
      }
    }
    while (bar_done < nph) { __builtin_amdgcn_s_barrier(); ++bar_done; }
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += kPanelWaves * 64)
      if (row_base + i < nrows) y[row_base + i] = ylds[i];
    if (g + 1 < ngen) {
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)((g + 1) * nb);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (wall_clock64() - t0 > 20000ull) break;
          __builtin_amdgcn_s_sleep(8);
        }
      }
      __syncthreads();
    }
  }
}

}  // namespace

void build_panel_image(Matrix *m, int P, int w, int pair, hipStream_t s) {
  auto b = std::make_unique<PanelImage>();
  b->P = P;
  b->w = w;
  b->npanels = (m->nrows_local + P - 1) / P;
  b->nib = (m->ncols + (1LL << w) - 1) >> w;
  if (b->nib < 1) b->nib = 1;
  const int64_t nseg = b->npanels * b->nib;
  constexpr int kSegPad = 10;  // trailing copies of the last boundary (the kernel peeks 2 K blocks ahead)
  DBuf<int> counts((size_t)nseg + 1);
  DBuf<int> chunks((size_t)nseg + 1);
  DBuf<int64_t> off64((size_t)nseg + 2);
  b->segc.alloc((size_t)nseg + 1 + kSegPad);
  SPL_HIP(hipMemsetAsync(counts.get(), 0, ((size_t)nseg + 1) * sizeof(int), s));
  const unsigned grid = blocks_for(m->nrows_local, 256);
  if (m->nrows_local > 0) {
    if (m->rowptr.get())
      hipLaunchKernelGGL(pnl_count_kernel<int>, dim3(grid), dim3(256), 0, s, m->nrows_local, m->rowptr.get(),
                         m->colidx.get(), P, w, b->nib, counts.get());
    else
      hipLaunchKernelGGL(pnl_count_kernel<int64_t>, dim3(grid), dim3(256), 0, s, m->nrows_local,
                         m->rowptr64.get(), m->colidx.get(), P, w, b->nib, counts.get());
  }
  hipLaunchKernelGGL(pnl_chunks_kernel, dim3(blocks_for(nseg, 256)), dim3(256), 0, s, nseg, counts.get(),
                     chunks.get(), pair);
  exclusive_scan_i32_to_i64(chunks.get(), off64.get(), nseg, s);
  int64_t nchunks = 0;
  SPL_HIP(hipMemcpyAsync(&nchunks, off64.get() + nseg, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  if (nchunks >= (int64_t)0x7fffffff - kPanelSlackChunks) throw DeviceError{SPL_ERROR_index_overflow};
  b->nchunks = nchunks;
  narrow_i64_to_i32(off64.get(), b->segc.get(), nseg + 1, s);
  hipLaunchKernelGGL(pnl_padseg_kernel, dim3(1), dim3(256), 0, s, (int64_t)kSegPad, (int)nchunks, b->segc.get(),
                     nseg + 1);
  const size_t entries = ((size_t)nchunks + kPanelSlackChunks) * 64;
  b->key.alloc(entries);
  b->val.alloc(entries);
  // padding: column 0 of the block, the dummy row P, value 0
  SPL_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(b->key.get()), P, entries, s));
  SPL_HIP(hipMemsetAsync(b->val.get(), 0, entries * sizeof(double), s));
  b->arrive.alloc(2);  // [0] generation rendezvous, [1] ring form: a bounded wait gave up
  SPL_HIP(hipMemsetAsync(counts.get(), 0, ((size_t)nseg + 1) * sizeof(int), s));
  if (m->nrows_local > 0) {
    if (m->rowptr.get())
      hipLaunchKernelGGL(pnl_fill_kernel<int>, dim3(grid), dim3(256), 0, s, m->nrows_local, m->rowptr.get(),
                         m->colidx.get(), m->val.get(), P, w, b->nib, b->segc.get(), counts.get(), b->key.get(),
                         b->val.get());
    else
      hipLaunchKernelGGL(pnl_fill_kernel<int64_t>, dim3(grid), dim3(256), 0, s, m->nrows_local,
                         m->rowptr64.get(), m->colidx.get(), m->val.get(), P, w, b->nib, b->segc.get(),
                         counts.get(), b->key.get(), b->val.get());
  }
  // cursor slots were handed out in arbitrary order: sort every padded segment by (column, row)
  hipLaunchKernelGGL(pnl_entryptr_kernel, dim3(blocks_for(nseg + 1, 256)), dim3(256), 0, s, nseg + 1,
                     b->segc.get(), off64.get());
  segmented_sort_pairs_u32(off64.get(), nseg, b->key.get(), b->val.get(), s);
  b->pair = pair ? 1 : 0;
  if (pair && nchunks > 0)
    hipLaunchKernelGGL(pnl_interleave_kernel, dim3((unsigned)(nchunks / 2)), dim3(128), 0, s, nchunks / 2,
                       b->key.get(), b->val.get());
  SPL_HIP(hipStreamSynchronize(s));
  delete m->panel;
  m->panel = b.release();
}

template <int U, int K, int ABL = 0>
static void launch_panel_as(const Matrix *m, const PanelImage *b, unsigned nb, size_t lds, const double *d_x,
                            double *d_y, int accumulate, hipStream_t s) {
  static std::atomic<uint64_t> set_{0};  // bit d: attribute set on device d (it is per device)
  if (!(set_.load(std::memory_order_acquire) >> (m->device & 63) & 1u)) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_panel_kernel<U, K, ABL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    set_.fetch_or(1ull << (m->device & 63), std::memory_order_release);
  }
  hipLaunchKernelGGL((spmv_panel_kernel<U, K, ABL>), dim3(nb), dim3(kPanelWaves * 64), lds, s, m->nrows_local,
                     b->npanels, b->P, b->w, b->nib, b->segc.get(), b->key.get(), b->val.get(), d_x, d_y,
                     accumulate, b->arrive.get(), (int64_t)b->nchunks << 6);
}

template <int U, int K>
static void launch_panelw_as(const Matrix *m, const PanelImage *b, unsigned nb, size_t lds, const double *d_x,
                             double *d_y, int accumulate, hipStream_t s) {
  static std::atomic<uint64_t> set_{0};
  if (!(set_.load(std::memory_order_acquire) >> (m->device & 63) & 1u)) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_panelw_kernel<U, K>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    set_.fetch_or(1ull << (m->device & 63), std::memory_order_release);
  }
  const int ns = b->nslices > 1 ? b->nslices : 1;
  if (ns > 1 && !accumulate) SPL_HIP(hipMemsetAsync(d_y, 0, (size_t)m->nrows_local * sizeof(double), s));
  hipLaunchKernelGGL((spmv_panelw_kernel<U, K>), dim3(nb), dim3(kPanelWaves * 64), lds, s, m->nrows_local, b->npanels,
                     b->P, b->w, b->nib, b->segc.get(), b->key.get(), b->val.get(), d_x, d_y, accumulate,
                     b->arrive.get(), (int64_t)(b->nchunks / 2) << 6, ns);
}


size_t panel_ring_lds_bytes(int P, int nl, int slots) {
  return ((((size_t)P + 1) * sizeof(double) + 15) & ~(size_t)15) + (size_t)nl * slots * (kRingUnitBytes + 8);
}

template <int NL, int D, int GD, int K, int S>
static void launch_panelr_as(const Matrix *m, const PanelImage *b, unsigned nb, const double *d_x, double *d_y,
                             int accumulate, hipStream_t s) {
  static std::atomic<uint64_t> set_{0};
  if (!(set_.load(std::memory_order_acquire) >> (m->device & 63) & 1u)) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_panelr_kernel<NL, D, GD, K, S>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    set_.fetch_or(1ull << (m->device & 63), std::memory_order_release);
  }
  hipLaunchKernelGGL((spmv_panelr_kernel<NL, D, GD, K, S>), dim3(nb), dim3(kPanelWaves * 64),
                     panel_ring_lds_bytes(b->P, NL, S), s, m->nrows_local, b->npanels, b->P, b->w, b->nib, b->segc.get(),
                     b->key.get(), b->val.get(), d_x, d_y, accumulate, b->arrive.get(), (int64_t)(b->nchunks / 2) << 6);
}

static int launch_panel_ring(const Matrix *m, const PanelImage *b, unsigned nb, const double *d_x, double *d_y,
                             int accumulate, hipStream_t s) {
  if (!b->pair) return SPL_ERROR_internal;
  const int nl = b->ring_nl, S = b->ring_slots, D = b->ring_depth, GD = b->ring_gather, K = b->kblocks;
  if (panel_ring_lds_bytes(b->P, nl, S) > 160 * 1024) return SPL_ERROR_argument_missing;
  SPL_HIP(hipMemsetAsync(b->arrive.get() + 1, 0, sizeof(unsigned), s));
#define SPL_RING(NLv, Dv, GDv, Kv, Sv) \
  if (nl == NLv && D == Dv && GD == GDv && K == Kv && S == Sv) { launch_panelr_as<NLv, Dv, GDv, Kv, Sv>(m, b, nb, d_x, d_y, accumulate, s); launched = true; }
  bool launched = false;
  // only shapes that compile without scratch: a spilled register with a gather still in flight would be stale
  SPL_RING(4, 4, 4, 2, 1) SPL_RING(4, 6, 4, 2, 1) SPL_RING(4, 8, 4, 2, 1) SPL_RING(4, 6, 3, 2, 1)
  SPL_RING(4, 6, 4, 1, 1) SPL_RING(4, 6, 4, 2, 3) SPL_RING(4, 4, 3, 2, 1) SPL_RING(4, 5, 4, 2, 1)
  SPL_RING(2, 8, 2, 2, 1) SPL_RING(8, 3, 4, 2, 1) SPL_RING(8, 4, 4, 2, 1)
  SPL_RING(4, 6, 4, 3, 1) SPL_RING(4, 6, 4, 4, 1) SPL_RING(4, 6, 4, 3, 3) SPL_RING(4, 6, 4, 4, 3) SPL_RING(4, 4, 4, 2, 3)
  SPL_RING(4, 6, 3, 2, 3) SPL_RING(4, 6, 4, 1, 3) SPL_RING(4, 4, 4, 4, 1) SPL_RING(4, 4, 4, 3, 1) SPL_RING(8, 4, 4, 4, 1)
  SPL_RING(8, 4, 4, 2, 2) SPL_RING(4, 4, 4, 8, 1) SPL_RING(4, 6, 4, 8, 1)
#undef SPL_RING
  if (!launched) return SPL_ERROR_argument_missing;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("spmv_panelr launch", e); return SPL_ERROR_device; }
  return SPL_OK;
}

int panel_ring_errors(const Matrix *m, hipStream_t s) {
  const PanelImage *b = m->panel;
  if (!b || !b->arrive.get()) return 0;
  unsigned e = 0;
  SPL_HIP(hipMemcpyAsync(&e, b->arrive.get() + 1, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  return (int)e;
}

int launch_spmv_panel(const Matrix *m, const double *d_x, double *d_y, int accumulate, hipStream_t s) {
  const PanelImage *b = m->panel;
  if (!b) return SPL_ERROR_internal;
  if (b->npanels == 0) return SPL_OK;
  if (m->nnz == 0) {
    if (!accumulate) SPL_HIP(hipMemsetAsync(d_y, 0, (size_t)m->nrows_local * sizeof(double), s));
    return SPL_OK;
  }
  const size_t lds = ((size_t)b->P + 1) * sizeof(double);
  if (lds > 160 * 1024) return SPL_ERROR_argument_missing;
  int cus = 0;
  SPL_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device));
  int64_t nb = cus;
  const int64_t tasks = b->npanels * (b->nslices > 1 && b->pair && !b->ring ? b->nslices : 1);
  if (nb > tasks) nb = tasks;
  SPL_HIP(hipMemsetAsync(b->arrive.get(), 0, sizeof(unsigned), s));
  if (b->ring) return launch_panel_ring(m, b, (unsigned)nb, d_x, d_y, accumulate, s);
  const int U = b->unroll, K = b->kblocks;
  if (b->ablate) {  // timing-only (wrong results): the two-stage kernel, 12 chunks, 2 blocks per phase
    switch (b->ablate) {
      case 1: launch_panel_as<12, 2, 1>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 2: launch_panel_as<12, 2, 2>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 3: launch_panel_as<12, 2, 3>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 4: launch_panel_as<12, 2, 4>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 5: launch_panel_as<12, 2, 5>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 6: launch_panel_as<12, 2, 6>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      default: launch_panel_as<12, 2, 7>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
    }
    hipError_t ea = hipGetLastError();
    if (ea != hipSuccess) { set_last_error("spmv_panel ablation launch", ea); return SPL_ERROR_device; }
    return SPL_OK;
  }
  if (b->pair) {  // paired storage: only the paired kernel reads it
    if (K == 1) {
      switch (U) {
        case 2: launch_panelw_as<2, 1>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
        case 4: launch_panelw_as<4, 1>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
        default: launch_panelw_as<3, 1>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      }
    } else {
      switch (U) {
        case 3: launch_panelw_as<3, 2>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
        case 4: launch_panelw_as<4, 2>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
        case 5: launch_panelw_as<5, 2>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
        default: launch_panelw_as<6, 2>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      }
    }
    hipError_t ew = hipGetLastError();
    if (ew != hipSuccess) { set_last_error("spmv_panelw launch", ew); return SPL_ERROR_device; }
    return SPL_OK;
  }
#define SPL_PNL(UU, KK) launch_panel_as<UU, KK>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s)
  if (K == 1) {
    switch (U) {
      case 4: SPL_PNL(4, 1); break;
      case 6: SPL_PNL(6, 1); break;
      case 8: SPL_PNL(8, 1); break;
      case 10: SPL_PNL(10, 1); break;
      default: SPL_PNL(12, 1); break;
    }
  } else {
    switch (U) {
      case 4: SPL_PNL(4, 2); break;
      case 6: SPL_PNL(6, 2); break;
      case 8: SPL_PNL(8, 2); break;
      case 10: SPL_PNL(10, 2); break;
      default: SPL_PNL(12, 2); break;
    }
  }
#undef SPL_PNL
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("spmv_panel launch", e); return SPL_ERROR_device; }
  return SPL_OK;
}

}  // namespace spl

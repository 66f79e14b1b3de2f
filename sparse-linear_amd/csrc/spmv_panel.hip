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
                                                         int *__restrict__ chunks) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nseg) chunks[i] = (segcount[i] + 63) >> 6;
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
// Touch the six 128-byte lines of one chunk of the matrix stream (256 B of keys, 512 B of values)
// through the scalar cache: the lines land in the XCD's L2 without taking a miss slot of the vector
// L1.  The loads return at any later time into the ONE register `t`, which therefore stays reserved
// (tied "+s" operand) from the first touch of a phase to the s_waitcnt lgkmcnt(0) at the start of the
// next: the compiler must never be free to reuse a register a scalar load is still going to write.
__device__ inline void pnl_touch_chunk(unsigned &t, const void *pk, const void *pv) {
  asm volatile(
      "s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x80\n\t"
      "s_load_dword %0, %2, 0x0\n\ts_load_dword %0, %2, 0x80\n\t"
      "s_load_dword %0, %2, 0x100\n\ts_load_dword %0, %2, 0x180"
      : "+s"(t) : "s"(pk), "s"(pv) : "memory");
}
__device__ inline void pnl_touch_join(unsigned &t) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t) : : "memory");
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
template <int U, int K, int PF, int ABL = 0>
__device__ inline void panel_phase(unsigned (&idC)[U], double (&aC)[U], unsigned (&idN)[U], double (&aN)[U],
                                   int cs, const int (&mid)[K > 1 ? K - 1 : 1], int ce, int64_t ib0, int w,
                                   const unsigned *__restrict__ key, const double *__restrict__ val,
                                   const double *__restrict__ x, double *yp, int wave, int nextlen, unsigned &touch,
                                   double &sink) {
  const int lane = threadIdx.x & 63;
  if (PF) pnl_touch_join(touch);  // the touches of the previous phase have returned
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
  const unsigned *kn = key + (((int64_t)(ce + wave)) << 6) + lane;
  const double *vn = val + (((int64_t)(ce + wave)) << 6) + lane;
#pragma unroll
  for (int u = 0; u < U; ++u) {  // ... then the next phase's stream, left in flight across the barrier
    idN[u] = __builtin_nontemporal_load(kn + (size_t)u * kPanelWaves * 64);
    if (ABL & 2) aN[u] = 1.0;
    else aN[u] = __builtin_nontemporal_load(vn + (size_t)u * kPanelWaves * 64);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (PF) {
    // lines of the phase after next, this wavefront's share, through the scalar cache
    const char *kb = reinterpret_cast<const char *>(key + (((int64_t)ce + nextlen) << 6));
    const char *vb = reinterpret_cast<const char *>(val + (((int64_t)ce + nextlen) << 6));
    // a chunk = 256 B of keys + 512 B of values = 2 + 4 lines of 128 B; wavefront i touches the
    // lines of chunks i, i + 16, ... like the loads that will follow
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t off = (int64_t)(wave + kPanelWaves * u);
      pnl_touch_chunk(touch, kb + off * 256, vb + off * 512);
    }
  }
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

template <int U, int K, int PF, int ABL = 0>
__global__ __launch_bounds__(kPanelWaves * 64) void spmv_panel_kernel(
    int64_t nrows, int64_t npanels, int P, int w, int64_t nib, const int *__restrict__ segc,
    const unsigned *__restrict__ key, const double *__restrict__ val, const double *__restrict__ x,
    double *__restrict__ y, int accumulate, unsigned *__restrict__ arrive) {
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
    unsigned touch = 0;
    double sink = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {  // prologue: this wavefront's first chunks of phase 0
      const int64_t k = (((int64_t)(c0 + wave + kPanelWaves * u)) << 6) + lane;
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
        panel_phase<U, K, PF, ABL>(idA, aA, idB, aB, cs, mid, ce, ib0, w, key, val, x, ylds, wave, cn - ce, touch, sink);
      }
      if (ph + 1 < nph) {
        const int64_t ib0 = (ph + 1) * K;
        int mid[K > 1 ? K - 1 : 1];
#pragma unroll
        for (int j = 0; j + 1 < K; ++j) { const int t = sp[ib0 + j + 1]; mid[j] = t < cend ? t : cend; }
        const int cs = sp[ib0];
        int ce = sp[ib0 + K]; ce = (ib0 + K < nib) ? ce : cend;
        int cn = sp[ib0 + 2 * K]; cn = (ib0 + 2 * K < nib) ? cn : cend;
        panel_phase<U, K, PF, ABL>(idB, aB, idA, aA, cs, mid, ce, ib0, w, key, val, x, ylds, wave, cn - ce, touch, sink);
      }
    }
    if (PF) pnl_touch_join(touch);
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


// ---- three-stage form -----------------------------------------------------------------------------
// The two-stage kernel above ends every phase with all of a CU's gathers drained: fold, barrier, key
// wait and address arithmetic pass before the next gathers reach the L2, and because the CUs of an
// XCD run in lockstep by design, its L2 — the unit whose request rate bounds this kernel
// (profiles/r02_gather_probe2.txt) — idles with them.  Here a wavefront issues the gathers of phase
// i+1 BEFORE it waits for those of phase i, and the stream of phase i+2 before that: three register
// sets rotate through the roles load -> gather -> fold, gathers are in flight at all times, and the
// barrier only keeps the wavefronts of a CU within one phase of each other (x window: two index
// blocks).  Every load is inline asm with counted vmcnt waits (vmcnt retires in order); a register an
// asm load is still going to write is tied ("+v") into the wait that precedes its first use, so the
// compiler never sees it as free in between.  A phase = one index block (K = 1).
__device__ inline unsigned pnl_ld_key(const unsigned *p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ inline double pnl_ld_val(const double *p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int N>
__device__ inline void pnl_wait_kv(unsigned &k, double &a) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(k), "+v"(a) : "n"(N) : "memory");
}
template <int N>
__device__ inline void pnl_wait_x(double &v) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}

template <int U>
struct PanelSet {
  unsigned id[U];
  double a[U];
  double x[U];
};

template <int U>
__device__ inline void pnl3_load(PanelSet<U> &L, int cs, int ce, int wave, int lane, const unsigned *__restrict__ key,
                                 const double *__restrict__ val, int64_t dummy_entry) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int c = cs + wave + kPanelWaves * u;
    const int64_t e = (c < ce ? ((int64_t)c << 6) : dummy_entry) + lane;  // wave-uniform select
    L.id[u] = pnl_ld_key(key + e);
    L.a[u] = pnl_ld_val(val + e);
  }
}

template <int U>
__device__ inline void pnl3_gather(PanelSet<U> &G, int cs, int ce, int wave, int64_t xbase,
                                   const double *__restrict__ x) {
  const double *xp[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool ok = cs + wave + kPanelWaves * u < ce;  // wave-uniform
    xp[u] = x + (ok ? xbase + (int64_t)(G.id[u] >> kRowBits) : 0);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) G.x[u] = pnl_gather_issue(xp[u]);
}

// one phase i: F holds phase i (its gathers in flight), G phase i+1 (its stream in flight), L is free
template <int U>
__device__ inline void panel3_phase(PanelSet<U> &F, PanelSet<U> &G, PanelSet<U> &L, int csF, int ceF, int ceG, int ceL,
                                    int64_t ibF, int w, const unsigned *__restrict__ key, const double *__restrict__ val,
                                    const double *__restrict__ x, double *yp, int wave, int lane, int64_t dummy_entry) {
  // a. the stream of phase i+2
  pnl3_load<U>(L, ceG, ceL, wave, lane, key, val, dummy_entry);
  // b. the keys of phase i+1: younger than them are gathers(i) [U] and stream(i+2) [2U]
#pragma unroll
  for (int u = 0; u < U; ++u) pnl_wait_kv<3 * U>(G.id[u], G.a[u]);
  // c. the gathers of phase i+1
  pnl3_gather<U>(G, ceF, ceG, wave, (ibF + 1) << w, x);
  // d. fold phase i: younger than its gather u are U-1-u gathers(i), stream(i+2) [2U], gathers(i+1) [U]
  if (U > 0) pnl_wait_x<4 * U - 1>(F.x[0]);
  if (U > 1) pnl_wait_x<4 * U - 2>(F.x[U > 1 ? 1 : 0]);
  if (U > 2) pnl_wait_x<4 * U - 3>(F.x[U > 2 ? 2 : 0]);
  if (U > 3) pnl_wait_x<4 * U - 4>(F.x[U > 3 ? 3 : 0]);
  if (U > 4) pnl_wait_x<4 * U - 5>(F.x[U > 4 ? 4 : 0]);
  if (U > 5) pnl_wait_x<4 * U - 6>(F.x[U > 5 ? 5 : 0]);
  if (U > 6) pnl_wait_x<4 * U - 7>(F.x[U > 6 ? 6 : 0]);
  if (U > 7) pnl_wait_x<4 * U - 8>(F.x[U > 7 ? 7 : 0]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (csF + wave + kPanelWaves * u >= ceF) break;  // wave-uniform
    pnl_fold(F.id[u], F.a[u] * F.x[u], yp);
  }
  for (int c = csF + wave + kPanelWaves * U; c < ceF; c += kPanelWaves) {  // tail of an over-long phase
    const unsigned id = pnl_ld_key(key + ((int64_t)c << 6) + lane);
    double a = pnl_ld_val(val + ((int64_t)c << 6) + lane);
    unsigned idw = id;
    pnl_wait_kv<0>(idw, a);
    double xv = pnl_gather_issue(x + (ibF << w) + (int64_t)(idw >> kRowBits));
    pnl_wait_x<0>(xv);
    pnl_fold(idw, a * xv, yp);
  }
  __builtin_amdgcn_s_barrier();  // pacing only
}

template <int U>
__device__ inline void pnl3_drain(PanelSet<U> &S) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    pnl_wait_kv<0>(S.id[u], S.a[u]);
    pnl_wait_x<0>(S.x[u]);
  }
}

template <int U>
__global__ __launch_bounds__(kPanelWaves * 64) void spmv_panel3_kernel(
    int64_t nrows, int64_t npanels, int P, int w, int64_t nib, const int *__restrict__ segc,
    const unsigned *__restrict__ key, const double *__restrict__ val, const double *__restrict__ x,
    double *__restrict__ y, int accumulate, unsigned *__restrict__ arrive, int64_t dummy_entry) {
  extern __shared__ __attribute__((aligned(16))) double ylds[];  // P + 1 doubles
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nb = gridDim.x;
  const int64_t ngen = (npanels + nb - 1) / nb;
  for (int64_t g = 0; g < ngen; ++g) {
    const int64_t p = g * nb + blockIdx.x;
    if (p >= npanels) break;  // only in the last generation: no rendezvous follows
    const int64_t row_base = p * P;
    const int *sp = segc + p * nib;
    const int cend = sp[nib];
    auto bound = [&](int64_t ib) -> int { const int t = sp[ib < nib ? ib : nib]; return ib < nib ? t : cend; };
    PanelSet<U> A, B, C;
    {  // prologue: stream(0) -> A, stream(1) -> B, then the gathers of phase 0
      const int c0 = sp[0], c1 = bound(1), c2 = bound(2);
      pnl3_load<U>(A, c0, c1, wave, lane, key, val, dummy_entry);
      pnl3_load<U>(B, c1, c2, wave, lane, key, val, dummy_entry);
#pragma unroll
      for (int u = 0; u < U; ++u) pnl_wait_kv<2 * U>(A.id[u], A.a[u]);
      pnl3_gather<U>(A, c0, c1, wave, 0, x);
#pragma unroll
      for (int u = 0; u < U; ++u) { C.id[u] = 0; C.a[u] = 0.0; C.x[u] = 0.0; B.x[u] = 0.0; }
    }
    for (int i = threadIdx.x; i <= P; i += kPanelWaves * 64)
      ylds[i] = (accumulate && i < P && row_base + i < nrows) ? y[row_base + i] : 0.0;
    __syncthreads();
    for (int64_t ph = 0; ph < nib; ph += 3) {
      {
        const int cs = sp[ph], ce = bound(ph + 1), ceG = bound(ph + 2), ceL = bound(ph + 3);
        panel3_phase<U>(A, B, C, cs, ce, ceG, ceL, ph, w, key, val, x, ylds, wave, lane, dummy_entry);
      }
      if (ph + 1 < nib) {
        const int cs = sp[ph + 1], ce = bound(ph + 2), ceG = bound(ph + 3), ceL = bound(ph + 4);
        panel3_phase<U>(B, C, A, cs, ce, ceG, ceL, ph + 1, w, key, val, x, ylds, wave, lane, dummy_entry);
      }
      if (ph + 2 < nib) {
        const int cs = sp[ph + 2], ce = bound(ph + 3), ceG = bound(ph + 4), ceL = bound(ph + 5);
        panel3_phase<U>(C, A, B, cs, ce, ceG, ceL, ph + 2, w, key, val, x, ylds, wave, lane, dummy_entry);
      }
    }
    pnl3_drain<U>(A);  // loads past the last phase went to the dummy chunk / x[0]: wait before the registers die
    pnl3_drain<U>(B);
    pnl3_drain<U>(C);
    __syncthreads();  // every wavefront's LDS adds are done
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

}  // namespace

void build_panel_image(Matrix *m, int P, int w, hipStream_t s) {
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
                     chunks.get());
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
  b->arrive.alloc(1);
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
  SPL_HIP(hipStreamSynchronize(s));
  delete m->panel;
  m->panel = b.release();
}

template <int U, int K, int PF, int ABL = 0>
static void launch_panel_as(const Matrix *m, const PanelImage *b, unsigned nb, size_t lds, const double *d_x,
                            double *d_y, int accumulate, hipStream_t s) {
  static bool set_ = false;
  if (!set_) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_panel_kernel<U, K, PF, ABL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    set_ = true;
  }
  hipLaunchKernelGGL((spmv_panel_kernel<U, K, PF, ABL>), dim3(nb), dim3(kPanelWaves * 64), lds, s, m->nrows_local,
                     b->npanels, b->P, b->w, b->nib, b->segc.get(), b->key.get(), b->val.get(), d_x, d_y,
                     accumulate, b->arrive.get());
}

template <int U>
static void launch_panel3_as(const Matrix *m, const PanelImage *b, unsigned nb, size_t lds, const double *d_x,
                             double *d_y, int accumulate, hipStream_t s) {
  static bool set_ = false;
  if (!set_) {
    SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_panel3_kernel<U>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    set_ = true;
  }
  hipLaunchKernelGGL((spmv_panel3_kernel<U>), dim3(nb), dim3(kPanelWaves * 64), lds, s, m->nrows_local, b->npanels,
                     b->P, b->w, b->nib, b->segc.get(), b->key.get(), b->val.get(), d_x, d_y, accumulate,
                     b->arrive.get(), (int64_t)b->nchunks << 6);
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
  if (nb > b->npanels) nb = b->npanels;
  SPL_HIP(hipMemsetAsync(b->arrive.get(), 0, sizeof(unsigned), s));
  const int U = b->unroll, K = b->kblocks, PF = b->prefetch;
  if (b->ablate) {  // timing-only (wrong results): the two-stage kernel, 12 chunks, 2 blocks per phase
    switch (b->ablate) {
      case 1: launch_panel_as<12, 2, 0, 1>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 2: launch_panel_as<12, 2, 0, 2>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 3: launch_panel_as<12, 2, 0, 3>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 4: launch_panel_as<12, 2, 0, 4>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 5: launch_panel_as<12, 2, 0, 5>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 6: launch_panel_as<12, 2, 0, 6>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      default: launch_panel_as<12, 2, 0, 7>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
    }
    hipError_t ea = hipGetLastError();
    if (ea != hipSuccess) { set_last_error("spmv_panel ablation launch", ea); return SPL_ERROR_device; }
    return SPL_OK;
  }
  if (b->stages == 3) {
    switch (U) {
      case 4: launch_panel3_as<4>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 5: launch_panel3_as<5>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      case 7: launch_panel3_as<7>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
      default: launch_panel3_as<6>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s); break;
    }
    hipError_t e3 = hipGetLastError();
    if (e3 != hipSuccess) { set_last_error("spmv_panel3 launch", e3); return SPL_ERROR_device; }
    return SPL_OK;
  }
#define SPL_PNL(UU, KK)                                                                       \
  do {                                                                                        \
    if (PF) launch_panel_as<UU, KK, 1>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s);     \
    else launch_panel_as<UU, KK, 0>(m, b, (unsigned)nb, lds, d_x, d_y, accumulate, s);        \
  } while (0)
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

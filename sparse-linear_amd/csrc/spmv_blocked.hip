// spmv_blocked.hip — column-blocked SpMV image for matrices without column locality.
//
// Why: for a uniformly random 1e7 x 1e7 matrix every x[c] gather misses the 4 MiB
// per-XCD L2 (x is 80 MB) and pulls a whole 128-byte line over the fabric:
// rocprofv3 counts 2.1e8 TCC_EA0_RDREQ_128B per SpMV = 27 GB moved for 2.6 GB of
// algorithmic bytes (profiles/r01_spmv_random_v0_pmc_summary.txt).  The fix is to
// bound the x window a wavefront gathers from at any moment so that it stays in L2.
//
// Image ("wave panels x column blocks"), built once per matrix in HBM:
//   * rows are cut into wave panels of R = 2^rw rows; columns into blocks of
//     W = 2^w columns (W*8 bytes of x: 1-2 MiB, a fraction of one XCD's L2);
//   * segment (p, cb) holds the entries of panel p whose column lies in block cb,
//     sorted by (row, column) — the CSR order restricted to the block — as a packed
//     32-bit key  (local_row << w | local_col)  plus the fp64 value: still 12 bytes
//     per entry, so the matrix stream is unchanged;
//   * segptr[p*ncb + cb] are the segment boundaries.
// Kernel: one wavefront per panel keeps the panel's y in LDS and walks the column
// blocks in ascending order; all wavefronts of the chip start together and advance
// at the same average rate, so at any time each XCD gathers from only a few
// neighbouring x blocks (loose lockstep, no barrier; any schedule is correct).
// Per 64 entries: coalesced non-temporal loads of key and value, one gather of x
// (L2 hit), then a segmented fold into LDS: runs of equal rows among adjacent lanes
// are summed left to right by their first lane, which reads y[row] from LDS, adds
// its run and writes it back.  No atomics; every y[r] receives a*x + y in ascending
// column order (blocks ascend, columns ascend inside a block), each multiply and add
// separately rounded: results are bit-identical to the reference order
// (Sparse.hs:447-451) and to the CSR-stream kernel.
#include "common.hpp"

namespace spl {

namespace {

constexpr int kStreamPad = 1024;  // >= 12 chunks of 64 entries

inline unsigned blocks_for(int64_t n, int per_block) {
  int64_t b = (n + per_block - 1) / per_block;
  return (unsigned)(b < 1 ? 1 : b);
}

// ---- image construction ------------------------------------------------------------------
template <typename PtrT>
__global__ __launch_bounds__(256) void blk_count_kernel(int64_t nrows, const PtrT *__restrict__ rowptr,
                                                        const int *__restrict__ colidx, int R, int w,
                                                        int64_t ncb, int *__restrict__ segcount) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int64_t base = (r / R) * ncb;
  for (PtrT k = rowptr[r]; k < rowptr[r + 1]; ++k) atomicAdd(&segcount[base + (colidx[k] >> w)], 1);
}

template <typename PtrT>
__global__ __launch_bounds__(256) void blk_fill_kernel(int64_t nrows, const PtrT *__restrict__ rowptr,
                                                       const int *__restrict__ colidx,
                                                       const double *__restrict__ val, int R, int w,
                                                       int64_t ncb, const int64_t *__restrict__ segptr,
                                                       int *__restrict__ cursor, int *__restrict__ key,
                                                       double *__restrict__ bval) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int64_t base = (r / R) * ncb;
  const int lr = (int)(r % R);
  const int wmask = (1 << w) - 1;
  for (PtrT k = rowptr[r]; k < rowptr[r + 1]; ++k) {
    const int c = colidx[k];
    const int64_t seg = base + (c >> w);
    const int64_t pos = segptr[seg] + atomicAdd(&cursor[seg], 1);
    key[pos] = (lr << w) | (c & wmask);
    bval[pos] = val[k];
  }
}

// ---- the kernels ------------------------------------------------------------------------------
// fold one 64-entry chunk (already multiplied) into the panel's y in LDS, reference order
__device__ inline void fold_chunk(int id, double prod, int w, double *yp) {
  const int lane = threadIdx.x & 63;
  const bool ok = id >= 0;
  const int lr = ok ? (id >> w) : 0x7fffffff;
  const int prev = __shfl_up(lr, 1, 64);
  const bool head = (lane == 0) || (prev != lr);
  const unsigned long long headmask = __ballot(head);
  double acc = ok ? yp[lr] : 0.0;
  acc = prod + acc;  // a * x + y   (Sparse.hs:449-451)
  unsigned long long f = headmask;
  for (int d = 1; d < 64; ++d) {
    f = (f << 1) & ~headmask;  // lanes that are the d-th follower of their run's head
    if (f == 0ull) break;
    const double np = __shfl_down(prod, d, 64);
    if (head && lane + d < 64 && ((f >> (lane + d)) & 1ull)) acc = np + acc;
  }
  if (head && ok) yp[lr] = acc;
  __builtin_amdgcn_wave_barrier();
}

// The same fold as ONE LDS instruction: ds_add_f64 of every lane's product into yp[row].
// Lanes of one instruction that hit the same address are applied one after the other by the
// LDS atomic unit; instructions of one wavefront execute in issue order.  Whether the
// same-address lanes are applied in ascending lane order (= ascending column, the
// reference order) is a hardware property: tests/test_gpu_spmv_blocked.py checks the
// result bit for bit against the oracle on rows with 3+ entries per chunk.
__device__ inline void fold_chunk_atomic(int id, double prod, int w, double *yp) {
  if (id >= 0)
    __builtin_amdgcn_ds_atomic_fadd_f64(
        (__attribute__((address_space(3))) double *)(yp + (id >> w)), prod);
}

// Hybrid fold: the first lane of every run of equal rows does a plain read-add-write, the
// followers (a minority: most rows have one entry per 64-entry chunk) add atomically in a
// second instruction.  Same order as the reference: head first, followers in lane order.
__device__ inline void fold_chunk_hybrid(int id, double prod, int w, double *yp) {
  const int lane = threadIdx.x & 63;
  const bool ok = id >= 0;
  const int lr = ok ? (id >> w) : 0x7fffffff;
  const int prev = __shfl_up(lr, 1, 64);
  const bool head = (lane == 0) || (prev != lr);
  if (ok && head) yp[lr] = prod + yp[lr];
  if (ok && !head)
    __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)(yp + lr), prod);
}

template <int FOLD>
__device__ inline void fold_any(int id, double prod, int w, double *yp) {
  if (FOLD == 1) fold_chunk_atomic(id, prod, w, yp);
  else if (FOLD == 2) fold_chunk_hybrid(id, prod, w, yp);
  else fold_chunk(id, prod, w, yp);
}

// Simple form (ablation baseline): one workgroup = 4 free-running wavefronts = 4 panels.
template <int UNROLL>
__global__ __launch_bounds__(256) void spmv_blocked_kernel(int64_t nrows, int64_t npanels, int R, int w,
                                                           int64_t ncb, const int64_t *__restrict__ segptr,
                                                           const int *__restrict__ key,
                                                           const double *__restrict__ val,
                                                           const double *__restrict__ x,
                                                           double *__restrict__ y, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) double ylds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t p = (int64_t)blockIdx.x * 4 + wave;
  if (p >= npanels) return;
  double *yp = ylds + (size_t)wave * R;
  const int64_t row_base = p * R;
  for (int i = lane; i < R; i += 64)
    yp[i] = (accumulate && row_base + i < nrows) ? y[row_base + i] : 0.0;
  __builtin_amdgcn_wave_barrier();
  const int wmask = (1 << w) - 1;
  const int64_t *sp = segptr + p * ncb;
  int64_t s = sp[0];
  for (int64_t cb = 0; cb < ncb; ++cb) {
    const int64_t e = sp[cb + 1];
    const double *xb = x + (cb << w);
    for (int64_t k0 = s; k0 < e; k0 += 64 * UNROLL) {
      int id[UNROLL];
      double a[UNROLL], xv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t k = k0 + u * 64 + lane;
        const bool ok = k < e;
        id[u] = ok ? __builtin_nontemporal_load(key + k) : -1;
        a[u] = ok ? __builtin_nontemporal_load(val + k) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) xv[u] = id[u] >= 0 ? xb[id[u] & wmask] : 0.0;
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (k0 + u * 64 >= e) break;  // wave-uniform
        fold_chunk(id[u], a[u] * xv[u], w, yp);
      }
    }
    s = e;
  }
  for (int i = lane; i < R; i += 64)
    if (row_base + i < nrows) y[row_base + i] = yp[i];
}

// Lockstep form (the product kernel): ONE workgroup of NW wavefronts per CU, each wavefront
// owning one panel (NW * R rows of y in LDS).  The workgroup crosses a barrier after every
// column block, so all gathers a CU issues at any moment fall into ONE x block; the CUs of an
// XCD start together and do statistically equal work per block, which keeps the XCD's live x
// window at one or two blocks (measured: gather L2 hit rate 56 % free-running -> 92 %).
// The matrix stream is software-pipelined ACROSS the barrier: while the gathers of block cb
// are in flight, the first U chunks of block cb+1 are already being loaded into a second
// register set (the barrier is a raw s_barrier, which does not drain vector memory), so a
// phase exposes one L2-hit gather latency instead of an HBM latency plus a gather latency.
// Barriers and the bounded generation rendezvous only order phases for locality; a
// wavefront touches nothing but its own panel, so no schedule can change the result.
// The x gathers of a phase are issued through inline asm so that they are ISSUED before the
// next block's stream loads (hipcc otherwise sinks them below those loads, and since vmcnt
// retires in order every fold would then wait for HBM-latency loads it does not need).
// The asm loads are invisible to the compiler's waitcnt pass, so each fold waits for its
// gather explicitly with a counted vmcnt: exactly 2*U stream loads are issued after the U
// gathers and before the first fold (pinned by the "memory" clobbers and sched_barriers).
__device__ inline double gather_issue(const double *p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int N>
__device__ inline void gather_wait(double &v) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}

// One phase of the lockstep kernel: gather + fold the chunks of block cb held in (idC, aC)
// while the first U chunks of block cb+1 are loaded into (idN, aN).  Register sets are passed
// by reference and swapped by the caller (ping-pong), so nothing is copied between phases and
// the next block's loads stay in flight across the barrier.
template <int U, int FOLD>
__device__ inline void lockstep_phase(int (&idC)[U], double (&aC)[U], int (&idN)[U], double (&aN)[U],
                                      int64_t s, int64_t e, int64_t s2, int w, int wmask,
                                      const int *__restrict__ key, const double *__restrict__ val,
                                      const double *__restrict__ xb, double *yp) {
  const int lane = threadIdx.x & 63;
  double xv[U];
  const double *xp[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {  // (compiler-counted waits for this block's keys land here)
    const bool ok = s + u * 64 + lane < e;
    idC[u] = ok ? idC[u] : -1;
    xp[u] = xb + (ok ? (idC[u] & wmask) : 0);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u) xv[u] = gather_issue(xp[u]);  // gathers first ...
  __builtin_amdgcn_sched_barrier(0);  // keep the issue order: vmcnt retires in order
  const int *kn = key + s2 + lane;      // the arrays carry kStreamPad entries of slack, so the
  const double *vn = val + s2 + lane;   // U chunks may run past the segment (masked when used)
#pragma unroll
  for (int u = 0; u < U; ++u) {  // ... then the next block's stream, left in flight
    idN[u] = __builtin_nontemporal_load(kn + u * 64);
    aN[u] = __builtin_nontemporal_load(vn + u * 64);
  }
  __builtin_amdgcn_sched_barrier(0);
  // younger than gather u at this point: U-1-u gathers + 2*U stream loads
  constexpr int Y = 3 * U - 1;
  gather_wait<Y>(xv[0]);
  if (U > 1) gather_wait<Y - 1>(xv[U > 1 ? 1 : 0]);
  if (U > 2) gather_wait<Y - 2>(xv[U > 2 ? 2 : 0]);
  if (U > 3) gather_wait<Y - 3>(xv[U > 3 ? 3 : 0]);
  if (U > 4) gather_wait<Y - 4>(xv[U > 4 ? 4 : 0]);
  if (U > 5) gather_wait<Y - 5>(xv[U > 5 ? 5 : 0]);
  if (U > 6) gather_wait<Y - 6>(xv[U > 6 ? 6 : 0]);
  if (U > 7) gather_wait<Y - 7>(xv[U > 7 ? 7 : 0]);
  if (U > 8) gather_wait<Y - 8>(xv[U > 8 ? 8 : 0]);
  if (U > 9) gather_wait<Y - 9>(xv[U > 9 ? 9 : 0]);
  if (U > 10) gather_wait<Y - 10>(xv[U > 10 ? 10 : 0]);
  if (U > 11) gather_wait<Y - 11>(xv[U > 11 ? 11 : 0]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (s + u * 64 >= e) break;  // wave-uniform
    fold_any<FOLD>(idC[u], aC[u] * xv[u], w, yp);
  }
  for (int64_t k0 = s + 64 * U; k0 < e; k0 += 64) {  // rare tail of an over-long segment
    const int64_t k = k0 + lane;
    const bool ok = k < e;
    const int id = ok ? __builtin_nontemporal_load(key + k) : -1;
    const double a = ok ? __builtin_nontemporal_load(val + k) : 0.0;
    fold_any<FOLD>(id, a * (id >= 0 ? xb[id & wmask] : 0.0), w, yp);
  }
  __builtin_amdgcn_s_barrier();  // pacing only: no fence, vector memory stays in flight
}

template <int U, int NW, int FOLD>
__global__ __launch_bounds__(NW * 64) void spmv_blocked_lockstep(
    int64_t nrows, int64_t npanels, int R, int w, int64_t ncb, int64_t ncols,
    const int64_t *__restrict__ segptr, const int *__restrict__ key, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, int accumulate, unsigned *__restrict__ arrive) {
  extern __shared__ __attribute__((aligned(16))) double ylds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: scalar loads below
  double *yp = ylds + (size_t)wave * R;
  const int wmask = (1 << w) - 1;
  const int64_t nb = gridDim.x;
  const int64_t ngen = (npanels + nb * NW - 1) / (nb * NW);
  (void)ncols;
  for (int64_t g = 0; g < ngen; ++g) {
    const int64_t p = (g * nb + blockIdx.x) * NW + wave;
    const bool have = p < npanels;
    const int64_t row_base = p * R;
    // segment boundaries of this panel come through the scalar cache (segptr has
    // npanels*ncb + 2 entries; a wavefront without a panel walks empty segments)
    const int64_t *sp = segptr + (have ? p : 0) * ncb;
    const int64_t s0 = sp[0];
    int idA[U], idB[U];
    double aA[U], aB[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // prologue: first chunks of block 0
      idA[u] = __builtin_nontemporal_load(key + s0 + lane + u * 64);
      aA[u] = __builtin_nontemporal_load(val + s0 + lane + u * 64);
    }
    if (have)
      for (int i = lane; i < R; i += 64)
        yp[i] = (accumulate && row_base + i < nrows) ? y[row_base + i] : 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int64_t cb = 0; cb < ncb; cb += 2) {
      const int64_t b0 = have ? sp[cb] : s0, b1 = have ? sp[cb + 1] : s0;
      lockstep_phase<U, FOLD>(idA, aA, idB, aB, b0, b1, b1, w, wmask, key, val, x + (cb << w), yp);
      if (cb + 1 < ncb) {
        const int64_t b2 = have ? sp[cb + 2] : s0;
        lockstep_phase<U, FOLD>(idB, aB, idA, aA, b1, b2, b2, w, wmask, key, val, x + ((cb + 1) << w), yp);
      }
    }
    if (have)
      for (int i = lane; i < R; i += 64)
        if (row_base + i < nrows) y[row_base + i] = yp[i];
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

void build_blocked_image(Matrix *m, int R, int w, hipStream_t s) {
  auto b = new BlockedImage();
  try {
    b->R = R;
    b->w = w;
    b->npanels = (m->nrows_local + R - 1) / R;
    b->ncb = (m->ncols + (1LL << w) - 1) >> w;
    if (b->ncb < 1) b->ncb = 1;
    const int64_t nseg = b->npanels * b->ncb;
    DBuf<int> counts((size_t)nseg);
    b->segptr.alloc((size_t)nseg + 2);  // +1 slack: the pipelined kernel peeks one block ahead
    b->key.alloc((size_t)m->nnz + kStreamPad);  // slack: the pipelined kernel loads whole chunk
    b->val.alloc((size_t)m->nnz + kStreamPad);  // groups past the last segment (never used)
    SPL_HIP(hipMemsetAsync(b->key.get() + m->nnz, 0xff, kStreamPad * sizeof(int), s));
    SPL_HIP(hipMemsetAsync(b->val.get() + m->nnz, 0, kStreamPad * sizeof(double), s));
    b->arrive.alloc(1);
    SPL_HIP(hipMemsetAsync(counts.get(), 0, (size_t)(nseg ? nseg : 1) * sizeof(int), s));
    const unsigned grid = blocks_for(m->nrows_local, 256);
    if (m->nrows_local > 0) {
      if (m->rowptr.get())
        hipLaunchKernelGGL(blk_count_kernel<int>, dim3(grid), dim3(256), 0, s, m->nrows_local, m->rowptr.get(),
                           m->colidx.get(), R, w, b->ncb, counts.get());
      else
        hipLaunchKernelGGL(blk_count_kernel<int64_t>, dim3(grid), dim3(256), 0, s, m->nrows_local,
                           m->rowptr64.get(), m->colidx.get(), R, w, b->ncb, counts.get());
    }
    exclusive_scan_i32_to_i64(counts.get(), b->segptr.get(), nseg, s);
    SPL_HIP(hipMemsetAsync(counts.get(), 0, (size_t)(nseg ? nseg : 1) * sizeof(int), s));
    if (m->nrows_local > 0) {
      if (m->rowptr.get())
        hipLaunchKernelGGL(blk_fill_kernel<int>, dim3(grid), dim3(256), 0, s, m->nrows_local, m->rowptr.get(),
                           m->colidx.get(), m->val.get(), R, w, b->ncb, b->segptr.get(), counts.get(),
                           b->key.get(), b->val.get());
      else
        hipLaunchKernelGGL(blk_fill_kernel<int64_t>, dim3(grid), dim3(256), 0, s, m->nrows_local,
                           m->rowptr64.get(), m->colidx.get(), m->val.get(), R, w, b->ncb, b->segptr.get(),
                           counts.get(), b->key.get(), b->val.get());
    }
    // cursor slots were handed out in arbitrary order: restore (row, column) order
    segmented_sort_pairs(b->segptr.get(), nseg, b->key.get(), b->val.get(), s);
    SPL_HIP(hipStreamSynchronize(s));
  } catch (...) {
    delete b;
    throw;
  }
  delete m->blocked;
  m->blocked = b;
}

// resident workgroups of the lockstep kernel: as many as the LDS (160 KiB) and the 32-wave
// limit of a CU admit
static int lockstep_per_cu(int NW, int R) {
  const size_t lds = (size_t)NW * (size_t)R * sizeof(double);
  int per_cu = (int)((160 * 1024) / (lds ? lds : 1));
  if (per_cu * NW > 32) per_cu = 32 / NW;
  return per_cu < 1 ? 1 : per_cu;
}

int launch_spmv_blocked(const Matrix *m, const double *d_x, double *d_y, int accumulate, int unroll,
                        hipStream_t s) {
  const BlockedImage *b = m->blocked;
  if (!b) return SPL_ERROR_internal;
  if (b->npanels == 0) return SPL_OK;
  hipError_t e;
  if (m->nnz == 0) {  // nothing to stream: y = 0 (or unchanged when accumulating)
    if (!accumulate) SPL_HIP(hipMemsetAsync(d_y, 0, (size_t)m->nrows_local * sizeof(double), s));
    return SPL_OK;
  }
  if (unroll < 0 || b->lockstep_waves == 0) {  // ablation baseline
    const int U = unroll < 0 ? -unroll : unroll;
    const size_t lds = (size_t)4 * (size_t)b->R * sizeof(double);
    if (lds > 160 * 1024) return SPL_ERROR_argument_missing;
    const unsigned grid = blocks_for(b->npanels, 4);
#define SPL_LAUNCH_BLOCKED(UU)                                                                          \
  do {                                                                                                  \
    static bool set_ = false;                                                                           \
    if (!set_) {                                                                                        \
      SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_blocked_kernel<UU>),             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));             \
      set_ = true;                                                                                      \
    }                                                                                                   \
    hipLaunchKernelGGL(spmv_blocked_kernel<UU>, dim3(grid), dim3(256), lds, s, m->nrows_local,          \
                       b->npanels, b->R, b->w, b->ncb, b->segptr.get(), b->key.get(), b->val.get(),     \
                       d_x, d_y, accumulate);                                                           \
  } while (0)
    switch (U) {
      case 1: SPL_LAUNCH_BLOCKED(1); break;
      case 2: SPL_LAUNCH_BLOCKED(2); break;
      case 8: SPL_LAUNCH_BLOCKED(8); break;
      default: SPL_LAUNCH_BLOCKED(4); break;
    }
#undef SPL_LAUNCH_BLOCKED
  } else {
    const int NW = b->lockstep_waves;
    const size_t lds = (size_t)NW * (size_t)b->R * sizeof(double);
    if (lds > 160 * 1024) return SPL_ERROR_argument_missing;
    int cus = 0;
    SPL_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device));
    int64_t nb = (int64_t)cus * lockstep_per_cu(NW, b->R);
    const int64_t need = (b->npanels + NW - 1) / NW;
    if (nb > need) nb = need;
    SPL_HIP(hipMemsetAsync(b->arrive.get(), 0, sizeof(unsigned), s));
#define SPL_LAUNCH_LS(UU, NN)                                                                            \
  do {                                                                                                   \
    if (b->fold == 1) SPL_LAUNCH_LS2(UU, NN, 1);                                                         \
    else if (b->fold == 2) SPL_LAUNCH_LS2(UU, NN, 2);                                                    \
    else SPL_LAUNCH_LS2(UU, NN, 0);                                                                      \
  } while (0)
#define SPL_LAUNCH_LS2(UU, NN, FF)                                                                         \
  do {                                                                                                   \
    static bool set_ = false;                                                                            \
    if (!set_) {                                                                                         \
      SPL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&spmv_blocked_lockstep<UU, NN, FF>),    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));              \
      set_ = true;                                                                                       \
    }                                                                                                    \
    hipLaunchKernelGGL((spmv_blocked_lockstep<UU, NN, FF>), dim3((unsigned)nb), dim3(NN * 64), lds,      \
                       s, m->nrows_local, b->npanels, b->R, b->w, b->ncb, m->ncols, b->segptr.get(),       \
                       b->key.get(), b->val.get(), d_x, d_y, accumulate, b->arrive.get());               \
  } while (0)
    if (NW == 16) {
      switch (unroll) {
        case 4: SPL_LAUNCH_LS(4, 16); break;
        case 8: SPL_LAUNCH_LS(8, 16); break;
        case 12: SPL_LAUNCH_LS(12, 16); break;
        default: SPL_LAUNCH_LS(10, 16); break;
      }
    } else if (NW == 4) {
      switch (unroll) {
        case 4: SPL_LAUNCH_LS(4, 4); break;
        case 8: SPL_LAUNCH_LS(8, 4); break;
        case 12: SPL_LAUNCH_LS(12, 4); break;
        default: SPL_LAUNCH_LS(10, 4); break;
      }
    } else {
      switch (unroll) {
        case 4: SPL_LAUNCH_LS(4, 8); break;
        case 8: SPL_LAUNCH_LS(8, 8); break;
        case 12: SPL_LAUNCH_LS(12, 8); break;
        default: SPL_LAUNCH_LS(10, 8); break;
      }
    }
#undef SPL_LAUNCH_LS
#undef SPL_LAUNCH_LS2
  }
  e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("spmv_blocked launch", e); return SPL_ERROR_device; }
  return SPL_OK;
}

}  // namespace spl

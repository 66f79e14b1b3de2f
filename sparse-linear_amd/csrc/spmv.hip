// spmv.hip — fp64 CSR SpMV for gfx950 (MI355X).
//
// Reference semantics: axpy_ / mulV, sparse-linear/src/Data/Matrix/Sparse.hs:433-471.
// The reference walks CSC columns and scatters `y[r] = a*x[c] + y[r]`; per output
// row that is the fold  acc <- a*x[c] + acc  over the row's entries in ascending
// column order starting from y0[r] (0 for mulV).  The kernels below read the
// row-major (CSR) image — bit for bit the CSC arrays of A^T — and keep exactly
// that per-row order and the separately rounded multiply and add (this file is
// compiled with -ffp-contract=off), so results are bit-identical to the
// reference order for every row shorter than one LDS chunk.
//
// Kernel `spmv_stream` ("CSR-stream"): HBM-bound by design.
//   * one wavefront owns 64 consecutive rows; its nnz range [S,E) is contiguous,
//     so colidx/val are read as fully coalesced 8/16-byte-per-lane streams
//     (non-temporal: each matrix byte is used once and must not evict x);
//   * products a*x[c] are staged in a wavefront-private LDS chunk (no
//     workgroup barrier anywhere: LDS operations of one wavefront complete in
//     issue order);
//   * lane l then folds the products of row l sequentially out of LDS — the
//     segmented reduction — and stores y[r] once;
//   * a chunk that lies entirely inside ONE long row is reduced by the whole
//     wavefront with shuffles instead;
//   * blockIdx is remapped so that each XCD (blocks b, b+8, ... share one)
//     walks a contiguous range of rows: neighbouring row blocks of banded /
//     stencil matrices then share their x lines in one XCD's L2.
//
// Algorithmic bytes per launch:
//   12*nnz + 4*(nrows+1) + 8*ncols + 8*nrows.
#include "common.hpp"

namespace spl {

namespace {

typedef int int2v __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

constexpr int kWavesPerBlock = 4;
constexpr int kRowsPerBlock = kWavesPerBlock * 64;

template <bool NT, typename T>
__device__ inline T stream_load(const T *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// flavours of the x gather (ablation): 0 plain, 1 non-temporal, 2 agent-scope (sc1, bypasses
// the CU's L1), 3 system-scope (sc0 sc1)
template <int GF>
__device__ inline double gather_load(const double *p) {
  if (GF == 1) return __builtin_nontemporal_load(p);
  if (GF == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (GF == 3) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return *p;
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// EPL: elements per lane per pass (chunk = 64*EPL products in LDS per wavefront)
// VW : elements per vector load (1, 2 or 4)
template <int EPL, int VW, bool NT, typename PtrT, int GF = 0, int ABL = 0>
__global__ __launch_bounds__(kWavesPerBlock * 64) void spmv_stream(
    int64_t nrows, int64_t nblocks, const PtrT *__restrict__ rowptr, const int *__restrict__ colidx,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    int accumulate) {
  constexpr int CH = 64 * EPL;
  constexpr int NV = EPL / VW;
  __shared__ double prod_all[kWavesPerBlock][CH];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // XCD-aware remap: blocks with equal (blockIdx % 8) share an XCD and get a
  // contiguous range of row blocks (speed only; any placement is correct).
  const int64_t per_xcd = gridDim.x >> 3;
  const int64_t rb = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (rb >= nblocks) return;
  const int64_t r0 = rb * kRowsPerBlock + (int64_t)wave * 64;
  if (r0 >= nrows) return;
  double *prod = prod_all[wave];

  const int64_t r = r0 + lane;
  const bool valid = r < nrows;
  const int64_t rc = valid ? r : nrows - 1;
  PtrT my_s = rowptr[rc];
  PtrT my_e = rowptr[rc + 1];
  const int nvalid = (nrows - r0) < 64 ? (int)(nrows - r0) : 64;
  const PtrT S = __shfl(my_s, 0, 64);
  const PtrT E = __shfl(my_e, nvalid - 1, 64);
  if (!valid) { my_s = E; my_e = E; }

  double acc = (accumulate && valid) ? y[r] : 0.0;

  for (PtrT b0 = S & ~(PtrT)(VW - 1); b0 < E; b0 += CH) {
    // ---- stream the chunk: coalesced vector loads, gather x, products to LDS
    int c[EPL];
    double a[EPL];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const PtrT k = b0 + (PtrT)((i * 64 + lane) * VW);
      if (k < E) {
        if (VW == 1) {
          c[i] = stream_load<NT>(colidx + k);
          a[i] = stream_load<NT>(val + k);
        } else if (VW == 2) {
          const int2v cv = stream_load<NT>(reinterpret_cast<const int2v *>(colidx + k));
          const double2v av = stream_load<NT>(reinterpret_cast<const double2v *>(val + k));
          c[2 * i] = cv.x; c[2 * i + 1] = cv.y;
          a[2 * i] = av.x; a[2 * i + 1] = av.y;
        } else {
          const int4v cv = stream_load<NT>(reinterpret_cast<const int4v *>(colidx + k));
          const double2v a0 = stream_load<NT>(reinterpret_cast<const double2v *>(val + k));
          const double2v a1 = stream_load<NT>(reinterpret_cast<const double2v *>(val + k + 2));
          c[4 * i] = cv.x; c[4 * i + 1] = cv.y; c[4 * i + 2] = cv.z; c[4 * i + 3] = cv.w;
          a[4 * i] = a0.x; a[4 * i + 1] = a0.y; a[4 * i + 2] = a1.x; a[4 * i + 3] = a1.y;
        }
#pragma unroll
        for (int j = 0; j < VW; ++j)  // tail of the last vector: never gather with a foreign index
          if (k + j >= E) { c[VW * i + j] = 0; a[VW * i + j] = 0.0; }
      } else {
#pragma unroll
        for (int j = 0; j < VW; ++j) { c[VW * i + j] = 0; a[VW * i + j] = 0.0; }
      }
    }
    double p[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      if (ABL >= 1) p[e] = a[e] * (double)(c[e] & 1);  // ablation: no x gather (timing only, wrong result)
      else p[e] = a[e] * gather_load<GF>(x + c[e]);
    }
    if (ABL == 2) {  // ablation: no LDS staging / fold either
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc += p[e];
      continue;
    }

    // ---- one long row covers the whole chunk: wavefront-wide reduction
    const bool covers = (my_s <= b0) && (my_e >= b0 + CH);
    const unsigned long long cover_mask = __ballot(covers);
    if (cover_mask != 0ull) {
      double part = 0.0;
#pragma unroll
      for (int e = 0; e < EPL; ++e) part += p[e];
      part = wave_sum(part);
      if (covers) acc = part + acc;
      continue;
    }

#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int off = (i * 64 + lane) * VW;
      if (VW == 1) {
        prod[off] = p[i];
      } else if (VW == 2) {
        double2v pv; pv.x = p[2 * i]; pv.y = p[2 * i + 1];
        *reinterpret_cast<double2v *>(prod + off) = pv;
      } else {
        double2v p0, p1;
        p0.x = p[4 * i]; p0.y = p[4 * i + 1]; p1.x = p[4 * i + 2]; p1.y = p[4 * i + 3];
        *reinterpret_cast<double2v *>(prod + off) = p0;
        *reinterpret_cast<double2v *>(prod + off + 2) = p1;
      }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- segmented reduction: lane folds its row's slice of the chunk in order
    const PtrT lo = (my_s > b0 ? my_s : b0) - b0;
    const PtrT hi = (my_e < b0 + CH ? my_e : b0 + CH) - b0;
    for (PtrT t = lo; t < hi; ++t) acc = prod[t] + acc;
    __builtin_amdgcn_wave_barrier();
  }
  if (valid) y[r] = acc;
}

// Classic sub-wavefront CSR kernel (G lanes per row, shuffle reduction): kept as
// the ablation baseline for the streaming kernel.  Summation order differs from
// the reference order (tolerance-checked only).
template <int G>
__global__ __launch_bounds__(256) void spmv_subwave(int64_t nrows, const int *__restrict__ rowptr,
                                                    const int *__restrict__ colidx,
                                                    const double *__restrict__ val,
                                                    const double *__restrict__ x,
                                                    double *__restrict__ y, int accumulate) {
  const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const int gl = threadIdx.x & (G - 1);
  const bool valid = gid < nrows;
  const int64_t r = valid ? gid : nrows - 1;
  const int s = rowptr[r], e = valid ? rowptr[r + 1] : s;
  double acc = 0.0;
  for (int k = s + gl; k < e; k += G) acc += val[k] * x[colidx[k]];
#pragma unroll
  for (int d = G >> 1; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (valid && gl == 0) y[r] = accumulate ? acc + y[r] : acc;
}

// Sparse x dense (mulM, Sparse.hs:473-498): C = A B for a row-major dense B (ncols x k).  The
// reference runs one axpy_ per column of B; here the matrix is read ONCE for all k columns: a
// group of G lanes owns a row, lane j of the group owns output column j, and every stored entry
// (r, c) gathers the k contiguous doubles B[c, :] (a full cache line for k >= 16 instead of 8
// useful bytes of one).  Each (row, column) accumulator still folds a*b + acc over the row's
// entries in ascending column order: bit-identical to the reference's per-column axpy_.
template <typename PtrT>
__global__ __launch_bounds__(256) void spmm_rowgroup_kernel(int64_t nrows, const PtrT *__restrict__ rowptr,
                                                            const int *__restrict__ colidx,
                                                            const double *__restrict__ val,
                                                            const double *__restrict__ B,
                                                            double *__restrict__ C, int k, int G,
                                                            int accumulate) {
  const int64_t gthread = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gthread / G;
  const int j0 = (int)(gthread % G);
  if (row >= nrows) return;
  const PtrT s = rowptr[row], e = rowptr[row + 1];
  for (int j = j0; j < k; j += G) {
    double acc = accumulate ? C[row * k + j] : 0.0;
    for (PtrT p = s; p < e; ++p) acc = val[p] * B[(int64_t)colidx[p] * k + j] + acc;
    C[row * k + j] = acc;
  }
}

int launch_spmm_impl(const Matrix *m, const double *d_B, double *d_C, int k, int accumulate, hipStream_t s) {
  if (m->nrows_local == 0 || k == 0) return SPL_OK;
  int G = 1;
  while (G < k && G < 64) G <<= 1;
  const int64_t threads = m->nrows_local * G;
  const int64_t grid = (threads + 255) / 256;
  if (grid > 0x7fffffffLL) return SPL_ERROR_internal;
  if (m->rowptr.get())
    hipLaunchKernelGGL(spmm_rowgroup_kernel<int>, dim3((unsigned)grid), dim3(256), 0, s, m->nrows_local,
                       m->rowptr.get(), m->colidx.get(), m->val.get(), d_B, d_C, k, G, accumulate);
  else
    hipLaunchKernelGGL(spmm_rowgroup_kernel<int64_t>, dim3((unsigned)grid), dim3(256), 0, s, m->nrows_local,
                       m->rowptr64.get(), m->colidx.get(), m->val.get(), d_B, d_C, k, G, accumulate);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("spmm launch", e); return SPL_ERROR_device; }
  return SPL_OK;
}

template <int EPL, int VW, bool NT, int GF = 0, int ABL = 0>
int launch_stream(const Matrix *m, const double *d_x, double *d_y, int accumulate, hipStream_t s) {
  const int64_t nblocks = (m->nrows_local + kRowsPerBlock - 1) / kRowsPerBlock;
  const int64_t grid = ((nblocks + 7) / 8) * 8;
  if (grid > 0x7fffffffLL) return SPL_ERROR_internal;
  if (m->rowptr.get()) {
    hipLaunchKernelGGL((spmv_stream<EPL, VW, NT, int, GF, ABL>), dim3((unsigned)grid), dim3(kWavesPerBlock * 64),
                       0, s, m->nrows_local, nblocks, m->rowptr.get(), m->colidx.get(), m->val.get(),
                       d_x, d_y, accumulate);
  } else {
    hipLaunchKernelGGL((spmv_stream<EPL, VW, NT, int64_t, GF, ABL>), dim3((unsigned)grid),
                       dim3(kWavesPerBlock * 64), 0, s, m->nrows_local, nblocks, m->rowptr64.get(),
                       m->colidx.get(), m->val.get(), d_x, d_y, accumulate);
  }
  return SPL_OK;
}

}  // namespace

int launch_spmm(const Matrix *m, const double *d_B, double *d_C, int k, int accumulate, hipStream_t s) {
  return launch_spmm_impl(m, d_B, d_C, k, accumulate, s);
}

int launch_spmv(const Matrix *m, const double *d_x, double *d_y, int accumulate, hipStream_t s) {
  if (m->nrows_local == 0) return SPL_OK;
  if (m->vw == 2) return launch_spmv_z(m, d_x, d_y, accumulate, s);
  int st = SPL_OK;
  if (m->variant == 15 || (m->variant == 0 && m->sell)) {
    if (!m->sell) return SPL_ERROR_argument_missing;
    return launch_spmv_sell(m, d_x, d_y, accumulate, s);
  }
  if (m->variant == 16 || (m->variant == 0 && m->panel && m->order_free)) {
    if (!m->panel) return SPL_ERROR_argument_missing;
    return launch_spmv_panel(m, d_x, d_y, accumulate, s);
  }
  if (m->variant == 8 || (m->variant == 0 && m->blocked)) {
    if (!m->blocked) return SPL_ERROR_argument_missing;
    return launch_spmv_blocked(m, d_x, d_y, accumulate, m->blocked_unroll, s);
  }
  switch (m->variant) {
    case 0:
    case 1: st = launch_stream<8, 2, true>(m, d_x, d_y, accumulate, s); break;
    case 2: st = launch_stream<8, 4, true>(m, d_x, d_y, accumulate, s); break;
    case 3: st = launch_stream<4, 2, true>(m, d_x, d_y, accumulate, s); break;
    case 4: st = launch_stream<16, 4, true>(m, d_x, d_y, accumulate, s); break;
    case 5: st = launch_stream<8, 1, true>(m, d_x, d_y, accumulate, s); break;
    case 6: st = launch_stream<8, 2, false>(m, d_x, d_y, accumulate, s); break;
    case 9: st = launch_stream<8, 2, true, 1>(m, d_x, d_y, accumulate, s); break;
    case 10: st = launch_stream<8, 2, true, 2>(m, d_x, d_y, accumulate, s); break;
    case 11: st = launch_stream<8, 2, true, 3>(m, d_x, d_y, accumulate, s); break;
    case 12: st = launch_stream<8, 2, true, 0, 1>(m, d_x, d_y, accumulate, s); break;  // timing ablations
    case 13: st = launch_stream<8, 2, true, 0, 2>(m, d_x, d_y, accumulate, s); break;
    case 14: st = launch_stream<16, 4, true, 0, 2>(m, d_x, d_y, accumulate, s); break;
    case 7: {
      if (!m->rowptr.get()) return SPL_ERROR_index_overflow;
      const int64_t threads = m->nrows_local * 16;
      const int64_t grid = (threads + 255) / 256;
      hipLaunchKernelGGL((spmv_subwave<16>), dim3((unsigned)grid), dim3(256), 0, s, m->nrows_local,
                         m->rowptr.get(), m->colidx.get(), m->val.get(), d_x, d_y, accumulate);
      break;
    }
    default: return SPL_ERROR_argument_missing;
  }
  if (st != SPL_OK) return st;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error("spmv launch", e); return SPL_ERROR_device; }
  return SPL_OK;
}

// which kernel spl_matrix_spmv_dev launches for this handle now: 0 CSR-stream (variants 1-7, 9-14
// report themselves), 8 column-blocked lockstep, 15 sliced ELL, 16 column-sorted panels
int spmv_kernel_in_use(const Matrix *m) {
  if ((m->variant == 15 || m->variant == 0) && m->sell) return 15;
  if (m->variant == 16 || (m->variant == 0 && m->panel && m->order_free)) return m->panel ? 16 : 0;
  if ((m->variant == 8 || m->variant == 0) && m->blocked) return 8;
  return m->variant;
}

// CUs the persistent images are laid out for.  In a multi-GPU step a communication kernel (RCCL's all-gather
// when its chunks run under the SpMV of the next chunk) wants CUs of its own: the lockstep kernels pace
// themselves by equal work per resident workgroup, so a workgroup that shares its CU falls behind and holds up
// its generation.  With r CUs reserved (spl_matrix_set_reserved_cus / SPL_SPMV_RESERVED_CUS) the images of a
// one-generation row block have exactly CUs - r panels (groups of wavefront panels), the grid is that much
// smaller, and the dispatcher has free CUs for the other kernel (tools/bench_reserved_cus.py).
int spmv_cus(const Matrix *m) {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device);
  int r = m->reserved_cus;
  if (r < 0) {
    const char *e = getenv("SPL_SPMV_RESERVED_CUS");
    r = e ? atoi(e) : 0;
  }
  if (r < 0) r = 0;
  if (r > cus - 8) r = cus - 8;
  return cus - r;
}

// Shape of the column-sorted panel image (spmv_panel.hip): one panel per workgroup, as tall as the
// LDS allows, the generations of the persistent grid full; index blocks of 2^17 columns (the key's
// 17 column bits) or fewer for narrow matrices.
void choose_panels(const Matrix *m, int *rows_per_panel, int *w, int *nslices) {
  const int cus = spmv_cus(m);
  const int64_t pmax = 20479;
  int64_t ngen = (m->nrows_local + (int64_t)cus * pmax - 1) / ((int64_t)cus * pmax);
  if (ngen < 1) ngen = 1;
  int64_t P = (m->nrows_local + ngen * cus - 1) / (ngen * cus);
  if (P < 64) P = 64;
  if (P > pmax) P = pmax;
  int ww = 17;
  while (ww > 4 && (1LL << ww) >= 2 * m->ncols) --ww;
  *w = ww;
  int ns = 1;
  // A row block too short to give every CU a tall panel (a rank's block at N = 4, 8): ns column slices per panel
  // (csrc/spmv_panel.hip "Column slices"), the largest power of two that still leaves ONE generation of
  // full-height panels — more slices mean more generations and more atomic adds into y (measured on the block of
  // a rank at N = 8: 1 / 2 / 4 / 8 slices 0.201 / 0.158 / 0.151 / 0.195 ms; at N = 4: 1 / 2 / 4: 0.286 / 0.260 / 0.310)
  const int64_t nib = (m->ncols + (1LL << ww) - 1) >> ww;
  if (nslices && ngen == 1 && P * 5 < pmax * 3 && nib >= 32) {
    int best = 1;
    for (int f = 2; f <= 8; f *= 2)
      if (cus % f == 0 && (int64_t)(cus / f) * pmax >= m->nrows_local) best = f;
    if (best > 1) {
      const int64_t ppg = cus / best;
      P = (m->nrows_local + ppg - 1) / ppg;
      if (P > pmax) P = pmax;
      ns = best;
    }
  }
  if (const char *ev = getenv("SPL_PANEL_SLICES")) {
    const int f = atoi(ev);
    if (nslices && f >= 1 && f != ns) {  // forced (experiments): panels sized for cus / f workgroups per generation
      ns = f;
      const int64_t ppg = cus / f > 0 ? cus / f : 1;
      int64_t ng = (m->nrows_local + ppg * pmax - 1) / (ppg * pmax);
      if (ng < 1) ng = 1;
      P = (m->nrows_local + ng * ppg - 1) / (ng * ppg);
      if (P < 64) P = 64;
      if (P > pmax) P = pmax;
    }
  }
  *rows_per_panel = (int)P;
  if (nslices) *nslices = ns;
}

void choose_panels(const Matrix *m, int *rows_per_panel, int *w) { choose_panels(m, rows_per_panel, w, nullptr); }

// The panel image beats the column-blocked one when a 128-byte line of x meets enough entries of a
// panel for lanes to share requests: entries per line = 16 * nnz/nrows * P / ncols.
// (measured on C2, P = 19 532: 0.63 per line, 0.99 ms vs 1.19 ms; tools/bench_spmv_variants.py)
// Order-free sums and rows that share no x lines: the column-sorted panels beat the CSR-stream kernel from the smallest
// sizes measured on (x = 1 MB: 2.0 vs 1.7 TB/s; 8 MB: 3.0 vs 1.2; 32 MB: 3.2 vs 0.77 — round 3 left everything below
// 32 MiB of x to the stream kernel: the R-MAT matrix of config C4, x = 8 MB, ran at 1.17 TB/s instead of 3.4)
bool panels_beat_stream(const Matrix *m) {
  if (!m->order_free || m->nrows_local < 1 || m->ncols < 1) return false;
  if (m->ncols * 8 < (1LL << 20)) return false;        // tiny: one launch of anything
  if (m->new_line_fraction < 0.5) return false;        // rows reuse their neighbours' lines: sliced ELL
  if (m->nnz < 4 * m->nrows_local) return false;       // too sparse for the panels' chunks
  return panels_pay(m);
}

bool panels_pay(const Matrix *m) {
  int P = 0, w = 0, ns = 1;
  choose_panels(m, &P, &w, &ns);
  if (m->nrows_local < 1 || m->ncols < 1) return false;
  const double per_line = 16.0 * ((double)m->nnz / (double)m->nrows_local) * (double)P / (double)m->ncols;
  return per_line >= 0.3;
}

// Blocking pays when x does not fit ONE XCD's L2 (4 MiB: a CU gathers through the L2 of its own XCD) and
// neighbouring rows do not share x lines (measured: random 1e7 gathers move 10x the algorithmic bytes; banded rows
// reuse lines).  Shape (profiles/r05_blocked_threshold_sweep.txt: every n = 2^17 .. 2^23 with 4 / 8 / 16 wavefronts per
// CU and x windows of 2^13 .. 2^18 columns, all 256 CUs busy): the kernel is fastest when a (panel, column block) SEGMENT
// holds about ten 64-entry chunks — the depth of its register pipeline; longer segments fall into the un-pipelined tail
// loop, shorter ones pay the fixed cost of a phase (barrier, segment pointers) per few entries — and, at equal segment
// length, with more wavefronts per CU.  So: for nw = 16, 8, 4 wavefronts per CU take the panel height that fills the chip
// (several full generations when the LDS is too small), then the widest x window <= 2 MiB whose segments stay within
// ~11 chunks; accept the first nw whose segments are at least half a pipeline long (a rank's short row block of a wide
// matrix ends at nw = 4: 1/8 of C2 0.197 ms with 4 x 1221 rows against 0.230 ms with 16 x 306).
// Round 4 left everything below 12 MiB of x to the CSR-stream kernel on the strength of a sweep whose blocked image had
// the default 16 x 1024-row shape (8 .. 128 of 256 CUs busy at n = 2^17 .. 2^21); shaped as above the blocked image wins
// from 4 MiB of x on: n = 2^19 0.071 vs 0.077 ms, 2^20 0.131 vs 0.224, 2^21 0.247 vs 0.617, R-MAT scale 20 0.194 vs 0.365.
void choose_blocking(const Matrix *m, int *rows_per_panel, int *w, int *waves) {
  *rows_per_panel = 0;
  *w = 0;
  *waves = 16;
  const int64_t x_bytes = m->ncols * 8;
  if (x_bytes < (4LL << 20)) return;         // x fits one XCD's L2: the CSR-stream kernel (2 MiB: 0.037 vs 0.036 ms)
  if (m->new_line_fraction < 0.5) return;    // rows reuse their neighbours' lines (banded, stencil)
  if (m->nnz < 4 * m->nrows_local) return;   // too sparse for 64-entry chunks per segment
  const int cus = spmv_cus(m);
  const double avg = (double)m->nnz / (double)(m->nrows_local > 0 ? m->nrows_local : 1);
  auto shape_for = [&](int nw, int64_t *R_out, int *w_out) -> double {  // -> entries per segment
    const int64_t rmax = (160 * 1024 / 8) / nw;  // rows of y per wavefront: nw wavefronts fill the LDS
    const int64_t slots = (int64_t)cus * nw;
    const int64_t ngen = (m->nrows_local + slots * rmax - 1) / (slots * rmax);  // generations of full panels
    int64_t R = (m->nrows_local + ngen * slots - 1) / (ngen * slots);
    if (R < 64) R = 64;
    int ww = 18;  // 2 MiB of x
    auto seg = [&](int wv) { return avg * (double)R / (double)((m->ncols + (1LL << wv) - 1) >> wv); };
    while (ww > 13 && seg(ww) > 700.0) --ww;
    *R_out = R;
    *w_out = ww;
    return seg(ww);
  };
  int64_t R = 0;
  int ww = 18, nw = 16;
  for (int cand : {16, 8, 4}) {
    nw = cand;
    const double seg = shape_for(cand, &R, &ww);
    if (seg >= 320.0) break;
  }
  *rows_per_panel = (int)R;
  *waves = nw;
  *w = ww;
}

}  // namespace spl

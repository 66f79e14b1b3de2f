// common.hpp — shared host-side plumbing of the gfx950 backend.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <memory>
#include <new>
#include <vector>

#include "../../include/sparse_linear_hip.h"

namespace spl {

// thread-local text of the last HIP failure, returned by spl_last_error()
void set_last_error(const char *where, hipError_t e);
void set_last_error_text(const char *text);

struct DeviceError {
  int status;
};

#define SPL_HIP(expr)                                                        \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      ::spl::set_last_error(#expr, e_);                                      \
      throw ::spl::DeviceError{e_ == hipErrorOutOfMemory ? SPL_ERROR_out_of_memory \
                                                         : SPL_ERROR_device}; \
    }                                                                        \
  } while (0)

// device memory for DBuf (device_pool.hip): large blocks are kept on release and reused
void *device_alloc(size_t bytes);
void device_free(void *p) noexcept;
size_t device_free_bytes();      // free memory as the driver reports it + what the pool would give back
size_t device_release_cached();  // returns the bytes given back to the driver
double device_alloc_seconds();  // seconds spent inside hipMalloc so far (pool misses)
// Non-blocking streams, kept for the life of the process (device_pool.hip): creating one costs 0.3 - 0.6 ms here and the
// first of a process 15 ms; the analysis takes four for its level structures, a factorisation seventeen per host thread.
hipStream_t pooled_stream_take(int device);            // an idle one of this device, or a new one (nullptr: creation failed)
void pooled_stream_give(int device, hipStream_t s);    // back, once its work is complete
void pooled_streams_prewarm(int device, int count);    // create streams until `count` are idle (another thread may take them meanwhile)

// RAII device buffer, movable
template <typename T>
struct DBuf {
  T *p = nullptr;
  size_t n = 0;
  DBuf() = default;
  explicit DBuf(size_t count) { alloc(count); }
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  DBuf(DBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DBuf &operator=(DBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DBuf() { release(); }
  void alloc(size_t count) {
    release();
    n = count;
    // +64 B slack: vector loads of the streaming kernels may read (never use) a
    // few elements past the logical end
    p = static_cast<T *>(device_alloc((count ? count : 1) * sizeof(T) + 64));
  }
  void release() {
    if (p) { device_free(p); p = nullptr; n = 0; }
  }
  T *get() const { return p; }
};

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Kernel attributes (hipFuncSetAttribute: dynamic LDS above 64 KB) are per DEVICE: a `static bool` would leave the second
// GPU of a process without them.  One bit per device in a mask per call site; true for the first caller on the current
// device.  (Two threads racing on the first use both set the attributes: the loser of fetch_or still proceeds only
// after its own check below, and setting them twice is harmless.)
inline bool first_use_on_this_device(std::atomic<uint64_t> &mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return true;
  const uint64_t bit = 1ull << (dev & 63);
  return (mask.load(std::memory_order_acquire) & bit) == 0;
}
inline void mark_used_on_this_device(std::atomic<uint64_t> &mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
}

// sets the device for the current thread and restores the previous one on exit
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  explicit DeviceGuard(int dev) {
    SPL_HIP(hipGetDevice(&prev));
    if (prev != dev) { SPL_HIP(hipSetDevice(dev)); changed = true; }
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

constexpr uint32_t kMatrixMagic = 0x53504C4Du;  // "SPLM"

// column-blocked image of a row block (spmv_blocked.hip)
struct BlockedImage {
  int R = 0, w = 0;               // panel = R rows (R << w must fit 31 bits), column block = 2^w columns
  int64_t npanels = 0, ncb = 0;
  DBuf<int64_t> segptr;           // npanels*ncb + 1
  DBuf<int> key;                  // nnz: (local_row << w) | local_col
  DBuf<double> val;               // nnz
  DBuf<unsigned> arrive;          // rendezvous counter of the persistent kernel
  int fold = 1;                   // 1: ds_add_f64 fold, 0: shuffle fold (both reference order)
  int lockstep_waves = 16;        // wavefronts (= panels) per lockstep workgroup: 16 or 8; 0 = ablation kernel
};

// workgroup-wide column-sorted panels of a row block (spmv_panel.hip): the order-free SpMV mode
struct PanelImage {
  int P = 0, w = 0;               // panel = P rows (one workgroup's LDS), index block = 2^w columns (w <= 17)
  int64_t npanels = 0, nib = 0, nchunks = 0;
  DBuf<int> segc;                 // npanels*nib + 1 (+ padding): first 64-entry chunk of each segment
  DBuf<unsigned> key;             // (local_col << 15) | local_row, segments padded to whole chunks
  DBuf<double> val;
  DBuf<unsigned> arrive;          // rendezvous counter between generations
  int unroll = 10;                // chunks per wavefront and register set
  int kblocks = 2;                // index blocks per phase (barrier to barrier)
  int pair = 0;                   // 1: paired storage (chunk pairs interleaved, 8-byte key / 16-byte value loads)
  int ablate = 0;                 // timing-only ablation bits (SPL_PANEL_ABLATE with SPL_ALLOW_ABLATION=1)
  int nslices = 1;                // paired form: column slices per panel (8 = one per XCD; csrc/spmv_panel.hip)
  int ring = 0;                   // 1: ring form (loader wavefronts hand units to gather wavefronts through LDS slots)
  int ring_nl = 4, ring_depth = 6, ring_gather = 4, ring_slots = 1;  // loaders, units in flight per loader / per gatherer, slots per loader
};

// sliced-ELL image of a row block (spmv_sell.hip)
struct SellImage {
  int64_t nslices = 0, entries = 0;
  DBuf<int64_t> sliceoff;  // nslices + 1, in units of 64 entries
  DBuf<int> col;
  DBuf<double> val;
};

// A device-resident block of rows [row0, row0+nrows_local) of a sparse matrix
// with nrows_global x ncols entries, stored row-major (CSR, int32 column
// indices, int64 row pointers relative to the block).
struct Matrix {
  uint32_t magic = kMatrixMagic;
  int device = 0;
  int64_t nrows_global = 0, ncols = 0, row0 = 0, nrows_local = 0, nnz = 0;
  DBuf<int64_t> rowptr64;  // nrows_local+1, always present
  DBuf<int> rowptr;        // nrows_local+1, present iff nnz < 2^31
  DBuf<int> colidx;        // nnz
  DBuf<double> val;        // nnz
  int vw = 1;              // doubles per stored value: 1 Double, 2 packed Complex Double (csrc/spmv_z.hip)
  int variant = 0;
  int64_t max_row_len = 0;
  double new_line_fraction = -1.0;  // share of entries whose x line the previous row did not touch; < 0: not measured yet
  BlockedImage *blocked = nullptr;  // built on demand (spl_matrix_build_blocked / auto)
  SellImage *sell = nullptr;        // built on demand (spl_matrix_optimize on regular matrices)
  PanelImage *panel = nullptr;      // built on demand (spl_matrix_build_panel / spl_matrix_set_spmv_order)
  int blocked_unroll = 0;  // 0 = default; < 0 selects the ablation kernel
  int reserved_cus = -1;    // spl_matrix_set_reserved_cus: CUs left to a communication kernel; < 0: SPL_SPMV_RESERVED_CUS or 0
  bool order_free = false;  // spl_matrix_set_spmv_order: variant 0 may use the panel image (1e-10, order-free sums)
  Matrix() = default;
  Matrix(const Matrix &) = delete;
  Matrix &operator=(const Matrix &) = delete;
  ~Matrix() { delete blocked; delete sell; delete panel; }
};

inline Matrix *as_matrix(void *h) {
  Matrix *m = static_cast<Matrix *>(h);
  return (m && m->magic == kMatrixMagic) ? m : nullptr;
}

// ---- device primitives (scan.hip) -------------------------------------------------
// exclusive prefix sum of n counts; out has n+1 entries (out[n] = total).
void exclusive_scan_i32_to_i64(const int *d_in, int64_t *d_out, int64_t n, hipStream_t s);
void exclusive_scan_i64(const int64_t *d_in, int64_t *d_out, int64_t n, hipStream_t s);
// radix_sort.hip: LSD radix sort of 64-bit keys by their bits [0, nbits); returns the buffer holding the result
size_t radix_sort_u64_temp_bytes(int64_t n);
unsigned long long *radix_sort_u64(unsigned long long *keys, unsigned long long *alt, int64_t n, int nbits, void *temp,
                                   hipStream_t s);
void narrow_i64_to_i32(const int64_t *d_in, int *d_out, int64_t n, hipStream_t s);
void widen_i32_to_i64(const int *d_in, int64_t *d_out, int64_t n, hipStream_t s);

// ---- CSR image construction (convert.hip) ------------------------------------------
// validate a CSC/CSR pointer+index pair on the device: ptr[0]==0, monotone,
// ptr[nmajor]==nnz, 0 <= idx < nminor.  Returns SPL_OK or SPL_ERROR_invalid_matrix.
int validate_compressed(const int *d_ptr, const int *d_idx, int64_t nmajor, int64_t nminor,
                        int64_t nnz, hipStream_t s);
// Order-preserving transpose of compressed arrays (nmajor slices over nminor
// indices): out_ptr64[nminor+1], out_idx[nnz], out_val[nnz]; inside each new
// slice the old major index ascends (Sparse.hs:301-329).
void transpose_compressed(const int *d_ptr, const int *d_idx, const double *d_val, int64_t nmajor,
                          int64_t nminor, int64_t nnz, int64_t *out_ptr64, int *out_idx,
                          double *out_val, hipStream_t s);
// sort every segment [ptr[i], ptr[i+1]) of (key, val) pairs by key ascending
void segmented_sort_pairs(const int64_t *d_ptr64, int64_t nseg, int *d_key, double *d_val,
                          hipStream_t s);
// same, but segments longer than max_len are left untouched (already sorted)
void segmented_sort_pairs_capped(const int64_t *d_ptr64, int64_t nseg, int *d_key, double *d_val,
                                 int64_t max_len, hipStream_t s);
void segmented_sort_pairs64(const int64_t *d_ptr64, int64_t nseg, int64_t *d_key, double *d_val,
                            hipStream_t s);
void segmented_sort_pairs_u32(const int64_t *d_ptr64, int64_t nseg, unsigned *d_key, double *d_val,
                              hipStream_t s);
// finish a Matrix whose rowptr64/colidx/val are filled: int32 pointers, stats
// the `zi` wrapper's note on a `di` Numeric object: which row pairs of the embedding it swapped
void numeric_set_pair_swap(void *Numeric, std::vector<char> &&flags);
const std::vector<char> *numeric_pair_swap(void *Numeric);  // nullptr: none
// ... and, for a complex symmetric matrix, the unit-modulus diagonal D = diag(u_r) of the congruence D A D it embedded
// symmetrically (umfpack_zi.hip): n pairs (re, im)
void numeric_set_pair_unit(void *Numeric, std::vector<double> &&u);
const std::vector<double> *numeric_pair_unit(void *Numeric);  // nullptr: none
// symbolic analysis of the real embedding of a complex matrix, ordered on the complex pattern (umfpack.hip)
int symbolic_of_embedding(int n, const int *Ap, const int *Ai, const int *Ep, const int *Ei, void **Symbolic);
double symbolic_tree_flops(void *Symbolic);  // LU flops of the multifrontal tree of a `di` analysis; 0: band path
int numeric_of_embedding(const int *Ep, const int *Ei, const double *Ex, void *Symbolic, void **Numeric, int native = 0);
bool symbolic_has_complex_tree(void *Symbolic);  // the analysis kept the tree of the complex pattern (native complex fronts possible)
uint64_t pattern_hash(const int *Ai, int64_t nnz);
// rectangular matrices (umfpack.hip, Symbolic::rectangular): analysed and "factored" as far as the reference's binding
// can observe — statuses; solves return UMFPACK_ERROR_invalid_system as UMFPACK's do
int symbolic_rectangular(int n_row, int n_col, const int *Ap, const int *Ai, void **Symbolic);
bool symbolic_is_rectangular(void *Symbolic);
int numeric_rectangular_of(void *Symbolic, const int *Ap, const int *Ai, const std::vector<char> &nonzero, void **Numeric,
                           const double *re = nullptr, const double *im = nullptr, int vstride = 1);  // values: small matrices get their numerical rank
bool numeric_is_rectangular(void *Numeric);
void finalize_matrix(Matrix *m, hipStream_t s);
int spmv_cus(const Matrix *m);  // CUs the persistent SpMV images are laid out for: the device's minus the reserved ones
void measure_locality(Matrix *m, hipStream_t s);  // fills new_line_fraction on first call

// ---- assembly (assemble.hip) ------------------------------------------------------------
int compress_device(int nrows, int ncols, int64_t nnz, const int *d_rows, const int *d_cols,
                    const double *d_vals, int *d_newptr, DBuf<int> &out_idx, DBuf<double> &out_val,
                    int64_t *nnz_out, int64_t *bad, hipStream_t s, bool check_only = false);
bool columns_sorted(const int *d_ptr, const int *d_idx, int64_t ncols, hipStream_t s);
void lin_device(double alpha, const int *Ap, const int *Ai, const double *Ax, double beta, const int *Bp,
                const int *Bi, const double *Bx, int64_t ncols, DBuf<int64_t> &Cp, DBuf<int> &Ci,
                DBuf<double> &Cx, int64_t *nnzC, hipStream_t s);
void lin_device_z(const double alpha[2], const int *Ap, const int *Ai, const double *Az, const double beta[2],
                  const int *Bp, const int *Bi, const double *Bz, int64_t ncols, DBuf<int64_t> &Cp, DBuf<int> &Ci,
                  DBuf<double> &Cz, int64_t *nnzC, hipStream_t s);

void kronecker_device(int nrowsB, const int *Ap, const int *Ai, const double *Ax, int64_t ncolsA, const int *Bp,
                      const int *Bi, const double *Bx, int64_t ncolsB, DBuf<int64_t> &Cp, DBuf<int> &Ci,
                      DBuf<double> &Cx, int64_t *nnzC, hipStream_t s);
void take_diag_device(const int *Ap, const int *Ai, const double *Ax, int n, double *d, hipStream_t s);
// place blocks at (row_off, col_off) of a result with ncolsC columns; vw = doubles per value (1 real, 2 complex)
void blocks_assemble_device(int nblocks, const int *ncols_b, const int *const *d_Bp, const int *const *d_Bi,
                            const double *const *d_Bx, int vw, const int *row_off, const int *col_off, int64_t ncolsC,
                            DBuf<int64_t> &Cp, DBuf<int> &Ci, DBuf<double> &Cx, int64_t *nnzC, hipStream_t s);

// ---- multifrontal LU without interchanges (multifrontal.hip, mf_symbolic.hpp) ------------------
namespace mf {
struct Tree;
struct Factors;
struct LevelService;
}  // namespace mf
// level structures of the nested dissection's large regions on the GPU (nd_levels.hip); nullptr: no usable device
std::unique_ptr<mf::LevelService> make_gpu_level_service(int n, int64_t max_adj);
size_t mf_device_bytes(const mf::Tree &T, int zm = 1);  // zm = 2: complex fronts (two planes)
mf::Factors *mf_factor(std::shared_ptr<const mf::Tree> tree, const int *d_Ap, const int *d_Ai, const double *d_Ax,
                       const int *d_Rp, const int *d_Rj, const double *d_Rx, const int *d_perm, const int *d_inv,
                       hipStream_t s, bool symmetric = false, bool zfront = false, bool pivot = false,
                       const double *d_rscale = nullptr);  // pivot: threshold pivoting inside the diagonal blocks, on
                                                           // candidates scaled by d_rscale (new ordering of `tree`)
int mf_singular(const mf::Factors *F);
// the chain matrices of the large fronts' pivot blocks (mf_chain.hpp): out[0] bytes of device memory, out[1] milliseconds
// their construction took, out[2] pivots per block (0: no chains)
void mf_chain_info(const mf::Factors *F, double out[3]);
void mf_solve(const mf::Factors *F, int sys, double *d_c, int k, size_t stride, hipStream_t s);
void mf_free(mf::Factors *F);

// ---- SpGEMM (spgemm.hip) ------------------------------------------------------------------
void spgemm_device(int64_t nrowsA, int64_t ncolsA, const int *Ap, const int *Ai, const double *Ax,
                   int64_t ncolsB, const int *Bp, const int *Bi, const double *Bx, DBuf<int64_t> &Cp,
                   DBuf<int> &Ci, DBuf<double> &Cx, int64_t *nnzC, int64_t *products, hipStream_t s);

// mm on packed Complex Double (spgemm_z.hip): pattern from the real kernels, values in the reference's order
void spgemm_device_z(int64_t nrowsA, int64_t ncolsA, const int *Ap, const int *Ai, const double *Az, int64_t ncolsB,
                     const int *Bp, const int *Bi, const double *Bz, DBuf<int64_t> &Cp, DBuf<int> &Ci,
                     DBuf<double> &Cz, int64_t *nnzC, hipStream_t s);

// ---- blocked band LU without interchanges (band_nopiv.hip) --------------------------------------
bool band_is_column_dominant(int n, const int *d_Ap, const int *d_Ai, const double *d_Ax, hipStream_t s);
// leading dimension of the band storage: >= kl+ku+1, padded so that the stride between the same row
// of adjacent columns (ldab-1 doubles) is an odd multiple of 128 bytes, not a power of two
inline int band_nopiv_ldab(int kl, int ku) {
  int l = kl + ku + 1;
  while ((l - 1) % 32 != 16) ++l;
  return l;
}
size_t band_nopiv_inverse_elems(int n);  // doubles needed for the per-block inverse factors
int band_nopiv_factor(int n, int kl, int ku, int ldab, double *d_AB, double *d_invs, const int *d_Ap, const int *d_Ai,
                      const double *d_Ax, const int *d_inv, hipStream_t s);
constexpr int kSolveGroup = 8;  // right-hand sides the blocked solves take through the band at once
void band_nopiv_solve(int sys, int n, int kl, int ku, int ldab, const double *d_AB, const double *d_invs, double *d_c,
                      int nrhs, size_t stride, hipStream_t s);

// ---- synthetic generators (generate.hip) --------------------------------------------
void generate_synthetic(Matrix *m, int kind, int64_t n_or_m, int K, uint64_t seed, hipStream_t s);
void generate_rmat_coo(uint64_t seed, int scale, uint32_t ta, uint32_t tb, uint32_t tc, int64_t nedges,
                       int *d_rows, int *d_cols, double *d_vals, hipStream_t s);
void generate_vector(uint64_t seed, int64_t j0, int64_t j1, double *d_x, hipStream_t s);

// ---- SpMV (spmv.hip) --------------------------------------------------------------------
// y = A x (accumulate == 0) or y <- A x + y, rows of the block; enqueued on s
int launch_spmv(const Matrix *m, const double *d_x, double *d_y, int accumulate, hipStream_t s);
// C (nrows_local x k, row-major) = A B (+ C) for a row-major dense B (ncols x k): one pass over A
int launch_spmm(const Matrix *m, const double *d_B, double *d_C, int k, int accumulate, hipStream_t s);
// Complex Double (spmv_z.hip): d_x, d_y packed (re, im) pairs
int launch_spmv_z(const Matrix *m, const double *d_x, double *d_y, int accumulate, hipStream_t s);
void fill_positions(int64_t n, double *d_out, hipStream_t s);
void gather_complex_values(int64_t n, const double *d_pos, const double *d_in, double *d_out, hipStream_t s);
int64_t sell_padded_entries(const Matrix *m, hipStream_t s);
void build_sell_image(Matrix *m, hipStream_t s);
int launch_spmv_sell(const Matrix *m, const double *d_x, double *d_y, int accumulate, hipStream_t s);
constexpr int kNumSpmvVariants = 17;  // 16 = column-sorted workgroup panels (order-free);  // 15 = sliced-ELL image;  // 12-14: timing-only ablations of the CSR-stream kernel (wrong results)  // 0 auto, 1-6 CSR-stream shapes, 7 sub-wavefront, 8 column-blocked
void build_blocked_image(Matrix *m, int rows_per_panel, int w, hipStream_t s);
int launch_spmv_blocked(const Matrix *m, const double *d_x, double *d_y, int accumulate, int unroll,
                        hipStream_t s);
void build_panel_image(Matrix *m, int rows_per_panel, int w, int pair, hipStream_t s);
int launch_spmv_panel(const Matrix *m, const double *d_x, double *d_y, int accumulate, hipStream_t s);
size_t panel_ring_lds_bytes(int P, int nl, int slots);
int panel_ring_errors(const Matrix *m, hipStream_t s);
// choose the blocked image's shape for this matrix (0,0 = blocking would not help)
void choose_blocking(const Matrix *m, int *rows_per_panel, int *w, int *waves);
void choose_panels(const Matrix *m, int *rows_per_panel, int *w);
void choose_panels(const Matrix *m, int *rows_per_panel, int *w, int *nslices);
int spmv_kernel_in_use(const Matrix *m);
bool panels_pay(const Matrix *m);
bool panels_beat_stream(const Matrix *m);  // order-free mode: the panel image instead of the CSR-stream kernel

}  // namespace spl

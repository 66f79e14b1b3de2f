// abi.hip — the extern "C" surface declared in include/sparse_linear_hip.h.
// Host-side marshalling only; every numerical step runs in a HIP kernel.
#include <stdio.h>
#include <vector>

#include "common.hpp"

namespace spl {

static thread_local char g_last_error[512] = "";

void set_last_error(const char *where, hipError_t e) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", where, hipGetErrorString(e));
}
void set_last_error_text(const char *text) { snprintf(g_last_error, sizeof(g_last_error), "%s", text); }

namespace {

// run f(), mapping C++ failures to status codes
template <typename F>
int guarded(F &&f) {
  try {
    return f();
  } catch (const DeviceError &e) {
    return e.status;
  } catch (const std::bad_alloc &) {
    return SPL_ERROR_out_of_memory;
  } catch (...) {
    return SPL_ERROR_internal;
  }
}

int current_device() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_last_error_text("no HIP device visible");
    throw DeviceError{SPL_ERROR_device};
  }
  int dev = 0;
  SPL_HIP(hipGetDevice(&dev));
  return dev;
}

template <typename T>
void upload(DBuf<T> &dst, const T *src, size_t n, hipStream_t s) {
  dst.alloc(n);
  if (n) SPL_HIP(hipMemcpyAsync(dst.get(), src, n * sizeof(T), hipMemcpyHostToDevice, s));
}

int check_tuple(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax) {
  if (nrows < 0 || ncols < 0) return SPL_ERROR_n_nonpositive;
  if (!Ap) return SPL_ERROR_argument_missing;
  const int nnz = Ap[ncols];
  if (nnz < 0) return SPL_ERROR_invalid_matrix;
  if (nnz > 0 && (!Ai || !Ax)) return SPL_ERROR_argument_missing;
  return SPL_OK;
}

// Build the row-major image of a CSC 5-tuple on the current device.
// out receives rows [row0,row1) chosen as the part-th of nparts nnz-balanced blocks.
int build_from_csc(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, int part,
                   int nparts, Matrix **out) {
  int st = check_tuple(nrows, ncols, Ap, Ai, Ax);
  if (st != SPL_OK) return st;
  if (nparts < 1 || part < 0 || part >= nparts) return SPL_ERROR_argument_missing;
  const int64_t nnz = Ap[ncols];
  const int dev = current_device();
  hipStream_t s = nullptr;

  DBuf<int> dAp, dAi;
  DBuf<double> dAx;
  upload(dAp, Ap, (size_t)ncols + 1, s);
  upload(dAi, Ai, (size_t)nnz, s);
  upload(dAx, Ax, (size_t)nnz, s);
  st = validate_compressed(dAp.get(), dAi.get(), ncols, nrows, nnz, s);
  if (st != SPL_OK) return st;

  Matrix *full = new Matrix();
  full->device = dev;
  full->nrows_global = nrows;
  full->ncols = ncols;
  full->row0 = 0;
  full->nrows_local = nrows;
  full->nnz = nnz;
  try {
    full->rowptr64.alloc((size_t)nrows + 1);
    full->colidx.alloc((size_t)nnz);
    full->val.alloc((size_t)nnz);
    transpose_compressed(dAp.get(), dAi.get(), dAx.get(), ncols, nrows, nnz, full->rowptr64.get(),
                         full->colidx.get(), full->val.get(), s);
    if (nparts == 1) {
      finalize_matrix(full, s);
      *out = full;
      return SPL_OK;
    }
    // nnz-balanced contiguous row blocks: block p starts at the first row whose
    // pointer is >= nnz*p/nparts (identical on every rank: same input, same rule)
    std::vector<int64_t> hptr((size_t)nrows + 1);
    SPL_HIP(hipMemcpy(hptr.data(), full->rowptr64.get(), ((size_t)nrows + 1) * sizeof(int64_t),
                      hipMemcpyDeviceToHost));
    auto boundary = [&](int p) -> int64_t {
      if (p <= 0) return 0;
      if (p >= nparts) return nrows;
      const int64_t target = (int64_t)(((__int128)nnz * p) / nparts);
      int64_t lo = 0, hi = nrows;
      while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if (hptr[(size_t)mid] < target) lo = mid + 1; else hi = mid;
      }
      return lo;
    };
    const int64_t r0 = boundary(part), r1 = boundary(part + 1);
    Matrix *blk = new Matrix();
    blk->device = dev;
    blk->nrows_global = nrows;
    blk->ncols = ncols;
    blk->row0 = r0;
    blk->nrows_local = r1 - r0;
    const int64_t k0 = hptr[(size_t)r0], k1 = hptr[(size_t)r1];
    blk->nnz = k1 - k0;
    try {
      std::vector<int64_t> rel((size_t)(r1 - r0) + 1);
      for (int64_t i = 0; i <= r1 - r0; ++i) rel[(size_t)i] = hptr[(size_t)(r0 + i)] - k0;
      upload(blk->rowptr64, rel.data(), rel.size(), s);
      blk->colidx.alloc((size_t)blk->nnz);
      blk->val.alloc((size_t)blk->nnz);
      if (blk->nnz) {
        SPL_HIP(hipMemcpyAsync(blk->colidx.get(), full->colidx.get() + k0, (size_t)blk->nnz * sizeof(int),
                               hipMemcpyDeviceToDevice, s));
        SPL_HIP(hipMemcpyAsync(blk->val.get(), full->val.get() + k0, (size_t)blk->nnz * sizeof(double),
                               hipMemcpyDeviceToDevice, s));
      }
      SPL_HIP(hipStreamSynchronize(s));
      finalize_matrix(blk, s);
    } catch (...) {
      delete blk;
      throw;
    }
    delete full;
    *out = blk;
    return SPL_OK;
  } catch (...) {
    delete full;
    throw;
  }
}

// y (host) = A x (host) [+ y]
int host_spmv(Matrix *m, int xlen, const double *x, int ylen, double *y, int accumulate) {
  if ((int64_t)xlen != m->ncols) return SPL_ERROR_dimension_mismatch;     // Sparse.hs:438-441
  if ((int64_t)ylen != m->nrows_local) return SPL_ERROR_dimension_mismatch;  // Sparse.hs:442-445
  if ((xlen > 0 && !x) || (ylen > 0 && !y)) return SPL_ERROR_argument_missing;
  DeviceGuard g(m->device);
  hipStream_t s = nullptr;
  DBuf<double> dx, dy;
  const size_t vw = (size_t)m->vw;  // packed complex vectors carry two doubles per entry
  upload(dx, x, (size_t)xlen * vw, s);
  if (accumulate) upload(dy, y, (size_t)ylen * vw, s); else dy.alloc((size_t)ylen * vw);
  int st = launch_spmv(m, dx.get(), dy.get(), accumulate, s);
  if (st != SPL_OK) return st;
  if (ylen) SPL_HIP(hipMemcpyAsync(y, dy.get(), (size_t)ylen * vw * sizeof(double), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  return SPL_OK;
}

}  // namespace
}  // namespace spl

using namespace spl;

extern "C" {

const char *spl_status_string(int status) {
  switch (status) {
    case SPL_OK: return "OK";
    case SPL_WARNING_singular_matrix: return "warning: singular matrix";
    case SPL_ERROR_out_of_memory: return "out of memory";
    case SPL_ERROR_invalid_handle: return "invalid handle";
    case SPL_ERROR_argument_missing: return "argument missing or invalid";
    case SPL_ERROR_n_nonpositive: return "negative dimension";
    case SPL_ERROR_invalid_matrix: return "invalid matrix (pointers not monotone or index out of range)";
    case SPL_ERROR_dimension_mismatch: return "dimension mismatch";
    case SPL_ERROR_index_out_of_bounds: return "index out of bounds";
    case SPL_ERROR_index_overflow: return "result does not fit 32-bit indices";
    case SPL_ERROR_device: return "HIP device error";
    case SPL_ERROR_internal: return "internal error";
    default: return "unknown status";
  }
}

int spl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *spl_last_error(void) { return g_last_error; }

void spl_free(void *p) { free(p); }

unsigned long long spl_release_cached_memory(void) { return (unsigned long long)device_release_cached(); }

double spl_device_alloc_seconds(void) { return device_alloc_seconds(); }

int spl_matrix_create(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, void **H) {
  return spl_matrix_create_rowblock(nrows, ncols, Ap, Ai, Ax, 0, 1, H);
}

int spl_matrix_create_rowblock(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax,
                               int part, int nparts, void **H) {
  if (!H) return SPL_ERROR_argument_missing;
  *H = nullptr;
  return guarded([&]() -> int {
    Matrix *m = nullptr;
    int st = build_from_csc(nrows, ncols, Ap, Ai, Ax, part, nparts, &m);
    if (st == SPL_OK) *H = m;
    return st;
  });
}

// Complex Double: the CSC 5-tuple with packed (re, im) values (what the reference passes for its complex
// instance, Umfpack/Internal.hs:124-132).  The row-major image is built by transposing the PATTERN with the
// entry positions as payload and gathering the 16-byte values along the permutation.
int spl_matrix_create_z(int nrows, int ncols, const int *Ap, const int *Ai, const double *Az, void **H) {
  if (!H) return SPL_ERROR_argument_missing;
  *H = nullptr;
  return guarded([&]() -> int {
    int st = check_tuple(nrows, ncols, Ap, Ai, Az);
    if (st != SPL_OK) return st;
    const int64_t nnz = Ap[ncols];
    const int dev = current_device();
    hipStream_t s = nullptr;
    DBuf<int> dAp, dAi;
    DBuf<double> dAz, dpos, dperm;
    upload(dAp, Ap, (size_t)ncols + 1, s);
    upload(dAi, Ai, (size_t)nnz, s);
    upload(dAz, Az, (size_t)nnz * 2, s);
    st = validate_compressed(dAp.get(), dAi.get(), ncols, nrows, nnz, s);
    if (st != SPL_OK) return st;
    dpos.alloc((size_t)nnz);
    dperm.alloc((size_t)nnz);
    fill_positions(nnz, dpos.get(), s);
    std::unique_ptr<Matrix> m(new Matrix());
    m->device = dev;
    m->nrows_global = nrows;
    m->ncols = ncols;
    m->row0 = 0;
    m->nrows_local = nrows;
    m->nnz = nnz;
    m->vw = 2;
    m->rowptr64.alloc((size_t)nrows + 1);
    m->colidx.alloc((size_t)nnz);
    m->val.alloc((size_t)nnz * 2);
    transpose_compressed(dAp.get(), dAi.get(), dpos.get(), ncols, nrows, nnz, m->rowptr64.get(), m->colidx.get(),
                         dperm.get(), s);
    gather_complex_values(nnz, dperm.get(), dAz.get(), m->val.get(), s);
    SPL_HIP(hipStreamSynchronize(s));
    finalize_matrix(m.get(), s);
    *H = m.release();
    return SPL_OK;
  });
}

int spl_matrix_is_complex(void *H) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  return m->vw == 2 ? 1 : 0;
}

int spl_matrix_create_csr(int64_t nrows_global, int64_t ncols, int64_t row0, int64_t nrows_local,
                          const int *rowptr, const int *colidx, const double *val, void **H) {
  if (!H) return SPL_ERROR_argument_missing;
  *H = nullptr;
  if (nrows_global < 0 || ncols < 0 || nrows_local < 0 || row0 < 0) return SPL_ERROR_n_nonpositive;
  if (row0 + nrows_local > nrows_global || ncols > 0x7fffffffLL) return SPL_ERROR_argument_missing;
  if (!rowptr) return SPL_ERROR_argument_missing;
  const int64_t nnz = rowptr[nrows_local];
  if (nnz < 0) return SPL_ERROR_invalid_matrix;
  if (nnz > 0 && (!colidx || !val)) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    const int dev = current_device();
    hipStream_t s = nullptr;
    Matrix *m = new Matrix();
    try {
      m->device = dev;
      m->nrows_global = nrows_global;
      m->ncols = ncols;
      m->row0 = row0;
      m->nrows_local = nrows_local;
      m->nnz = nnz;
      DBuf<int> dptr;
      upload(dptr, rowptr, (size_t)nrows_local + 1, s);
      upload(m->colidx, colidx, (size_t)nnz, s);
      upload(m->val, val, (size_t)nnz, s);
      int st = validate_compressed(dptr.get(), m->colidx.get(), nrows_local, ncols, nnz, s);
      if (st != SPL_OK) { delete m; return st; }
      m->rowptr64.alloc((size_t)nrows_local + 1);
      widen_i32_to_i64(dptr.get(), m->rowptr64.get(), nrows_local + 1, s);
      finalize_matrix(m, s);
    } catch (...) {
      delete m;
      throw;
    }
    *H = m;
    return SPL_OK;
  });
}

int spl_matrix_create_synthetic(int kind, int64_t n_or_m, int K, uint64_t seed, int64_t row0,
                                int64_t row1, void **H) {
  if (!H) return SPL_ERROR_argument_missing;
  *H = nullptr;
  if (kind < 0 || kind > 3 || n_or_m <= 0) return SPL_ERROR_argument_missing;
  if (kind == 0 && (K < 1 || K > 64)) return SPL_ERROR_argument_missing;
  int64_t n = n_or_m;
  if (kind == 2) n = n_or_m * n_or_m;
  if (kind == 3) n = n_or_m * n_or_m * n_or_m;
  if (n > 0x7fffffffLL) return SPL_ERROR_index_overflow;
  if (row0 < 0 || row1 < row0 || row1 > n) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    const int dev = current_device();
    Matrix *m = new Matrix();
    try {
      m->device = dev;
      m->nrows_global = n;
      m->ncols = n;
      m->row0 = row0;
      m->nrows_local = row1 - row0;
      generate_synthetic(m, kind, n_or_m, K, seed, nullptr);
      finalize_matrix(m, nullptr);
    } catch (...) {
      delete m;
      throw;
    }
    *H = m;
    return SPL_OK;
  });
}

int spl_matrix_create_rmat(int scale, int edge_factor, double a, double b, double c, uint64_t seed,
                           void **H) {
  if (!H) return SPL_ERROR_argument_missing;
  *H = nullptr;
  if (scale < 1 || scale > 30 || edge_factor < 1 || a < 0 || b < 0 || c < 0 || a + b + c > 1.0)
    return SPL_ERROR_argument_missing;
  const int64_t n = 1LL << scale;
  const int64_t nedges = n * edge_factor;
  if (nedges >= 0x7fffffffLL) return SPL_ERROR_index_overflow;
  return guarded([&]() -> int {
    const int dev = current_device();
    hipStream_t s = nullptr;
    auto thr = [](double p) -> uint32_t {
      const double v = p * 4294967296.0;
      return v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v;
    };
    DBuf<int> dr((size_t)nedges), dc((size_t)nedges), dptr((size_t)n + 1), oidx;
    DBuf<double> dv((size_t)nedges), oval;
    generate_rmat_coo(seed, scale, thr(a), thr(a + b), thr(a + b + c), nedges, dr.get(), dc.get(), dv.get(), s);
    // row-major image: compress with the roles of rows and columns exchanged
    int64_t nz = 0, bad = -1;
    int st = compress_device((int)n, (int)n, nedges, dc.get(), dr.get(), dv.get(), dptr.get(), oidx, oval, &nz,
                             &bad, s);
    if (st != SPL_OK) return st;
    Matrix *m = new Matrix();
    try {
      m->device = dev;
      m->nrows_global = n;
      m->ncols = n;
      m->row0 = 0;
      m->nrows_local = n;
      m->nnz = nz;
      m->rowptr64.alloc((size_t)n + 1);
      widen_i32_to_i64(dptr.get(), m->rowptr64.get(), n + 1, s);
      m->colidx = std::move(oidx);
      m->val = std::move(oval);
      finalize_matrix(m, s);
    } catch (...) {
      delete m;
      throw;
    }
    *H = m;
    return SPL_OK;
  });
}

int spl_matrix_spgemm(void *HA, void *HB, void **HC, int64_t *products) {
  Matrix *A = as_matrix(HA), *B = as_matrix(HB);
  if (!A || !B) return SPL_ERROR_invalid_handle;
  if (!HC) return SPL_ERROR_argument_missing;
  *HC = nullptr;
  if (A->vw != 1 || B->vw != 1) return SPL_ERROR_argument_missing;  // complex products: through the host mirror
  if (A->ncols != B->nrows_global || B->row0 != 0 || B->nrows_local != B->nrows_global)
    return SPL_ERROR_dimension_mismatch;  // Sparse.hs:694 (B must be whole; A may be a row block)
  if (!A->rowptr.get() || !B->rowptr.get()) return SPL_ERROR_index_overflow;
  return guarded([&]() -> int {
    DeviceGuard g(A->device);
    hipStream_t s = nullptr;
    Matrix *C = new Matrix();
    try {
      C->device = A->device;
      C->nrows_global = A->nrows_global;
      C->ncols = B->ncols;
      C->row0 = A->row0;
      C->nrows_local = A->nrows_local;
      // rows of A*B = columns of (A*B)^T = B^T * A^T: the CSR arrays of B and A are the CSC
      // arrays of B^T and A^T, so the column-wise kernel runs on them unchanged
      spgemm_device(B->ncols, B->nrows_global, B->rowptr.get(), B->colidx.get(), B->val.get(), A->nrows_local,
                    A->rowptr.get(), A->colidx.get(), A->val.get(), C->rowptr64, C->colidx, C->val, &C->nnz,
                    products, s);
      finalize_matrix(C, s);
    } catch (...) {
      delete C;
      throw;
    }
    *HC = C;
    return SPL_OK;
  });
}


// ---- device-resident forms of lin / transpose / compress (round 3) -----------------------------------------
// The host 5-tuple entry points (spl_lin, spl_transpose, spl_compress) pay PCIe and marshalling for kernels that
// take a few milliseconds; these work handle to handle.  A handle holds the ROW-major image; `lin` merges along
// the major index whichever it is (Sparse.hs:401-431 applied to the transposes), so the same kernels serve.
int spl_matrix_lin(void *HA, const double alpha[2], void *HB, const double beta[2], void **HC) {
  Matrix *A = as_matrix(HA), *B = as_matrix(HB);
  if (!A || !B) return SPL_ERROR_invalid_handle;
  if (!HC || !alpha || !beta) return SPL_ERROR_argument_missing;
  *HC = nullptr;
  if (A->nrows_global != B->nrows_global || A->ncols != B->ncols || A->row0 != B->row0 || A->nrows_local != B->nrows_local)
    return SPL_ERROR_dimension_mismatch;  // Sparse.hs:408-409
  if (A->vw != B->vw || A->device != B->device) return SPL_ERROR_argument_missing;
  if (A->vw == 1 && (alpha[1] != 0.0 || beta[1] != 0.0)) return SPL_ERROR_argument_missing;  // complex scalars: spl_matrix_to_complex first
  if (!A->rowptr.get() || !B->rowptr.get()) return SPL_ERROR_index_overflow;
  return guarded([&]() -> int {
    DeviceGuard g(A->device);
    hipStream_t s = nullptr;
    std::unique_ptr<Matrix> C(new Matrix());
    C->device = A->device;
    C->nrows_global = A->nrows_global;
    C->ncols = A->ncols;
    C->row0 = A->row0;
    C->nrows_local = A->nrows_local;
    C->vw = A->vw;
    if (A->vw == 1)
      lin_device(alpha[0], A->rowptr.get(), A->colidx.get(), A->val.get(), beta[0], B->rowptr.get(), B->colidx.get(),
                 B->val.get(), A->nrows_local, C->rowptr64, C->colidx, C->val, &C->nnz, s);
    else
      lin_device_z(alpha, A->rowptr.get(), A->colidx.get(), A->val.get(), beta, B->rowptr.get(), B->colidx.get(),
                   B->val.get(), A->nrows_local, C->rowptr64, C->colidx, C->val, &C->nnz, s);
    finalize_matrix(C.get(), s);
    *HC = C.release();
    return SPL_OK;
  });
}

namespace {
__global__ __launch_bounds__(256) void promote_complex_kernel(const double *__restrict__ x, int64_t n, double *__restrict__ z) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { z[2 * i] = x[i]; z[2 * i + 1] = 0.0; }  // cmap (:+ 0)
}
}  // namespace

// the Complex Double handle of a real one: same pattern, values (x :+ 0)
int spl_matrix_to_complex(void *H, void **HZ) {
  Matrix *A = as_matrix(H);
  if (!A) return SPL_ERROR_invalid_handle;
  if (!HZ) return SPL_ERROR_argument_missing;
  *HZ = nullptr;
  if (A->vw != 1) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    DeviceGuard g(A->device);
    hipStream_t s = nullptr;
    std::unique_ptr<Matrix> C(new Matrix());
    C->device = A->device;
    C->nrows_global = A->nrows_global;
    C->ncols = A->ncols;
    C->row0 = A->row0;
    C->nrows_local = A->nrows_local;
    C->nnz = A->nnz;
    C->vw = 2;
    C->rowptr64.alloc((size_t)A->nrows_local + 1);
    SPL_HIP(hipMemcpyAsync(C->rowptr64.get(), A->rowptr64.get(), ((size_t)A->nrows_local + 1) * sizeof(int64_t),
                           hipMemcpyDeviceToDevice, s));
    C->colidx.alloc((size_t)A->nnz);
    C->val.alloc((size_t)A->nnz * 2);
    if (A->nnz) {
      SPL_HIP(hipMemcpyAsync(C->colidx.get(), A->colidx.get(), (size_t)A->nnz * sizeof(int), hipMemcpyDeviceToDevice, s));
      int64_t blocks = (A->nnz + 255) / 256;
      if (blocks > 65536) blocks = 65536;
      hipLaunchKernelGGL(promote_complex_kernel, dim3((unsigned)blocks), dim3(256), 0, s, A->val.get(), A->nnz, C->val.get());
    }
    finalize_matrix(C.get(), s);
    *HZ = C.release();
    return SPL_OK;
  });
}

// handle of the transpose (Sparse.hs:301-329 on the device, no host round trip); whole matrices only
int spl_matrix_transpose(void *H, void **HT) {
  Matrix *A = as_matrix(H);
  if (!A) return SPL_ERROR_invalid_handle;
  if (!HT) return SPL_ERROR_argument_missing;
  *HT = nullptr;
  if (A->vw != 1 || A->row0 != 0 || A->nrows_local != A->nrows_global) return SPL_ERROR_argument_missing;
  if (!A->rowptr.get()) return SPL_ERROR_index_overflow;
  return guarded([&]() -> int {
    DeviceGuard g(A->device);
    hipStream_t s = nullptr;
    std::unique_ptr<Matrix> C(new Matrix());
    C->device = A->device;
    C->nrows_global = C->nrows_local = A->ncols;
    C->ncols = A->nrows_global;
    C->nnz = A->nnz;
    C->rowptr64.alloc((size_t)A->ncols + 1);
    C->colidx.alloc((size_t)A->nnz);
    C->val.alloc((size_t)A->nnz);
    transpose_compressed(A->rowptr.get(), A->colidx.get(), A->val.get(), A->nrows_local, A->ncols, A->nnz,
                         C->rowptr64.get(), C->colidx.get(), C->val.get(), s);
    finalize_matrix(C.get(), s);
    *HT = C.release();
    return SPL_OK;
  });
}

// COO triples in device memory -> handle (compress / fromTriples, Sparse.hs:184-280: bounds checked rows first,
// then columns; duplicates summed in input order).  *bad receives the first offending position on
// SPL_ERROR_index_out_of_bounds (may be NULL).
int spl_matrix_compress_dev(int nrows, int ncols, int64_t ntriples, const int *d_rows, const int *d_cols,
                            const double *d_vals, void **H, int64_t *bad) {
  if (!H) return SPL_ERROR_argument_missing;
  *H = nullptr;
  if (nrows < 0 || ncols < 0) return SPL_ERROR_n_nonpositive;
  if (ntriples < 0 || (ntriples > 0 && (!d_rows || !d_cols || !d_vals))) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    const int dev = current_device();
    hipStream_t s = nullptr;
    // the row-major image of A is the column-major image of A^T: compress with the roles of rows and columns
    // exchanged.  The reference checks rows before columns (Sparse.hs:196-212): keep its order of complaints.
    {
      DBuf<int> none_i;
      DBuf<double> none_v;
      DBuf<int> probe((size_t)ncols + 1);
      int64_t nz = 0, where = -1;
      int st = compress_device(nrows, ncols, ntriples, d_rows, d_cols, d_vals, probe.get(), none_i, none_v, &nz, &where, s,
                               /*check_only=*/true);
      if (st != SPL_OK) { if (bad) *bad = where; return st; }
    }
    std::unique_ptr<Matrix> C(new Matrix());
    C->device = dev;
    C->nrows_global = C->nrows_local = nrows;
    C->ncols = ncols;
    DBuf<int> ptr32((size_t)nrows + 1);
    int64_t where = -1;
    int st = compress_device(ncols, nrows, ntriples, d_cols, d_rows, d_vals, ptr32.get(), C->colidx, C->val, &C->nnz, &where, s, false);
    if (st != SPL_OK) { if (bad) *bad = where; return st; }
    C->rowptr64.alloc((size_t)nrows + 1);
    widen_i32_to_i64(ptr32.get(), C->rowptr64.get(), (int64_t)nrows + 1, s);
    finalize_matrix(C.get(), s);
    *H = C.release();
    return SPL_OK;
  });
}

void spl_matrix_free(void **H) {
  if (!H || !*H) return;
  Matrix *m = as_matrix(*H);
  *H = nullptr;
  if (!m) return;
  int prev = -1;
  const bool have_prev = hipGetDevice(&prev) == hipSuccess;
  (void)hipSetDevice(m->device);  // may run on a finalizer thread with another current device
  m->magic = 0;
  delete m;
  if (have_prev) (void)hipSetDevice(prev);
}

int spl_matrix_info(void *H, int64_t info[8]) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if (!info) return SPL_ERROR_argument_missing;
  info[0] = m->nrows_global;
  info[1] = m->ncols;
  info[2] = m->row0;
  info[3] = m->nrows_local;
  info[4] = m->nnz;
  info[5] = m->device;
  const int kern = spmv_kernel_in_use(m);
  info[6] = kern == 8 ? m->blocked->R : kern == 16 ? m->panel->P : kern == 15 ? -64 : 0;  // -64: sliced-ELL image
  info[7] = kern == 8 ? m->blocked->w : kern == 16 ? m->panel->w : 0;
  return SPL_OK;
}

int spl_matrix_spmv_kernel(void *H) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  return spmv_kernel_in_use(m);
}

int spl_matrix_export_csr(void *H, int64_t *rowptr, int *colidx, double *val) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if (!rowptr || (m->nnz > 0 && (!colidx || !val))) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    DeviceGuard g(m->device);
    SPL_HIP(hipMemcpy(rowptr, m->rowptr64.get(), ((size_t)m->nrows_local + 1) * sizeof(int64_t),
                      hipMemcpyDeviceToHost));
    if (m->nnz) {
      SPL_HIP(hipMemcpy(colidx, m->colidx.get(), (size_t)m->nnz * sizeof(int), hipMemcpyDeviceToHost));
      SPL_HIP(hipMemcpy(val, m->val.get(), (size_t)m->nnz * (size_t)m->vw * sizeof(double), hipMemcpyDeviceToHost));
    }
    return SPL_OK;
  });
}

int spl_matrix_export_csr_rows(void *H, int64_t row0, int64_t row1, int64_t *rowptr, int64_t capacity, int *colidx,
                               double *val) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if (m->vw != 1) return SPL_ERROR_argument_missing;
  if (!rowptr || row0 < 0 || row1 < row0 || row1 > m->nrows_local || capacity < 0) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    DeviceGuard g(m->device);
    SPL_HIP(hipMemcpy(rowptr, m->rowptr64.get() + row0, ((size_t)(row1 - row0) + 1) * sizeof(int64_t),
                      hipMemcpyDeviceToHost));
    const int64_t a = rowptr[0], b = rowptr[row1 - row0];
    if (b - a > capacity || (b > a && (!colidx || !val))) return SPL_ERROR_argument_missing;
    if (b > a) {
      SPL_HIP(hipMemcpy(colidx, m->colidx.get() + a, (size_t)(b - a) * sizeof(int), hipMemcpyDeviceToHost));
      SPL_HIP(hipMemcpy(val, m->val.get() + a, (size_t)(b - a) * sizeof(double), hipMemcpyDeviceToHost));
    }
    return SPL_OK;
  });
}

int spl_matrix_export_csc(void *H, int64_t *colptr, int *rowidx, double *val) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if (m->vw != 1) return SPL_ERROR_argument_missing;  // complex handles: SpMV only
  if (!colptr || (m->nnz > 0 && (!rowidx || !val))) return SPL_ERROR_argument_missing;
  if (!m->rowptr.get()) return SPL_ERROR_index_overflow;
  return guarded([&]() -> int {
    DeviceGuard g(m->device);
    hipStream_t s = nullptr;
    DBuf<int64_t> dcp((size_t)m->ncols + 1);
    DBuf<int> dri((size_t)m->nnz);
    DBuf<double> dv((size_t)m->nnz);
    transpose_compressed(m->rowptr.get(), m->colidx.get(), m->val.get(), m->nrows_local, m->ncols, m->nnz,
                         dcp.get(), dri.get(), dv.get(), s);
    SPL_HIP(hipMemcpy(colptr, dcp.get(), ((size_t)m->ncols + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (m->nnz) {
      SPL_HIP(hipMemcpy(rowidx, dri.get(), (size_t)m->nnz * sizeof(int), hipMemcpyDeviceToHost));
      SPL_HIP(hipMemcpy(val, dv.get(), (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToHost));
    }
    return SPL_OK;
  });
}

int spl_matrix_mulv(void *H, int xlen, const double *x, double *y) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  return guarded([&]() -> int { return host_spmv(m, xlen, x, (int)m->nrows_local, y, 0); });
}

int spl_matrix_gaxpy(void *H, int xlen, const double *x, int ylen, double *y) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  return guarded([&]() -> int { return host_spmv(m, xlen, x, ylen, y, 1); });
}

int spl_matrix_spmv_dev(void *H, const double *d_x, double *d_y, int accumulate, void *stream) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if ((m->ncols > 0 && !d_x) || (m->nrows_local > 0 && !d_y)) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    DeviceGuard g(m->device);
    return launch_spmv(m, d_x, d_y, accumulate, as_stream(stream));
  });
}

int spl_matrix_set_variant(void *H, int variant) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if (m->vw != 1) return variant == 0 ? SPL_OK : SPL_ERROR_argument_missing;  // complex: the one native kernel
  if (variant < 0 || variant >= kNumSpmvVariants) return SPL_ERROR_argument_missing;
  if (variant >= 12 && variant <= 14) {
    // timing-only ablations of the CSR-stream kernel (they do NOT compute A x): refuse them unless
    // a profiling session asks explicitly
    const char *ok = getenv("SPL_ALLOW_ABLATION");
    if (!(ok && ok[0] == '1')) return SPL_ERROR_argument_missing;
  }
  if (variant == 8 && !m->blocked) {
    int st = spl_matrix_build_blocked(H, 0, 0, 0);
    if (st != SPL_OK) return st;
  }
  if (variant == 16 && !m->panel) {
    int st = spl_matrix_build_panel(H, 0, 0, 0, 0);
    if (st != SPL_OK) return st;
  }
  if (variant == 15 && !m->sell) {
    int st = guarded([&]() -> int {
      DeviceGuard g(m->device);
      build_sell_image(m, nullptr);
      return SPL_OK;
    });
    if (st != SPL_OK) return st;
  }
  m->variant = variant;
  return SPL_OK;
}

int spl_matrix_build_blocked(void *H, int rows_per_panel, int cols_log2, int unroll) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if (m->vw != 1) return SPL_ERROR_argument_missing;
  int waves = 16;
  if (rows_per_panel == 0 && cols_log2 == 0) {
    {
      int st = guarded([&]() -> int { DeviceGuard g(m->device); measure_locality(m, nullptr); return SPL_OK; });
      if (st != SPL_OK) return st;
    }
    choose_blocking(m, &rows_per_panel, &cols_log2, &waves);
    if (rows_per_panel == 0) { rows_per_panel = 1024; cols_log2 = 18; waves = 16; }  // explicit request: default shape
  }
  if (const char *ev = getenv("SPL_BLOCKED_LOCKSTEP")) waves = atoi(ev);
  if (waves != 0 && waves != 8 && waves != 4) waves = 16;
  if (rows_per_panel < 1 || rows_per_panel > 20480 || cols_log2 < 4 || cols_log2 > 26 ||
      ((int64_t)rows_per_panel << cols_log2) > 0x7fffffffLL)
    return SPL_ERROR_argument_missing;
  // the lockstep workgroup keeps `waves` panels in one CU's LDS (160 KiB)
  while (waves > 0 && (size_t)waves * (size_t)rows_per_panel * sizeof(double) > 160 * 1024)
    waves = waves == 16 ? 8 : waves == 8 ? 4 : 0;
  if (waves == 0 && (size_t)4 * (size_t)rows_per_panel * sizeof(double) > 160 * 1024)
    return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    DeviceGuard g(m->device);
    build_blocked_image(m, rows_per_panel, cols_log2, nullptr);
    if (unroll == 0) {
      // register pipeline depth: the smallest of {4,8,10,12} chunks that covers the mean segment
      // (longer segments take the un-pipelined tail loop; a deeper pipeline only streams entries
      // of the next segment it cannot use yet — measured at C2: 10 chunks 1.19 ms, 12 chunks 1.22 ms)
      const double seg = (double)m->nnz / (double)(m->blocked->npanels * m->blocked->ncb > 0
                                                       ? m->blocked->npanels * m->blocked->ncb : 1);
      const double want = seg / 64.0;
      unroll = want <= 4 ? 4 : want <= 8 ? 8 : want <= 10 ? 10 : 12;
    }
    m->blocked_unroll = unroll;
    m->blocked->lockstep_waves = waves;
    if (const char *ev = getenv("SPL_BLOCKED_FOLD")) m->blocked->fold = atoi(ev);
    return SPL_OK;
  });
}

int spl_matrix_build_panel(void *H, int rows_per_panel, int cols_log2, int unroll, int form) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if (m->vw != 1) return SPL_ERROR_argument_missing;
  const bool auto_shape = rows_per_panel == 0 && cols_log2 == 0, form_was_default = form == 0, unroll_was_default = unroll == 0;
  int nslices = 1;
  if (rows_per_panel == 0 && cols_log2 == 0) choose_panels(m, &rows_per_panel, &cols_log2, (form == 0 || form == 4 || form == 5) ? &nslices : nullptr);
  else if (const char *ev = getenv("SPL_PANEL_SLICES")) { if (form == 4 || form == 5) nslices = atoi(ev) >= 1 ? atoi(ev) : 1; }
  if (rows_per_panel < 1 || rows_per_panel > 20479 || cols_log2 < 4 || cols_log2 > 17)
    return SPL_ERROR_argument_missing;
  if (form != 0 && form != 1 && form != 2 && form != 4 && form != 5 && (form < 6 || form > 10)) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    DeviceGuard g(m->device);
    // form: 1 / 2 = one chunk per load, 1 / 2 index blocks per phase; 4 / 5 = paired storage (a pair of
    // chunks per 8-byte key / 16-byte value load; unroll then counts pairs), 1 / 2 index blocks per phase.
    // Default: paired, a phase's x window 2 MiB at most (measured on C2, tools/bench_spmv_variants.py:
    // paired 0.90 ms, one chunk per load 0.96 ms)
    // 6 / 7 = ring form on the paired storage (loader + gather wavefronts, csrc/spmv_panel.hip), 1 / 2 index
    // blocks per phase; unroll then counts the units a loader keeps in flight (0: 6)
    if (form == 0) form = cols_log2 >= 17 ? 5 : 4;
    const bool pair = form >= 4;
    const bool ring = form >= 6;
    // ring forms: 6 / 7 / 8 / 9 / 10 = 1 / 2 / 3 / 4 / 8 index blocks per phase
    const int kblocks = form == 10 ? 8 : form >= 8 ? form - 5 : (form == 2 || form == 5 || form == 7) ? 2 : 1;
    if (ring) {
      int nl = 4, slots = 1;
      if (const char *ev = getenv("SPL_PANEL_RING_NL")) nl = atoi(ev);
      if (const char *ev = getenv("SPL_PANEL_RING_SLOTS")) slots = atoi(ev);
      if (nl < 1 || slots < 1 || panel_ring_lds_bytes(rows_per_panel, nl, slots) > 160 * 1024) return SPL_ERROR_argument_missing;
    }
    build_panel_image(m, rows_per_panel, cols_log2, pair ? 1 : 0, nullptr);
    PanelImage *b = m->panel;
    b->kblocks = kblocks;
    b->nslices = pair && !ring ? nslices : 1;
    if (ring) {
      b->ring = 1;
      b->ring_depth = unroll > 0 ? unroll : 6;
      if (const char *ev = getenv("SPL_PANEL_RING_NL")) b->ring_nl = atoi(ev);
      if (const char *ev = getenv("SPL_PANEL_RING_SLOTS")) b->ring_slots = atoi(ev);
      if (const char *ev = getenv("SPL_PANEL_RING_GD")) b->ring_gather = atoi(ev);
      b->unroll = 0;
      b->ablate = 0;
      return SPL_OK;
    }
    if (unroll == 0) {
      // units (chunks or pairs) per wavefront and phase: the 16 wavefronts share a phase's units evenly and a
      // phase's length varies by a unit or two.  More loads in flight than the mean needs cost time (the
      // stream then queues in front of the gathers in the CU's L1): the nearest count, and the few longer
      // phases take the un-pipelined tail loop (measured on C2: 5 pairs 0.90 ms, 6 pairs 0.97 ms).
      const double per_wave = (double)b->nchunks * kblocks / (double)(b->npanels * b->nib > 0 ? b->npanels * b->nib : 1) / 16.0;
      if (pair) {
        unroll = (int)(per_wave / 2.0 + 0.5);
        unroll = unroll < 2 ? 2 : unroll > 6 ? 6 : unroll;
      } else {
        const int want = (int)(per_wave + 0.5);
        unroll = want <= 4 ? 4 : want <= 6 ? 6 : want <= 8 ? 8 : want <= 10 ? 10 : 12;
      }
    }
    const bool autotune = auto_shape && form_was_default && unroll_was_default && pair && m->nnz > (int64_t)1 << 22;
    if (const char *ev = getenv("SPL_PANEL_UNROLL")) unroll = atoi(ev);
    b->unroll = unroll;
    if (autotune && !getenv("SPL_PANEL_UNROLL") && !(getenv("SPL_PANEL_TUNE") && getenv("SPL_PANEL_TUNE")[0] == '0')) {
      // The register sets per wavefront and the index blocks per phase are worth 5-10 % either way and the
      // best pair sits next to the heuristic one (C2: 5 pairs, 2 blocks: 0.90 ms; 6 pairs: 0.97; 4: 0.98;
      // 3 pairs, 1 block: 0.96): time the neighbours once (the image is the same for all of them; about a
      // hundred launches on a scratch vector) and keep the fastest.  SPL_PANEL_TUNE=0 keeps the heuristic.
      DBuf<double> tx((size_t)m->ncols), ty((size_t)m->nrows_local);
      SPL_HIP(hipMemsetAsync(tx.get(), 0, (size_t)m->ncols * sizeof(double), nullptr));
      hipEvent_t e0, e1;
      SPL_HIP(hipEventCreate(&e0));
      SPL_HIP(hipEventCreate(&e1));
      struct Cand { int k, u; };
      std::vector<Cand> cands;
      for (int du = -1; du <= 1; ++du) {
        const int u2 = unroll + du;
        if (kblocks == 2 && u2 >= 3 && u2 <= 6) cands.push_back({2, u2});
        if (kblocks == 1 && u2 >= 2 && u2 <= 4) cands.push_back({1, u2});
      }
      if (kblocks == 2) {
        for (int u1 = (unroll + 1) / 2; u1 <= (unroll + 1) / 2 + 1; ++u1)
          if (u1 >= 2 && u1 <= 4) cands.push_back({1, u1});
      }
      float best = 0.f;
      int best_k = kblocks, best_u = unroll;
      float heur = 0.f;
      for (const Cand &c : cands) {
        b->kblocks = c.k;
        b->unroll = c.u;
        for (int w = 0; w < 2; ++w) (void)launch_spmv_panel(m, tx.get(), ty.get(), 0, nullptr);
        SPL_HIP(hipEventRecord(e0, nullptr));
        for (int r = 0; r < 8; ++r) (void)launch_spmv_panel(m, tx.get(), ty.get(), 0, nullptr);
        SPL_HIP(hipEventRecord(e1, nullptr));
        SPL_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        SPL_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (c.k == kblocks && c.u == unroll) heur = ms;
        if (best == 0.f || ms < best) { best = ms; best_k = c.k; best_u = c.u; }
      }
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      if (heur > 0.f && best > 0.99f * heur) { best_k = kblocks; best_u = unroll; }  // within noise: keep the heuristic
      if (getenv("SPL_PANEL_VERBOSE"))
        fprintf(stderr, "[panel] heuristic %d pairs x %d blocks: %.4f ms; chosen %d x %d: %.4f ms\n", unroll, kblocks,
                heur / 8.f, best_u, best_k, best / 8.f);
      b->kblocks = best_k;
      b->unroll = best_u;
    }
    b->ablate = 0;
    {
      const char *ok = getenv("SPL_ALLOW_ABLATION"), *ab = getenv("SPL_PANEL_ABLATE");
      if (ok && ok[0] == '1' && ab && !pair) b->ablate = atoi(ab) & 7;
    }
    return SPL_OK;
  });
}

int spl_matrix_panel_errors(void *H) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  int out = 0;
  int st = guarded([&]() -> int { DeviceGuard g(m->device); out = panel_ring_errors(m, nullptr); return SPL_OK; });
  return st != SPL_OK ? st : out;
}

int spl_matrix_set_spmv_order(void *H, int order) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if (order != SPL_ORDER_REFERENCE && order != SPL_ORDER_FREE) return SPL_ERROR_argument_missing;
  m->order_free = order == SPL_ORDER_FREE;
  return SPL_OK;
}

int spl_matrix_set_reserved_cus(void *H, int reserved) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if (reserved < 0) return SPL_ERROR_argument_missing;
  m->reserved_cus = reserved;
  return SPL_OK;
}

int spl_matrix_optimize(void *H) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if (m->vw != 1) return SPL_OK;  // complex: the CSR-stream kernel of spmv_z.hip is the one there is
  int R = 0, w = 0, waves = 16;
  {
    int st = guarded([&]() -> int { DeviceGuard g(m->device); measure_locality(m, nullptr); return SPL_OK; });
    if (st != SPL_OK) return st;
  }
  if (panels_beat_stream(m)) return spl_matrix_build_panel(H, 0, 0, 0, 0);
  choose_blocking(m, &R, &w, &waves);
  if (R == 0) {
    // no blocking needed.  Regular rows with column locality (banded, stencil): the sliced-ELL
    // image turns the x gathers into near-coalesced loads; build it when padding stays < 1/8.
    if (m->nnz > 0 && m->new_line_fraction < 0.5 && m->nrows_local >= 64 * 256) {
      return guarded([&]() -> int {
        DeviceGuard g(m->device);
        const int64_t padded = sell_padded_entries(m, nullptr);
        if (padded - m->nnz <= m->nnz / 8) build_sell_image(m, nullptr);
        return SPL_OK;
      });
    }
    return SPL_OK;  // the CSR-stream kernel is already the right one
  }
  // order-free sums allowed (spl_matrix_set_spmv_order): the column-sorted panels send fewer requests
  // to the L2 when a panel is tall enough for lines of x to meet several of its entries
  if (m->order_free && panels_pay(m)) return spl_matrix_build_panel(H, 0, 0, 0, 0);
  return spl_matrix_build_blocked(H, 0, 0, 0);  // 0,0: the same choice, including wavefronts per CU
}

int spl_vector_synthetic_dev(uint64_t seed, int64_t j0, int64_t j1, double *d_x, void *stream) {
  if (j1 < j0) return SPL_ERROR_argument_missing;
  if (j1 > j0 && !d_x) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    (void)current_device();
    generate_vector(seed, j0, j1, d_x, as_stream(stream));
    return SPL_OK;
  });
}

// ---- one-shot operations ------------------------------------------------------------------

int spl_gaxpy(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, int xlen,
              const double *x, int ylen, double *y) {
  // the reference checks dimensions before touching anything (Sparse.hs:438-445)
  if (nrows >= 0 && ncols >= 0 && (xlen != ncols || ylen != nrows)) return SPL_ERROR_dimension_mismatch;
  return guarded([&]() -> int {
    Matrix *m = nullptr;
    int st = build_from_csc(nrows, ncols, Ap, Ai, Ax, 0, 1, &m);
    if (st != SPL_OK) return st;
    try {
      st = host_spmv(m, xlen, x, ylen, y, 1);
    } catch (...) {
      delete m;
      throw;
    }
    delete m;
    return st;
  });
}

int spl_mulv(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, int xlen,
             const double *x, double *y) {
  if (nrows >= 0 && ncols >= 0 && xlen != ncols) return SPL_ERROR_dimension_mismatch;
  return guarded([&]() -> int {
    Matrix *m = nullptr;
    int st = build_from_csc(nrows, ncols, Ap, Ai, Ax, 0, 1, &m);
    if (st != SPL_OK) return st;
    try {
      st = host_spmv(m, xlen, x, nrows, y, 0);
    } catch (...) {
      delete m;
      throw;
    }
    delete m;
    return st;
  });
}

int spl_gaxpy_t(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, int xlen,
                const double *x, int ylen, double *y) {
  if (nrows >= 0 && ncols >= 0 && (xlen != nrows || ylen != ncols)) return SPL_ERROR_dimension_mismatch;
  int st = check_tuple(nrows, ncols, Ap, Ai, Ax);
  if (st != SPL_OK) return st;
  // the CSC arrays of A are the CSR arrays of A^T: no conversion at all
  void *h = nullptr;
  st = spl_matrix_create_csr(ncols, nrows, 0, ncols, Ap, Ai, Ax, &h);
  if (st != SPL_OK) return st;
  st = spl_matrix_gaxpy(h, xlen, x, ylen, y);
  spl_matrix_free(&h);
  return st;
}

int spl_mulm(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, int brows, int bcols,
             const double *B, double *C) {
  if (nrows >= 0 && ncols >= 0 && ncols != brows) return SPL_ERROR_dimension_mismatch;  // Sparse.hs:478
  if (bcols < 0) return SPL_ERROR_n_nonpositive;
  if ((int64_t)brows * bcols > 0 && !B) return SPL_ERROR_argument_missing;
  if ((int64_t)nrows * bcols > 0 && !C) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    Matrix *m = nullptr;
    int st = build_from_csc(nrows, ncols, Ap, Ai, Ax, 0, 1, &m);
    if (st != SPL_OK) return st;
    try {
      hipStream_t s = nullptr;
      const size_t nb = (size_t)brows * bcols, nc = (size_t)nrows * bcols;
      DBuf<double> dB, dC(nc);
      upload(dB, B, nb, s);
      // the reference runs one axpy_ per column of B (Sparse.hs:482-488); the fused kernel reads A
      // once for all columns and keeps each column's evaluation order
      st = launch_spmm(m, dB.get(), dC.get(), bcols, 0, s);
      if (st == SPL_OK && nc) SPL_HIP(hipMemcpyAsync(C, dC.get(), nc * sizeof(double), hipMemcpyDeviceToHost, s));
      SPL_HIP(hipStreamSynchronize(s));
    } catch (...) {
      delete m;
      throw;
    }
    delete m;
    return st;
  });
}

int spl_matrix_spmm_dev(void *H, const double *d_B, double *d_C, int k, int accumulate, void *stream) {
  Matrix *m = as_matrix(H);
  if (!m) return SPL_ERROR_invalid_handle;
  if (m->vw != 1) return SPL_ERROR_argument_missing;
  if (k < 0) return SPL_ERROR_n_nonpositive;
  if (k > 0 && ((m->ncols > 0 && !d_B) || (m->nrows_local > 0 && !d_C))) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    DeviceGuard g(m->device);
    return launch_spmm(m, d_B, d_C, k, accumulate, as_stream(stream));
  });
}

int spl_transpose(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, int *Tp, int *Ti,
                  double *Tx) {
  int st = check_tuple(nrows, ncols, Ap, Ai, Ax);
  if (st != SPL_OK) return st;
  if (!Tp) return SPL_ERROR_argument_missing;
  const int64_t nnz = Ap[ncols];
  if (nnz > 0 && (!Ti || !Tx)) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    (void)current_device();
    hipStream_t s = nullptr;
    DBuf<int> dAp, dAi, dTi((size_t)nnz), dTp((size_t)nrows + 1);
    DBuf<double> dAx, dTx((size_t)nnz);
    DBuf<int64_t> dTp64((size_t)nrows + 1);
    upload(dAp, Ap, (size_t)ncols + 1, s);
    upload(dAi, Ai, (size_t)nnz, s);
    upload(dAx, Ax, (size_t)nnz, s);
    int v = validate_compressed(dAp.get(), dAi.get(), ncols, nrows, nnz, s);
    if (v != SPL_OK) return v;
    transpose_compressed(dAp.get(), dAi.get(), dAx.get(), ncols, nrows, nnz, dTp64.get(), dTi.get(),
                         dTx.get(), s);
    narrow_i64_to_i32(dTp64.get(), dTp.get(), (int64_t)nrows + 1, s);
    SPL_HIP(hipMemcpyAsync(Tp, dTp.get(), ((size_t)nrows + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    if (nnz) {
      SPL_HIP(hipMemcpyAsync(Ti, dTi.get(), (size_t)nnz * sizeof(int), hipMemcpyDeviceToHost, s));
      SPL_HIP(hipMemcpyAsync(Tx, dTx.get(), (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    SPL_HIP(hipStreamSynchronize(s));
    return SPL_OK;
  });
}

}  // extern "C"

// ---- assembly / SpGEMM one-shots ------------------------------------------------------------

namespace {

struct DeviceCsc {
  DBuf<int> p, i;
  DBuf<double> x;
  int64_t nnz = 0;
};

// upload + validate a CSC 5-tuple; sort its columns if a caller violated the
// ascending-row invariant (the reference's SPA does not care about input order)
int upload_csc(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, DeviceCsc &d,
               hipStream_t s) {
  int st = check_tuple(nrows, ncols, Ap, Ai, Ax);
  if (st != SPL_OK) return st;
  d.nnz = Ap[ncols];
  upload(d.p, Ap, (size_t)ncols + 1, s);
  upload(d.i, Ai, (size_t)d.nnz, s);
  upload(d.x, Ax, (size_t)d.nnz, s);
  st = validate_compressed(d.p.get(), d.i.get(), ncols, nrows, d.nnz, s);
  if (st != SPL_OK) return st;
  if (!columns_sorted(d.p.get(), d.i.get(), ncols, s)) {
    DBuf<int64_t> p64((size_t)ncols + 1);
    widen_i32_to_i64(d.p.get(), p64.get(), (int64_t)ncols + 1, s);
    segmented_sort_pairs(p64.get(), ncols, d.i.get(), d.x.get(), s);
    SPL_HIP(hipStreamSynchronize(s));
  }
  return SPL_OK;
}

// copy a device result into malloc()'d host arrays (adoptable by `fromForeign False`)
int download_result(int64_t ncols, int64_t nnz, const int64_t *dCp64, const int *dCi, const double *dCx,
                    int **Cp, int **Ci, double **Cx, hipStream_t s) {
  if (nnz >= 0x7fffffffLL) return SPL_ERROR_index_overflow;
  int *hp = (int *)malloc(((size_t)ncols + 1) * sizeof(int));
  int *hi = (int *)malloc((size_t)(nnz ? nnz : 1) * sizeof(int));
  double *hx = (double *)malloc((size_t)(nnz ? nnz : 1) * sizeof(double));
  if (!hp || !hi || !hx) { free(hp); free(hi); free(hx); return SPL_ERROR_out_of_memory; }
  try {
    DBuf<int> dCp32((size_t)ncols + 1);
    narrow_i64_to_i32(dCp64, dCp32.get(), ncols + 1, s);
    SPL_HIP(hipMemcpyAsync(hp, dCp32.get(), ((size_t)ncols + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    if (nnz) {
      SPL_HIP(hipMemcpyAsync(hi, dCi, (size_t)nnz * sizeof(int), hipMemcpyDeviceToHost, s));
      SPL_HIP(hipMemcpyAsync(hx, dCx, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    SPL_HIP(hipStreamSynchronize(s));
  } catch (...) {
    free(hp); free(hi); free(hx);
    throw;
  }
  *Cp = hp; *Ci = hi; *Cx = hx;
  return SPL_OK;
}

}  // namespace

extern "C" {

int spl_spgemm(int nrowsA, int ncolsA, const int *Ap, const int *Ai, const double *Ax, int nrowsB,
               int ncolsB, const int *Bp, const int *Bi, const double *Bx, int *nrowsC, int *ncolsC,
               int **Cp, int **Ci, double **Cx) {
  if (!nrowsC || !ncolsC || !Cp || !Ci || !Cx) return SPL_ERROR_argument_missing;
  *Cp = nullptr; *Ci = nullptr; *Cx = nullptr;
  if (nrowsA >= 0 && ncolsA >= 0 && nrowsB >= 0 && ncolsB >= 0 && ncolsA != nrowsB)
    return SPL_ERROR_dimension_mismatch;  // Sparse.hs:694
  return guarded([&]() -> int {
    (void)current_device();
    hipStream_t s = nullptr;
    DeviceCsc A, B;
    int st = upload_csc(nrowsA, ncolsA, Ap, Ai, Ax, A, s);
    if (st != SPL_OK) return st;
    st = upload_csc(nrowsB, ncolsB, Bp, Bi, Bx, B, s);
    if (st != SPL_OK) return st;
    DBuf<int64_t> dCp;
    DBuf<int> dCi;
    DBuf<double> dCx;
    int64_t nnzC = 0;
    spgemm_device(nrowsA, ncolsA, A.p.get(), A.i.get(), A.x.get(), ncolsB, B.p.get(), B.i.get(), B.x.get(),
                  dCp, dCi, dCx, &nnzC, nullptr, s);
    st = download_result(ncolsB, nnzC, dCp.get(), dCi.get(), dCx.get(), Cp, Ci, Cx, s);
    if (st != SPL_OK) return st;
    *nrowsC = nrowsA;
    *ncolsC = ncolsB;
    return SPL_OK;
  });
}

// hcat / vcat / fromBlocks / fromBlocksDiag (Sparse.hs:500-595): nblocks CSC blocks placed at (row_off, col_off)
// of an nrowsC x ncolsC result; see include/sparse_linear_hip.h
int spl_assemble_blocks(int nblocks, const int *nrows, const int *ncols, const int *const *Ap, const int *const *Ai,
                        const double *const *Ax, int value_width, const int *row_off, const int *col_off, int nrowsC,
                        int ncolsC, int **Cp, int **Ci, double **Cx) {
  if (!Cp || !Ci || !Cx) return SPL_ERROR_argument_missing;
  *Cp = nullptr; *Ci = nullptr; *Cx = nullptr;
  if (nblocks < 0 || nrowsC < 0 || ncolsC < 0 || (value_width != 1 && value_width != 2)) return SPL_ERROR_argument_missing;
  if (nblocks > 0 && (!nrows || !ncols || !Ap || !Ai || !Ax || !row_off || !col_off)) return SPL_ERROR_argument_missing;
  for (int b = 0; b < nblocks; ++b) {
    if (nrows[b] < 0 || ncols[b] < 0 || row_off[b] < 0 || col_off[b] < 0 || (int64_t)row_off[b] + nrows[b] > nrowsC ||
        (int64_t)col_off[b] + ncols[b] > ncolsC)
      return SPL_ERROR_dimension_mismatch;
    if (!Ap[b]) return SPL_ERROR_argument_missing;
  }
  return guarded([&]() -> int {
    (void)current_device();
    hipStream_t s = nullptr;
    std::vector<DBuf<int>> dp((size_t)nblocks), di((size_t)nblocks);
    std::vector<DBuf<double>> dx((size_t)nblocks);
    std::vector<const int *> pp((size_t)nblocks), pi((size_t)nblocks);
    std::vector<const double *> px((size_t)nblocks);
    for (int b = 0; b < nblocks; ++b) {
      const int64_t nz = Ap[b][ncols[b]];
      if (nz < 0 || (nz > 0 && (!Ai[b] || !Ax[b]))) return SPL_ERROR_argument_missing;
      upload(dp[(size_t)b], Ap[b], (size_t)ncols[b] + 1, s);
      upload(di[(size_t)b], Ai[b], (size_t)nz, s);
      upload(dx[(size_t)b], Ax[b], (size_t)nz * (size_t)value_width, s);
      int st = validate_compressed(dp[(size_t)b].get(), di[(size_t)b].get(), ncols[b], nrows[b], nz, s);
      if (st != SPL_OK) return st;
      pp[(size_t)b] = dp[(size_t)b].get();
      pi[(size_t)b] = di[(size_t)b].get();
      px[(size_t)b] = dx[(size_t)b].get();
    }
    DBuf<int64_t> dCp;
    DBuf<int> dCi;
    DBuf<double> dCx;
    int64_t nnzC = 0;
    blocks_assemble_device(nblocks, ncols, pp.data(), pi.data(), px.data(), value_width, row_off, col_off, ncolsC, dCp, dCi,
                           dCx, &nnzC, s);
    if (nnzC >= 0x7fffffffLL) return SPL_ERROR_index_overflow;
    // download (values are value_width doubles per entry)
    int *hp = (int *)malloc(((size_t)ncolsC + 1) * sizeof(int));
    int *hi = (int *)malloc((size_t)(nnzC ? nnzC : 1) * sizeof(int));
    double *hx = (double *)malloc((size_t)(nnzC ? nnzC : 1) * (size_t)value_width * sizeof(double));
    if (!hp || !hi || !hx) { free(hp); free(hi); free(hx); return SPL_ERROR_out_of_memory; }
    try {
      DBuf<int> dCp32((size_t)ncolsC + 1);
      narrow_i64_to_i32(dCp.get(), dCp32.get(), (int64_t)ncolsC + 1, s);
      SPL_HIP(hipMemcpyAsync(hp, dCp32.get(), ((size_t)ncolsC + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
      if (nnzC) {
        SPL_HIP(hipMemcpyAsync(hi, dCi.get(), (size_t)nnzC * sizeof(int), hipMemcpyDeviceToHost, s));
        SPL_HIP(hipMemcpyAsync(hx, dCx.get(), (size_t)nnzC * (size_t)value_width * sizeof(double), hipMemcpyDeviceToHost, s));
      }
      SPL_HIP(hipStreamSynchronize(s));
    } catch (...) {
      free(hp); free(hi); free(hx);
      throw;
    }
    *Cp = hp; *Ci = hi; *Cx = hx;
    return SPL_OK;
  });
}

int spl_lin(double alpha, int nrowsA, int ncolsA, const int *Ap, const int *Ai, const double *Ax,
            double beta, int nrowsB, int ncolsB, const int *Bp, const int *Bi, const double *Bx,
            int *nrowsC, int *ncolsC, int **Cp, int **Ci, double **Cx) {
  if (!nrowsC || !ncolsC || !Cp || !Ci || !Cx) return SPL_ERROR_argument_missing;
  *Cp = nullptr; *Ci = nullptr; *Cx = nullptr;
  if (nrowsA >= 0 && ncolsA >= 0 && nrowsB >= 0 && ncolsB >= 0 && (nrowsA != nrowsB || ncolsA != ncolsB))
    return SPL_ERROR_dimension_mismatch;  // Sparse.hs:408-409
  return guarded([&]() -> int {
    (void)current_device();
    hipStream_t s = nullptr;
    DeviceCsc A, B;
    int st = upload_csc(nrowsA, ncolsA, Ap, Ai, Ax, A, s);
    if (st != SPL_OK) return st;
    st = upload_csc(nrowsB, ncolsB, Bp, Bi, Bx, B, s);
    if (st != SPL_OK) return st;
    DBuf<int64_t> dCp;
    DBuf<int> dCi;
    DBuf<double> dCx;
    int64_t nnzC = 0;
    lin_device(alpha, A.p.get(), A.i.get(), A.x.get(), beta, B.p.get(), B.i.get(), B.x.get(), ncolsA, dCp,
               dCi, dCx, &nnzC, s);
    st = download_result(ncolsA, nnzC, dCp.get(), dCi.get(), dCx.get(), Cp, Ci, Cx, s);
    if (st != SPL_OK) return st;
    *nrowsC = nrowsA;
    *ncolsC = ncolsA;
    return SPL_OK;
  });
}

// upload + validate a packed-complex CSC 5-tuple; unsorted columns are sorted by row with the values following
// their positions
static int upload_csc_z(int nrows, int ncols, const int *Ap, const int *Ai, const double *Az, DeviceCsc &d,
                        hipStream_t s) {
  int st = check_tuple(nrows, ncols, Ap, Ai, Az);
  if (st != SPL_OK) return st;
  d.nnz = Ap[ncols];
  upload(d.p, Ap, (size_t)ncols + 1, s);
  upload(d.i, Ai, (size_t)d.nnz, s);
  upload(d.x, Az, (size_t)d.nnz * 2, s);
  st = validate_compressed(d.p.get(), d.i.get(), ncols, nrows, d.nnz, s);
  if (st != SPL_OK) return st;
  if (!columns_sorted(d.p.get(), d.i.get(), ncols, s)) {
    DBuf<int64_t> p64((size_t)ncols + 1);
    DBuf<double> pos((size_t)d.nnz), sorted((size_t)d.nnz * 2);
    widen_i32_to_i64(d.p.get(), p64.get(), (int64_t)ncols + 1, s);
    fill_positions(d.nnz, pos.get(), s);
    segmented_sort_pairs(p64.get(), ncols, d.i.get(), pos.get(), s);
    gather_complex_values(d.nnz, pos.get(), d.x.get(), sorted.get(), s);
    SPL_HIP(hipStreamSynchronize(s));
    d.x = std::move(sorted);
  }
  return SPL_OK;
}

// copy a packed-complex device result into malloc()'d host arrays
static int download_result_z(int64_t ncols, int64_t nnz, const int64_t *dCp64, const int *dCi, const double *dCz,
                             int **Cp, int **Ci, double **Cz, hipStream_t s) {
  if (nnz >= 0x7fffffffLL) return SPL_ERROR_index_overflow;
  int *hp = (int *)malloc(((size_t)ncols + 1) * sizeof(int));
  int *hi = (int *)malloc((size_t)(nnz ? nnz : 1) * sizeof(int));
  double *hz = (double *)malloc((size_t)(nnz ? nnz : 1) * 2 * sizeof(double));
  if (!hp || !hi || !hz) { free(hp); free(hi); free(hz); return SPL_ERROR_out_of_memory; }
  try {
    DBuf<int> dCp32((size_t)ncols + 1);
    narrow_i64_to_i32(dCp64, dCp32.get(), ncols + 1, s);
    SPL_HIP(hipMemcpyAsync(hp, dCp32.get(), ((size_t)ncols + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    if (nnz) {
      SPL_HIP(hipMemcpyAsync(hi, dCi, (size_t)nnz * sizeof(int), hipMemcpyDeviceToHost, s));
      SPL_HIP(hipMemcpyAsync(hz, dCz, (size_t)nnz * 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    SPL_HIP(hipStreamSynchronize(s));
  } catch (...) {
    free(hp); free(hi); free(hz);
    throw;
  }
  *Cp = hp; *Ci = hi; *Cz = hz;
  return SPL_OK;
}

int spl_spgemm_z(int nrowsA, int ncolsA, const int *Ap, const int *Ai, const double *Az, int nrowsB, int ncolsB,
                 const int *Bp, const int *Bi, const double *Bz, int *nrowsC, int *ncolsC, int **Cp, int **Ci,
                 double **Cz) {
  if (!nrowsC || !ncolsC || !Cp || !Ci || !Cz) return SPL_ERROR_argument_missing;
  *Cp = nullptr; *Ci = nullptr; *Cz = nullptr;
  if (nrowsA >= 0 && ncolsA >= 0 && nrowsB >= 0 && ncolsB >= 0 && ncolsA != nrowsB)
    return SPL_ERROR_dimension_mismatch;  // Sparse.hs:694
  return guarded([&]() -> int {
    (void)current_device();
    hipStream_t s = nullptr;
    DeviceCsc A, B;
    int st = upload_csc_z(nrowsA, ncolsA, Ap, Ai, Az, A, s);
    if (st != SPL_OK) return st;
    st = upload_csc_z(nrowsB, ncolsB, Bp, Bi, Bz, B, s);
    if (st != SPL_OK) return st;
    DBuf<int64_t> dCp;
    DBuf<int> dCi;
    DBuf<double> dCz;
    int64_t nnzC = 0;
    spgemm_device_z(nrowsA, ncolsA, A.p.get(), A.i.get(), A.x.get(), ncolsB, B.p.get(), B.i.get(), B.x.get(), dCp, dCi,
                    dCz, &nnzC, s);
    st = download_result_z(ncolsB, nnzC, dCp.get(), dCi.get(), dCz.get(), Cp, Ci, Cz, s);
    if (st != SPL_OK) return st;
    *nrowsC = nrowsA;
    *ncolsC = ncolsB;
    return SPL_OK;
  });
}

int spl_lin_z(const double alpha[2], int nrowsA, int ncolsA, const int *Ap, const int *Ai, const double *Az,
              const double beta[2], int nrowsB, int ncolsB, const int *Bp, const int *Bi, const double *Bz,
              int *nrowsC, int *ncolsC, int **Cp, int **Ci, double **Cz) {
  if (!nrowsC || !ncolsC || !Cp || !Ci || !Cz || !alpha || !beta) return SPL_ERROR_argument_missing;
  *Cp = nullptr; *Ci = nullptr; *Cz = nullptr;
  if (nrowsA >= 0 && ncolsA >= 0 && nrowsB >= 0 && ncolsB >= 0 && (nrowsA != nrowsB || ncolsA != ncolsB))
    return SPL_ERROR_dimension_mismatch;  // Sparse.hs:408-409
  return guarded([&]() -> int {
    (void)current_device();
    hipStream_t s = nullptr;
    DeviceCsc A, B;
    int st = upload_csc_z(nrowsA, ncolsA, Ap, Ai, Az, A, s);
    if (st != SPL_OK) return st;
    st = upload_csc_z(nrowsB, ncolsB, Bp, Bi, Bz, B, s);
    if (st != SPL_OK) return st;
    DBuf<int64_t> dCp;
    DBuf<int> dCi;
    DBuf<double> dCz;
    int64_t nnzC = 0;
    lin_device_z(alpha, A.p.get(), A.i.get(), A.x.get(), beta, B.p.get(), B.i.get(), B.x.get(), ncolsA, dCp, dCi,
                 dCz, &nnzC, s);
    st = download_result_z(ncolsA, nnzC, dCp.get(), dCi.get(), dCz.get(), Cp, Ci, Cz, s);
    if (st != SPL_OK) return st;
    *nrowsC = nrowsA;
    *ncolsC = ncolsA;
    return SPL_OK;
  });
}

int spl_kronecker(int nrowsA, int ncolsA, const int *Ap, const int *Ai, const double *Ax, int nrowsB,
                  int ncolsB, const int *Bp, const int *Bi, const double *Bx, int *nrowsC, int *ncolsC,
                  int **Cp, int **Ci, double **Cx) {
  if (!nrowsC || !ncolsC || !Cp || !Ci || !Cx) return SPL_ERROR_argument_missing;
  *Cp = nullptr; *Ci = nullptr; *Cx = nullptr;
  if (nrowsA < 0 || ncolsA < 0 || nrowsB < 0 || ncolsB < 0) return SPL_ERROR_n_nonpositive;
  if ((int64_t)nrowsA * nrowsB >= 0x7fffffffLL || (int64_t)ncolsA * ncolsB >= 0x7fffffffLL)
    return SPL_ERROR_index_overflow;  // the seam is int32 (Foreign.hs:24-28)
  return guarded([&]() -> int {
    (void)current_device();
    hipStream_t s = nullptr;
    DeviceCsc A, B;
    int st = upload_csc(nrowsA, ncolsA, Ap, Ai, Ax, A, s);
    if (st != SPL_OK) return st;
    st = upload_csc(nrowsB, ncolsB, Bp, Bi, Bx, B, s);
    if (st != SPL_OK) return st;
    DBuf<int64_t> dCp;
    DBuf<int> dCi;
    DBuf<double> dCx;
    int64_t nnzC = 0;
    kronecker_device(nrowsB, A.p.get(), A.i.get(), A.x.get(), ncolsA, B.p.get(), B.i.get(), B.x.get(), ncolsB,
                     dCp, dCi, dCx, &nnzC, s);
    st = download_result((int64_t)ncolsA * ncolsB, nnzC, dCp.get(), dCi.get(), dCx.get(), Cp, Ci, Cx, s);
    if (st != SPL_OK) return st;
    *nrowsC = nrowsA * nrowsB;
    *ncolsC = ncolsA * ncolsB;
    return SPL_OK;
  });
}

int spl_take_diag(int nrows, int ncols, const int *Ap, const int *Ai, const double *Ax, double *d) {
  if (nrows < 0 || ncols < 0) return SPL_ERROR_n_nonpositive;
  const int n = nrows < ncols ? nrows : ncols;
  if (n > 0 && !d) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    (void)current_device();
    hipStream_t s = nullptr;
    DeviceCsc A;
    int st = upload_csc(nrows, ncols, Ap, Ai, Ax, A, s);
    if (st != SPL_OK) return st;
    if (n == 0) return SPL_OK;
    DBuf<double> dd((size_t)n);
    take_diag_device(A.p.get(), A.i.get(), A.x.get(), n, dd.get(), s);
    SPL_HIP(hipMemcpyAsync(d, dd.get(), (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    SPL_HIP(hipStreamSynchronize(s));
    SPL_HIP(hipGetLastError());
    return SPL_OK;
  });
}

int spl_compress(int nrows, int ncols, int64_t nnz, const int *rows, const int *cols, const double *vals,
                 int *Ap, int **Ai, double **Ax, int64_t *bad) {
  if (!Ap || !Ai || !Ax) return SPL_ERROR_argument_missing;
  *Ai = nullptr; *Ax = nullptr;
  if (nrows < 0 || ncols < 0 || nnz < 0) return SPL_ERROR_n_nonpositive;
  if (nnz >= 0x7fffffffLL) return SPL_ERROR_index_overflow;
  if (nnz > 0 && (!rows || !cols || !vals)) return SPL_ERROR_argument_missing;
  return guarded([&]() -> int {
    (void)current_device();
    hipStream_t s = nullptr;
    DBuf<int> dr, dc, dptr((size_t)ncols + 1), oidx;
    DBuf<double> dv, oval;
    upload(dr, rows, (size_t)nnz, s);
    upload(dc, cols, (size_t)nnz, s);
    upload(dv, vals, (size_t)nnz, s);
    int64_t nz = 0;
    int st = compress_device(nrows, ncols, nnz, dr.get(), dc.get(), dv.get(), dptr.get(), oidx, oval, &nz,
                             bad, s);
    if (st != SPL_OK) return st;
    int *hi = (int *)malloc((size_t)(nz ? nz : 1) * sizeof(int));
    double *hx = (double *)malloc((size_t)(nz ? nz : 1) * sizeof(double));
    if (!hi || !hx) { free(hi); free(hx); return SPL_ERROR_out_of_memory; }
    try {
      SPL_HIP(hipMemcpyAsync(Ap, dptr.get(), ((size_t)ncols + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
      if (nz) {
        SPL_HIP(hipMemcpyAsync(hi, oidx.get(), (size_t)nz * sizeof(int), hipMemcpyDeviceToHost, s));
        SPL_HIP(hipMemcpyAsync(hx, oval.get(), (size_t)nz * sizeof(double), hipMemcpyDeviceToHost, s));
      }
      SPL_HIP(hipStreamSynchronize(s));
    } catch (...) {
      free(hi); free(hx);
      throw;
    }
    *Ai = hi;
    *Ax = hx;
    return SPL_OK;
  });
}

}  // extern "C"

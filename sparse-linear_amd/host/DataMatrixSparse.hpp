// DataMatrixSparse.hpp — compiled-language host side above the C ABI.
//
// The reference's host language is Haskell (no GHC in this pipeline), so this
// header mirrors the operator interface of `Data.Matrix.Sparse` and
// `Numeric.LinearAlgebra.Umfpack` in C++: same names, argument order and error behaviour as
//   sparse-linear/src/Data/Matrix/Sparse.hs:13-30   (export list)
//   suitesparse/src/Numeric/LinearAlgebra/Umfpack.hs:5-13
// Every hot-path operation crosses include/sparse_linear_hip.h / include/umfpack_hip.h into
// the gfx950 kernels, narrowing 64-bit indices to int32 at the seam exactly as
// `withConstMatrix` does (Data/Matrix/Sparse/Foreign.hs:24-41).  Header-only; link with
// -lsparse_linear_hip.  Errors that the reference raises with errorWithStackTrace / error
// are thrown as std::runtime_error with the same message prefix.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <algorithm>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/sparse_linear_hip.h"
#include "../../include/umfpack_hip.h"

namespace Data { namespace Matrix { namespace Sparse {

using Int = int64_t;  // Haskell Int

// data Matrix v a = Matrix { ncols, nrows, pointers, indices, values }   (Sparse.hs:67-76)
struct Matrix {
  Int ncols = 0, nrows = 0;
  std::vector<Int> pointers;   // ncols + 1, last = number of stored entries
  std::vector<Int> indices;    // row ids, strictly ascending inside a column
  std::vector<double> values;
  bool operator==(const Matrix &o) const {  // deriving Eq (Sparse.hs:78)
    return ncols == o.ncols && nrows == o.nrows && pointers == o.pointers && indices == o.indices &&
           values == o.values;
  }
};

inline Int nonZero(const Matrix &m) { return m.pointers.back(); }  // Sparse.hs:115-117

namespace detail {
inline void oops(const std::string &fn, const std::string &msg) { throw std::runtime_error(fn + ": " + msg); }
inline void check(const char *where, int status) {  // Umfpack.hs:67: negative status is fatal
  if (status < 0) oops(where, std::string(spl_status_string(status)) + " [" + spl_last_error() + "]");
}
// withConstMatrix (Foreign.hs:24-41): fresh CInt copies of pointers / indices
struct Const {
  int nrows, ncols;
  std::vector<int> p, i;
  const double *x;
  explicit Const(const Matrix &m)
      : nrows((int)m.nrows), ncols((int)m.ncols), p(m.pointers.begin(), m.pointers.end()),
        i(m.indices.begin(), m.indices.end()), x(m.values.data()) {}
};
// fromForeign False (Foreign.hs:43-88): adopt malloc()'d arrays, then free() them
inline Matrix adopt(int nr, int nc, int *cp, int *ci, double *cx) {
  Matrix c;
  c.nrows = nr;
  c.ncols = nc;
  c.pointers.assign(cp, cp + nc + 1);
  const Int nz = c.pointers.back();
  c.indices.assign(ci, ci + nz);
  c.values.assign(cx, cx + nz);
  spl_free(cp); spl_free(ci); spl_free(cx);
  return c;
}
}  // namespace detail

// compress (Sparse.hs:184-255)
inline Matrix compress(Int nrows, Int ncols, const std::vector<Int> &rows, const std::vector<Int> &cols,
                       const std::vector<double> &vals) {
  if (rows.size() != cols.size()) detail::oops("compress", "row and column array lengths differ");
  if (rows.size() != vals.size()) detail::oops("compress", "row and value array lengths differ");
  std::vector<int> r(rows.size()), c(cols.size());
  for (size_t k = 0; k < rows.size(); ++k) {
    r[k] = (rows[k] < INT32_MIN || rows[k] > INT32_MAX) ? -1 : (int)rows[k];
    c[k] = (cols[k] < INT32_MIN || cols[k] > INT32_MAX) ? -1 : (int)cols[k];
  }
  std::vector<int> ap((size_t)ncols + 1);
  int *ai = nullptr;
  double *ax = nullptr;
  int64_t bad = -1;
  const int st = spl_compress((int)nrows, (int)ncols, (int64_t)r.size(), r.data(), c.data(), vals.data(), ap.data(),
                              &ai, &ax, &bad);
  if (st == SPL_ERROR_index_out_of_bounds) {
    const bool row_bad = !(rows[(size_t)bad] >= 0 && rows[(size_t)bad] < nrows);
    detail::oops("compress", std::string(row_bad ? "row" : "column") + " index out of bounds at " + std::to_string(bad));
  }
  detail::check("compress", st);
  Matrix m;
  m.nrows = nrows;
  m.ncols = ncols;
  m.pointers.assign(ap.begin(), ap.end());
  m.indices.assign(ai, ai + ap[(size_t)ncols]);
  m.values.assign(ax, ax + ap[(size_t)ncols]);
  spl_free(ai); spl_free(ax);
  return m;
}

// fromTriples / (><) (Sparse.hs:357-369)
inline Matrix fromTriples(Int nr, Int nc, const std::vector<std::tuple<Int, Int, double>> &triples) {
  std::vector<Int> r, c;
  std::vector<double> v;
  for (const auto &t : triples) { r.push_back(std::get<0>(t)); c.push_back(std::get<1>(t)); v.push_back(std::get<2>(t)); }
  return compress(nr, nc, r, c, v);
}

// diag / ident / zeros (Sparse.hs:650-677): host-side index plumbing
inline Matrix diag(const std::vector<double> &values) {
  Matrix m;
  m.ncols = m.nrows = (Int)values.size();
  m.pointers.resize(values.size() + 1);
  m.indices.resize(values.size());
  for (size_t k = 0; k <= values.size(); ++k) m.pointers[k] = (Int)k;
  for (size_t k = 0; k < values.size(); ++k) m.indices[k] = (Int)k;
  m.values = values;
  return m;
}
inline Matrix ident(Int n) { return diag(std::vector<double>((size_t)n, 1.0)); }
inline Matrix zeros(Int nrows, Int ncols) {
  Matrix m;
  m.nrows = nrows;
  m.ncols = ncols;
  m.pointers.assign((size_t)ncols + 1, 0);
  return m;
}

// transpose (Sparse.hs:301-329)
inline Matrix transpose(const Matrix &a) {
  detail::Const c(a);
  const Int nz = nonZero(a);
  std::vector<int> tp((size_t)a.nrows + 1), ti((size_t)(nz ? nz : 1));
  Matrix t;
  t.nrows = a.ncols;
  t.ncols = a.nrows;
  t.values.resize((size_t)(nz ? nz : 1));
  detail::check("transpose", spl_transpose(c.nrows, c.ncols, c.p.data(), c.i.data(), c.x, tp.data(), ti.data(),
                                           t.values.data()));
  t.values.resize((size_t)nz);
  t.pointers.assign(tp.begin(), tp.end());
  t.indices.assign(ti.begin(), ti.begin() + nz);
  return t;
}

// axpy_ (Sparse.hs:433-453): in place  ys <- A xs + ys
inline void axpy_(const Matrix &a, const std::vector<double> &xs, std::vector<double> &ys) {
  if ((Int)xs.size() != a.ncols)
    detail::oops("axpy_", "column dimension " + std::to_string(a.ncols) + " does not match operand dimension " +
                              std::to_string(xs.size()));
  if ((Int)ys.size() != a.nrows)
    detail::oops("axpy_", "row dimension " + std::to_string(a.nrows) + " does not match result dimension " +
                              std::to_string(ys.size()));
  detail::Const c(a);
  detail::check("axpy_", spl_gaxpy(c.nrows, c.ncols, c.p.data(), c.i.data(), c.x, (int)xs.size(), xs.data(),
                                   (int)ys.size(), ys.data()));
}
// axpy (Sparse.hs:455-462), mulV (Sparse.hs:464-471)
inline std::vector<double> axpy(const Matrix &a, const std::vector<double> &x, std::vector<double> y) {
  axpy_(a, x, y);
  return y;
}
inline std::vector<double> mulV(const Matrix &a, const std::vector<double> &x) {
  return axpy(a, x, std::vector<double>((size_t)a.nrows, 0.0));
}

// mm / (*) (Sparse.hs:691-702)
inline Matrix mm(const Matrix &a, const Matrix &b) {
  if (a.ncols != b.nrows) detail::oops("mm", "inner dimension mismatch");
  detail::Const ca(a), cb(b);
  int nr = 0, nc = 0, *cp = nullptr, *ci = nullptr;
  double *cx = nullptr;
  detail::check("mm", spl_spgemm(ca.nrows, ca.ncols, ca.p.data(), ca.i.data(), ca.x, cb.nrows, cb.ncols, cb.p.data(),
                                 cb.i.data(), cb.x, &nr, &nc, &cp, &ci, &cx));
  return detail::adopt(nr, nc, cp, ci, cx);
}
inline Matrix operator*(const Matrix &a, const Matrix &b) { return mm(a, b); }

// lin (Sparse.hs:426-431), (+), (-) (Sparse.hs:107-108)
inline Matrix lin(double alpha, const Matrix &a, double beta, const Matrix &b) {
  if (a.nrows != b.nrows) detail::oops("glin", "row number mismatch");
  if (a.ncols != b.ncols) detail::oops("glin", "column number mismatch");
  detail::Const ca(a), cb(b);
  int nr = 0, nc = 0, *cp = nullptr, *ci = nullptr;
  double *cx = nullptr;
  detail::check("glin", spl_lin(alpha, ca.nrows, ca.ncols, ca.p.data(), ca.i.data(), ca.x, beta, cb.nrows, cb.ncols,
                                cb.p.data(), cb.i.data(), cb.x, &nr, &nc, &cp, &ci, &cx));
  return detail::adopt(nr, nc, cp, ci, cx);
}
inline Matrix operator+(const Matrix &a, const Matrix &b) { return lin(1.0, a, 1.0, b); }
inline Matrix operator-(const Matrix &a, const Matrix &b) { return lin(1.0, a, -1.0, b); }

// kronecker (Sparse.hs:597-634)
inline Matrix kronecker(const Matrix &a, const Matrix &b) {
  detail::Const ca(a), cb(b);
  int nr = 0, nc = 0, *cp = nullptr, *ci = nullptr;
  double *cx = nullptr;
  detail::check("kronecker", spl_kronecker(ca.nrows, ca.ncols, ca.p.data(), ca.i.data(), ca.x, cb.nrows, cb.ncols,
                                           cb.p.data(), cb.i.data(), cb.x, &nr, &nc, &cp, &ci, &cx));
  return detail::adopt(nr, nc, cp, ci, cx);
}

// takeDiag (Sparse.hs:636-648)
inline std::vector<double> takeDiag(const Matrix &a) {
  detail::Const ca(a);
  std::vector<double> d((size_t)std::min(a.nrows, a.ncols), 0.0);
  detail::check("takeDiag", spl_take_diag(ca.nrows, ca.ncols, ca.p.data(), ca.i.data(), ca.x, d.data()));
  return d;
}

// hcat / vcat / fromBlocks (Sparse.hs:504-587): one device assembly (spl_assemble_blocks); nullptr = Nothing
namespace detail {
inline Matrix assemble(const std::vector<const Matrix *> &blocks, const std::vector<int> &row_off,
                       const std::vector<int> &col_off, Int nrows, Int ncols) {
  const int k = (int)blocks.size();
  std::vector<Const> cs;
  cs.reserve((size_t)k);
  for (const Matrix *m : blocks) cs.emplace_back(*m);
  std::vector<int> nr((size_t)k), nc((size_t)k);
  std::vector<const int *> ap((size_t)k), ai((size_t)k);
  std::vector<const double *> ax((size_t)k);
  for (int b = 0; b < k; ++b) {
    nr[(size_t)b] = cs[(size_t)b].nrows; nc[(size_t)b] = cs[(size_t)b].ncols;
    ap[(size_t)b] = cs[(size_t)b].p.data(); ai[(size_t)b] = cs[(size_t)b].i.data(); ax[(size_t)b] = cs[(size_t)b].x;
  }
  int *cp = nullptr, *ci = nullptr;
  double *cx = nullptr;
  check("assemble", spl_assemble_blocks(k, nr.data(), nc.data(), ap.data(), ai.data(), ax.data(), 1, row_off.data(),
                                        col_off.data(), (int)nrows, (int)ncols, &cp, &ci, &cx));
  return adopt((int)nrows, (int)ncols, cp, ci, cx);
}
}  // namespace detail

inline Matrix hcat(const std::vector<Matrix> &mats) {
  if (mats.empty()) detail::oops("hcat", "empty list");
  std::vector<const Matrix *> bs;
  std::vector<int> ro, co;
  Int w = 0;
  for (const Matrix &m : mats) {
    if (m.nrows != mats[0].nrows) detail::oops("hcat", "nrows mismatch");
    bs.push_back(&m); ro.push_back(0); co.push_back((int)w);
    w += m.ncols;
  }
  return detail::assemble(bs, ro, co, mats[0].nrows, w);
}
inline Matrix hjoin(const Matrix &a, const Matrix &b) { return hcat({a, b}); }

inline Matrix vcat(const std::vector<Matrix> &mats) {
  if (mats.empty()) detail::oops("vcat", "empty list");
  std::vector<const Matrix *> bs;
  std::vector<int> ro, co;
  Int h = 0;
  for (const Matrix &m : mats) {
    if (m.ncols != mats[0].ncols) detail::oops("vcat", "ncols mismatch");
    bs.push_back(&m); ro.push_back((int)h); co.push_back(0);
    h += m.nrows;
  }
  return detail::assemble(bs, ro, co, h, mats[0].ncols);
}
inline Matrix vjoin(const Matrix &a, const Matrix &b) { return vcat({a, b}); }

// fromBlocks (Sparse.hs:563-587): rectangular list of lists, nullptr = a zero block
inline Matrix fromBlocks(const std::vector<std::vector<const Matrix *>> &rows) {
  size_t ncb = 0;
  for (const auto &r : rows) ncb = std::max(ncb, r.size());
  std::vector<Int> heights(rows.size(), -1), widths(ncb, -1);
  for (size_t r = 0; r < rows.size(); ++r)
    for (size_t c = 0; c < rows[r].size(); ++c)
      if (const Matrix *m = rows[r][c]) {
        if (heights[r] >= 0 && heights[r] != m->nrows) detail::oops("fromBlocks", "incompatible heights");
        if (widths[c] >= 0 && widths[c] != m->ncols) detail::oops("fromBlocks", "incompatible widths");
        heights[r] = m->nrows;
        widths[c] = m->ncols;
      }
  for (Int h : heights) if (h < 0) detail::oops("fromBlocks", "underspecified heights");
  for (Int w : widths) if (w < 0) detail::oops("fromBlocks", "underspecified widths");
  std::vector<Int> roff(rows.size() + 1, 0), coff(ncb + 1, 0);
  for (size_t r = 0; r < rows.size(); ++r) roff[r + 1] = roff[r] + heights[r];
  for (size_t c = 0; c < ncb; ++c) coff[c + 1] = coff[c] + widths[c];
  for (const auto &r : rows) if (coff[r.size()] != coff[rows[0].size()]) detail::oops("vcat", "ncols mismatch");
  std::vector<const Matrix *> bs;
  std::vector<int> ro, co;
  for (size_t r = 0; r < rows.size(); ++r)
    for (size_t c = 0; c < rows[r].size(); ++c)
      if (rows[r][c]) { bs.push_back(rows[r][c]); ro.push_back((int)roff[r]); co.push_back((int)coff[c]); }
  return detail::assemble(bs, ro, co, roff.back(), coff[rows[0].size()]);
}

}}}  // namespace Data::Matrix::Sparse

namespace Numeric { namespace LinearAlgebra { namespace Umfpack {

using Data::Matrix::Sparse::Matrix;
namespace detail = Data::Matrix::Sparse::detail;

enum UmfpackMode { UmfpackNormal = 0, UmfpackTrans = 1 };  // Umfpack.hs:85

// ForeignPtr with the UMFPACK free function as finalizer on the void* cell (Umfpack.hs:63-65)
struct Analysis {
  std::shared_ptr<void *> fsym;
};
struct Factors {
  std::shared_ptr<void *> fnum;
  int status = 0;
  // what the object holds now (spl_umfpack_stats): path, n, kl, ku, device bytes, flops, fronts
  struct Stats {
    int path = -1, n = 0, kl = 0, ku = 0, fronts = 0;
    double device_bytes = 0, flops = 0;
  };
  Stats stats() const {
    double out[8];
    Stats s;
    if (spl_umfpack_stats(*fnum, out) != 0) detail::oops("Factors::stats", "invalid Numeric object");
    s.path = (int)out[0]; s.n = (int)out[1]; s.kl = (int)out[2]; s.ku = (int)out[3];
    s.device_bytes = out[4]; s.flops = out[5]; s.fronts = (int)out[6];
    return s;
  }
};

// gives the device blocks the library keeps for reuse back to the driver (spl_release_cached_memory)
inline unsigned long long releaseCachedMemory() { return spl_release_cached_memory(); }

inline Analysis analyze(const Matrix &mat) {  // Umfpack.hs:60-69
  detail::Const c(mat);
  auto cell = std::shared_ptr<void *>(new void *(nullptr), [](void **p) { umfpack_di_free_symbolic(p); delete p; });
  const int st = umfpack_di_symbolic(c.nrows, c.ncols, c.p.data(), c.i.data(), c.x, cell.get(), nullptr, nullptr);
  umfpack_di_report_status(nullptr, st);
  if (st < 0) detail::oops("analyze", "umfpack_symbolic failed");
  return Analysis{cell};
}

inline Factors factor(const Matrix &mat, const Analysis &an) {  // Umfpack.hs:71-83
  detail::Const c(mat);
  auto cell = std::shared_ptr<void *>(new void *(nullptr), [](void **p) { umfpack_di_free_numeric(p); delete p; });
  const int st = umfpack_di_numeric(c.p.data(), c.i.data(), c.x, *an.fsym, cell.get(), nullptr, nullptr);
  umfpack_di_report_status(nullptr, st);
  if (st < 0) detail::oops("factor", "umfpack_numeric failed");
  return Factors{cell, st};
}

inline std::vector<double> linearSolve_(const Factors &fact, UmfpackMode mode, const Matrix &mat,
                                        const std::vector<double> &b) {  // Umfpack.hs:87-102
  detail::Const c(mat);
  std::vector<double> soln((size_t)mat.ncols, 0.0);  // MV.replicate ncols 0
  const int st = umfpack_di_solve((int)mode, c.p.data(), c.i.data(), c.x, soln.data(), b.data(), *fact.fnum, nullptr,
                                  nullptr);
  umfpack_di_report_status(nullptr, st);
  if (st < 0) detail::oops("linearSolve_", "umfpack_solve failed");
  return soln;
}

// map (linearSolve_ fact mode mat) bs (Umfpack.hs:103-108) in one pass of all right-hand sides
// through the factors (spl_umfpack_di_solve_many)
inline std::vector<std::vector<double>> linearSolveMany_(const Factors &fact, UmfpackMode mode, const Matrix &mat,
                                                         const std::vector<std::vector<double>> &bs) {
  detail::Const c(mat);
  const size_t k = bs.size(), n = (size_t)mat.ncols;
  std::vector<double> B(k * (size_t)mat.nrows), X(k * n, 0.0);
  for (size_t j = 0; j < k; ++j) {
    if (bs[j].size() != (size_t)mat.nrows) detail::oops("linearSolveMany_", "right-hand side of the wrong length");
    std::copy(bs[j].begin(), bs[j].end(), B.begin() + j * (size_t)mat.nrows);
  }
  const int st = spl_umfpack_di_solve_many((int)mode, c.p.data(), c.i.data(), c.x, (int)k, X.data(), B.data(), *fact.fnum);
  umfpack_di_report_status(nullptr, st);
  if (st < 0) detail::oops("linearSolveMany_", "umfpack_solve failed");
  std::vector<std::vector<double>> xs(k);
  for (size_t j = 0; j < k; ++j) xs[j].assign(X.begin() + j * n, X.begin() + (j + 1) * n);
  return xs;
}

inline std::vector<std::vector<double>> linearSolve(const Matrix &mat, const std::vector<std::vector<double>> &bs) {
  const Factors fact = factor(mat, analyze(mat));  // Umfpack.hs:38-46
  return linearSolveMany_(fact, UmfpackNormal, mat, bs);
}

// (<\>) (Umfpack.hs:48-50)
inline std::vector<double> solve(const Matrix &mat, const std::vector<double> &b) { return linearSolve(mat, {b})[0]; }

}}}  // namespace Numeric::LinearAlgebra::Umfpack

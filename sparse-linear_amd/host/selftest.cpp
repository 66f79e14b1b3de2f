// selftest.cpp — the reference's hspec items (sparse-linear/tests/Sparse.hs, suitesparse/tests/
// test-umfpack.hs) and closed-form known answers driven through the C++ host mirror, i.e. a
// compiled-language client of the C ABI with no Python and no torch in the process.
// Exit code 0 = all passed.  Needs a GPU (the backend has no CPU path).
#include <cmath>
#include <cstdio>

#include "DataMatrixSparse.hpp"

using namespace Data::Matrix::Sparse;
namespace U = Numeric::LinearAlgebra::Umfpack;

static int failures = 0;
#define EXPECT(cond)                                                      \
  do {                                                                    \
    if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
  } while (0)

int main() {
  if (spl_device_count() < 1) { std::printf("selftest needs a GPU\n"); return 2; }
  // mulV: ident `mulV` v == v  (tests/Sparse.hs:41-47)
  {
    std::vector<double> v;
    for (int i = 0; i < 1000; ++i) v.push_back(0.5 * i - 3);
    EXPECT(mulV(ident(1000), v) == v);
  }
  // fromTriples sums duplicates, keeps explicit zeros, sorts rows (Sparse.hs:184-280)
  {
    Matrix m = fromTriples(3, 2, {{1, 0, 2.0}, {1, 0, -2.0}, {0, 1, 0.0}, {2, 1, 5.0}, {0, 1, 0.0}});
    EXPECT((m.pointers == std::vector<Int>{0, 1, 3}));
    EXPECT((m.indices == std::vector<Int>{1, 0, 2}));
    EXPECT((m.values == std::vector<double>{0.0, 0.0, 5.0}));
    bool threw = false;
    try { fromTriples(2, 2, {{0, 0, 1.0}, {2, 0, 1.0}}); } catch (const std::runtime_error &e) {
      threw = std::string(e.what()).find("compress: row index out of bounds at 1") == 0;
    }
    EXPECT(threw);
  }
  // transpose . diag == diag (tests/Sparse.hs:56-59); ctrans fixtures (:61-69, real ones)
  {
    Matrix d = diag({1, 2, 3, 4, 5});
    EXPECT(transpose(d) == d);
    Matrix h = fromTriples(2, 2, {{0, 0, 2}, {0, 1, -1}, {1, 0, -1}, {1, 1, 2}});
    EXPECT(transpose(h) == h);
  }
  // mul: ident * a == a, a * ident == a, associativity, cancellation keeps a stored zero (:75-102)
  {
    Matrix a = fromTriples(4, 3, {{0, 0, 1}, {3, 0, -2}, {1, 1, 4}, {2, 2, 5}, {0, 2, 7}});
    Matrix b = fromTriples(3, 5, {{0, 0, 2}, {1, 1, 3}, {2, 4, -1}, {0, 3, 1}, {2, 0, 6}});
    Matrix c = fromTriples(5, 2, {{0, 0, 1}, {4, 1, 2}, {3, 1, -3}, {1, 0, 5}});
    EXPECT(ident(4) * a == a);
    EXPECT(a * ident(3) == a);
    EXPECT((a * b) * c == a * (b * c));
    Matrix p = fromTriples(1, 2, {{0, 0, 1.0}, {0, 1, -1.0}}), q = fromTriples(2, 1, {{0, 0, 1.0}, {1, 0, 1.0}});
    Matrix z = p * q;
    EXPECT((z.pointers == std::vector<Int>{0, 1}) && z.values == std::vector<double>{0.0});
    bool threw = false;
    try { a * a; } catch (const std::runtime_error &e) { threw = std::string(e.what()) == "mm: inner dimension mismatch"; }
    EXPECT(threw);
  }
  // addition: a + zeros == a; a - a == 0-valued a; commutativity (:49-54,147-165)
  {
    Matrix a = fromTriples(3, 3, {{0, 0, 1}, {2, 1, 3}, {1, 2, -4}});
    Matrix b = fromTriples(3, 3, {{0, 0, 5}, {1, 2, 7}, {2, 2, 1}});
    EXPECT(a + zeros(3, 3) == a);
    Matrix z = a - a;
    EXPECT(z.pointers == a.pointers && z.indices == a.indices && z.values == std::vector<double>(3, 0.0));
    EXPECT(a + b == b + a);
  }
  // config C1: 2-D Poisson on a 100 x 100 grid, A = I (x) T + T (x) I assembled from its stencil;
  // closed forms: nnz = 5 n^2 - 4 n, A*1 = 4 - #neighbours, A v = lambda v
  {
    const Int n = 100;
    std::vector<std::tuple<Int, Int, double>> t;
    for (Int iy = 0; iy < n; ++iy)
      for (Int ix = 0; ix < n; ++ix) {
        const Int r = iy * n + ix;
        t.emplace_back(r, r, 4.0);
        if (ix > 0) t.emplace_back(r, r - 1, -1.0);
        if (ix + 1 < n) t.emplace_back(r, r + 1, -1.0);
        if (iy > 0) t.emplace_back(r, r - n, -1.0);
        if (iy + 1 < n) t.emplace_back(r, r + n, -1.0);
      }
    Matrix A = fromTriples(n * n, n * n, t);
    EXPECT(nonZero(A) == 5 * n * n - 4 * n);
    std::vector<double> y = mulV(A, std::vector<double>((size_t)(n * n), 1.0));
    bool ok = true;
    for (Int iy = 0; iy < n; ++iy)
      for (Int ix = 0; ix < n; ++ix) {
        const double nb = (ix > 0) + (ix + 1 < n) + (iy > 0) + (iy + 1 < n);
        ok &= y[(size_t)(iy * n + ix)] == 4.0 - nb;
      }
    EXPECT(ok);
    const double h = M_PI / (n + 1), lam = 4 - 4 * std::cos(h);
    std::vector<double> v((size_t)(n * n));
    for (Int iy = 0; iy < n; ++iy)
      for (Int ix = 0; ix < n; ++ix) v[(size_t)(iy * n + ix)] = std::sin((iy + 1) * h) * std::sin((ix + 1) * h);
    std::vector<double> av = mulV(A, v);
    double err = 0;
    for (size_t k = 0; k < v.size(); ++k) err = std::fmax(err, std::fabs(av[k] - lam * v[k]));
    EXPECT(err < 1e-13);
    // solve with a manufactured solution (Umfpack.hs:48-50)
    std::vector<double> b = mulV(A, v), x = U::solve(A, b);
    double rel = 0;
    for (size_t k = 0; k < v.size(); ++k) rel = std::fmax(rel, std::fabs(x[k] - v[k]) / std::fabs(v[k] + x[k]));
    EXPECT(rel < 1e-10);  // closeness predicate of feast/tests/test-feast.hs:17-19
  }
  // the reference's own construction of that matrix: kronecker (ident n) T + kronecker T (ident n);
  // takeDiag; several right-hand sides in one batched solve
  {
    const Int n = 30;
    std::vector<std::tuple<Int, Int, double>> t;
    for (Int i = 0; i < n; ++i) {
      t.emplace_back(i, i, 2.0);
      if (i > 0) t.emplace_back(i, i - 1, -1.0);
      if (i + 1 < n) t.emplace_back(i, i + 1, -1.0);
    }
    Matrix T = fromTriples(n, n, t);
    Matrix A = kronecker(ident(n), T) + kronecker(T, ident(n));
    EXPECT(A.nrows == n * n && nonZero(A) == 5 * n * n - 4 * n);
    EXPECT(takeDiag(A) == std::vector<double>((size_t)(n * n), 4.0));
    EXPECT(kronecker(ident(3), ident(4)) == ident(12));
    // hcat / vcat / fromBlocks on the device: [[I2, 0], [0, I3]] == I5; (A | A)^T == A^T over A^T
    const Matrix i2 = ident(2), i3 = ident(3);
    EXPECT(fromBlocks({{&i2, nullptr}, {nullptr, &i3}}) == ident(5));
    EXPECT(transpose(hjoin(A, A)) == vjoin(transpose(A), transpose(A)));
    EXPECT(nonZero(vcat({A, A, A})) == 3 * nonZero(A));
    std::vector<std::vector<double>> xs, bs;
    for (int j = 0; j < 3; ++j) {
      std::vector<double> x((size_t)(n * n));
      for (size_t k = 0; k < x.size(); ++k) x[k] = 1.0 + 0.001 * (double)((k * (size_t)(j + 7)) % 97);
      xs.push_back(x);
      bs.push_back(mulV(A, x));
    }
    std::vector<std::vector<double>> got = U::linearSolve(A, bs);
    double rel = 0;
    for (size_t j = 0; j < xs.size(); ++j)
      for (size_t k = 0; k < xs[j].size(); ++k) rel = std::fmax(rel, std::fabs(got[j][k] - xs[j][k]) / std::fabs(xs[j][k]));
    EXPECT(rel < 1e-10);
  }
  // ident <\> v == v, exactly (suitesparse/tests/test-umfpack.hs:16-19)
  {
    std::vector<double> v{3.5, -1.25, 1e6, 0.0, 7.0};
    EXPECT(U::solve(ident(5), v) == v);
    U::Factors f = U::factor(ident(5), U::analyze(ident(5)));
    EXPECT(U::linearSolve_(f, U::UmfpackTrans, ident(5), v) == v);
    const U::Factors::Stats st = f.stats();
    EXPECT(st.n == 5 && st.path == 1 && st.kl == 0 && st.ku == 0 && st.fronts == 0 && st.device_bytes > 0);
  }
  // work arrays of 256 KiB and more are kept for reuse (at most 2 GiB of them); releasing gives them back once
  EXPECT(U::releaseCachedMemory() <= (size_t)2 << 30);
  EXPECT(U::releaseCachedMemory() == 0);
  std::printf(failures ? "selftest: %d FAILED\n" : "selftest: all passed\n", failures);
  return failures ? 1 : 0;
}

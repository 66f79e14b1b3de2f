// auction_transversal_probe.hip — NOT part of the library (round 3, negative result kept for the record): the
// maximum-product transversal of static pivoting (static_pivot.hpp) by a Jacobi auction with epsilon-scaling on the GPU.
// Wired into factor_static_pivot it finished the easy cases in milliseconds (kkt2d, 78 125 unknowns: 64 rounds, 4 ms —
// where the host's greedy start is as fast) and did NOT finish the ones the host is slow on within 2e6 rounds / 30 s
// (perm2d and tiny2d at 4e5 unknowns: price wars at eps = 0.02 over a cost range of 14 .. 27; the host's shortest
// augmenting paths take 13 .. 19 s there and stay the algorithm in use).  DESIGN.md section 4.6.
//
// static_pivot.hpp finds the transversal by successive shortest augmenting paths on the host: exact, sequential, and —
// on a mesh matrix whose rows arrive in random order with values over six orders of magnitude — 13 to 19 s at 4e5
// unknowns, the slowest thing this library does (tools/fuzz_lu_scale.py, family perm2d).  The same assignment problem
//     minimise  sum_j c(row(j), j),   c_ij = log max_i |a_ij| - log |a_ij| >= 0
// is solved here by Bertsekas' AUCTION algorithm with epsilon-scaling, in its Jacobi form, which is parallel over the
// columns: every column without a row bids for the row it likes best at the current prices (value -c_ij - p_i), raising
// that row's price by the margin to its second choice plus eps; a row takes its highest bid (ties: the smallest column)
// and drops its previous column, which bids again in the next round.  When every column has a row, the matching is
// within n eps of the optimum and the prices are dual variables: with u_i = -p_i and v_j = min_i (c_ij + p_i),
// u_i + v_j <= c_ij everywhere and >= c_ij - eps on the matched entries, i.e. B = Dr P A Dc has |b_ij| <= 1 and
// |b_jj| >= exp(-eps) — what the static-pivoting stage needs (its factors are checked by every solve anyway).
// eps goes from a quarter of the cost range down to kFinalEps in steps of 4; prices survive a change of eps, the
// assignment does not.  One round = three small launches over the list of bidding columns (bid, resolve ties, assign),
// the next list is built by atomic append; the host looks at its length every few rounds.
// A matrix without a perfect matching lets the prices grow without end: rounds and seconds are bounded, and a run that
// does not finish leaves the work to the exact host algorithm (which also reports structural singularity).
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "common.hpp"
#include "static_pivot.hpp"

namespace spl {
namespace {

constexpr double kFinalEps = 0.02;  // |b_jj| >= exp(-0.02) = 0.98 of the largest scaled entry of its row and column
constexpr double kAbsent = 1e300;   // cost of a stored zero

__device__ __forceinline__ unsigned long long ordered_bits(double d) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// c_p = log(column maximum) - log |a_p|; cmax[j]; *bad |= 1 for an empty / zero / non-finite column; *crange = largest cost
__global__ __launch_bounds__(256) void auction_cost_kernel(int n, const int *__restrict__ Ap, const double *__restrict__ Ax,
                                                           double *__restrict__ c, double *__restrict__ cmax,
                                                           int *__restrict__ bad, unsigned long long *__restrict__ crange) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  double m = 0.0;
  for (int p = Ap[j]; p < Ap[j + 1]; ++p) m = fmax(m, fabs(Ax[p]));
  if (!(m > 0.0) || !(m < 1e308)) {
    atomicOr(bad, 1);
    return;
  }
  cmax[j] = m;
  const double lm = log(m);
  double worst = 0.0;
  for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
    const double a = fabs(Ax[p]);
    const double cp = a > 0.0 ? lm - log(a) : kAbsent;
    c[p] = cp;
    if (cp < kAbsent) worst = fmax(worst, cp);
  }
  atomicMax(crange, ordered_bits(worst));
}

__global__ __launch_bounds__(256) void auction_iota_kernel(int n, int *__restrict__ list) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n) list[j] = j;
}

// every listed column bids for its best row
__global__ __launch_bounds__(256) void auction_bid_kernel(const int *__restrict__ list, const int *__restrict__ count,
                                                          const int *__restrict__ Ap, const int *__restrict__ Ai,
                                                          const double *__restrict__ c, const double *__restrict__ price,
                                                          double eps, double lone, int *__restrict__ target,
                                                          double *__restrict__ bidv, unsigned long long *__restrict__ bidval,
                                                          int *__restrict__ bad) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= *count) return;
  const int j = list[t];
  double best = -INFINITY, second = -INFINITY;
  int bi = -1;
  for (int p = Ap[j]; p < Ap[j + 1]; ++p) {
    const double cp = c[p];
    if (cp >= kAbsent) continue;
    const int i = Ai[p];
    const double v = -cp - price[i];
    if (v > best) {
      second = best;
      best = v;
      bi = i;
    } else if (v > second) {
      second = v;
    }
  }
  if (bi < 0) {
    atomicOr(bad, 1);
    target[j] = -1;
    return;
  }
  if (second == -INFINITY) second = best - lone;  // a column with one entry: any price is worth paying
  const double bid = price[bi] + (best - second) + eps;
  target[j] = bi;
  bidv[j] = bid;
  atomicMax(&bidval[bi], ordered_bits(bid));
}

// among the highest bids for a row the smallest column wins
__global__ __launch_bounds__(256) void auction_resolve_kernel(const int *__restrict__ list, const int *__restrict__ count,
                                                              const int *__restrict__ target, const double *__restrict__ bidv,
                                                              const unsigned long long *__restrict__ bidval,
                                                              int *__restrict__ bidder) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= *count) return;
  const int j = list[t], i = target[j];
  if (i >= 0 && ordered_bits(bidv[j]) == bidval[i]) atomicMin(&bidder[i], j);
}

// winners take their rows (the previous owner bids again), losers bid again
__global__ __launch_bounds__(256) void auction_assign_kernel(const int *__restrict__ list, const int *__restrict__ count,
                                                             const int *__restrict__ target, const double *__restrict__ bidv,
                                                             unsigned long long *__restrict__ bidval, int *__restrict__ bidder,
                                                             double *__restrict__ price, int *__restrict__ owner,
                                                             int *__restrict__ assigned, int *__restrict__ next,
                                                             int *__restrict__ next_count) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= *count) return;
  const int j = list[t], i = target[j];
  if (i < 0) return;
  if (bidder[i] == j) {  // (only this thread writes row i below; the losers of row i compare bidder[i] with themselves)
    const int old = owner[i];
    owner[i] = j;
    assigned[j] = i;
    price[i] = bidv[j];
    bidval[i] = 0ull;
    bidder[i] = 0x7fffffff;
    if (old >= 0) {
      assigned[old] = -1;
      next[atomicAdd(next_count, 1)] = old;
    }
  } else {
    next[atomicAdd(next_count, 1)] = j;
  }
}

}  // namespace

// d_Ap / d_Ai / d_Ax: CSC arrays of A on the device (int32 pointers); h_*: the same on the host (for the duals).
// true: T holds a perfect matching with its scalings; false: not finished within the bounds (or an empty column):
// the caller runs the exact host algorithm.
bool auction_transversal(int n, const int *d_Ap, const int *d_Ai, const double *d_Ax, const int *h_Ap, const int *h_Ai,
                         const double *h_Ax, sp::Transversal &T, double max_seconds, hipStream_t s) {
  if (n <= 0) return false;
  const auto t_start = std::chrono::steady_clock::now();
  auto seconds = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
  const int64_t nnz = h_Ap[n];
  const size_t N = (size_t)n;
  DBuf<double> c((size_t)nnz), cmax(N), price(N), bidv(N);
  DBuf<unsigned long long> bidval(N + 1);  // [n]: the cost range
  DBuf<int> bidder(N), owner(N), assigned(N), target(N), list0(N), list1(N), ctl(4);  // ctl: count0, count1, bad
  SPL_HIP(hipMemsetAsync(ctl.get(), 0, 4 * sizeof(int), s));
  SPL_HIP(hipMemsetAsync(bidval.get(), 0, (N + 1) * sizeof(unsigned long long), s));
  SPL_HIP(hipMemsetAsync(price.get(), 0, N * sizeof(double), s));
  SPL_HIP(hipMemsetAsync(bidder.get(), 0x7f, N * sizeof(int), s));  // 0x7f7f7f7f: larger than any column
  const unsigned gn = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(auction_cost_kernel, dim3(gn), dim3(256), 0, s, n, d_Ap, d_Ax, c.get(), cmax.get(), ctl.get() + 2,
                     bidval.get() + n);
  int h_ctl[4] = {0, 0, 0, 0};
  unsigned long long h_range = 0;
  SPL_HIP(hipMemcpyAsync(h_ctl, ctl.get(), 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipMemcpyAsync(&h_range, bidval.get() + n, sizeof h_range, hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  if (h_ctl[2]) return false;
  double crange;
  {
    const unsigned long long b = (h_range >> 63) ? (h_range & 0x7fffffffffffffffull) : ~h_range;
    std::memcpy(&crange, &b, sizeof crange);
  }
  if (!(crange >= 0.0) || !(crange < 1e6)) return false;
  const double lone = 2.0 * crange + 1.0;
  int *lists[2] = {list0.get(), list1.get()};
  int64_t rounds = 0;
  for (double eps = std::max(crange / 4.0, kFinalEps);; eps = std::max(eps / 4.0, kFinalEps)) {
    // a phase: everybody bids again at the prices reached so far
    SPL_HIP(hipMemsetAsync(owner.get(), 0xff, N * sizeof(int), s));
    SPL_HIP(hipMemsetAsync(assigned.get(), 0xff, N * sizeof(int), s));
    hipLaunchKernelGGL(auction_iota_kernel, dim3(gn), dim3(256), 0, s, n, lists[0]);
    int cur = 0, count = n;
    SPL_HIP(hipMemcpyAsync(ctl.get(), &count, sizeof(int), hipMemcpyHostToDevice, s));
    while (count > 0) {
      const unsigned g = (unsigned)((count + 255) / 256);
      for (int k = 0; k < 16; ++k) {  // (the list only shrinks: the grid of the first round of the batch covers the others)
        int *cnt = ctl.get() + cur, *ncnt = ctl.get() + (cur ^ 1);
        SPL_HIP(hipMemsetAsync(ncnt, 0, sizeof(int), s));
        hipLaunchKernelGGL(auction_bid_kernel, dim3(g), dim3(256), 0, s, lists[cur], cnt, d_Ap, d_Ai, c.get(), price.get(), eps,
                           lone, target.get(), bidv.get(), bidval.get(), ctl.get() + 2);
        hipLaunchKernelGGL(auction_resolve_kernel, dim3(g), dim3(256), 0, s, lists[cur], cnt, target.get(), bidv.get(),
                           bidval.get(), bidder.get());
        hipLaunchKernelGGL(auction_assign_kernel, dim3(g), dim3(256), 0, s, lists[cur], cnt, target.get(), bidv.get(),
                           bidval.get(), bidder.get(), price.get(), owner.get(), assigned.get(), lists[cur ^ 1], ncnt);
        cur ^= 1;
        ++rounds;
      }
      SPL_HIP(hipMemcpyAsync(h_ctl, ctl.get(), 4 * sizeof(int), hipMemcpyDeviceToHost, s));
      SPL_HIP(hipStreamSynchronize(s));
      if (h_ctl[2]) return false;  // a column without a usable entry
      count = h_ctl[cur];
      if (rounds > 2000000 || seconds() > max_seconds) return false;
    }
    if (eps <= kFinalEps) break;
  }
  SPL_HIP(hipGetLastError());
  // matching and prices -> the transversal and its scalings
  std::vector<int> h_assigned(N);
  std::vector<double> h_price(N);
  SPL_HIP(hipMemcpyAsync(h_assigned.data(), assigned.get(), N * sizeof(int), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipMemcpyAsync(h_price.data(), price.get(), N * sizeof(double), hipMemcpyDeviceToHost, s));
  SPL_HIP(hipStreamSynchronize(s));
  T.row_of_col.assign(N, -1);
  T.col_of_row.assign(N, -1);
  for (int j = 0; j < n; ++j) {
    const int i = h_assigned[(size_t)j];
    if (i < 0 || i >= n || T.col_of_row[(size_t)i] >= 0) return false;  // (cannot happen: checked, not assumed)
    T.row_of_col[(size_t)j] = i;
    T.col_of_row[(size_t)i] = j;
  }
  T.dr.resize(N);
  T.dc.resize(N);
  for (int i = 0; i < n; ++i) T.dr[(size_t)i] = std::exp(-h_price[(size_t)i]);
  for (int j = 0; j < n; ++j) {
    double m = 0.0;
    for (int p = h_Ap[j]; p < h_Ap[j + 1]; ++p) m = std::max(m, std::fabs(h_Ax[p]));
    const double lm = std::log(m);
    double v = std::numeric_limits<double>::infinity();
    for (int p = h_Ap[j]; p < h_Ap[j + 1]; ++p) {
      const double a = std::fabs(h_Ax[p]);
      if (a > 0.0) v = std::min(v, lm - std::log(a) + h_price[(size_t)h_Ai[p]]);
    }
    T.dc[(size_t)j] = std::exp(v) / m;
    if (!std::isfinite(T.dc[(size_t)j]) || !(T.dc[(size_t)j] > 0.0) || !(T.dr[(size_t)j] > 0.0)) return false;  // out of range
  }
  if (getenv("SPL_MF_TIMING"))
    fprintf(stderr, "[static pivot] auction on the device: %lld rounds, %.3f s, cost range %.2f\n", (long long)rounds, seconds(),
            crange);
  return true;
}

}  // namespace spl

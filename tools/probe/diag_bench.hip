// diag_bench.hip — the 64 x 64 diagonal-block factorisation with its two explicit inverses (dense_lu_kernels.hpp), which
// sits on the chain of every block step of a front: time per launch and accuracy of the unblocked form of rounds 1 - 3
// (Band::piv = 2) against the blocked one of round 4 (piv = 0): |L U - A|, |inv(L) L - I|, |U inv(U) - I|.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../sparse-linear_amd/csrc -I../../include diag_bench.hip -o diag_bench
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "dense_lu_kernels.hpp"

namespace spl { void set_last_error(const char *what, hipError_t e) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); } }
using namespace spl;

int main(int argc, char **argv) {
  const int n = 64, ld = 80, jb = argc > 1 ? atoi(argv[1]) : 64;
  std::vector<double> A((size_t)ld * n, 0.0);
  srand(1);
  for (int j = 0; j < jb; ++j) for (int i = 0; i < jb; ++i) A[(size_t)i + (size_t)j * ld] = (i == j) ? 9.0 : (rand() / (double)RAND_MAX - 0.5);
  double *dA, *dinv; int *dsing;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dinv, 2 * 64 * 64 * 8); hipMalloc(&dsing, 4);
  hipMemset(dsing, 0, 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_lu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileBytes));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int piv : {2, 0}) {
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    Band b = dense_view(dA, jb, ld, 0, 0, piv);
    hipLaunchKernelGGL(diag_lu_kernel, dim3(1), dim3(256), 2 * kTileBytes, 0, b, 0, jb, dsing, dinv, dinv + 64 * 64);
    std::vector<double> F(A.size()), iv(2 * 64 * 64);
    hipMemcpy(F.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(iv.data(), dinv, iv.size() * 8, hipMemcpyDeviceToHost);
    auto Lf = [&](int i, int k) { return i > k ? F[(size_t)i + (size_t)k * ld] : (i == k ? 1.0 : 0.0); };
    auto Uf = [&](int k, int j) { return k <= j ? F[(size_t)k + (size_t)j * ld] : 0.0; };
    double e_lu = 0, e_l = 0, e_u = 0;
    for (int i = 0; i < jb; ++i) for (int j = 0; j < jb; ++j) {
      double s = 0, sl = 0, su = 0;
      for (int k = 0; k < jb; ++k) { s += Lf(i, k) * Uf(k, j); sl += iv[(size_t)i + (size_t)k * 64] * Lf(k, j); su += Uf(i, k) * iv[4096 + (size_t)k + (size_t)j * 64]; }
      e_lu = fmax(e_lu, fabs(s - A[(size_t)i + (size_t)j * ld]));
      e_l = fmax(e_l, fabs(sl - (i == j)));
      e_u = fmax(e_u, fabs(su - (i == j)));
    }
    // identity padding of the inverses beyond jb
    double e_pad = 0;
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) if (i >= jb || j >= jb) {
      e_pad = fmax(e_pad, fabs(iv[(size_t)i + (size_t)j * 64] - (i == j)));
      e_pad = fmax(e_pad, fabs(iv[4096 + (size_t)i + (size_t)j * 64] - (i == j)));
    }
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(diag_lu_kernel, dim3(1), dim3(256), 2 * kTileBytes, 0, b, 0, jb, dsing, dinv, dinv + 64 * 64);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(diag_lu_kernel, dim3(1), dim3(256), 2 * kTileBytes, 0, b, 0, jb, dsing, dinv, dinv + 64 * 64);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s, jb = %d: %.1f us per launch (200 back to back, launch gap included); |LU - A| %.2e  |inv(L) L - I| %.2e  |U inv(U) - I| %.2e  padding %.1e\n",
           piv == 2 ? "unblocked (rounds 1 - 3)" : "blocked by 16 (round 4) ", jb, ms * 1e3 / 200, e_lu, e_l, e_u, e_pad);
  }
  // ---- the complex twin (two planes zoff apart, diag_lu_kernel_z)
  {
    const size_t plane = (size_t)ld * n;
    std::vector<double> Z(2 * plane, 0.0);
    for (int j = 0; j < jb; ++j) for (int i = 0; i < jb; ++i) {
      Z[(size_t)i + (size_t)j * ld] = (i == j) ? 9.0 : (rand() / (double)RAND_MAX - 0.5);
      Z[plane + (size_t)i + (size_t)j * ld] = (i == j) ? 2.0 : (rand() / (double)RAND_MAX - 0.5);
    }
    double *dZ, *dzi;
    hipMalloc(&dZ, Z.size() * 8); hipMalloc(&dzi, (size_t)kInvBlockZ * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_lu_kernel_z), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDiagLdsZ);
    for (int piv : {2, 0}) {
      hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice);
      Band b = dense_view(dZ, jb, ld, 0, plane, piv);
      hipLaunchKernelGGL(diag_lu_kernel_z, dim3(1), dim3(256), kDiagLdsZ, 0, b, 0, jb, dsing, dzi, dzi + 2 * 64 * 64);
      std::vector<double> F(Z.size()), iv((size_t)kInvBlockZ);
      hipMemcpy(F.data(), dZ, Z.size() * 8, hipMemcpyDeviceToHost);
      hipMemcpy(iv.data(), dzi, iv.size() * 8, hipMemcpyDeviceToHost);
      typedef std::pair<double, double> cd;
      auto mul = [](cd a, cd c) { return cd(a.first * c.first - a.second * c.second, a.first * c.second + a.second * c.first); };
      auto Fz = [&](int i, int j) { return cd(F[(size_t)i + (size_t)j * ld], F[plane + (size_t)i + (size_t)j * ld]); };
      auto Lf = [&](int i, int k) { return i > k ? Fz(i, k) : cd(i == k ? 1.0 : 0.0, 0.0); };
      auto Uf = [&](int k, int j) { return k <= j ? Fz(k, j) : cd(0.0, 0.0); };
      auto IL = [&](int i, int k) { return cd(iv[(size_t)i + (size_t)k * 64], iv[4096 + (size_t)i + (size_t)k * 64]); };
      auto IU = [&](int i, int k) { return cd(iv[8192 + (size_t)i + (size_t)k * 64], iv[8192 + 4096 + (size_t)i + (size_t)k * 64]); };
      double e_lu = 0, e_l = 0, e_u = 0;
      for (int i = 0; i < jb; ++i) for (int j = 0; j < jb; ++j) {
        cd s(0, 0), sl(0, 0), su(0, 0);
        for (int k = 0; k < jb; ++k) {
          cd t = mul(Lf(i, k), Uf(k, j)); s.first += t.first; s.second += t.second;
          t = mul(IL(i, k), Lf(k, j)); sl.first += t.first; sl.second += t.second;
          t = mul(Uf(i, k), IU(k, j)); su.first += t.first; su.second += t.second;
        }
        e_lu = fmax(e_lu, hypot(s.first - Z[(size_t)i + (size_t)j * ld], s.second - Z[plane + (size_t)i + (size_t)j * ld]));
        e_l = fmax(e_l, hypot(sl.first - (i == j), sl.second));
        e_u = fmax(e_u, hypot(su.first - (i == j), su.second));
      }
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(diag_lu_kernel_z, dim3(1), dim3(256), kDiagLdsZ, 0, b, 0, jb, dsing, dzi, dzi + 2 * 64 * 64);
      hipEventRecord(e0, 0);
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(diag_lu_kernel_z, dim3(1), dim3(256), kDiagLdsZ, 0, b, 0, jb, dsing, dzi, dzi + 2 * 64 * 64);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("complex, %s, jb = %d: %.1f us per launch; |LU - A| %.2e  |inv(L) L - I| %.2e  |U inv(U) - I| %.2e\n",
             piv == 2 ? "unblocked" : "blocked by 16", jb, ms * 1e3 / 200, e_lu, e_l, e_u);
    }
  }
  return 0;
}

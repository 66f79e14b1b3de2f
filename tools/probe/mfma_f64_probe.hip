// Probe the operand / accumulator layout of v_mfma_f64_16x16x4_f64 on gfx950 by brute force.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4v __attribute__((ext_vector_type(4)));
__global__ void k(const double* a, const double* b, double* c) {
  int l = threadIdx.x;
  double4v acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) c[l * 4 + r] = acc[r];
}
int main() {
  double ha[64], hb[64], hc[256];
  for (int l = 0; l < 64; ++l) { ha[l] = 1 + l * 0.5; hb[l] = 100 + l * 3; }
  double *da, *db, *dc;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 2048);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  k<<<1, 64>>>(da, db, dc);
  hipMemcpy(hc, dc, 2048, hipMemcpyDeviceToHost);
  // hypotheses
  auto Aidx = [](int h, int l, int& i, int& kk) { if (h == 0) { i = l % 16; kk = l / 16; } else { i = l / 4; kk = l % 4; } };
  auto Bidx = [](int h, int l, int& kk, int& j) { if (h == 0) { kk = l / 16; j = l % 16; } else { kk = l % 4; j = l / 4; } };
  for (int ha_ = 0; ha_ < 2; ++ha_) for (int hb_ = 0; hb_ < 2; ++hb_) for (int hcx = 0; hcx < 4; ++hcx) {
    double A[16][4], B[4][16], C[16][16] = {};
    for (int l = 0; l < 64; ++l) { int i, kk, j; Aidx(ha_, l, i, kk); A[i][kk] = ha[l]; Bidx(hb_, l, kk, j); B[kk][j] = hb[l]; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int kk = 0; kk < 4; ++kk) C[i][j] += A[i][kk] * B[kk][j];
    bool ok = true;
    for (int l = 0; l < 64 && ok; ++l) for (int r = 0; r < 4; ++r) {
      int i, j;
      if (hcx == 0) { i = 4 * (l / 16) + r; j = l % 16; }
      else if (hcx == 1) { i = l % 16; j = 4 * (l / 16) + r; }
      else if (hcx == 2) { i = (l / 16) + 4 * r; j = l % 16; }
      else { i = l % 16; j = (l / 16) + 4 * r; }
      if (C[i][j] != hc[l * 4 + r]) { ok = false; break; }
    }
    if (ok) printf("MATCH: A-hyp %d  B-hyp %d  C-hyp %d\n", ha_, hb_, hcx);
  }
  printf("probe done; c[0..7]= %g %g %g %g | %g %g %g %g\n", hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7]);
  return 0;
}

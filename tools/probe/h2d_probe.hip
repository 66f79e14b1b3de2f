// h2d_probe.hip — pageable host-to-device copies by size: hipMemcpy against hipMemcpyAsync + hipStreamSynchronize (the analysis and
// the first factorisation upload the graph, the tree and the lists of the levels from std::vector storage).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 h2d_probe.hip -o h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
  const size_t maxb = (size_t)256 << 20;
  std::vector<char> h(maxb);
  memset(h.data(), 1, maxb);
  char *d;
  CK(hipMalloc(&d, maxb));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (size_t b : {(size_t)4 << 10, (size_t)64 << 10, (size_t)1 << 20, (size_t)8 << 20, (size_t)32 << 20, (size_t)256 << 20}) {
    const int reps = b <= ((size_t)1 << 20) ? 200 : 10;
    double t[2];
    for (int mode = 0; mode < 2; ++mode) {
      for (int w = 0; w < 2; ++w) { if (mode) { CK(hipMemcpyAsync(d, h.data(), b, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); } else CK(hipMemcpy(d, h.data(), b, hipMemcpyHostToDevice)); }
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r) {
        if (mode) { CK(hipMemcpyAsync(d, h.data(), b, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }
        else CK(hipMemcpy(d, h.data(), b, hipMemcpyHostToDevice));
      }
      t[mode] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    }
    printf("%9zu KB: hipMemcpy %9.1f us = %6.2f GB/s   hipMemcpyAsync + sync %9.1f us = %6.2f GB/s\n", b >> 10, t[0], b / t[0] * 1e-3, t[1], b / t[1] * 1e-3);
  }
  // many small copies queued, one wait at the end (what upload_vec did per level)
  {
    const size_t b = 64 << 10;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 400; ++r) CK(hipMemcpyAsync(d + (size_t)r * b, h.data() + (size_t)r * b, b, hipMemcpyHostToDevice, s));
    const double tq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipStreamSynchronize(s));
    const double ta = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("400 x 64 KB hipMemcpyAsync: queued in %.1f us (%.2f us each), complete after %.1f us\n", tq, tq / 400, ta);
  }
  return 0;
}

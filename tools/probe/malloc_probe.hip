// malloc_probe.hip — how long hipMalloc / hipFree of very large blocks take (diagnostic)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <atomic>
#include <thread>
#include <algorithm>
__global__ void tiny_kernel(double *p) { p[blockIdx.x * 256 + threadIdx.x] += 1.0; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  size_t fr, tot;
  hipMemGetInfo(&fr, &tot);
  printf("free %.1f GB of %.1f GB\n", fr * 1e-9, tot * 1e-9);
  {  // first thing in a fresh process: nothing of ours has been released yet
    void *p = nullptr;
    double t0 = now();
    (void)hipMalloc(&p, (size_t)250e9);
    double t1 = now();
    (void)hipFree(p);
    double t2 = now();
    (void)hipMalloc(&p, (size_t)250e9);
    double t3 = now();
    (void)hipFree(p);
    printf("fresh process: 250 GB hipMalloc %.1f ms, hipFree %.1f ms, the same again at once %.1f ms\n", (t1 - t0) * 1e3,
           (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    std::this_thread::sleep_for(std::chrono::seconds(8));
    t0 = now();
    (void)hipMalloc(&p, (size_t)250e9);
    t1 = now();
    (void)hipFree(p);
    printf("and 8 s after that release: %.1f ms\n", (t1 - t0) * 1e3);
    std::this_thread::sleep_for(std::chrono::seconds(8));
  }
  for (double gb : {1.0, 8.0, 32.0, 64.0, 100.0, 128.0, 172.0, 250.0}) {
    void *p = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&p, (size_t)(gb * 1e9));
    double t1 = now();
    if (e != hipSuccess) { printf("%.0f GB: %s\n", gb, hipGetErrorString(e)); continue; }
    hipMemset(p, 0, 1 << 20);
    hipDeviceSynchronize();
    double t2 = now();
    hipFree(p);
    double t3 = now();
    printf("one block of %5.0f GB: hipMalloc %8.1f ms, first touch %6.1f ms, hipFree %8.1f ms\n", gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
  }
  {  // the pattern of the 200^3 factorisation: 172 + 42 + 41 GB
    void *p[3];
    double t0 = now();
    hipMalloc(&p[0], (size_t)172e9);
    double t1 = now();
    hipMalloc(&p[1], (size_t)42e9);
    double t2 = now();
    hipMalloc(&p[2], (size_t)41e9);
    double t3 = now();
    printf("172 + 42 + 41 GB: %.1f + %.1f + %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    for (int i = 0; i < 3; ++i) hipFree(p[i]);
  }
  {  // does a large hipMalloc on another thread hold up kernel launches on this one?
    double *buf;
    (void)hipMalloc(&buf, 1 << 20);
    std::atomic<int> done{0};
    void *big = nullptr;
    double tm = 0;
    std::thread th([&] {
      double t0 = now();
      (void)hipMalloc(&big, (size_t)200e9);
      tm = now() - t0;
      done = 1;
    });
    int launches = 0;
    double worst = 0, t0 = now();
    while (!done) {
      double a = now();
      hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, 0, buf);
      (void)hipStreamSynchronize(0);
      worst = std::max(worst, now() - a);
      ++launches;
    }
    th.join();
    printf("200 GB on a second thread: %.1f ms; meanwhile %d launch+sync pairs in %.1f ms, slowest %.2f ms\n", tm * 1e3,
           launches, (now() - t0) * 1e3, worst * 1e3);
    (void)hipFree(big);
    (void)hipFree(buf);
  }
  {  // per-block times
    std::vector<void *> v(32);
    for (auto &q : v) {
      double t0 = now();
      (void)hipMalloc(&q, (size_t)8e9);
      printf("%.0f ", (now() - t0) * 1e3);
    }
    printf("ms per 8 GB block\n");
    for (auto &q : v) (void)hipFree(q);
  }
  {  // the same total in blocks of 8 GB
    std::vector<void *> v(32);
    double t0 = now();
    for (auto &q : v) hipMalloc(&q, (size_t)8e9);
    double t1 = now();
    printf("32 x 8 GB: %.1f ms\n", (t1 - t0) * 1e3);
    for (auto &q : v) hipFree(q);
  }
  return 0;
}

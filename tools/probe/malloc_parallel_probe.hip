// malloc_parallel_probe.hip — is the cost of a very large hipMalloc (1.3 s for 171 + 84 GB at config C5 in a fresh
// process on this pool) something several threads can share?  One block of G GB | k blocks of G / k GB one after the
// other | the same from k threads at once.  hipcc --offload-arch=gfx950 -O2 -o malloc_parallel_probe malloc_parallel_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 170.0;
  const int k = argc > 2 ? atoi(argv[2]) : 8;
  (void)hipFree(nullptr);
  const size_t total = (size_t)(gb * 1e9), part = total / (size_t)k;
  const int first = argc > 3 ? atoi(argv[3]) : 0;  // which variant goes first (the first allocation of a process may differ)
  for (int it = 0; it < 3; ++it) {
    const int mode = (first + it) % 3;
    std::vector<void *> p((size_t)k, nullptr);
    const double t0 = now();
    if (mode == 0) {
      if (hipMalloc(&p[0], total) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    } else if (mode == 1) {
      for (int i = 0; i < k; ++i)
        if (hipMalloc(&p[(size_t)i], part) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    } else {
      std::vector<std::thread> th;
      for (int i = 0; i < k; ++i) th.emplace_back([&, i] { (void)hipMalloc(&p[(size_t)i], part); });
      for (auto &t : th) t.join();
    }
    const double t1 = now();
    // touch the first and the last byte of every block (a kernel would fault on an unmapped page)
    for (void *q : p)
      if (q) (void)hipMemset(q, 0, 4096);
    (void)hipDeviceSynchronize();
    const double t2 = now();
    for (void *q : p)
      if (q) (void)hipFree(q);
    const double t3 = now();
    printf("%s: %.1f GB in %d block(s): hipMalloc %.3f s, first touch %.3f s, hipFree %.3f s\n",
           mode == 0 ? "one block" : mode == 1 ? "sequential blocks" : "blocks from threads", gb, mode == 0 ? 1 : k, t1 - t0,
           t2 - t1, t3 - t2);
    // let the driver finish wiping what was just released before the next variant is timed
    std::this_thread::sleep_for(std::chrono::seconds(8));
  }
  return 0;
}

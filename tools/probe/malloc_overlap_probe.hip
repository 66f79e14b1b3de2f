// malloc_overlap_probe.hip — can work go on while another thread waits in a large allocation (round 4)?
// A hipMalloc that reaches into memory the driver is still wiping waits for it (malloc_probe.hip); a one-shot
// factorisation at config C5 pays 2 - 3.5 s for that before its first kernel.  Thread B allocates `gb` GB in slabs of
// `slab` GB one after the other (hipMalloc, or hipMemCreate + hipMemMap into one reserved range with mode = vmm) right
// after the process released as much; thread A meanwhile launches a ~100 us kernel in a loop, synchronising every 8th.
// Printed: per-slab times, and what thread A saw — launches done while B was busy, its longest stall.
// hipcc --offload-arch=gfx950 -O2 -o malloc_overlap_probe malloc_overlap_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void spin_kernel(double *out, int iters) {
  double a = threadIdx.x * 1e-3;
  for (int i = 0; i < iters; ++i) a = a * 1.0000001 + 1e-9;
  if (a == 12345.678) out[0] = a;
}
__global__ void touch_kernel(char *p, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096; i < n; i += (size_t)gridDim.x * blockDim.x * 4096) p[i] = 1;
}

int main(int argc, char **argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 200.0;
  const double slab_gb = argc > 2 ? atof(argv[2]) : 8.0;
  const bool vmm = argc > 3 && !strcmp(argv[3], "vmm");
  const bool nosync = argc > 4 && !strcmp(argv[4], "nosync");
  (void)hipFree(nullptr);
  double *d_out = nullptr;
  (void)hipMalloc(&d_out, 4096);
  hipStream_t sa;
  (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  // calibrate the kernel to ~100 us
  int iters = 20000;
  for (int rep = 0; rep < 3; ++rep) {
    const double t0 = now();
    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, sa, d_out, iters);
    (void)hipStreamSynchronize(sa);
    const double dt = now() - t0;
    if (rep == 2) printf("spin kernel: %d iterations, %.1f us with its launch and sync\n", iters, dt * 1e6);
  }
  // dirty memory: allocate, touch, release
  {
    void *p = nullptr;
    const size_t total = (size_t)(gb * 1e9);
    double t0 = now();
    if (hipMalloc(&p, total) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    double t1 = now();
    hipLaunchKernelGGL(touch_kernel, dim3(4096), dim3(256), 0, 0, (char *)p, total);
    (void)hipDeviceSynchronize();
    double t2 = now();
    (void)hipFree(p);
    double t3 = now();
    printf("dirtying: hipMalloc %.3f s, touch %.3f s, hipFree %.3f s\n", t1 - t0, t2 - t1, t3 - t2);
  }
  std::atomic<int> b_busy{1};
  std::vector<double> slab_t;
  const size_t slab = (size_t)(slab_gb * 1e9) / (2 << 20) * (2 << 20);
  const int nslab = (int)(gb / slab_gb);
  double b_t0 = 0, b_t1 = 0;
  std::thread B([&] {
    b_t0 = now();
    if (!vmm) {
      std::vector<void *> ps;
      for (int i = 0; i < nslab; ++i) {
        void *p = nullptr;
        const double t0 = now();
        if (hipMalloc(&p, slab) != hipSuccess) { printf("slab hipMalloc failed\n"); break; }
        slab_t.push_back(now() - t0);
        ps.push_back(p);
      }
      b_t1 = now();
      b_busy = 0;
      for (void *p : ps) (void)hipFree(p);
    } else {
      hipDeviceptr_t base = 0;
      hipMemAllocationProp prop = {};
      prop.type = hipMemAllocationTypePinned;
      prop.location.type = hipMemLocationTypeDevice;
      prop.location.id = 0;
      size_t gran = 0;
      (void)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
      printf("vmm granularity %zu\n", gran);
      const size_t sl = slab / gran * gran;
      if (hipMemAddressReserve(&base, sl * (size_t)nslab, 0, 0, 0) != hipSuccess) { printf("reserve failed\n"); b_busy = 0; return; }
      std::vector<hipMemGenericAllocationHandle_t> hs;
      hipMemAccessDesc acc = {};
      acc.location = prop.location;
      acc.flags = hipMemAccessFlagsProtReadWrite;
      for (int i = 0; i < nslab; ++i) {
        hipMemGenericAllocationHandle_t h;
        const double t0 = now();
        if (hipMemCreate(&h, sl, &prop, 0) != hipSuccess) { printf("hipMemCreate failed\n"); break; }
        if (hipMemMap((hipDeviceptr_t)((char *)base + sl * (size_t)i), sl, 0, h, 0) != hipSuccess) { printf("hipMemMap failed\n"); break; }
        if (hipMemSetAccess((hipDeviceptr_t)((char *)base + sl * (size_t)i), sl, &acc, 1) != hipSuccess) { printf("hipMemSetAccess failed\n"); break; }
        slab_t.push_back(now() - t0);
        hs.push_back(h);
      }
      b_t1 = now();
      b_busy = 0;
      // touch the whole range from one kernel: the mapping is one address range
      hipLaunchKernelGGL(touch_kernel, dim3(4096), dim3(256), 0, 0, (char *)base, sl * hs.size());
      printf("touch of the mapped range: %s\n", hipGetErrorString(hipDeviceSynchronize()));
      (void)hipMemUnmap(base, sl * hs.size());
      for (auto h : hs) (void)hipMemRelease(h);
      (void)hipMemAddressFree(base, sl * (size_t)nslab);
    }
  });
  // thread A
  int launches = 0;
  double worst = 0.0, a_t0 = now(), last = a_t0, worst_launch = 0.0;
  while (b_busy.load()) {
    const double l0 = now();
    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, sa, d_out, iters);
    const double l1 = now();
    worst_launch = l1 - l0 > worst_launch ? l1 - l0 : worst_launch;
    ++launches;
    if (!nosync && (launches & 7) == 0) (void)hipStreamSynchronize(sa);
    if (nosync && (launches & 63) == 0) {  // keep the queue bounded without a HIP call that might take a lock
      while (hipStreamQuery(sa) == hipErrorNotReady && b_busy.load()) std::this_thread::yield();
    }
    const double t = now();
    worst = t - last > worst ? t - last : worst;
    last = t;
  }
  (void)hipStreamSynchronize(sa);
  const double a_t1 = now();
  B.join();
  double sum = 0, mx = 0;
  for (double t : slab_t) { sum += t; mx = t > mx ? t : mx; }
  printf("%s, %d slabs of %.1f GB: total %.3f s (sum of calls %.3f s, slowest %.3f s)\n", vmm ? "hipMemCreate+Map" : "hipMalloc",
         (int)slab_t.size(), slab_gb, b_t1 - b_t0, sum, mx);
  printf("per slab ms:");
  for (double t : slab_t) printf(" %.0f", t * 1e3);
  printf("\nthread A (%s): %d launches in %.3f s = %.1f us each; longest stall of an iteration %.3f s, of a launch call %.3f s\n",
         nosync ? "query only" : "sync every 8", launches, a_t1 - a_t0, (a_t1 - a_t0) / (launches ? launches : 1) * 1e6, worst, worst_launch);
  return 0;
}

// Probe 4: can scalar loads (SQC -> L2) pull an HBM-cold stream into the XCD's L2 ahead of the vector
// loads that consume it, and at what rate?
// Per workgroup (16 wavefronts, one per CU) a private cold region of REGION bytes.
//   phase 1: every wavefront touches its share with s_load_dword, one per STEP bytes (64 or 128), at most
//            DEPTH outstanding; cycles until all returned (s_memtime).
//   phase 2: every wavefront reads its share with coalesced dwordx2 loads (8 in flight); cycles.
// mode 0: phase 2 only (cold: HBM latency);  mode 1: phase 1 then phase 2;  mode 2: phase 2 twice (warm L2
// through the vector path, the reference for "L2-hit speed").
// build: hipcc -O3 --offload-arch=gfx950 scalar_touch_probe.hip -o scalar_touch_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

template <int STEP, int DEPTH>
__device__ inline void touch_range(const char *p, int bytes) {
  unsigned t = 0;
  for (int o = 0; o < bytes; o += STEP * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const char *q = p + o + d * STEP;
      asm volatile("s_load_dword %0, %1, 0x0" : "+s"(t) : "s"(q) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t) : : "memory");
  }
}

__device__ inline double read_range(const double *p, int doubles, int lane) {
  double acc = 0.0;
  for (int o = 0; o < doubles; o += 64 * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + o + u * 64 + lane);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  return acc;
}

template <int STEP, int DEPTH>
__global__ __launch_bounds__(1024) void probe(const double *__restrict__ buf, size_t region_doubles, int mode,
                                              unsigned long long *__restrict__ stamps, double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const double *mine = buf + (size_t)blockIdx.x * region_doubles + (size_t)wave * (region_doubles / 16);
  const int share = (int)(region_doubles / 16);
  __syncthreads();
  const unsigned long long t0 = now();
  double acc = 0.0;
  if (mode == 1) touch_range<STEP, DEPTH>(reinterpret_cast<const char *>(mine), share * 8);
  if (mode == 2) acc += read_range(mine, share, lane);
  __syncthreads();
  const unsigned long long t1 = now();
  acc += read_range(mine, share, lane);
  __syncthreads();
  const unsigned long long t2 = now();
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = t2 - t1; }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int STEP, int DEPTH>
void run(const char *name, int wgs, const double *buf, size_t region_doubles, int mode, unsigned long long *d_st, double *out) {
  hipLaunchKernelGGL((probe<STEP, DEPTH>), dim3(wgs), dim3(1024), 0, 0, buf, region_doubles, mode, d_st, out);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(wgs * 2);
  hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<unsigned long long> a, b;
  for (int i = 0; i < wgs; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
  std::sort(a.begin(), a.end());
  std::sort(b.begin(), b.end());
  const double bytes = region_doubles * 8.0;
  // s_memtime ticks at 100 MHz on this part: report microseconds and GB/s per CU
  printf("%-46s %3d WGs: phase1 median %7.2f us | phase2 median %7.2f us = %6.1f GB/s per CU\n", name, wgs, a[wgs / 2] / 100.0,
         b[wgs / 2] / 100.0, bytes / (b[wgs / 2] / 100.0) / 1e3);
}

int main() {
  const size_t region = (size_t)256 << 10;  // 256 KiB per workgroup
  const size_t rd = region / 8;
  double *buf, *out;
  unsigned long long *st;
  const int maxw = 256;
  // 64 fresh slabs so that every launch reads memory no cache holds (each launch uses its own slab)
  const size_t slab = (size_t)maxw * rd;
  hipMalloc(&buf, slab * 8 * 64);
  hipMemset(buf, 0, slab * 8 * 64);
  hipMalloc(&out, (size_t)maxw * 1024 * 8);
  hipMalloc(&st, maxw * 2 * 8);
  // flush caches between: a 1 GiB memset touches more than L2 + MALL
  double *flush;
  hipMalloc(&flush, (size_t)1 << 30);
  int k = 0;
  auto fresh = [&]() { hipMemset(flush, 1, (size_t)1 << 30); hipDeviceSynchronize(); return buf + (size_t)(k++ % 64) * slab; };
  for (int wgs : {8, 256}) {
    run<128, 8>("cold vector read (no touch)", wgs, fresh(), rd, 0, st, out);
    run<128, 8>("vector read twice (second = L2-warm)", wgs, fresh(), rd, 2, st, out);
    run<128, 4>("scalar touch per 128 B, 4 outstanding/wave", wgs, fresh(), rd, 1, st, out);
    run<128, 8>("scalar touch per 128 B, 8 outstanding/wave", wgs, fresh(), rd, 1, st, out);
    run<128, 14>("scalar touch per 128 B, 14 outstanding/wave", wgs, fresh(), rd, 1, st, out);
    run<64, 8>("scalar touch per 64 B, 8 outstanding/wave", wgs, fresh(), rd, 1, st, out);
    run<64, 14>("scalar touch per 64 B, 14 outstanding/wave", wgs, fresh(), rd, 1, st, out);
  }
  return 0;
}

// Probe 5: does any cache policy / width let a CU's HBM stream and its L2 gathers overlap instead of
// adding up?  Same mix as tcp_mix_probe M (every wavefront: 8 gathers + 12 stream loads of 512 B per
// iteration), with the load flavours varied.  time(G alone) and time(S alone) are printed per flavour.
// build: hipcc -O3 --offload-arch=gfx950 tcp_policy_probe.hip -o tcp_policy_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

// GP: gather policy 0 plain, 1 sc1, 2 nt, 3 sc0 sc1;  SP: stream policy 0 nt, 1 plain, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt
template <int GP>
__device__ inline double gload(const double *p) {
  double v;
  if (GP == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (GP == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (GP == 2) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if (GP == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int SP>
__device__ inline double sload(const double *p) {
  double v;
  if (SP == 0) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  if (SP == 1) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (SP == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (SP == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if (SP == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}
typedef double double2v __attribute__((ext_vector_type(2)));
__device__ inline double2v sload4(const double *p) {
  double2v v;
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int US, int UG, int GP, int SP, int WIDE>
__global__ __launch_bounds__(1024) void mix_kernel(const double *__restrict__ stream, size_t per_wg_doubles,
                                                   const double *__restrict__ table, uint32_t mask, int iters,
                                                   double *__restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  const double *base = stream + (size_t)blockIdx.x * per_wg_doubles;
  double acc = 0.0;
  size_t pos = (size_t)wave * (WIDE ? 128 : 64) + (WIDE ? 2 * lane : lane);
  const size_t stride = (size_t)16 * (WIDE ? 128 : 64);
  for (int it = 0; it < iters; ++it) {
    double g[UG > 0 ? UG : 1], v[US > 0 ? US : 1];
#pragma unroll
    for (int u = 0; u < UG; ++u) {
      s = s * 1664525u + 1013904223u;
      g[u] = gload<GP>(table + ((s >> 7) & mask));
    }
    if (WIDE) {
#pragma unroll
      for (int u = 0; u < US; u += 2) {
        const double2v t = sload4(base + pos);
        v[u] = t.x;
        v[u + 1 < US ? u + 1 : u] = t.y;
        pos += stride;
        if (pos >= per_wg_doubles) pos -= per_wg_doubles;
      }
    } else {
#pragma unroll
      for (int u = 0; u < US; ++u) {
        v[u] = sload<SP>(base + pos);
        pos += stride;
        if (pos >= per_wg_doubles) pos -= per_wg_doubles;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < UG; ++u) { asm volatile("" : "+v"(g[u])); acc += g[u]; }
#pragma unroll
    for (int u = 0; u < US; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u]; }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int US, int UG, int GP, int SP, int WIDE>
float run(const double *stream, size_t per_wg, const double *table, uint32_t mask, int iters, double *out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((mix_kernel<US, UG, GP, SP, WIDE>), dim3(256), dim3(1024), 0, 0, stream, per_wg, table, mask, 2, out);
  hipEventRecord(a);
  hipLaunchKernelGGL((mix_kernel<US, UG, GP, SP, WIDE>), dim3(256), dim3(1024), 0, 0, stream, per_wg, table, mask, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

template <int GP, int SP, int WIDE>
void combo(const char *name, const double *stream, size_t per_wg, const double *table, uint32_t mask, double *out) {
  const int iters = 200;
  const float g = run<0, 8, GP, SP, WIDE>(stream, per_wg, table, mask, iters, out);
  const float s = run<12, 0, GP, SP, WIDE>(stream, per_wg, table, mask, iters, out);
  const float m = run<12, 8, GP, SP, WIDE>(stream, per_wg, table, mask, iters, out);
  printf("%-44s G %.3f  S %.3f  mixed %.3f  (sum %.3f, max %.3f)  mixed/sum %.2f\n", name, g, s, m, g + s, g > s ? g : s, m / (g + s));
}

int main() {
  double *out, *table, *stream;
  hipMalloc(&out, (size_t)256 * 1024 * sizeof(double));
  const size_t tn = (size_t)1 << 18;
  hipMalloc(&table, tn * sizeof(double));
  std::vector<double> h(tn, 1.0);
  hipMemcpy(table, h.data(), tn * sizeof(double), hipMemcpyHostToDevice);
  const size_t per_wg = (size_t)12 << 17;
  hipMalloc(&stream, 256 * per_wg * sizeof(double));
  hipMemset(stream, 0, 256 * per_wg * sizeof(double));
  const uint32_t mask = (uint32_t)(tn - 1);
  combo<0, 0, 0>("gathers plain, stream nt", stream, per_wg, table, mask, out);
  combo<0, 1, 0>("gathers plain, stream plain", stream, per_wg, table, mask, out);
  combo<0, 2, 0>("gathers plain, stream sc1", stream, per_wg, table, mask, out);
  combo<0, 3, 0>("gathers plain, stream sc0 sc1", stream, per_wg, table, mask, out);
  combo<0, 4, 0>("gathers plain, stream sc0 sc1 nt", stream, per_wg, table, mask, out);
  combo<1, 0, 0>("gathers sc1, stream nt", stream, per_wg, table, mask, out);
  combo<3, 0, 0>("gathers sc0 sc1, stream nt", stream, per_wg, table, mask, out);
  combo<1, 2, 0>("gathers sc1, stream sc1", stream, per_wg, table, mask, out);
  combo<0, 0, 1>("gathers plain, stream nt dwordx4", stream, per_wg, table, mask, out);
  combo<1, 0, 1>("gathers sc1, stream nt dwordx4", stream, per_wg, table, mask, out);
  return 0;
}

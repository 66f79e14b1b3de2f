// Probe 5 (round 3): does a matrix stream fed through LDS-DMA (global_load_lds_dwordx4: HBM -> LDS with no
// VGPR destination) share the CU's vector-L1 miss capacity with VGPR gathers the way VGPR stream loads do
// (tcp_mix_probe rows M / W: the times ADD), or does it run beside them (time = max)?
//
// One 16-wavefront workgroup per CU.  Wavefronts 0 .. NL-1 are LOADERS: each keeps D one-KiB LDS-DMA pieces in
// flight (counted vmcnt) into its own ring in LDS; wavefronts NL .. 15 are GATHERERS: UG random 8-byte gathers
// from a 2 MiB L2-resident table per iteration, results summed in registers.  Amounts per workgroup are those
// of tcp_mix_probe (per iteration 96 KiB of stream and 8192 gathers = C2's 12 B per gathered entry), whoever
// carries them.  Rows:
//   L   LDS-DMA stream alone (NL = 1, 2, 4, 8; D = 8 .. 32; nt or default policy), nobody reads the ring
//   Lc  the same with the 16-NL other wavefronts reading the ring with ds_read_b128 (no handshake: timing only)
//   G   gathers alone from the 16-NL gather wavefronts
//   X   both together: sum or max?
//   V   reference: the same split with VGPR stream loads (global_load_dwordx4 nt) in the loader wavefronts
// build: hipcc -O3 --offload-arch=gfx950 lds_dma_mix_probe.hip -o lds_dma_mix_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ inline void glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// MODE bit 0: loaders run; bit 1: gatherers gather; bit 2: non-loader wavefronts read the ring (ds_read_b128);
// bit 3: loaders use VGPR loads (dwordx4 nt) instead of LDS-DMA
template <int NL, int D, int UG, int NT, int MODE>
__global__ __launch_bounds__(1024) void mix_kernel(const char *__restrict__ stream, size_t per_wg_bytes,
                                                   const double *__restrict__ table, uint32_t mask, int iters,
                                                   double *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char ring[];  // NL * D * 2 KiB (twice the depth in flight)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int SLOTS = 2 * D;
  double acc = 0.0;
  if (wave < NL) {
    if (MODE & 1) {
      // this loader's share: pieces wave, wave + NL, ... of the workgroup's stream; iters * 96 pieces per workgroup
      const size_t npieces = (size_t)iters * 96 / NL;
      const char *base = stream + (size_t)blockIdx.x * per_wg_bytes;
      const size_t wrap = per_wg_bytes >> 10;  // pieces in the workgroup's part of the buffer
      size_t piece = wave;
      const unsigned ring0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)ring + (unsigned)wave * SLOTS * 1024u;
      if (MODE & 8) {
        f4 v[D];
        f4 s4 = {0, 0, 0, 0};
        for (size_t i = 0; i < npieces; i += D) {
#pragma unroll
          for (int d = 0; d < D; ++d) {
            v[d] = __builtin_nontemporal_load((const f4 *)(base + (piece << 10)) + lane);
            piece += NL;
            if (piece >= wrap) piece -= wrap;
          }
#pragma unroll
          for (int d = 0; d < D; ++d) s4 += v[d];
        }
        acc = s4.x + s4.y + s4.z + s4.w;
      } else {
        int slot = 0;
        for (size_t i = 0; i < npieces; ++i) {
          glds16<NT>(base + (piece << 10) + lane * 16, ring0 + (unsigned)slot * 1024u);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
          piece += NL;
          if (piece >= wrap) piece -= wrap;
          slot = slot + 1 == SLOTS ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  } else {
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    // gathers: iters * 8192 per workgroup over (16 - NL) * 64 lanes
    const int ng = (int)(((size_t)iters * 8192 / ((16 - NL) * 64) + UG - 1) / UG);
    const f4 *rp = (const f4 *)ring;
    f4 r4 = {0, 0, 0, 0};
    for (int it = 0; it < ng; ++it) {
      double g[UG];
      if (MODE & 2) {
#pragma unroll
        for (int u = 0; u < UG; ++u) {
          s = s * 1664525u + 1013904223u;
          g[u] = table[(s >> 7) & mask];
        }
      }
      if (MODE & 4) {  // 12 B per gather = 0.75 ds_read_b128 per gather and lane: 3 reads per 4 gathers
#pragma unroll
        for (int u = 0; u < (UG * 3) / 4; ++u) r4 += rp[((it * 8 + u) * 64 + lane) & (NL * SLOTS * 64 - 1)];
      }
      if (MODE & 2) {
#pragma unroll
        for (int u = 0; u < UG; ++u) acc += g[u];
      }
    }
    acc += r4.x + r4.y + r4.z + r4.w;
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int NL, int D, int UG, int NT, int MODE>
float run(int wgs, const char *stream, size_t per_wg, const double *table, uint32_t mask, int iters, double *out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const size_t lds = (size_t)NL * 2 * D * 1024;
  auto k = mix_kernel<NL, D, UG, NT, MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), lds, 0, stream, per_wg, table, mask, 2, out);
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), lds, 0, stream, per_wg, table, mask, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("launch error: %s\n", hipGetErrorString(e));
  hipEventDestroy(a);
  hipEventDestroy(b);
  return ms;
}

template <int NL, int D, int NT>
void rows(const char *stream, size_t per_wg, const double *table, uint32_t mask, double *out) {
  const int iters = 192;  // 192 * 96 KiB = 18.9 MB of stream and 1.57 M gathers per workgroup
  const double sbytes = 256.0 * iters * 96 * 1024, gathers = 256.0 * iters * 8192, clk = 2.4e9;
  const float l = run<NL, D, 8, NT, 1>(256, stream, per_wg, table, mask, iters, out);
  const float lc = run<NL, D, 8, NT, 1 | 4>(256, stream, per_wg, table, mask, iters, out);
  const float g = run<NL, D, 8, NT, 2>(256, stream, per_wg, table, mask, iters, out);
  const float x = run<NL, D, 8, NT, 1 | 2>(256, stream, per_wg, table, mask, iters, out);
  const float xc = run<NL, D, 8, NT, 1 | 2 | 4>(256, stream, per_wg, table, mask, iters, out);
  printf("NL=%d D=%2d %s: L %.3f ms (%.2f TB/s, %2d KiB in flight/CU)  Lc %.3f  G %.3f ms (%.3f g/clk/CU)  X %.3f ms  Xc %.3f ms"
         "  [sum %.3f max %.3f]\n",
         NL, D, NT ? "nt " : "def", l, sbytes / l / 1e9, NL * D, lc, g, gathers / (g * 1e-3) / 256 / clk, x, xc, l + g,
         l > g ? l : g);
  fflush(stdout);
}

template <int NL, int D>
void vrows(const char *stream, size_t per_wg, const double *table, uint32_t mask, double *out) {
  const int iters = 192;
  const double sbytes = 256.0 * iters * 96 * 1024;
  const float l = run<NL, D, 8, 1, 1 | 8>(256, stream, per_wg, table, mask, iters, out);
  const float g = run<NL, D, 8, 1, 2>(256, stream, per_wg, table, mask, iters, out);
  const float x = run<NL, D, 8, 1, 1 | 2 | 8>(256, stream, per_wg, table, mask, iters, out);
  printf("V NL=%d D=%2d VGPR dwordx4 nt stream in the loader wavefronts: S %.3f ms (%.2f TB/s)  G %.3f  X %.3f ms  [sum %.3f max %.3f]\n",
         NL, D, l, sbytes / l / 1e9, g, x, l + g, l > g ? l : g);
  fflush(stdout);
}

int main() {
  double *out, *table;
  char *stream;
  CHECK(hipMalloc(&out, (size_t)256 * 1024 * sizeof(double)));
  const size_t tn = (size_t)1 << 18;
  CHECK(hipMalloc(&table, tn * sizeof(double)));
  std::vector<double> h(tn, 1.0);
  CHECK(hipMemcpy(table, h.data(), tn * sizeof(double), hipMemcpyHostToDevice));
  const size_t per_wg = (size_t)24 << 20;  // 24 MiB per workgroup -> 6 GiB for 256: every piece is read once
  CHECK(hipMalloc(&stream, 256 * per_wg));
  CHECK(hipMemset(stream, 0, 256 * per_wg));
  const uint32_t mask = (uint32_t)(tn - 1);
  rows<1, 16, 1>(stream, per_wg, table, mask, out);
  rows<1, 32, 1>(stream, per_wg, table, mask, out);
  rows<2, 16, 1>(stream, per_wg, table, mask, out);
  rows<4, 8, 1>(stream, per_wg, table, mask, out);
  rows<4, 16, 1>(stream, per_wg, table, mask, out);
  rows<8, 8, 1>(stream, per_wg, table, mask, out);
  rows<2, 16, 0>(stream, per_wg, table, mask, out);
  rows<4, 8, 0>(stream, per_wg, table, mask, out);
  vrows<2, 16>(stream, per_wg, table, mask, out);
  vrows<4, 8>(stream, per_wg, table, mask, out);
  vrows<8, 8>(stream, per_wg, table, mask, out);
  if (getenv("PROBE_MORE")) {
    vrows<3, 8>(stream, per_wg, table, mask, out);
    vrows<3, 16>(stream, per_wg, table, mask, out);
    vrows<4, 4>(stream, per_wg, table, mask, out);
    vrows<4, 12>(stream, per_wg, table, mask, out);
    vrows<4, 16>(stream, per_wg, table, mask, out);
    vrows<6, 4>(stream, per_wg, table, mask, out);
    vrows<6, 8>(stream, per_wg, table, mask, out);
    vrows<8, 4>(stream, per_wg, table, mask, out);
    vrows<12, 4>(stream, per_wg, table, mask, out);
  }
  return 0;
}

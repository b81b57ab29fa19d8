// Does a wave's VALU / LDS-store / global-load work overlap with the partner wave's fp32 MFMAs on the same SIMD?
// 512-thread workgroups, one per CU (LDS-limited).  Waves 0-3 run MFMAs, waves 4-7 run the "other" work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NOPS = 0, int PRIO = 0>   // PRIO: s_setprio of the partner waves; NOPS: s_nop 15 after every MFMA of the MFMA waves;  bit0: MFMA waves active, bit1: VALU in partner, bit2: LDS stores in partner, bit3: global loads in partner
__global__ __launch_bounds__(512) void k(float* out, const float* src, int iters, long long* cyc) {
  __shared__ float big[32768];
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (MODE & 1) {
      f32x16 acc[4];
      for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
      float a = threadIdx.x * 0.001f, b = 1.0f + blockIdx.x * 0.01f;
      const long long t0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            if (NOPS >= 1) asm volatile("s_nop 15");
            if (NOPS >= 2) asm volatile("s_nop 15");
            if (NOPS >= 3) asm volatile("s_nop 15");
            if (NOPS >= 4) asm volatile("s_nop 7");
          }
      }
      const long long t1 = __builtin_amdgcn_s_memtime();
      if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) cyc[wave] = t1 - t0;
      for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) r += acc[i][e];
    }
  } else {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    f32x4 g = {0, 0, 0, 0};
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    const long long p0 = __builtin_amdgcn_s_memtime();
    const int piters = iters / 4;      // the partner finishes early: its cycles are all spent beside running MFMAs
    for (int it = 0; it < piters; ++it) {
      if (MODE & 2) {
#pragma unroll
        for (int u = 0; u < 12; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = v[i] * 1.0001f + 0.5f;     // 96 VALU per iteration (32 MFMAs)
      }
      if (MODE & 4) {
#pragma unroll
        for (int u = 0; u < 8; ++u) big[(u * 2048 + threadIdx.x + it) & 32767] = v[u];
        f32x4 w = {v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(&big[((u * 2048 + threadIdx.x * 4 + it * 4) & 32764)]) = w;
      }
      if (MODE & 8) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          f32x4 t = *reinterpret_cast<const f32x4*>(src + ((size_t)(blockIdx.x * 64 + it) * 8192 + u * 2048 + (threadIdx.x - 256) * 4) % (1 << 24));
          g += t;
        }
      }
    }
    for (int i = 0; i < 8; ++i) r += v[i];
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) cyc[wave] = __builtin_amdgcn_s_memtime() - p0;
    r += g[0] + g[1] + g[2] + g[3] + big[threadIdx.x];
  }
  if (r == 123.456f) out[threadIdx.x] = r;
}

static long long* g_cyc;
template <int MODE, int NOPS = 0, int PRIO = 0> float run(float* out, const float* src, int iters) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<MODE, NOPS, PRIO><<<256, 512>>>(out, src, iters, g_cyc);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < 5; ++i) k<MODE, NOPS, PRIO><<<256, 512>>>(out, src, iters, g_cyc);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  long long h[8] = {0}; (void)hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("   [mfma-wave cycles per MFMA: %.1f, implied clock %.2f GHz; partner wave: %.0f cycles per iteration]  ", (double)h[0] / (iters * 32.0), (double)h[0] / (ms / 5 * 1e6), (double)h[4] / (iters / 4));
  return ms / 5 * 1000.f;
}
int main() {
  float *out, *src; (void)hipMalloc(&out, 4096); (void)hipMalloc(&g_cyc, 64); (void)hipMemset(g_cyc, 0, 64); (void)hipMalloc(&src, (size_t)(1 << 24) * 4 + 65536);
  (void)hipMemset(src, 0, (size_t)(1 << 24) * 4 + 65536);
  const int iters = 2000;   // 2000 x 32 MFMAs x 64 cyc = 4.1M cycles ~ 2 ms
  printf("mfma only        %8.1f us\n", run<1>(out, src, iters));
  printf("mfma+valu nop16  %8.1f us\n", run<3, 1>(out, src, iters));
  printf("mfma+valu nop32  %8.1f us\n", run<3, 2>(out, src, iters));
  printf("mfma+valu nop48  %8.1f us\n", run<3, 3>(out, src, iters));
  printf("mfma+valu nop56  %8.1f us\n", run<3, 4>(out, src, iters));
  printf("mfma+valu prio3  %8.1f us\n", run<3, 0, 3>(out, src, iters));
  printf("mfma+all  prio3  %8.1f us\n", run<15, 0, 3>(out, src, iters));
  printf("mfma+all  prio1  %8.1f us\n", run<15, 0, 1>(out, src, iters));
  printf("mfma+all  nop48  %8.1f us\n", run<15, 3>(out, src, iters));
  printf("valu only        %8.1f us\n", run<2>(out, src, iters));
  printf("mfma + valu      %8.1f us\n", run<3>(out, src, iters));
  printf("lds only         %8.1f us\n", run<4>(out, src, iters));
  printf("mfma + lds       %8.1f us\n", run<5>(out, src, iters));
  printf("loads only       %8.1f us\n", run<8>(out, src, iters));
  printf("mfma + loads     %8.1f us\n", run<9>(out, src, iters));
  printf("valu+lds+loads   %8.1f us\n", run<14>(out, src, iters));
  printf("mfma + all       %8.1f us\n", run<15>(out, src, iters));
  return 0;
}

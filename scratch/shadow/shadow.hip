// What can ONE wave issue in the shadow of its own fp32 MFMAs (v_mfma_f32_32x32x2_f32, 16 passes = 64 cycles)?
// 256-thread workgroups (one wave per SIMD), one per CU.  After every MFMA the wave issues K other instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int K>   // KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: ds_write_b32, 3: s_nop 0, 4: ds_read_b128, 5: v_mov/v_add u32
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  __shared__ float big[16384];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + blockIdx.x * 0.01f;
  float v[16];
  f32x2 w[16];
  for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x + i; w[i] = f32x2{v[i], v[i] + 1}; }
  float4 rd = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const int r = (u * 4 + i + j) & 15;
          if (KIND == 0) v[r] = v[r] * 1.0001f + 0.5f;
          if (KIND == 1) w[r] = w[r] * 1.0001f + 0.5f;
          if (KIND == 2) big[(threadIdx.x + 256 * ((u * 4 + i + j) & 31)) & 16383] = v[r];
          if (KIND == 3) asm volatile("s_nop 0");
          if (KIND == 4) { float4 t = *reinterpret_cast<const float4*>(&big[(threadIdx.x * 4 + 1024 * ((u + j) & 7)) & 16380]); rd.x += t.x; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) cyc[threadIdx.x >> 6] = t1 - t0;
  float r = rd.x;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) r += acc[i][e];
  for (int i = 0; i < 16; ++i) r += v[i] + w[i].x + w[i].y;
  if (r == 123.456f) out[threadIdx.x] = r + big[threadIdx.x];
}

static long long* g_cyc;
template <int KIND, int K> void run(const char* name, float* out) {
  const int iters = 500;
  k<KIND, K><<<256, 256>>>(out, iters, g_cyc);
  (void)hipDeviceSynchronize();
  long long h[4] = {0};
  (void)hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-14s x%-2d per MFMA: %6.1f cycles per MFMA\n", name, K, (double)h[0] / (iters * 32.0));
}
int main() {
  float* out; (void)hipMalloc(&out, 4096); (void)hipMalloc(&g_cyc, 64);
  run<0, 0>("none", out);
  run<0, 2>("v_fma_f32", out); run<0, 4>("v_fma_f32", out); run<0, 8>("v_fma_f32", out); run<0, 12>("v_fma_f32", out); run<0, 16>("v_fma_f32", out);
  run<1, 4>("v_pk_fma_f32", out); run<1, 8>("v_pk_fma_f32", out); run<1, 12>("v_pk_fma_f32", out);
  run<2, 2>("ds_write_b32", out); run<2, 4>("ds_write_b32", out); run<2, 8>("ds_write_b32", out);
  run<4, 1>("ds_read_b128", out); run<4, 2>("ds_read_b128", out);
  run<3, 4>("s_nop 0", out); run<3, 16>("s_nop 0", out);
  return 0;
}

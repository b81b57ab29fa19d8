// Does the MFMA SHAPE change the clock the chip holds under an fp32 matrix loop?  (MI355X_MICROARCH.md, DVFS give-back item 7,
// reports 1.12-1.15x the FLOP/s for the 16x16x32 bf16 form over 32x32x16 at equal cycles per FLOP, on random data.)
// Bare loops, operands in registers, random data, NW waves per SIMD, 8 independent accumulators per wave, ~50 ms per launch:
//   f32 32x32x2 (64 cycles, 4096 FLOP)  vs  f32 16x16x4 (32 cycles, 2048 FLOP);   bf16 32x32x16 vs 16x16x32 for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void loop(const float* seed, float* out, int iters) {
  const int l = threadIdx.x + blockIdx.x * 256;
  float a[8], b[8];
  for (int k = 0; k < 8; ++k) { a[k] = seed[(l * 8 + k) & 65535]; b[k] = seed[(l * 8 + k + 7777) & 65535]; }
  float s = 0.f;
  if constexpr (SHAPE == 0) {                 // f32 32x32x2
    f32x16 acc[8];
    for (int k = 0; k < 8; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + 3) & 7], acc[k], 0, 0, 0);
    for (int k = 0; k < 8; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  } else if constexpr (SHAPE == 1) {          // f32 16x16x4
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < 2 * iters; ++it)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[(k + 3) & 7], acc[k], 0, 0, 0);
    for (int k = 0; k < 8; ++k) for (int e = 0; e < 4; ++e) s += acc[k][e];
  } else if constexpr (SHAPE == 2) {          // bf16 32x32x16
    bf16x8 av[4], bv[4];
    for (int k = 0; k < 4; ++k) for (int e = 0; e < 8; ++e) { av[k][e] = (__bf16)a[(k + e) & 7]; bv[k][e] = (__bf16)b[(k * 3 + e) & 7]; }
    f32x16 acc[8];
    for (int k = 0; k < 8; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    for (int it = 0; it < 2 * iters; ++it)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[k & 3], bv[(k + 1) & 3], acc[k], 0, 0, 0);
    for (int k = 0; k < 8; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  } else {                                    // bf16 16x16x32
    bf16x8 av[4], bv[4];
    for (int k = 0; k < 4; ++k) for (int e = 0; e < 8; ++e) { av[k][e] = (__bf16)a[(k + e) & 7]; bv[k][e] = (__bf16)b[(k * 3 + e) & 7]; }
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < 4 * iters; ++it)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[k & 3], bv[(k + 1) & 3], acc[k], 0, 0, 0);
    for (int k = 0; k < 8; ++k) for (int e = 0; e < 4; ++e) s += acc[k][e];
  }
  out[l] = s;
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256;          // 256 = one wave per SIMD, 512 = two
  float *seed, *out;
  (void)hipMalloc(&seed, 65536 * 4); (void)hipMalloc(&out, (size_t)wgs * 256 * 4);
  float* h = (float*)malloc(65536 * 4);
  srand(1);
  for (int i = 0; i < 65536; ++i) h[i] = (float)(rand() & 0xffff) / 32768.f - 1.f;
  (void)hipMemcpy(seed, h, 65536 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 100000;                                // x 8 MFMAs x 64 cycles = 51 M cycles ~ 25 ms per launch
  const char* names[4] = {"f32 32x32x2 ", "f32 16x16x4 ", "bf16 32x32x16", "bf16 16x16x32"};
  const double flop_per_iter[4] = {8 * 4096.0, 2 * 8 * 2048.0, 2 * 8 * 32768.0, 4 * 8 * 16384.0};
  for (int rep = 0; rep < 2; ++rep)
    for (int sh = 0; sh < 4; ++sh) {
      float ms = 0.f;
      for (int r = 0; r < 6; ++r) {       // back to back: the clock settles
        (void)hipEventRecord(e0);
        if (sh == 0) hipLaunchKernelGGL(loop<0>, dim3(wgs), dim3(256), 0, 0, seed, out, iters);
        if (sh == 1) hipLaunchKernelGGL(loop<1>, dim3(wgs), dim3(256), 0, 0, seed, out, iters);
        if (sh == 2) hipLaunchKernelGGL(loop<2>, dim3(wgs), dim3(256), 0, 0, seed, out, iters);
        if (sh == 3) hipLaunchKernelGGL(loop<3>, dim3(wgs), dim3(256), 0, 0, seed, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
      }
      const double tf = flop_per_iter[sh] * iters * 64.0 /*lanes->wave: per wave */ / 64.0 * (double)wgs * 4 / (ms * 1e-3) / 1e12;
      printf("%s  waves/SIMD %d  %7.2f ms  %8.1f TFLOP/s\n", names[sh], wgs / 256, ms, tf);
    }
  return 0;
}

// v_mfma_f32_4x4x1_16B_f32: operand layout check and issue rate (round 3: would a 64 -> 3 channel 7x7 layer run on it?)
//   block b = lanes 4b .. 4b+3;  D_b[i][j] += A_b[i] * B_b[j];  A_b[i] from lane 4b+i, B_b[j] from lane 4b+j,
//   D_b[i][j] in VGPR i of lane 4b+j  (the assumption to verify)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[l * 4 + i] = acc[i];
}

// rate: NI independent accumulators, operands from registers (REG) or one ds_read_b128 per four MFMAs (LDS)
template <int NACC, bool LDS>
__global__ __launch_bounds__(512) void rate(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float sm[8192];
  const int l = threadIdx.x;
  for (int i = l; i < 8192; i += 512) sm[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[NACC];
  for (int k = 0; k < NACC; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 av = {1.f, 2.f, 3.f, 4.f}, bv = {0.5f, 0.25f, 0.125f, 1.f};
  const float* p = sm + (l & 63) * 20;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
      if (LDS) av = *reinterpret_cast<const f32x4*>(p + ((it + k) & 7) * 1280);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], bv[e], acc[k], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * 512 + l] = s;
}

int main() {
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  std::vector<float> ha(64), hb(64), hd(256);
  for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f * (l + 1); }
  hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int blk = 0; blk < 16; ++blk)
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < 4; ++i) {
        const float want = ha[4 * blk + i] * hb[4 * blk + j], got = hd[(4 * blk + j) * 4 + i];
        if (want != got) { if (bad < 5) printf("mismatch blk %d i %d j %d want %g got %g\n", blk, i, j, want, got); ++bad; }
      }
  printf("layout D_b[i][j] in VGPR i of lane 4b+j: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  auto run = [&](auto kern, const char* name, int nacc) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)iters * nacc * 4;             // per wave
    const double flops = n_mfma * 512.0 * 8 * 256;              // 512 FLOP per instruction, 8 waves, 256 workgroups
    printf("%-28s %8.3f ms  %6.1f TFLOP/s  (%.2f ns per MFMA per wave; 2 waves per SIMD)\n", name, ms, flops / ms / 1e9, ms * 1e6 / n_mfma);
  };
  run(rate<4, false>, "4 accumulators, registers", 4);
  run(rate<8, false>, "8 accumulators, registers", 8);
  run(rate<4, true>, "4 accumulators, b128/4 MFMA", 4);
  run(rate<8, true>, "8 accumulators, b128/4 MFMA", 8);
  return 0;
}

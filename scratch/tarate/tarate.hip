// Texture-path throughput of one CU for the gather patterns of the Winograd kernels: 512-thread workgroups (8 waves),
// one per CU, every wave issues N independent buffer loads back to back; cycles per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PAT 0: dword, 8 pixels x 8 channels per instruction (32-byte runs, pixels 1 KB apart)          [wino_kernel]
// PAT 1: dword, 4 pixels x 16 channels (64-byte runs)                                            [wino_wgrad_kernel]
// PAT 2: dwordx2, 16 pixels x 4 lanes (32-byte runs)
// PAT 3: dwordx4, 8 pixels x 8 lanes (128-byte runs = whole lines)                               [wino43_input_kernel]
// PAT 4: dwordx4, 1 KB contiguous                                                                [wino43_kernel]
// PAT 5: dwordx4, 64 pixels x 16 bytes (every lane its own line)
// PAT 6: dwordx2, 8 pixels x 8 lanes (64-byte runs)
template <int PAT>
__global__ __launch_bounds__(512) void k(const float* src, float* out, int iters, long long* cyc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1u << 30, 0x00020000);
  unsigned off;
  const unsigned pix = 1024;   // bytes per pixel (256 channels)
  const unsigned wbase = (blockIdx.x * 8 + wave) * 64 * pix;
  if (PAT == 0) off = wbase + (lane >> 3) * pix + (lane & 7) * 4;
  if (PAT == 1) off = wbase + (lane & 3) * pix + (lane >> 2) * 4;
  if (PAT == 2) off = wbase + (lane >> 2) * pix + (lane & 3) * 8;
  if (PAT == 3) off = wbase + (lane >> 3) * pix + (lane & 7) * 16;
  if (PAT == 4) off = wbase + lane * 16;
  if (PAT == 5) off = wbase + lane * pix;
  if (PAT == 6) off = wbase + (lane >> 3) * pix + (lane & 7) * 8;
  float acc = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const unsigned so = (it & 15) * 4096 * 16;    // walk through the rows so that lines are not re-used
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (PAT <= 1) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, so + u * 65536 * 4, 0));
      else if (PAT == 2 || PAT == 6) { f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, so + u * 65536 * 4, 0)); acc += v.x + v.y; }
      else { f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, so + u * 65536 * 4, 0)); acc += v.x + v.w; }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && blockIdx.x == 7) cyc[wave] = t1 - t0;
  if (acc == 123.456f) out[tid] = acc;
}

template <int PAT> void run(const char* name, const float* src, float* out, long long* cyc) {
  const int iters = 200;
  k<PAT><<<256, 512>>>(src, out, iters, cyc);
  (void)hipDeviceSynchronize();
  k<PAT><<<256, 512>>>(src, out, iters, cyc);
  (void)hipDeviceSynchronize();
  long long h[8];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  long long mx = 0; for (int i = 0; i < 8; ++i) mx = h[i] > mx ? h[i] : mx;
  printf("%-52s %6.1f cycles per wave-instruction (8 waves -> %5.1f per CU instruction)\n", name, (double)mx / (iters * 16.0), (double)mx / (iters * 16.0 * 8));
}
int main() {
  float *src, *out; long long* cyc;
  (void)hipMalloc(&src, (size_t)1 << 30); (void)hipMemset(src, 0, (size_t)1 << 30);
  (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 64);
  run<0>("dword   8 px x 32 B", src, out, cyc);
  run<1>("dword   4 px x 64 B", src, out, cyc);
  run<2>("dwordx2 16 px x 32 B", src, out, cyc);
  run<6>("dwordx2 8 px x 64 B", src, out, cyc);
  run<3>("dwordx4 8 px x 128 B", src, out, cyc);
  run<4>("dwordx4 1 KB contiguous", src, out, cyc);
  run<5>("dwordx4 64 px x 16 B", src, out, cyc);
  return 0;
}

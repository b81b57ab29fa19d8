// LDS-DMA facts the igemm16 DMA path relies on (gfx950), checked on the GPU:
//   1. buffer_load_dwordx4 ... lds: lane L's 16 bytes land at M0 base + 16 L;
//   2. a lane whose offset is out of range (>= num_records) writes ZEROS to its 16 bytes (not "nothing");
//   3. a counted s_waitcnt vmcnt(N) + s_barrier orders the DMA data for ds_reads of OTHER waves.
//   hipcc --offload-arch=gfx950 -O3 dma_test.hip -o dma_test && ./dma_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const float* src, float* dst, unsigned bytes, int rounds) {
  __shared__ __attribute__((aligned(16))) unsigned char buf[2][4096];
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 2048; i += 256) reinterpret_cast<float*>(buf)[i] = -7.f;      // poison
  __syncthreads();
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    const int b = r & 1;
    // every fourth lane is "masked": out of range
    unsigned voff = (unsigned)((r * 256 + tid) * 16);
    if ((tid & 3) == 3) voff = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(buf[b] + wave * 1024), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // read what ANOTHER wave wrote
    const int t2 = (tid + 64) & 255;
    const f32x4 v = *reinterpret_cast<const f32x4*>(buf[b] + t2 * 16);
    acc += v[0] + v[1] + v[2] + v[3];
    dst[(r * 256 + tid) * 4 + 0] = v[0]; dst[(r * 256 + tid) * 4 + 1] = v[1];
    dst[(r * 256 + tid) * 4 + 2] = v[2]; dst[(r * 256 + tid) * 4 + 3] = v[3];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (acc == 12345.f) dst[0] = acc;
}
int main() {
  const int rounds = 8, n = rounds * 256 * 4;
  std::vector<float> h(n), out(n);
  for (int i = 0; i < n; ++i) h[i] = 1.f + i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, (unsigned)(n * 4), rounds);
  hipMemcpy(out.data(), o, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < rounds; ++r)
    for (int t = 0; t < 256; ++t) {
      const int t2 = (t + 64) & 255;
      for (int e = 0; e < 4; ++e) {
        const float want = (t2 & 3) == 3 ? 0.f : 1.f + (r * 256 + t2) * 4 + e;
        const float got = out[(r * 256 + t) * 4 + e];
        if (got != want && bad++ < 8) printf("round %d thread %d elem %d: got %g want %g\n", r, t, e, got, want);
      }
    }
  printf("dma_test: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
  return bad != 0;
}

// Does buffer_store_dwordx4 with an SGPR scalar offset misplace data when its data registers are rewritten right after it?
// (round 5: wino42_kernel's epilogue wrote dwords 1 and 3 of some lanes with the NEXT store's values; profiles/LOG.md.)
// Each lane stores NP pixels of 16 bytes: value(p, c) = 1000 p + 10 lane + c, computed by VALU instructions into the SAME four
// registers for every p, through (a) voffset only, (b) SGPR soffset.  The host counts mismatches per form.
//   hipcc --offload-arch=gfx950 -O3 bufstore.hip -o bufstore && ./bufstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NP = 16;

template <bool SOFF>
__global__ __launch_bounds__(256) void k(float* dst, int pix_stride_bytes, float scale) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(dst);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  float* q = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(NP * pix_stride_bytes), 0x00020000);
  const int lane = threadIdx.x;
  const unsigned voff = lane * 16;
  float base = 10.f * lane * scale;          // scale = 1: keeps the compiler from folding the values
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    f32x4 v;
    v[0] = __builtin_fmaf(1000.f * p, scale, base + 0.f);
    v[1] = __builtin_fmaf(1000.f * p, scale, base + 1.f);
    v[2] = __builtin_fmaf(1000.f * p, scale, base + 2.f);
    v[3] = __builtin_fmaf(1000.f * p, scale, base + 3.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.f ? v[c] : __builtin_fmaf(v[c], scale, 0.f);
    if (SOFF) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, p * pix_stride_bytes, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff + (unsigned)(p * pix_stride_bytes), 0, 0);
  }
}

int main() {
  const int stride = 256 * 16;               // bytes between pixels: 256 lanes x 16 bytes
  float* d;
  hipMalloc(&d, NP * stride);
  std::vector<float> h(NP * stride / 4);
  for (int form = 0; form < 2; ++form) {
    hipMemset(d, 0xff, NP * stride);
    if (form) hipLaunchKernelGGL(k<true>, dim3(1), dim3(256), 0, 0, d, stride, 1.f);
    else hipLaunchKernelGGL(k<false>, dim3(1), dim3(256), 0, 0, d, stride, 1.f);
    hipMemcpy(h.data(), d, NP * stride, hipMemcpyDeviceToHost);
    int bad = 0, first = -1;
    for (int p = 0; p < NP; ++p)
      for (int l = 0; l < 256; ++l)
        for (int c = 0; c < 4; ++c) {
          const float want = 1000.f * p + 10.f * l + c, got = h[(p * stride + l * 16) / 4 + c];
          if (got != want) { if (first < 0) first = (p * 256 + l) * 4 + c; ++bad; }
        }
    std::printf("%s: %d wrong values%s\n", form ? "SGPR soffset" : "voffset only", bad, bad ? "" : " (all right)");
    if (bad) std::printf("  first wrong: pixel %d lane %d dword %d\n", first / 1024, (first / 4) % 256, first % 4);
  }
  return 0;
}

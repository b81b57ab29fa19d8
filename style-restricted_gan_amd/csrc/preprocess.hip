// GPU side of the reference's input pipeline:
//   CenterCrop(178) -> Resize(128) -> RandomHorizontalFlip -> ToTensor -> MinMax(True)
// (05-train notebook cell 9; MinMax = pyfiles/util.py:108-155) applied to a batch of decoded uint8 RGB images.
//
// Resize on a PIL image is Pillow's antialiased BILINEAR resample (third-party: Pillow, `Resample.c`): a horizontal then a
// vertical pass, each with per-output-pixel windows [xmin, xmin+n) of 22-bit fixed-point coefficients and an 8-bit
// rounded / clipped result in between.  The host (srgan_amd/data.py) builds the window / coefficient tables exactly as
// Pillow's precompute_coeffs + normalize_coeffs_8bpc do; the kernels below repeat its integer arithmetic, so the resized
// bytes are bit-identical.  ToTensor and MinMax are fp32 IEEE operations (correctly rounded division), also exact:
//   x = u / 255,  r = (x - min) / ((max - min) + 1e-8),  out = r * 2 - 1      (min / max over the whole image).
#include "common.h"

namespace srgan {

constexpr int PREC_BITS = 32 - 8 - 2;      // Pillow: PRECISION_BITS

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PREC_BITS;                          // arithmetic shift, as Pillow's clip8 lookup index
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass over the cropped rows: tmp[b][y][xx][c], y in [0, crop_h)
__global__ void prep_resize_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ tmp,
                                     const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int B, int Hs,
                                     int Ws, int top, int left, int crop_h, int out_w) {
  const long long total = (long long)B * crop_h * out_w;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int xx = (int)(idx % out_w);
    long long r = idx / out_w;
    const int y = (int)(r % crop_h);
    const int b = (int)(r / crop_h);
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + xx * ksize;
    const unsigned char* row = src + (((size_t)b * Hs + top + y) * Ws + left + xmin) * 3;
    int s0 = 1 << (PREC_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
      const int w = k[x];
      s0 += row[3 * x + 0] * w;
      s1 += row[3 * x + 1] * w;
      s2 += row[3 * x + 2] * w;
    }
    unsigned char* o = tmp + idx * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
  }
}

// vertical pass: u8[b][yy][xx][c]; per-image min / max of the bytes (exact: u -> u/255 is monotonic)
__global__ void prep_resize_v_kernel(const unsigned char* __restrict__ tmp, unsigned char* __restrict__ u8,
                                     const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int B,
                                     int crop_h, int out_h, int out_w, int* __restrict__ minmax) {
  const int b = blockIdx.y;
  const int per = out_h * out_w;
  int lo = 255, hi = 0;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < per; idx += gridDim.x * blockDim.x) {
    const int xx = idx % out_w, yy = idx / out_w;
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int* k = kk + yy * ksize;
    const unsigned char* col = tmp + (((size_t)b * crop_h + ymin) * out_w + xx) * 3;
    int s0 = 1 << (PREC_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
      const int w = k[y];
      const unsigned char* p = col + (size_t)y * out_w * 3;
      s0 += p[0] * w;
      s1 += p[1] * w;
      s2 += p[2] * w;
    }
    const int v0 = clip8(s0), v1 = clip8(s1), v2 = clip8(s2);
    unsigned char* o = u8 + ((size_t)b * per + idx) * 3;
    o[0] = (unsigned char)v0; o[1] = (unsigned char)v1; o[2] = (unsigned char)v2;
    lo = min(lo, min(v0, min(v1, v2)));
    hi = max(hi, max(v0, max(v1, v2)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o, 64));
    hi = max(hi, __shfl_xor(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&minmax[2 * b], lo);
    atomicMax(&minmax[2 * b + 1], hi);
  }
}

__global__ void prep_minmax_init_kernel(int* minmax, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { minmax[2 * i] = 255; minmax[2 * i + 1] = 0; }
}

// flip + ToTensor + MinMax -> dst[b][y][x][c] fp32 (NHWC)
__global__ void prep_normalise_kernel(const unsigned char* __restrict__ u8, const int* __restrict__ minmax,
                                      const unsigned char* __restrict__ flip, float* __restrict__ dst, int B, int out_h,
                                      int out_w, int minmax_on, int mean0) {
  const long long total = (long long)B * out_h * out_w * 3;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % 3);
    long long r = idx / 3;
    const int x = (int)(r % out_w); r /= out_w;
    const int y = (int)(r % out_h);
    const int b = (int)(r / out_h);
    const int sx = (flip && flip[b]) ? out_w - 1 - x : x;
    const float v = (float)u8[(((size_t)b * out_h + y) * out_w + sx) * 3 + c] / 255.0f;     // ToTensor
    float o = v;
    if (minmax_on) {
      const float lo = (float)minmax[2 * b] / 255.0f, hi = (float)minmax[2 * b + 1] / 255.0f;
      o = (v - lo) / ((hi - lo) + 1e-8f);
      if (mean0) o = o * 2.0f - 1.0f;
    }
    dst[idx] = o;
  }
}

}  // namespace srgan

using namespace srgan;

extern "C" size_t srgan_preprocess_workspace(int B, int crop_h, int out_h, int out_w) {
  if (B <= 0 || crop_h <= 0 || out_h <= 0 || out_w <= 0) return 0;
  const size_t tmp = (size_t)B * crop_h * out_w * 3, u8 = (size_t)B * out_h * out_w * 3;
  return round_up((long long)tmp, 256) + round_up((long long)u8, 256) + round_up((long long)B * 2 * sizeof(int), 256);
}

extern "C" int srgan_preprocess_u8(const unsigned char* src, int B, int Hs, int Ws, int top, int left, int crop_h, int crop_w,
                                   int out_h, int out_w, const int* h_bounds, const int* h_coeffs, int h_ksize,
                                   const int* v_bounds, const int* v_coeffs, int v_ksize, const unsigned char* flip,
                                   int minmax, int mean0, float* dst, void* ws, size_t ws_bytes, void* stream) {
  SRGAN_REQUIRE(src && dst && h_bounds && h_coeffs && v_bounds && v_coeffs && ws, "preprocess: null pointer");
  SRGAN_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && crop_h > 0 && crop_w > 0 && out_h > 0 && out_w > 0, "preprocess: bad shape");
  SRGAN_REQUIRE(top >= 0 && left >= 0 && top + crop_h <= Hs && left + crop_w <= Ws, "preprocess: crop window outside the image");
  SRGAN_REQUIRE(h_ksize > 0 && v_ksize > 0, "preprocess: empty coefficient tables");
  SRGAN_REQUIRE(ws_bytes >= srgan_preprocess_workspace(B, crop_h, out_h, out_w), "preprocess: workspace too small");
  hipStream_t st = as_stream(stream);
  unsigned char* tmp = (unsigned char*)ws;
  unsigned char* u8 = tmp + round_up((long long)B * crop_h * out_w * 3, 256);
  int* mm = (int*)(u8 + round_up((long long)B * out_h * out_w * 3, 256));
  hipLaunchKernelGGL(prep_minmax_init_kernel, dim3((B + 255) / 256), dim3(256), 0, st, mm, B);
  const long long nh = (long long)B * crop_h * out_w;
  hipLaunchKernelGGL(prep_resize_h_kernel, dim3((unsigned)std::min<long long>(ceil_div(nh, 256), 8192)), dim3(256), 0, st, src, tmp,
                     h_bounds, h_coeffs, h_ksize, B, Hs, Ws, top, left, crop_h, out_w);
  hipLaunchKernelGGL(prep_resize_v_kernel, dim3((unsigned)std::min<long long>(ceil_div(out_h * out_w, 256), 64), (unsigned)B),
                     dim3(256), 0, st, tmp, u8, v_bounds, v_coeffs, v_ksize, B, crop_h, out_h, out_w, mm);
  const long long nt = (long long)B * out_h * out_w * 3;
  hipLaunchKernelGGL(prep_normalise_kernel, dim3((unsigned)std::min<long long>(ceil_div(nt, 256), 8192)), dim3(256), 0, st, u8, mm,
                     flip, dst, B, out_h, out_w, minmax, mean0);
  return check_launch("preprocess");
}

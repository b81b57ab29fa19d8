// Stride-1 convolutions with THREE input channels on the exact-fp32 MFMA: the generator's 7x7 RGB input layer
// (reference pyfiles/model.py:212, `nn.Conv2d(3, 64, kernel_size=7, stride=1, padding=3)`) and -- with the flipped,
// transposed filter -- the input gradient of its 7x7 RGB output layer (model.py:232).
//
// The implicit GEMM serves 3-channel inputs through a scalar gather: K = 147 (ky, kx, c) entries per output pixel, one
// buffer load and ~10 address operations each, and on this chip vector work inside a multiply loop is ADDED to the MFMA
// time: 48 TFLOP/s.  Here the gather disappears:
//   * a workgroup owns 16 x 32 output pixels x 64 output channels; the (16+6) x (32+6) x 3 input halo (10 KB) and the
//     64-channel filter block (39 KB) are staged ONCE in LDS,
//   * K is laid out (ky, 22-entry row): a row holds the 21 (kx, c) taps plus one zero-weight pad, so the two k values of
//     an MFMA step never straddle rows and the halo element of (pixel, k) sits at  pixel base + compile-time offset:
//     the A fragment is one ds_read_b32 with an immediate offset, no address arithmetic at all,
//   * wave w multiplies rows 2w, 2w+1 of the tile (2 x 32 pixels) by the 64 channels: 4 accumulators, 77 steps of
//     2 + 2 LDS reads and 4 MFMAs, fully unrolled,
//   * an accumulator lane is an output CHANNEL: every store instruction writes 2 pixels x 32 contiguous channels.
#include <algorithm>
#include <cstdlib>
#include "common.h"

namespace srgan {

struct RgbinParams {
  const float* x;     // [NB][H][W][3]
  const float* wp;    // [KH * KR][O]
  const float* bias;  // [O] or null
  float* y;           // [NB][Ho][Wo][O]
  int NB, H, W, Ho, Wo, O, pad;
  int tiles_x, tiles_y, o_blocks;
  int act;
  float slope;
};

template <int KH, int KW, int CI>
__global__ __launch_bounds__(512) void rgbin_conv_kernel(RgbinParams p) {
  constexpr int TR = 16, TC = 32;
  constexpr int HR = TR + KH - 1, HC = TC + KW - 1, HW = HC * CI;      // halo rows, columns, floats per halo row
  constexpr int KR = (KW * CI + 1) & ~1, KP = KH * KR;                 // K entries per filter row (even), in total
  __shared__ float halo[HR * HW + 8];
  __shared__ float wl[KP * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y; b /= p.tiles_y;
  const int ob = b % p.o_blocks, n = b / p.o_blocks;
  const int X0 = tx * TC, Y0 = ty * TR;

  for (int e = tid; e < KP * 64; e += 512) wl[e] = p.wp[(size_t)(e >> 6) * p.O + ob * 64 + (e & 63)];
  {
    const float* img = p.x + (size_t)n * p.H * p.W * CI;
    const int gx0 = (X0 - p.pad) * CI;
    for (int e = tid; e < HR * HW + 8; e += 512) {
      const int r = e / HW, j = e - r * HW;
      const int gy = Y0 - p.pad + r, gx = gx0 + j;
      float v = 0.f;
      if (r < HR && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)(p.W * CI)) v = img[(size_t)gy * p.W * CI + gx];
      halo[e] = v;
    }
  }
  __syncthreads();

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // A: lane (lr, lh) = pixel column lr, k parity lh;  B: lane = output channel lr, k parity lh
  const float* a0 = halo + (2 * wave) * HW + lr * CI + lh;
  const float* a1 = a0 + HW;
  const float* bw = wl + lh * 64 + lr;
#pragma unroll
  for (int ky = 0; ky < KH; ++ky)
#pragma unroll
    for (int s = 0; s < KR / 2; ++s) {
      const float va0 = a0[ky * HW + 2 * s], va1 = a1[ky * HW + 2 * s];
      const float vb0 = bw[(ky * KR + 2 * s) * 64], vb1 = bw[(ky * KR + 2 * s) * 64 + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va0, vb0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va0, vb1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va1, vb0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(va1, vb1, acc[1][1], 0, 0, 0);
    }

  // D[i = pixel column][j = channel]: lane = channel lr, register e = pixel column 8 * (e / 4) + 4 * lh + e % 4
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = Y0 + 2 * wave + i;
    if (oy >= p.Ho) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int oc = ob * 64 + j * 32 + lr;
      const float bv = p.bias ? p.bias[oc] : 0.f;
      float* row = p.y + ((size_t)(n * p.Ho + oy) * p.Wo) * p.O + oc;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ox = X0 + 8 * (e >> 2) + 4 * lh + (e & 3);
        if (ox < p.Wo) row[(size_t)ox * p.O] = apply_act(acc[i][j][e] + bv, p.act, p.slope);
      }
    }
  }
}

// packed filter: wp[(ky * KR + kx * CI + c)][o] = w[o][c][ky][kx] (through the weight strides), zero in the pad entry
__global__ void rgbin_pack_kernel(const float* w, float* dst, long long sO, long long sI, long long sH, long long sW, int O,
                                  int CI, int KH, int KW) {
  const int KR = (KW * CI + 1) & ~1;
  const long long total = (long long)KH * KR * O;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(idx % O);
    const int k = (int)(idx / O);
    const int ky = k / KR, j = k - ky * KR;
    float v = 0.f;
    if (j < KW * CI) {
      const int kx = j / CI, c = j - kx * CI;
      v = w[o * sO + c * sI + ky * sH + kx * sW];
    }
    dst[idx] = v;
  }
}

bool rgbin_applicable(const srgan_conv_desc* d) {
  static const bool off = std::getenv("SRGAN_NO_RGBIN") != nullptr;
  if (off || compute_bf16()) return false;
  return d->I == 3 && d->kh == 7 && d->kw == 7 && d->stride == 1 && d->pad_mode == SRGAN_PAD_ZERO && d->O % 64 == 0 &&
         d->Wo >= 32 && d->Ho >= 16 && (long long)d->N * d->Ho * d->Wo * d->O < (1LL << 31);
}

size_t rgbin_packed_elems(const srgan_conv_desc* d) { return (size_t)d->kh * ((d->kw * d->I + 1) & ~1) * d->O; }

int rgbin_pack(const srgan_conv_desc* d, const float* w, float* dst, hipStream_t st) {
  const long long total = (long long)rgbin_packed_elems(d);
  hipLaunchKernelGGL(rgbin_pack_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 1024)), dim3(256), 0, st, w, dst,
                     d->sO, d->sI, d->sH, d->sW, d->O, d->I, d->kh, d->kw);
  return check_launch("rgbin_pack_kernel");
}

int rgbin_run(const srgan_conv_desc* d, const float* x, const float* packed, const float* bias, float* y, int act, float slope,
              hipStream_t st) {
  SRGAN_REQUIRE(rgbin_applicable(d), "rgb-input conv: layer not applicable");
  RgbinParams p{};
  p.x = x; p.wp = packed; p.bias = bias; p.y = y;
  p.NB = d->N; p.H = d->Hi; p.W = d->Wi; p.Ho = d->Ho; p.Wo = d->Wo; p.O = d->O; p.pad = d->pad;
  p.tiles_x = (int)ceil_div(d->Wo, 32); p.tiles_y = (int)ceil_div(d->Ho, 16); p.o_blocks = d->O / 64;
  p.act = act; p.slope = slope;
  const long long grid = (long long)p.tiles_x * p.tiles_y * p.o_blocks * d->N;
  SRGAN_REQUIRE(grid < (1LL << 31), "rgb-input conv: grid too large");
  ProfToken tok = prof_begin(20, 2.0 * d->N * d->Ho * d->Wo * (double)d->O * d->kh * d->kw * d->I, st);
  hipLaunchKernelGGL((rgbin_conv_kernel<7, 7, 3>), dim3((unsigned)grid), dim3(512), 0, st, p);
  prof_end(tok, st);
  return check_launch("rgbin_conv_kernel");
}

}  // namespace srgan

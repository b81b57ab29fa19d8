// The 7x7 / stride-1 / zero-pad-3 convolutions with THREE output channels on 128x128-class maps: the generator's RGB output layer
// (pyfiles/model.py:232, 247-248: conv7(64 -> 3) before tanh) and -- with the flipped, transposed filter -- the input gradient of
// its 7x7 RGB input layer (model.py:212).  Round 3.
//
// Every 32-wide MFMA formulation pads the 3 output channels to 32 (the 7x1 row convolution + shift-add of conv_igemm.hip gets to
// 21 / 32 by making (cout, kx) the N index, at the price of a 21-float-per-pixel intermediate through HBM: 1.63 GB per launch).
// v_mfma_f32_4x4x1_16B_f32 is the matrix instruction with a 4-wide N: 16 independent 4x4 blocks per wave,
//     D_b[i][j] += A_b[i] * B_b[j],   A_b[i] from lane 4b + i,  B_b[j] from lane 4b + j,  D_b[i][j] in VGPR i of lane 4b + j
// (layout and rate measured: scratch/mfma4/mfma4.hip -- 8 cycles per instruction, the same 64 FLOP per cycle and SIMD as the
// 32x32x2 form).  Here block b = four consecutive output pixels of a row, j = output channel (3 of 4 used: 75 % of the pipe),
// one instruction per reduce index (ky, kx, c): a DIRECT convolution, nothing but x, w and y ever leaves the chip.
//
//   workgroup (8 waves) = 32 rows x 64 columns of one image; lane = pixel column, wave w = rows 4w .. 4w + 3 (4 accumulators
//   of 4 registers);
//   the reduce channels are walked in quads: the (32 + 6) x (64 + 6) halo of ONE channel quad is 16 bytes per pixel in LDS
//   (42.5 KB, double-buffered, consecutive lanes read consecutive 16-byte slots: conflict-free ds_read_b128), so the A operands of
//   the four instructions (pixel, tap, c .. c + 3) are one read; the whole packed filter (7 x 7 x C x 4 floats = 50 KB at C = 64)
//   sits in LDS as [ky][quad][kx][cout][4 c] and the B operands of a (ky, quad) -- 7 reads -- stay in registers for the wave's
//   four rows: 5 LDS reads per 16 instructions;
//   global loads of the next quad's halo are issued before the current quad's 784 instructions per wave and written to the other
//   buffer after them: one barrier per quad.
// 128x128 maps at batch 32 are 256 workgroups: one round.
#include <algorithm>
#include <cstdlib>
#include "common.h"

namespace srgan {
namespace {

constexpr int RO_TR = 32, RO_TC = 64, RO_K = 7, RO_PAD = 3;
constexpr int RO_HR = RO_TR + RO_K - 1, RO_HC = RO_TC + RO_K - 1;      // 38 x 70 halo pixels
constexpr int RO_HALO = RO_HR * RO_HC;                                  // 2660 pixels = float4 slots per buffer
constexpr int RO_LOADS = (RO_HALO + 511) / 512;                         // halo pixels per thread and quad (6)
constexpr int RO_MAXQ = 16;                                             // up to 64 reduce channels (the filter must fit LDS)

struct RgboutParams {
  const float* x;      // [NB][H][W][C]
  const float* wp;     // [ky][quad][kx][4 couts][4 c]
  const float* bias;   // [O] or null
  float* y;            // [NB][Ho][Wo][O]
  int NB, H, W, C, Ho, Wo, O, ncq, tiles_x, tiles_y;
};

__global__ __launch_bounds__(512) void rgbout_conv_kernel(RgboutParams p) {
  __shared__ f32x4 halo[2 * RO_HALO];                                   // [2][38 x 70 pixels] x 16 bytes = 85 KB
  __shared__ __attribute__((aligned(16))) float wl[RO_K * RO_MAXQ * RO_K * 16];      // [ky][quad][kx][4 couts][4 c]: 50 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int n = b / p.tiles_y;
  const int X0 = tx * RO_TC, Y0 = ty * RO_TR;

  const int nw = RO_K * p.ncq * RO_K * 4;                               // float4 count of the packed filter
  for (int e = tid; e < nw; e += 512) reinterpret_cast<f32x4*>(wl)[e] = reinterpret_cast<const f32x4*>(p.wp)[e];

  // this thread's halo pixels: byte-free offsets of channel quad 0, or -1 outside the image (zero padding)
  long long hoff[RO_LOADS];
#pragma unroll
  for (int j = 0; j < RO_LOADS; ++j) {
    const int hp = tid + 512 * j;
    const int r = hp / RO_HC, c = hp - r * RO_HC;
    const int gy = Y0 - RO_PAD + r, gx = X0 - RO_PAD + c;
    const bool ok = hp < RO_HALO && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    hoff[j] = ok ? (((long long)n * p.H + gy) * p.W + gx) * p.C : -1;
  }
  f32x4 stage[RO_LOADS];
  auto load_quad = [&](int cq) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < RO_LOADS; ++j)
      stage[j] = hoff[j] >= 0 ? *reinterpret_cast<const f32x4*>(p.x + hoff[j] + cq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_quad = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < RO_LOADS; ++j) {
      const int hp = tid + 512 * j;
      if (hp < RO_HALO) halo[buf * RO_HALO + hp] = stage[j];
    }
  };

  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_quad(0);
  store_quad(0);
  __syncthreads();
  const int co4 = (lane & 3) * 4;
  for (int cq = 0; cq < p.ncq; ++cq) {
    if (cq + 1 < p.ncq) load_quad(cq + 1);
    const f32x4* H = halo + (cq & 1) * RO_HALO + (4 * wave) * RO_HC + lane;
#pragma unroll 1
    for (int ky = 0; ky < RO_K; ++ky) {
      f32x4 B[RO_K];
      const float* wq = wl + ((ky * p.ncq + cq) * RO_K) * 16 + co4;
#pragma unroll
      for (int kx = 0; kx < RO_K; ++kx) B[kx] = *reinterpret_cast<const f32x4*>(wq + kx * 16);
#pragma unroll
      for (int kx = 0; kx < RO_K; ++kx) {
        f32x4 A[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) A[t] = H[(t + ky) * RO_HC + kx];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[t][e], B[kx][e], acc[t], 0, 0, 0);
      }
    }
    if (cq + 1 < p.ncq) store_quad((cq + 1) & 1);
    __syncthreads();
  }

  // D[pixel 4b + i][cout j] sits in register i of lane 4b + j: lane = (pixel group, cout)
  const int co = lane & 3, pg = lane >> 2;
  if (co < p.O) {
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int oy = Y0 + 4 * wave + t;
      if (oy >= p.Ho) continue;
      float* row = p.y + (((size_t)n * p.Ho + oy) * p.Wo) * p.O + co;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ox = X0 + 4 * pg + i;
        if (ox < p.Wo) row[(size_t)ox * p.O] = acc[t][i] + bv;
      }
    }
  }
}

// packed filter [ky][quad][kx][4 couts][4 c] = w[cout][4 quad + c][ky][kx] through the weight strides (zero for cout >= O)
__global__ void rgbout_pack_kernel(const float* w, float* dst, long long sO, long long sI, long long sH, long long sW, int O, int ncq) {
  const int total = RO_K * ncq * RO_K * 16;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int e = idx & 3, co = (idx >> 2) & 3;
    int r = idx >> 4;
    const int kx = r % RO_K; r /= RO_K;
    const int cq = r % ncq;
    const int ky = r / ncq;
    dst[idx] = co < O ? w[co * sO + (cq * 4 + e) * sI + ky * sH + kx * sW] : 0.f;
  }
}

}  // namespace

bool rgbout_applicable(const srgan_conv_desc* d) {
  static const bool off = std::getenv("SRGAN_NO_RGBOUT") != nullptr;
  if (off) return false;
  return d->O >= 1 && d->O <= 4 && d->kh == RO_K && d->kw == RO_K && d->stride == 1 && d->pad == RO_PAD && d->pad_mode == SRGAN_PAD_ZERO &&
         d->I % 4 == 0 && d->I >= 16 && d->I <= 4 * RO_MAXQ && d->Wo >= 64 && d->Ho >= 32 && d->Hi == d->Ho &&
         d->Wi == d->Wo && (long long)d->N * d->Hi * d->Wi * d->I < (1LL << 31);
}

size_t rgbout_packed_elems(const srgan_conv_desc* d) { return (size_t)RO_K * (d->I / 4) * RO_K * 16; }

int rgbout_pack(const srgan_conv_desc* d, const float* w, float* dst, hipStream_t st) {
  const int total = (int)rgbout_packed_elems(d);
  hipLaunchKernelGGL(rgbout_pack_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 256)), dim3(256), 0, st, w, dst, d->sO, d->sI, d->sH,
                     d->sW, d->O, d->I / 4);
  return check_launch("rgbout_pack_kernel");
}

int rgbout_run(const srgan_conv_desc* d, const float* x, const float* packed, const float* bias, float* y, hipStream_t st) {
  SRGAN_REQUIRE(rgbout_applicable(d), "rgb-output conv: layer not applicable");
  RgboutParams p{};
  p.x = x; p.wp = packed; p.bias = bias; p.y = y;
  p.NB = d->N; p.H = d->Hi; p.W = d->Wi; p.C = d->I; p.Ho = d->Ho; p.Wo = d->Wo; p.O = d->O; p.ncq = d->I / 4;
  p.tiles_x = (int)ceil_div(d->Wo, RO_TC); p.tiles_y = (int)ceil_div(d->Ho, RO_TR);
  const long long grid = (long long)p.tiles_x * p.tiles_y * d->N;
  SRGAN_REQUIRE(grid < (1LL << 31), "rgb-output conv: grid too large");
  ProfToken tok = prof_begin(28, 2.0 * d->N * d->Ho * d->Wo * (double)d->O * d->kh * d->kw * d->I, st);
  hipLaunchKernelGGL(rgbout_conv_kernel, dim3((unsigned)grid), dim3(512), 0, st, p);
  prof_end(tok, st);
  return check_launch("rgbout_conv_kernel");
}

}  // namespace srgan

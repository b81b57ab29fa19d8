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

// S (round 6): stride 1 or 2 -- the discriminator's first layer (nn.Conv2d(3, 64, 4, 2, 1), model.py:302) and the encoder's
// (nn.Conv2d(3, 64, 7, 2, 1), model.py:445) ran on the scalar-gather implicit GEMM at ~1 TB/s of their own OUTPUT (70 us for a
// result that takes 20 us to write); the same tile with a (S TR + KH - S) x (S TC + KW - S) halo and pixels S entries apart
template <int KH, int KW, int CI, int S = 1>
__global__ __launch_bounds__(512) void rgbin_conv_kernel(RgbinParams p) {
  constexpr int TR = 16, TC = 32;
  constexpr int HR = S * (TR - 1) + KH, HC = S * (TC - 1) + KW, HW = HC * CI;      // halo rows, columns, floats per halo row
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
    const int gx0 = (S * X0 - p.pad) * CI;
    for (int e = tid; e < HR * HW + 8; e += 512) {
      const int r = e / HW, j = e - r * HW;
      const int gy = S * Y0 - p.pad + r, gx = gx0 + j;
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
  const float* a0 = halo + (S * 2 * wave) * HW + lr * (S * CI) + lh;
  const float* a1 = a0 + S * HW;
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

// ---- bf16 compute mode (round 4): the same layers on v_mfma_f32_32x32x16_bf16 ----
// In the bf16 mode these layers used to stay on the fp32 kernel above: 108 us per launch at batch 32 for a result that takes 25 us
// to write.  Here the halo is parked as bf16 with FOUR channels per pixel (RGB + a zero), so that the 8 consecutive k of a lane's
// fragment -- two neighbouring pixels x 4 channels -- are 16 contiguous bytes at an 8-byte aligned address (two ds_read_b64;
// with 3 channels per pixel they would start at odd multiples of 2 bytes); K is laid out (ky, 32-entry row): 7 (kx) x 4 (c)
// taps + 4 zero-weight entries, two 16-deep K steps per filter row, 14 steps in all (the fp32 kernel: 77 steps of 2).  The filter
// block lives in LDS as [64 output channels][232] bf16 (row stride 29 x 16 bytes: conflict-free ds_read_b128 fragments).  Same
// tile, wave roles and epilogue as above; the products are 11x cheaper, what is left is the write of the result.
constexpr int R16_HC = 40;            // halo columns: 32 + 6 + 2 (the zero-weight entries read one pixel further)
constexpr int R16_KPS = 232;          // bf16 per packed filter row: 7 x 32 + 8 pad

struct Rgbin16Params {
  const float* x;            // [NB][H][W][3]
  const unsigned short* wp;  // [O][R16_KPS] bf16
  const float* bias;         // [O] or null
  float* y;                  // [NB][Ho][Wo][O] fp32, or bf16 (OUT16)
  int NB, H, W, Ho, Wo, O, pad;
  int tiles_x, tiles_y, o_blocks;
  int act;
  float slope;
};

// OUT16 (round 6): the 64-channel result is written as bf16 (16-bit activation storage: the generator's first norm reads it, or --
// the input gradient of the RGB output layer -- its last norm's backward); lanes pair up so that a half-wave writes a pixel's
// whole 128-byte line (even lane: channels (lr, lr + 1) of block 0, odd lane: (32 + lr - 1, 32 + lr) of block 1).
template <bool OUT16>
__global__ __launch_bounds__(512) void rgbin16_conv_kernel(Rgbin16Params p) {
  constexpr int TR = 16, TC = 32, KH = 7, HR = TR + KH - 1;
  __shared__ __attribute__((aligned(16))) unsigned short halo[HR * R16_HC * 4 + 32];
  __shared__ __attribute__((aligned(16))) unsigned short wl[64 * R16_KPS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y; b /= p.tiles_y;
  const int ob = b % p.o_blocks, n = b / p.o_blocks;
  const int X0 = tx * TC, Y0 = ty * TR;

  {
    const f32x4* src = reinterpret_cast<const f32x4*>(p.wp + (size_t)ob * 64 * R16_KPS);
    f32x4* dst = reinterpret_cast<f32x4*>(wl);
    for (int e = tid; e < 64 * R16_KPS / 8; e += 512) dst[e] = src[e];
  }
  {
    const float* img = p.x + (size_t)n * p.H * p.W * 3;
    for (int e = tid; e < HR * R16_HC + 8; e += 512) {
      const int r = e / R16_HC, c = e - r * R16_HC;
      const int gy = Y0 - p.pad + r, gx = X0 - p.pad + c;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < HR && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
        const float* q = img + ((size_t)gy * p.W + gx) * 3;
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
      }
      *reinterpret_cast<bf16x4*>(&halo[e * 4]) = __builtin_convertvector(v, bf16x4);
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
  // A: lane (lr, lh) = pixel column lr, k = 8 lh .. 8 lh + 7 of the step = pixels lr + 4 s + 2 lh, + 1 of filter row ky
  const unsigned short* a0 = halo + ((2 * wave) * R16_HC + lr + 2 * lh) * 4;
  const unsigned short* bw = wl + lr * R16_KPS + 8 * lh;
#pragma unroll
  for (int ky = 0; ky < KH; ++ky)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned short* a = a0 + ((i + ky) * R16_HC + 4 * s) * 4;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(a), hi = *reinterpret_cast<const bf16x4*>(a + 4);
        fa[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bw + j * 32 * R16_KPS + ky * 32 + 16 * s);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }

#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = Y0 + 2 * wave + i;
    if (oy >= p.Ho) continue;
    if constexpr (OUT16) {
      const bool odd = lr & 1;
      const float bv0 = p.bias ? p.bias[ob * 64 + lr] : 0.f, bv1 = p.bias ? p.bias[ob * 64 + 32 + lr] : 0.f;
      __bf16* row = reinterpret_cast<__bf16*>(p.y) + ((size_t)(n * p.Ho + oy) * p.Wo) * p.O + ob * 64 + (odd ? 32 + lr - 1 : lr);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v0 = apply_act(acc[i][0][e] + bv0, p.act, p.slope), v1 = apply_act(acc[i][1][e] + bv1, p.act, p.slope);
        const float got = __shfl_xor(odd ? v0 : v1, 1);
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 pr;
        pr[0] = (__bf16)(odd ? got : v0);
        pr[1] = (__bf16)(odd ? v1 : got);
        const int ox = X0 + 8 * (e >> 2) + 4 * lh + (e & 3);
        if (ox < p.Wo) *reinterpret_cast<bf16x2*>(row + (size_t)ox * p.O) = pr;
      }
      continue;
    }
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

// bf16 packed filter: wp[o][ky * 32 + kx * 4 + c] = w[o][c][ky][kx] (through the weight strides), zero elsewhere
__global__ void rgbin16_pack_kernel(const float* w, unsigned short* dst, long long sO, long long sI, long long sH, long long sW, int O) {
  const long long total = (long long)O * R16_KPS;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(idx / R16_KPS), k = (int)(idx - (long long)o * R16_KPS);
    const int ky = k >> 5, kx = (k & 31) >> 2, c = k & 3;
    float v = 0.f;
    if (ky < 7 && kx < 7 && c < 3) v = w[o * sO + c * sI + ky * sH + kx * sW];
    reinterpret_cast<__bf16*>(dst)[idx] = (__bf16)v;
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

// ---- weight gradient of the 7x7 stride-1 layers between a 3-channel and a 64-channel tensor ----
//   R[ch][(ky, kx, c)] = sum over pixels  C64[pixel][ch] * T3[pixel + (ky - 3, kx - 3)][c]
// SWAP 0: the RGB INPUT layer (model.py:212): C64 = dy, T3 = x, dW[o = ch][c][ky][kx] = R.
// SWAP 1: the RGB OUTPUT layer (model.py:232): C64 = x, T3 = dy and the offsets change sign:
//         dW[o = c][i = ch][ky][kx] = R[ch][(6 - ky, 6 - kx, c)].
// A GEMM with M = 64 channels, N = 154 (-> 160) taps laid out as in the forward kernel, K = pixels.  A workgroup walks over
// 16 x 32 pixel tiles (persistent, one per CU); the T3 halo of a tile sits in LDS (double-buffered, one barrier per tile) and
// the B fragment of (pixel, tap) is halo[pixel base (immediate) + tap offset (one register per 32-tap tile)]; the A fragment
// (32 channels of two pixels) comes straight from global memory, 128 contiguous bytes per pixel, reloaded in place one
// 32-pixel row ahead.  Every wave accumulates ALL 2 x 5 output tiles (160 registers) over its two rows of each tile; at
// the end the eight partial results meet in LDS and one slab per workgroup goes to the split-K reduce of conv_igemm.hip.
struct RgbWgradParams {
  const float* c64;   // [NB][H][W][64]
  const float* t3;    // [NB][H][W][3]
  float* slab;        // [gridDim.x][Cdpad][NNpad]
  int NB, H, W, tiles_x, tiles_y, ntiles;
};

template <int SWAP>
__global__ __launch_bounds__(512) void rgb_wgrad_kernel(RgbWgradParams p) {
  constexpr int KH = 7, KW = 7, CI = 3, TR = 16, TC = 32;
  constexpr int HR = TR + KH - 1, HC = TC + KW - 1, HW = HC * CI, HSZ = HR * HW + 8;
  constexpr int KR = 22, KP = KH * KR;      // 154 taps, padded to 5 x 32
  __shared__ float lds[8 * 1024];           // 2 halo buffers (2 x 2516 floats); the final reduction reuses all 32 KB
  float* halo = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;

  // B role: lane = tap 32 nt + lr, pixel parity lh
  int tapoff[5];
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    const int k = nt * 32 + lr;
    const int ky = k / KR, j = k - ky * KR;
    tapoff[nt] = (k < KP && j < KW * CI) ? ky * HW + j + lh * CI : lh * CI;      // pad taps read a valid slot, their column is dropped
  }
  f32x16 acc[2][5];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.c64), 0, (unsigned)((size_t)p.NB * p.H * p.W * 64 * 4), 0x00020000);
  // A role: lane = channel lr (+32 for the second row tile), pixel parity lh; a unit = one 32-pixel row of a tile
  float areg[2][16];
  const unsigned alane = (unsigned)(lh * 64 + lr) * 4u;
  auto unit_base = [&](int t, int u) -> unsigned {          // byte offset of pixel (row 2 wave + u, column 0) of tile t
    const int tx = t % p.tiles_x;
    int r = t / p.tiles_x;
    const int ty = r % p.tiles_y, n = r / p.tiles_y;
    return (unsigned)(((n * p.H + ty * TR + 2 * wave + u) * p.W + tx * TC) * 64) * 4u;
  };
  auto load_a = [&](unsigned base, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < 2; ++it)
      areg[it][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_c, alane + it * 128, base + s * 512, 0));
  };
  auto fill_halo = [&](int t, int buf) __attribute__((always_inline)) {
    const int tx = t % p.tiles_x;
    int r = t / p.tiles_x;
    const int ty = r % p.tiles_y, n = r / p.tiles_y;
    const float* img = p.t3 + (size_t)n * p.H * p.W * CI;
    const int gx0 = (tx * TC - 3) * CI, gy0 = ty * TR - 3;
    float* h = halo + buf * HSZ;
    for (int e = tid; e < HSZ; e += 512) {
      const int rr = e / HW, j = e - rr * HW;
      const int gy = gy0 + rr, gx = gx0 + j;
      float v = 0.f;
      if (rr < HR && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)(p.W * CI)) v = img[(size_t)gy * p.W * CI + gx];
      h[e] = v;
    }
  };

  int t = blockIdx.x;
  if (t < p.ntiles) {
    fill_halo(t, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) load_a(unit_base(t, 0), s);
  }
  int buf = 0;
  for (; t < p.ntiles; t += gridDim.x) {
    __syncthreads();                                 // halo[buf] filled; everyone is done with halo[buf ^ 1]
    const int tn = t + gridDim.x;
    if (tn < p.ntiles) fill_halo(tn, buf ^ 1);
    const float* h = halo + buf * HSZ + (2 * wave) * HW;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // the unit after this one: the second row of the tile, or the first row of the next tile (after the last tile the
      // same row is loaded once more and never used: no branch inside the unrolled steps)
      const unsigned nbase = u == 0 ? unit_base(t, 1) : unit_base(tn < p.ntiles ? tn : t, 0);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        float vb[5];
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) vb[nt] = h[u * HW + 2 * s * CI + tapoff[nt]];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
          for (int nt = 0; nt < 5; ++nt)
            acc[it][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[it][s], vb[nt], acc[it][nt], 0, 0, 0);
        load_a(nbase, s);                            // in place: lands a whole row later
      }
    }
    buf ^= 1;
  }

  // ---- the eight partial results meet in LDS, one 32 x 32 output tile at a time ----
  float* slab = p.slab + (size_t)blockIdx.x * (SWAP ? 4 * (KH * KW * 64) : 64 * (KH * KW * CI));
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
      __syncthreads();
      // D[i = channel][j = tap]: lane = tap lr, register e = channel 8 * (e / 4) + 4 * lh + e % 4
#pragma unroll
      for (int e = 0; e < 16; ++e) lds[wave * 1024 + (8 * (e >> 2) + 4 * lh + (e & 3)) * 32 + lr] = acc[it][nt][e];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int el = tid + 512 * q;                // element (channel, tap) of the 32 x 32 tile
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) v += lds[w8 * 1024 + el];
        const int ch = it * 32 + (el >> 5), k = nt * 32 + (el & 31);
        const int ky = k / KR, j = k - ky * KR;
        if (k < KP && j < KW * CI) {
          const int kx = j / CI, c = j - kx * CI;
          if (SWAP) slab[(size_t)c * (KH * KW * 64) + ((KH - 1 - ky) * KW + (KW - 1 - kx)) * 64 + ch] = v;
          else slab[(size_t)ch * (KH * KW * CI) + (ky * KW + kx) * CI + c] = v;
        }
      }
    }
  if (SWAP) {      // the slab has 4 channel rows (Cdpad of the reduce): row 3 is never written above
    for (int e = tid; e < KH * KW * 64; e += 512) slab[(size_t)3 * (KH * KW * 64) + e] = 0.f;
  }
}

// ---- the same weight gradient on the bf16 MFMA (bf16 compute mode; round 6) ----
// In the bf16 mode the two kernels above were the last exact-fp32 products of the 7x7 RGB layers: 8 launches x 124 us per train
// step for a GEMM whose bf16 matrix time is a few us -- what bounds it then is reading the 64-channel tensor once (134 MB at
// batch 32).  Same GEMM (M = 64 channels, N = 154 -> 160 taps, K = pixels), same slab layout, on v_mfma_f32_32x32x16_bf16:
//   * a workgroup (4 waves) walks a range of 4 x 32 pixel patches; per patch the 64-channel tensor's 128 pixels are read ONCE as
//     fp32 (8 x 16 bytes per thread, in flight under the previous patch's products), rounded to bf16 (RNE) and parked in LDS as
//     [pixel][64 channels + pad] (192-byte rows, as halo16_wgrad_kernel); the K index is the pixel, so the A fragment -- 32 channels
//     of 16 pixels -- is two transposing reads (ds_read_b64_tr_b16);
//   * the 3-channel tensor's (4 + 6) x (32 + 6) pixel halo is parked as bf16 [row][column][3]; the B fragment of (tap, 16 pixels) is
//     eight 2-byte reads at  pixel base + tap offset + 3 j  (a lane = a tap, as in the fp32 kernel);
//   * wave w owns patch row w: 2 K steps x (2 x 5) products per patch, all 2 x 5 output tiles in 160 accumulator registers; at the
//     end the four partial results meet in LDS and one slab per workgroup goes to the split-K sum of conv_igemm.hip.
struct RgbWgrad16Params {
  const void* c64;    // [NB][H][W][64] fp32, or bf16 (C16)
  const float* t3;    // [NB][H][W][3]
  float* slab;        // [gridDim.x][Cdpad][NNpad]
  int NB, H, W, tiles_x, tiles_y, patches, per_split;
};

template <int SWAP, bool C16>
__global__ __launch_bounds__(256, 2) void rgb_wgrad16_kernel(RgbWgrad16Params p) {
  constexpr int KH = 7, KW = 7, CI = 3, PR = 4, PC = 32;
  constexpr int HR = PR + KH - 1, HC = PC + KW - 1, HWE = HC * CI, HEL = HR * HWE;      // 10 x 38 pixels, 114 entries per row, 1140
  constexpr int KR = 22, KP = KH * KR;      // 154 taps, padded to 5 x 32
  constexpr int GPS = 192;                  // bytes per pixel row of the 64-channel bf16 tile
  constexpr int CT = PR * PC * GPS, HT = (HEL * 2 + 15) & ~15;
  static_assert(2 * CT + 2 * HT >= 4 * 1024 * 4, "the final reduction reuses the operand buffers");
  typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * CT + 2 * HT];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int li = lane & 15, q4 = li >> 2, pq = li & 3, g1 = (lane >> 4) & 1;
  const int p_begin = blockIdx.x * p.per_split, p_end = min(p_begin + p.per_split, p.patches);

  constexpr int CSZ = C16 ? 2 : 4;
  const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.c64), 0, (unsigned)((size_t)p.NB * p.H * p.W * 64 * CSZ), 0x00020000);
  const auto rs_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.t3), 0, (unsigned)((size_t)p.NB * p.H * p.W * CI * 4), 0x00020000);
  constexpr unsigned kOutside = 0x80000000u;
  const int hcg = tid & 7, hpl = tid >> 3;           // 8 channels of one of the 32 pixels of a patch row

  f32x4 cl[PR], ch[PR];
  float tv[5];
  auto issue = [&](int patch) __attribute__((always_inline)) {
    int r = patch;
    const int tx = r % p.tiles_x; r /= p.tiles_x;
    const int ty = r % p.tiles_y;
    const int nb = r / p.tiles_y;
    const int Y0 = ty * PR, X0 = tx * PC;
#pragma unroll
    for (int g = 0; g < PR; ++g) {
      const unsigned off = (unsigned)((((nb * p.H + Y0 + g) * p.W + X0 + hpl) * 64 + hcg * 8) * CSZ);
      cl[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_c, off, 0, 0));
      if constexpr (!C16) ch[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_c, off, 16, 0));
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int e = tid + 256 * i;
      const int rr = e / HWE, j = e - rr * HWE;
      const int gy = Y0 - 3 + rr, gx = (X0 - 3) * CI + j;
      const bool ok = e < HEL && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W * CI;
      const unsigned off = ok ? (unsigned)(((nb * p.H + gy) * p.W * CI + gx) * 4) : kOutside;
      tv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_t, off, 0, 0));
    }
  };
  auto park = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < PR; ++g) {
      if constexpr (C16) {
        *reinterpret_cast<f32x4*>(&smem[buf * CT + (g * PC + hpl) * GPS + hcg * 16]) = cl[g];
      } else {
        const bf16x4 a = __builtin_convertvector(cl[g], bf16x4), b = __builtin_convertvector(ch[g], bf16x4);
        bf16x8 v;
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        *reinterpret_cast<bf16x8*>(&smem[buf * CT + (g * PC + hpl) * GPS + hcg * 16]) = v;
      }
    }
    __bf16* h = reinterpret_cast<__bf16*>(&smem[2 * CT + buf * HT]);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int e = tid + 256 * i;
      if (e < HEL) h[e] = (__bf16)tv[i];
    }
  };

  // B role: lane = tap 32 nt + lr, pixels 8 lh .. 8 lh + 7 of the K step
  int tapoff[5];
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    const int k = nt * 32 + lr;
    const int ky = k / KR, j = k - ky * KR;
    tapoff[nt] = ((k < KP && j < KW * CI) ? ky * HWE + j : 0) + lh * 8 * CI;     // pad taps read a valid slot, their column is dropped
  }
  // A role: transposing read of pixels 8 lh + q4 (+ 4), channels 16 g1 + 4 pq .. + 3 of a 32-channel half
  const int a_lane = (wave * PC + 8 * lh + q4) * GPS + (16 * g1 + 4 * pq) * 2;

  f32x16 acc[2][5];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (p_begin < p_end) {
    issue(p_begin);
    park(0);
    __syncthreads();
  }
  for (int pt = p_begin; pt < p_end; ++pt) {
    const int buf = (pt - p_begin) & 1;
    if (pt + 1 < p_end) issue(pt + 1);
    const unsigned char* A = smem + buf * CT + a_lane;
    const unsigned short* hb = reinterpret_cast<const unsigned short*>(smem + 2 * CT + buf * HT) + wave * HWE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const unsigned char* a = A + (16 * s) * GPS + it * 64;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(a));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(a + 4 * GPS));
        af[it] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) {
        const unsigned short* b = hb + 16 * s * CI + tapoff[nt];
        unsigned w4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w4[j] = (unsigned)b[(2 * j) * CI] | ((unsigned)b[(2 * j + 1) * CI] << 16);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 wv;
        wv[0] = w4[0]; wv[1] = w4[1]; wv[2] = w4[2]; wv[3] = w4[3];
        const bf16x8 bf = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
        for (int it = 0; it < 2; ++it) acc[it][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[it], bf, acc[it][nt], 0, 0, 0);
      }
    }
    if (pt + 1 < p_end) park(buf ^ 1);
    __syncthreads();
  }

  // ---- the four partial results meet in LDS, one 32 x 32 output tile at a time ----
  float* red = reinterpret_cast<float*>(smem);
  float* slab = p.slab + (size_t)blockIdx.x * (SWAP ? 4 * (KH * KW * 64) : 64 * (KH * KW * CI));
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
      __syncthreads();
      // D[i = channel][j = tap]: lane = tap lr, register e = channel 8 * (e / 4) + 4 * lh + e % 4
#pragma unroll
      for (int e = 0; e < 16; ++e) red[wave * 1024 + (8 * (e >> 2) + 4 * lh + (e & 3)) * 32 + lr] = acc[it][nt][e];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int el = tid + 256 * q;                // element (channel, tap) of the 32 x 32 tile
        const float v = (red[el] + red[1024 + el]) + (red[2048 + el] + red[3072 + el]);
        const int chn = it * 32 + (el >> 5), k = nt * 32 + (el & 31);
        const int ky = k / KR, j = k - ky * KR;
        if (k < KP && j < KW * CI) {
          const int kx = j / CI, c = j - kx * CI;
          if (SWAP) slab[(size_t)c * (KH * KW * 64) + ((KH - 1 - ky) * KW + (KW - 1 - kx)) * 64 + chn] = v;
          else slab[(size_t)chn * (KH * KW * CI) + (ky * KW + kx) * CI + c] = v;
        }
      }
    }
  if (SWAP) {      // the slab has 4 channel rows (Cdpad of the reduce): row 3 is never written above
    for (int e = tid; e < KH * KW * 64; e += 256) slab[(size_t)3 * (KH * KW * 64) + e] = 0.f;
  }
}

// bf16 mode: patches of 4 x 32 pixels in at most 256 contiguous ranges (one workgroup per CU; measured on the step: 256 ranges
// 965.0, 512 962.9, 1024 960.2 images/s, fp32 kernels 947.7 -- more ranges only add slab traffic)
static bool rgb_wgrad16_on() {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_RGB_WGRAD16");
  return !off && compute_bf16();
}
static void rgb_wgrad16_plan(const srgan_conv_desc* d, int* patches, int* per_split, int* splits) {
  static const int max_splits = (int)SRGAN_AB_INT("SRGAN_RGB_WGRAD16_SPLITS", 256);
  *patches = d->N * (d->Hi / 4) * (d->Wi / 32);
  *per_split = (int)ceil_div(*patches, std::min(*patches, max_splits));
  *splits = (int)ceil_div(*patches, *per_split);
}

// kind 0: RGB input layer (I == 3, O == 64); kind 1: RGB output layer (I == 64, O == 3)
int rgb_wgrad_kind(const srgan_conv_desc* d) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_RGBIN");
  // (also in the bf16 compute mode: the RGB layers are documented to stay fp32 there, and these exact-fp32 MFMA kernels are 2-3x
  // faster than the scalar-gather GEMM / VALU kernels they would otherwise fall back to: 3.1 + 1.7 ms -> 1.4 + 1.0 ms per step)
  if (off) return -1;
  if (d->kh != 7 || d->kw != 7 || d->stride != 1 || d->pad != 3 || d->pad_mode != SRGAN_PAD_ZERO) return -1;
  if (d->Hi % 16 != 0 || d->Wi % 32 != 0 || (long long)d->N * d->Hi * d->Wi * 64 >= (1LL << 29)) return -1;
  if (d->I == 3 && d->O == 64) return 0;
  if (d->I == 64 && d->O == 3) return 1;
  return -1;
}

// slab geometry for the reduce: [splits][Cdpad][NNpad]
void rgb_wgrad_slab(const srgan_conv_desc* d, int* splits, int* Cdpad, int* NNpad) {
  const long long tiles = (long long)d->N * (d->Hi / 16) * (d->Wi / 32);
  *splits = (int)std::min<long long>(tiles, 256);
  if (rgb_wgrad16_on()) {
    int patches, per;
    rgb_wgrad16_plan(d, &patches, &per, splits);
  }
  *Cdpad = rgb_wgrad_kind(d) == 1 ? 4 : 64;
  *NNpad = d->kh * d->kw * d->I;
}

bool rgb_wgrad16_served(const srgan_conv_desc* d) { return rgb_wgrad_kind(d) >= 0 && rgb_wgrad16_on(); }

// c64_bf16: the 64-channel tensor (kind 0: dy, kind 1: x) is bf16 (bf16 mode only)
int rgb_wgrad_run(const srgan_conv_desc* d, const void* x, const void* dy, float* slab, hipStream_t st, bool c64_bf16) {
  const int kind = rgb_wgrad_kind(d);
  SRGAN_REQUIRE(kind >= 0, "rgb weight gradient: layer not applicable");
  SRGAN_REQUIRE(!c64_bf16 || rgb_wgrad16_on(), "rgb weight gradient: a bf16 tensor needs the bf16 compute mode's kernel");
  RgbWgradParams p{};
  p.c64 = static_cast<const float*>(kind == 0 ? dy : x); p.t3 = static_cast<const float*>(kind == 0 ? x : dy); p.slab = slab;
  p.NB = d->N; p.H = d->Hi; p.W = d->Wi; p.tiles_x = d->Wi / 32; p.tiles_y = d->Hi / 16;
  p.ntiles = p.NB * p.tiles_x * p.tiles_y;
  int splits, cd, nn;
  rgb_wgrad_slab(d, &splits, &cd, &nn);
  ProfToken tok = prof_begin(21, 2.0 * d->N * d->Ho * d->Wo * (double)d->O * d->kh * d->kw * d->I, st);
  if (rgb_wgrad16_on()) {
    RgbWgrad16Params q{};
    q.c64 = kind == 0 ? dy : x; q.t3 = p.t3; q.slab = slab;
    q.NB = d->N; q.H = d->Hi; q.W = d->Wi; q.tiles_x = d->Wi / 32; q.tiles_y = d->Hi / 4;
    int s16 = 0;
    rgb_wgrad16_plan(d, &q.patches, &q.per_split, &s16);
    SRGAN_REQUIRE(s16 == splits, "rgb weight gradient: split plan changed between sizing and launch");
    if (kind == 0 && c64_bf16) hipLaunchKernelGGL((rgb_wgrad16_kernel<0, true>), dim3((unsigned)splits), dim3(256), 0, st, q);
    else if (kind == 0) hipLaunchKernelGGL((rgb_wgrad16_kernel<0, false>), dim3((unsigned)splits), dim3(256), 0, st, q);
    else if (c64_bf16) hipLaunchKernelGGL((rgb_wgrad16_kernel<1, true>), dim3((unsigned)splits), dim3(256), 0, st, q);
    else hipLaunchKernelGGL((rgb_wgrad16_kernel<1, false>), dim3((unsigned)splits), dim3(256), 0, st, q);
  } else if (kind == 0) hipLaunchKernelGGL(rgb_wgrad_kernel<0>, dim3((unsigned)splits), dim3(512), 0, st, p);
  else hipLaunchKernelGGL(rgb_wgrad_kernel<1>, dim3((unsigned)splits), dim3(512), 0, st, p);
  prof_end(tok, st);
  return check_launch("rgb_wgrad_kernel");
}

bool rgbin_applicable(const srgan_conv_desc* d) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_RGBIN");
  if (off) return false;
  const bool k7s1 = d->kh == 7 && d->kw == 7 && d->stride == 1;
  // round 6: the strided 3-channel layers (D's first conv 4x4 / 2 / pad 1, E's first conv 7x7 / 2 / pad 1)
  static const bool no_s2 = SRGAN_AB_SET("SRGAN_NO_RGBIN_S2");
  const bool s2 = !no_s2 && d->stride == 2 && ((d->kh == 4 && d->kw == 4) || (d->kh == 7 && d->kw == 7));
  return d->I == 3 && (k7s1 || s2) && d->pad_mode == SRGAN_PAD_ZERO && d->O % 64 == 0 &&
         d->Wo >= 32 && d->Ho >= 16 && (long long)d->N * d->Ho * d->Wo * d->O < (1LL << 31);
}

size_t rgbin_packed_elems(const srgan_conv_desc* d) { return (size_t)d->kh * ((d->kw * d->I + 1) & ~1) * d->O; }

// bf16 mode: the 3 -> 64k-channel 7x7 / pad-3 layers on rgbin16_conv_kernel (its packed filter, O x 232 bf16, fits the fp32
// kernel's allocation of 7 x 22 x O floats)
static bool rgbin16_mode(const srgan_conv_desc* d) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_RGBIN16");
  return !off && compute_bf16() && d->I == 3 && d->kh == 7 && d->kw == 7 && d->pad == 3 && d->stride == 1;
}

int rgbin_pack(const srgan_conv_desc* d, const float* w, float* dst, hipStream_t st) {
  if (rgbin16_mode(d)) {
    const long long total16 = (long long)d->O * R16_KPS;
    hipLaunchKernelGGL(rgbin16_pack_kernel, dim3((unsigned)std::min<long long>(ceil_div(total16, 256), 1024)), dim3(256), 0, st, w,
                       reinterpret_cast<unsigned short*>(dst), d->sO, d->sI, d->sH, d->sW, d->O);
    return check_launch("rgbin16_pack_kernel");
  }
  const long long total = (long long)rgbin_packed_elems(d);
  hipLaunchKernelGGL(rgbin_pack_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 1024)), dim3(256), 0, st, w, dst,
                     d->sO, d->sI, d->sH, d->sW, d->O, d->I, d->kh, d->kw);
  return check_launch("rgbin_pack_kernel");
}

bool rgbin16_served(const srgan_conv_desc* d) { return rgbin_applicable(d) && rgbin16_mode(d); }

int rgbin_run(const srgan_conv_desc* d, const float* x, const float* packed, const float* bias, float* y, int act, float slope,
              hipStream_t st, bool dst16) {
  SRGAN_REQUIRE(rgbin_applicable(d), "rgb-input conv: layer not applicable");
  SRGAN_REQUIRE(!dst16 || rgbin16_mode(d), "rgb-input conv: a bf16 result needs the bf16 compute mode's kernel");
  RgbinParams p{};
  p.x = x; p.wp = packed; p.bias = bias; p.y = y;
  p.NB = d->N; p.H = d->Hi; p.W = d->Wi; p.Ho = d->Ho; p.Wo = d->Wo; p.O = d->O; p.pad = d->pad;
  p.tiles_x = (int)ceil_div(d->Wo, 32); p.tiles_y = (int)ceil_div(d->Ho, 16); p.o_blocks = d->O / 64;
  p.act = act; p.slope = slope;
  const long long grid = (long long)p.tiles_x * p.tiles_y * p.o_blocks * d->N;
  SRGAN_REQUIRE(grid < (1LL << 31), "rgb-input conv: grid too large");
  ProfToken tok = prof_begin(20, 2.0 * d->N * d->Ho * d->Wo * (double)d->O * d->kh * d->kw * d->I, st);
  if (rgbin16_mode(d)) {
    Rgbin16Params q{};
    q.x = x; q.wp = reinterpret_cast<const unsigned short*>(packed); q.bias = bias; q.y = y;
    q.NB = p.NB; q.H = p.H; q.W = p.W; q.Ho = p.Ho; q.Wo = p.Wo; q.O = p.O; q.pad = p.pad;
    q.tiles_x = p.tiles_x; q.tiles_y = p.tiles_y; q.o_blocks = p.o_blocks; q.act = act; q.slope = slope;
    if (dst16) hipLaunchKernelGGL(rgbin16_conv_kernel<true>, dim3((unsigned)grid), dim3(512), 0, st, q);
    else hipLaunchKernelGGL(rgbin16_conv_kernel<false>, dim3((unsigned)grid), dim3(512), 0, st, q);
    prof_end(tok, st);
    return check_launch("rgbin16_conv_kernel");
  }
  if (d->stride == 2 && d->kh == 4) hipLaunchKernelGGL((rgbin_conv_kernel<4, 4, 3, 2>), dim3((unsigned)grid), dim3(512), 0, st, p);
  else if (d->stride == 2) hipLaunchKernelGGL((rgbin_conv_kernel<7, 7, 3, 2>), dim3((unsigned)grid), dim3(512), 0, st, p);
  else hipLaunchKernelGGL((rgbin_conv_kernel<7, 7, 3>), dim3((unsigned)grid), dim3(512), 0, st, p);
  prof_end(tok, st);
  return check_launch("rgbin_conv_kernel");
}

}  // namespace srgan

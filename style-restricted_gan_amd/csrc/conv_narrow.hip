// Direct (non-GEMM) convolution kernels for layers with a NARROW output: Cout <= 4, stride 1, zero padding,
// Cin % 16 == 0.  In the SRGAN generator this is `up_convs[-1]`, the 7x7 64->3 RGB head (pyfiles/model.py:232),
// whose forward and weight gradient would waste 29/32 of every 32x32 MFMA tile and, as an implicit GEMM, re-read
// each input pixel 49 times.  Here a workgroup stages an (8+k-1) x (32+k-1) halo tile of 16 input channels in LDS
// once and every pixel / weight of the tile is produced from it on the vector ALUs (the fp32 VALU rate equals the
// fp32 MFMA rate on gfx950, so nothing is lost by leaving the matrix pipe).
//
//   narrow_conv_fwd    one thread per output pixel, CO accumulators, weights through wave-uniform (scalar) loads
//   narrow_conv_wgrad  thread = (channel of the 16-chunk, tap group); accumulates dW over all tiles of its slice
//                      in registers and writes one slab that wgrad_reduce_kernel sums (deterministic)
#include <algorithm>
#include "common.h"

namespace srgan {

constexpr int NT_H = 8, NT_W = 32;     // output tile (pixels) per workgroup pass
constexpr int NCH = 16;                // input channels staged per pass
constexpr int NPIX = NCH + 4;          // LDS floats per staged pixel (pad: conflict-free ds_read_b128 across lanes)

struct NarrowParams {
  const float* x;      // [N][H][W][Ci]
  const float* wp;     // fwd: packed [Ci/16][T][16][4]
  const float* bias;   // [CO] or null
  const float* dy;     // wgrad: [N][Ho][Wo][CO]
  float* y;            // fwd: [N][Ho][Wo][CO]
  float* slab;         // wgrad: [blocks][4][T*Ci]
  int N, H, W, Ci, Ho, Wo, CO, kh, kw, pad;
  int tiles_x, tiles_y, tiles_per_block, splits;   // wgrad work split
  // generic forward only: column padding (rows use `pad`) and the placement of output pixel (oy, ox) in the destination image:
  // dst[n][oy * oy_mul + oy_off][ox * ox_mul + ox_off] of an Hd x Wd image (the phase images of a stride-2 input gradient)
  int pad_x, Hd, Wd, oy_mul, ox_mul, oy_off, ox_off;
};

// stage the halo tile of channels [c0, c0+16) for output tile origin (oy0, ox0) of image n
__device__ __forceinline__ void stage_tile(const NarrowParams& p, float* tile, int n, int oy0, int ox0, int c0) {
  const int th = NT_H + p.kh - 1, tw = NT_W + p.kw - 1;
  const int total = th * tw * (NCH / 4);
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int q = idx & 3, pix = idx >> 2;
    const int ty = pix / tw, tx = pix - ty * tw;
    const int y = oy0 + ty - p.pad, x = ox0 + tx - p.pad_x;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
      v = *reinterpret_cast<const f32x4*>(p.x + ((size_t)(n * p.H + y) * p.W + x) * p.Ci + c0 + q * 4);
    *reinterpret_cast<f32x4*>(&tile[pix * NPIX + q * 4]) = v;
  }
}

template <int CO>
__global__ __launch_bounds__(256) void narrow_conv_fwd_kernel(NarrowParams p) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int n = blockIdx.z;
  const int oy0 = blockIdx.y * NT_H, ox0 = blockIdx.x * NT_W;
  const int py = threadIdx.x >> 5, px = threadIdx.x & 31;
  const int tw = NT_W + p.kw - 1;
  const int T = p.kh * p.kw;
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = p.bias ? p.bias[o] : 0.f;
  for (int c0 = 0; c0 < p.Ci; c0 += NCH) {
    __syncthreads();
    stage_tile(p, tile, n, oy0, ox0, c0);
    __syncthreads();
    const float* wq = p.wp + (size_t)(c0 / NCH) * T * NCH * 4;
    for (int ky = 0; ky < p.kh; ++ky) {
      for (int kx = 0; kx < p.kw; ++kx) {
        const float* src = &tile[((py + ky) * tw + (px + kx)) * NPIX];
        const float* wt = wq + (ky * p.kw + kx) * NCH * 4;      // wave-uniform -> scalar loads
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + q * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int o = 0; o < CO; ++o) acc[o] = fmaf(v[e], wt[(q * 4 + e) * 4 + o], acc[o]);
        }
      }
    }
  }
  const int oy = oy0 + py, ox = ox0 + px;
  if (oy < p.Ho && ox < p.Wo) {
    float* dst = p.y + ((size_t)(n * p.Hd + oy * p.oy_mul + p.oy_off) * p.Wd + ox * p.ox_mul + p.ox_off) * CO;
#pragma unroll
    for (int o = 0; o < CO; ++o) dst[o] = acc[o];
  }
}

// ---- register-blocked forward (v2) for K x K kernels, K = KW (7 for the RGB head) ---------------------------------
// Each thread produces 4 horizontally adjacent output pixels x CO channels.  The halo tile is staged CHANNEL-PLANAR in
// LDS ([c][y][x], x contiguous): for one (channel, ky) a thread reads 12 consecutive floats (3 conflict-free
// ds_read_b128) and uses them for 4 pixels x KW taps x CO outputs = 84 FMAs (CO=3), with the KW*CO weights of that
// (channel, ky) coming from one scalar (wave-uniform) load burst -- 28 FMAs per LDS read instead of 12.
constexpr int N2_ROWS = 8, N2_COLS = 128, N2_CH = 8;        // output tile 8 x 128 pixels, 8 channels per pass (4 measured slower)

template <int CO, int KW>
__global__ __launch_bounds__(256) void narrow_conv_fwd4_kernel(NarrowParams p) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  constexpr int TH = N2_ROWS + KW - 1;
  constexpr int TWP = N2_COLS + 8;                 // staged row: 128 + 6 halo, padded to a multiple of 4
  constexpr int WROW = ((KW * CO + 3) / 4) * 4;    // weights per (channel, ky), padded for aligned scalar loads
  const int n = blockIdx.z;
  const int oy0 = blockIdx.y * N2_ROWS, ox0 = blockIdx.x * N2_COLS;
  const int py = threadIdx.x >> 5, pg = threadIdx.x & 31;     // row in tile, 4-pixel group
  float acc[4][CO];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[q][o] = p.bias ? p.bias[o] : 0.f;

  for (int c0 = 0; c0 < p.Ci; c0 += N2_CH) {
    __syncthreads();
    // stage: NHWC global (float4 = 4 channels of a pixel) -> planar LDS
    for (int idx = threadIdx.x; idx < TH * (N2_COLS + KW - 1) * (N2_CH / 4); idx += 256) {
      const int half = idx % (N2_CH / 4);
      const int pix = idx / (N2_CH / 4);
      const int ty = pix / (N2_COLS + KW - 1), tx = pix - ty * (N2_COLS + KW - 1);
      const int y = oy0 + ty - p.pad, x = ox0 + tx - p.pad;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
        v = *reinterpret_cast<const f32x4*>(p.x + ((size_t)(n * p.H + y) * p.W + x) * p.Ci + c0 + half * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[((half * 4 + e) * TH + ty) * TWP + tx] = v[e];
    }
    __syncthreads();
    const float* wq = p.wp + (size_t)c0 * KW * WROW;          // [channel][ky][WROW]
    // Software pipeline over the (channel, ky) steps: the 3 LDS reads and the scalar weight burst of step s+1 are issued
    // before the 84 FMAs of step s (scalar loads return out of order, so their wait is lgkmcnt(0): issued a whole step
    // ahead, that wait is free; issued right before use it exposed the full scalar-cache latency at 2 waves per SIMD).
    float xs[2][12], wv[2][WROW];
    // `rel` = step relative to the bases (a compile-time constant after unrolling: immediate offsets, one base each)
    auto fetch = [&](const float* rowb, const float* wb, int rel, int buf) __attribute__((always_inline)) {
      const int c = rel / KW, ky = rel - c * KW;
      const float* row = rowb + (c * TH + ky) * TWP;
      *reinterpret_cast<f32x4*>(xs[buf]) = *reinterpret_cast<const f32x4*>(row);
      *reinterpret_cast<f32x4*>(xs[buf] + 4) = *reinterpret_cast<const f32x4*>(row + 4);
      *reinterpret_cast<f32x4*>(xs[buf] + 8) = *reinterpret_cast<const f32x4*>(row + 8);
      const float* wt = wb + rel * WROW;                        // wave-uniform -> scalar loads
#pragma unroll
      for (int i = 0; i < WROW; ++i) wv[buf][i] = wt[i];
    };
    constexpr int STEPS = N2_CH * KW, UNR = 2 * KW;             // body = 2 channels, buffers alternate statically
    static_assert(STEPS % UNR == 0, "N2_CH must be even");
    const float* rowb = &tile[py * TWP + pg * 4];
    const float* wb = wq;
    fetch(rowb, wb, 0, 0);
#pragma unroll 1
    for (int s0 = 0; s0 < STEPS; s0 += UNR) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int cur = u & 1;
        if (u + 1 < UNR || s0 + UNR < STEPS) fetch(rowb, wb, u + 1, cur ^ 1);    // u + 1 == UNR: first step of the next body
#pragma unroll
        for (int kx = 0; kx < KW; ++kx)
#pragma unroll
          for (int o = 0; o < CO; ++o) {
            const float w = wv[cur][kx * CO + o];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q][o] = fmaf(xs[cur][q + kx], w, acc[q][o]);
          }
      }
      rowb += 2 * TH * TWP;
      wb += UNR * WROW;
    }
  }
  const int oy = oy0 + py, ox = ox0 + pg * 4;
  if (oy < p.Ho) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ox + q < p.Wo) {
        float* dst = p.y + ((size_t)(n * p.Ho + oy) * p.Wo + ox + q) * CO;
#pragma unroll
        for (int o = 0; o < CO; ++o) dst[o] = acc[q][o];
      }
  }
}

// wp[ci][ky][WROW]: wp[(ci*KW + ky)*WROW + kx*CO + o] = W[o][ci][ky][kx]
__global__ void narrow_pack4_kernel(const float* w, float* wp, long long sO, long long sI, long long sH, long long sW,
                                    int CO, int Ci, int KW, int WROW) {
  const int total = Ci * KW * WROW;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx % WROW;
    const int r = idx / WROW;
    const int ky = r % KW, ci = r / KW;
    const int kx = j / CO, o = j - kx * CO;
    wp[idx] = (kx < KW) ? w[o * sO + ci * sI + ky * sH + kx * sW] : 0.f;
  }
}

static bool narrow4_applicable(const srgan_conv_desc* d) {
  return d->O == 3 && d->kh == 7 && d->kw == 7 && (d->I % N2_CH) == 0 && d->Wo >= 64;
}

// wp[cc][tap][c][4] = W[o][cc*16+c][ky][kx] (o < CO, zero otherwise)
__global__ void narrow_pack_kernel(const float* w, float* wp, long long sO, long long sI, long long sH, long long sW,
                                   int CO, int Ci, int kh, int kw) {
  const int total = Ci * kh * kw * 4;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int o = idx & 3;
    int r = idx >> 2;
    const int c = r % NCH; r /= NCH;
    const int tap = r % (kh * kw);
    const int cc = r / (kh * kw);
    const int ky = tap / kw, kx = tap - ky * kw;
    wp[idx] = o < CO ? w[o * sO + (cc * NCH + c) * sI + ky * sH + kx * sW] : 0.f;
  }
}

template <int CO, int TJ>
__global__ __launch_bounds__(256) void narrow_conv_wgrad_kernel(NarrowParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int th = NT_H + p.kh - 1, tw = NT_W + p.kw - 1;
  float* tile = smem;                       // [th][tw][NPIX]
  float* dyt = smem + th * tw * NPIX;       // [256][4]
  const int cc = blockIdx.x;                // channel chunk
  const int split = blockIdx.y % p.splits, n = blockIdx.y / p.splits;
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int T = p.kh * p.kw;
  int off[TJ];
  bool valid[TJ];
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int tap = g + 16 * j;
    valid[j] = tap < T;
    const int ky = valid[j] ? tap / p.kw : 0, kx = valid[j] ? tap - ky * p.kw : 0;
    off[j] = (ky * tw + kx) * NPIX + c;
  }
  float acc[TJ][CO];
#pragma unroll
  for (int j = 0; j < TJ; ++j)
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[j][o] = 0.f;

  const int ntiles = p.tiles_x * p.tiles_y;
  const int t0 = split * p.tiles_per_block, t1 = min(ntiles, t0 + p.tiles_per_block);
  for (int t = t0; t < t1; ++t) {
    const int oy0 = (t / p.tiles_x) * NT_H, ox0 = (t % p.tiles_x) * NT_W;
    __syncthreads();
    stage_tile(p, tile, n, oy0, ox0, cc * NCH);
    {
      const int py = threadIdx.x >> 5, px = threadIdx.x & 31;
      const int oy = oy0 + py, ox = ox0 + px;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (oy < p.Ho && ox < p.Wo) {
        const float* src = p.dy + ((size_t)(n * p.Ho + oy) * p.Wo + ox) * CO;
#pragma unroll
        for (int o = 0; o < CO; ++o) v[o] = src[o];
      }
      *reinterpret_cast<f32x4*>(&dyt[threadIdx.x * 4]) = v;
    }
    __syncthreads();
    for (int py = 0; py < NT_H; ++py) {
#pragma unroll 4
      for (int px = 0; px < NT_W; ++px) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(&dyt[(py * NT_W + px) * 4]);   // broadcast read
        const float* base = &tile[(py * tw + px) * NPIX];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const float xv = base[off[j]];
#pragma unroll
          for (int o = 0; o < CO; ++o) acc[j][o] = fmaf(d[o], xv, acc[j][o]);
        }
      }
    }
  }
  const size_t nn_total = (size_t)T * p.Ci;
  float* slab = p.slab + (size_t)blockIdx.y * 4 * nn_total;     // [n*splits+split][4][T*Ci]; chunks write disjoint ci
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    if (!valid[j]) continue;
    const int tap = g + 16 * j;
#pragma unroll
    for (int o = 0; o < CO; ++o) slab[(size_t)o * nn_total + (size_t)tap * p.Ci + cc * NCH + c] = acc[j][o];
  }
}

// Small feature maps with many channels (PatchGAN last_layer: 8x8x512 -> 7x7x1, k4 p1): one WAVE per output pixel,
// lanes stride over the input channels (float4, coalesced), taps looped, 64-lane shuffle reduction at the end.
// wp: [CO][Kpad], K = (tap, channel) contiguous (pack_weights_kernel mode 0 layout).
template <int CO>
__global__ __launch_bounds__(256) void narrow_wave_fwd_kernel(NarrowParams p, int Kpad) {
  const int pix = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int M = p.N * p.Ho * p.Wo;
  if (pix >= M) return;
  const int n = pix / (p.Ho * p.Wo);
  const int rem = pix - n * (p.Ho * p.Wo);
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = 0.f;
  for (int ky = 0; ky < p.kh; ++ky) {
    const int y = oy + ky - p.pad;
    if ((unsigned)y >= (unsigned)p.H) continue;
    for (int kx = 0; kx < p.kw; ++kx) {
      const int x = ox + kx - p.pad;
      if ((unsigned)x >= (unsigned)p.W) continue;
      const float* src = p.x + ((size_t)(n * p.H + y) * p.W + x) * p.Ci;
      const float* wt = p.wp + (size_t)(ky * p.kw + kx) * p.Ci;
      for (int c = lane * 4; c < p.Ci; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
#pragma unroll
        for (int o = 0; o < CO; ++o) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(wt + (size_t)o * Kpad + c);
          acc[o] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    const float t = wave_sum(acc[o]);
    if (lane == 0) p.y[(size_t)pix * CO + o] = t + (p.bias ? p.bias[o] : 0.f);
  }
}

bool narrow_wave_applicable(const srgan_conv_desc* d) {
  return d->O <= 4 && d->stride == 1 && d->pad_mode == SRGAN_PAD_ZERO && (d->I % 4) == 0 && d->I >= 128 &&
         d->Ho * d->Wo <= 1024;
}

// wp: pack_weights_kernel(mode 0) output with Npad = O
int narrow_wave_fwd(const srgan_conv_desc* d, const float* x, const float* wp, int Kpad, const float* bias, float* y,
                    hipStream_t st) {
  NarrowParams p{};
  p.x = x; p.wp = wp; p.bias = bias; p.y = y;
  p.N = d->N; p.H = d->Hi; p.W = d->Wi; p.Ci = d->I; p.Ho = d->Ho; p.Wo = d->Wo; p.CO = d->O;
  p.kh = d->kh; p.kw = d->kw; p.pad = d->pad;
  p.pad_x = d->pad; p.Hd = d->Ho; p.Wd = d->Wo; p.oy_mul = p.ox_mul = 1; p.oy_off = p.ox_off = 0;
  const long long M = (long long)d->N * d->Ho * d->Wo;
  dim3 grid((unsigned)ceil_div(M, 4));
  switch (d->O) {
    case 1: hipLaunchKernelGGL(narrow_wave_fwd_kernel<1>, grid, dim3(256), 0, st, p, Kpad); break;
    case 2: hipLaunchKernelGGL(narrow_wave_fwd_kernel<2>, grid, dim3(256), 0, st, p, Kpad); break;
    case 3: hipLaunchKernelGGL(narrow_wave_fwd_kernel<3>, grid, dim3(256), 0, st, p, Kpad); break;
    default: hipLaunchKernelGGL(narrow_wave_fwd_kernel<4>, grid, dim3(256), 0, st, p, Kpad); break;
  }
  return check_launch("narrow_wave_fwd_kernel");
}

bool narrow_applicable(const srgan_conv_desc* d) {
  // (Ho*Wo >= 9: a 1x1 output would leave 255 of 256 lanes idle -- those are dense heads, see dense_head_applicable)
  return d->O <= 4 && d->stride == 1 && d->pad_mode == SRGAN_PAD_ZERO && (d->I % NCH) == 0 && d->kh <= 8 && d->kw <= 8 &&
         d->Ho * d->Wo >= 9;
}

size_t narrow_workspace(const srgan_conv_desc* d) {
  const size_t pack = (size_t)d->I * d->kh * d->kw * 4 * sizeof(float);
  const int tiles = (int)(ceil_div(d->Wo, NT_W) * ceil_div(d->Ho, NT_H));
  int splits = 1;
  while ((long long)(d->I / NCH) * d->N * splits < 512 && splits * 2 <= tiles) splits *= 2;
  const size_t slab = (size_t)d->N * splits * 4 * d->kh * d->kw * d->I * sizeof(float);
  return std::max(pack, slab) + 1024;
}

int narrow_pack(const srgan_conv_desc* d, const float* w, float* wp, hipStream_t st) {
  if (narrow4_applicable(d)) {
    const int wrow = ((d->kw * d->O + 3) / 4) * 4;
    const int total4 = d->I * d->kw * wrow;
    hipLaunchKernelGGL(narrow_pack4_kernel, dim3((total4 + 255) / 256), dim3(256), 0, st, w, wp, d->sO, d->sI, d->sH, d->sW,
                       d->O, d->I, d->kw, wrow);
    return check_launch("narrow_pack4_kernel");
  }
  const int total = d->I * d->kh * d->kw * 4;
  hipLaunchKernelGGL(narrow_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, wp, d->sO, d->sI, d->sH, d->sW, d->O,
                     d->I, d->kh, d->kw);
  return check_launch("narrow_pack_kernel");
}

int narrow_fwd_packed(const srgan_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, hipStream_t st) {
  NarrowParams p{};
  p.x = x; p.wp = wp; p.bias = bias; p.y = y;
  p.N = d->N; p.H = d->Hi; p.W = d->Wi; p.Ci = d->I; p.Ho = d->Ho; p.Wo = d->Wo; p.CO = d->O;
  p.kh = d->kh; p.kw = d->kw; p.pad = d->pad;
  p.pad_x = d->pad; p.Hd = d->Ho; p.Wd = d->Wo; p.oy_mul = p.ox_mul = 1; p.oy_off = p.ox_off = 0;
  if (narrow4_applicable(d)) {
    const size_t sh4 = (size_t)N2_CH * (N2_ROWS + 6) * (N2_COLS + 8) * sizeof(float);
    dim3 g4((unsigned)ceil_div(d->Wo, N2_COLS), (unsigned)ceil_div(d->Ho, N2_ROWS), (unsigned)d->N);
    hipLaunchKernelGGL((narrow_conv_fwd4_kernel<3, 7>), g4, dim3(256), sh4, st, p);
    return check_launch("narrow_conv_fwd4_kernel");
  }
  const size_t shmem = (size_t)(NT_H + d->kh - 1) * (NT_W + d->kw - 1) * NPIX * sizeof(float);
  dim3 grid((unsigned)ceil_div(d->Wo, NT_W), (unsigned)ceil_div(d->Ho, NT_H), (unsigned)d->N);
  switch (d->O) {
    case 1: hipLaunchKernelGGL(narrow_conv_fwd_kernel<1>, grid, dim3(256), shmem, st, p); break;
    case 2: hipLaunchKernelGGL(narrow_conv_fwd_kernel<2>, grid, dim3(256), shmem, st, p); break;
    case 3: hipLaunchKernelGGL(narrow_conv_fwd_kernel<3>, grid, dim3(256), shmem, st, p); break;
    default: hipLaunchKernelGGL(narrow_conv_fwd_kernel<4>, grid, dim3(256), shmem, st, p); break;
  }
  return check_launch("narrow_conv_fwd_kernel");
}

// One phase image of a stride-2 input gradient (conv_igemm.hip, narrow_s2_*): the generic kernel with separate row / column
// padding and its output pixels scattered with stride 2 into the Hd x Wd destination.  wp from narrow_pack(f, ...).
int narrow_fwd_strided(const srgan_conv_desc* f, int pad_x, const float* x, const float* wp, float* y, int Hd, int Wd, int oy_off,
                       int ox_off, hipStream_t st) {
  NarrowParams p{};
  p.x = x; p.wp = wp; p.bias = nullptr; p.y = y;
  p.N = f->N; p.H = f->Hi; p.W = f->Wi; p.Ci = f->I; p.Ho = f->Ho; p.Wo = f->Wo; p.CO = f->O;
  p.kh = f->kh; p.kw = f->kw; p.pad = f->pad;
  p.pad_x = pad_x; p.Hd = Hd; p.Wd = Wd; p.oy_mul = p.ox_mul = 2; p.oy_off = oy_off; p.ox_off = ox_off;
  const size_t shmem = (size_t)(NT_H + f->kh - 1) * (NT_W + f->kw - 1) * NPIX * sizeof(float);
  dim3 grid((unsigned)ceil_div(f->Wo, NT_W), (unsigned)ceil_div(f->Ho, NT_H), (unsigned)f->N);
  switch (f->O) {
    case 1: hipLaunchKernelGGL(narrow_conv_fwd_kernel<1>, grid, dim3(256), shmem, st, p); break;
    case 2: hipLaunchKernelGGL(narrow_conv_fwd_kernel<2>, grid, dim3(256), shmem, st, p); break;
    case 3: hipLaunchKernelGGL(narrow_conv_fwd_kernel<3>, grid, dim3(256), shmem, st, p); break;
    default: hipLaunchKernelGGL(narrow_conv_fwd_kernel<4>, grid, dim3(256), shmem, st, p); break;
  }
  return check_launch("narrow_conv_fwd_kernel (phase)");
}

// returns the number of slabs written ([slabs][4][T*I]) through *n_slabs
int narrow_wgrad(const srgan_conv_desc* d, const float* x, const float* dy, void* ws, int* n_slabs, hipStream_t st) {
  NarrowParams p{};
  p.x = x; p.dy = dy; p.slab = (float*)ws;
  p.N = d->N; p.H = d->Hi; p.W = d->Wi; p.Ci = d->I; p.Ho = d->Ho; p.Wo = d->Wo; p.CO = d->O;
  p.kh = d->kh; p.kw = d->kw; p.pad = d->pad;
  p.pad_x = d->pad; p.Hd = d->Ho; p.Wd = d->Wo; p.oy_mul = p.ox_mul = 1; p.oy_off = p.ox_off = 0;
  p.tiles_x = (int)ceil_div(d->Wo, NT_W); p.tiles_y = (int)ceil_div(d->Ho, NT_H);
  const int tiles = p.tiles_x * p.tiles_y;
  int splits = 1;
  while ((long long)(d->I / NCH) * d->N * splits < 512 && splits * 2 <= tiles) splits *= 2;
  p.splits = splits;
  p.tiles_per_block = (int)ceil_div(tiles, splits);
  *n_slabs = d->N * splits;
  const size_t shmem = ((size_t)(NT_H + d->kh - 1) * (NT_W + d->kw - 1) * NPIX + 256 * 4) * sizeof(float);
  dim3 grid((unsigned)(d->I / NCH), (unsigned)(d->N * splits), 1);
  const int T = d->kh * d->kw;
  const int tj = (int)ceil_div(T, 16);
#define NARROW_WG(CO_, TJ_) hipLaunchKernelGGL((narrow_conv_wgrad_kernel<CO_, TJ_>), grid, dim3(256), shmem, st, p)
#define NARROW_WG_CO(TJ_)                         \
  switch (d->O) {                                 \
    case 1: NARROW_WG(1, TJ_); break;             \
    case 2: NARROW_WG(2, TJ_); break;             \
    case 3: NARROW_WG(3, TJ_); break;             \
    default: NARROW_WG(4, TJ_); break;            \
  }
  if (tj <= 1) { NARROW_WG_CO(1) } else if (tj == 2) { NARROW_WG_CO(2) } else if (tj == 3) { NARROW_WG_CO(3) } else { NARROW_WG_CO(4) }
#undef NARROW_WG_CO
#undef NARROW_WG
  return check_launch("narrow_conv_wgrad_kernel");
}


// ---- dense head: a valid conv whose kernel covers the whole input (Ho = Wo = 1) is a Linear layer on the
// NHWC-flattened input: out[b][o] = bias[o] + sum_k x[b][k] * wp[o][k], k = (ky, kx, ci).  Used by the PatchGAN
// classification heads (Conv2d(512, n_class, 8) on an 8x8 map, model.py:330-331): M = batch, K = 32768.
__global__ __launch_bounds__(256) void dense_head_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                         const float* __restrict__ bias, float* __restrict__ y, int K,
                                                         int Kpad, int O) {
  __shared__ float red[16];
  const int o = blockIdx.x, b = blockIdx.y;
  const float* xr = x + (size_t)b * K;
  const float* wr = wp + (size_t)o * Kpad;
  float s = 0.f;
  if ((K & 3) == 0) {
    for (int k = threadIdx.x * 4; k < K; k += 256 * 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(xr + k);
      const f32x4 w = *reinterpret_cast<const f32x4*>(wr + k);
      s += a[0] * w[0] + a[1] * w[1] + a[2] * w[2] + a[3] * w[3];
    }
  } else {
    for (int k = threadIdx.x; k < K; k += 256) s += xr[k] * wr[k];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) y[(size_t)b * O + o] = s + (bias ? bias[o] : 0.f);
}

bool dense_head_applicable(const srgan_conv_desc* d) {
  return d->Ho == 1 && d->Wo == 1 && d->pad == 0 && d->kh == d->Hi && d->kw == d->Wi && d->O <= 64;
}

int dense_head_fwd(const srgan_conv_desc* d, const float* x, const float* wp, int Kpad, const float* bias, float* y,
                   hipStream_t st) {
  const int K = d->kh * d->kw * d->I;
  hipLaunchKernelGGL(dense_head_kernel, dim3((unsigned)d->O, (unsigned)d->N), dim3(256), 0, st, x, wp, bias, y, K, Kpad, d->O);
  return check_launch("dense_head_kernel");
}

}  // namespace srgan

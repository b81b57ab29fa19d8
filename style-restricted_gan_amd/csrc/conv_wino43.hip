// Winograd F(4x4,3x3) for the 3x3 / stride-1 / zero-pad-1 layers whose maps are multiples of 4 (the generator's residual
// trunk, reference pyfiles/model.py:188-201): 36 multiplies per 16 outputs instead of 144 -- 4x fewer MFMA FLOPs than the
// implicit GEMM, 1.78x fewer than F(2x2,3x3) (conv_wino.hip), every product still an exact fp32 product on
// v_mfma_f32_32x32x2_f32.       Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A,   4x4 output tile, 6x6 input patch.
//
// Two measured properties of the fp32 matrix path on gfx950 shape the design (scratch/coissue, scratch/shadow):
//   * v_mfma_f32_32x32x2_f32 and the vector ALU do not overlap: a wave's own VALU instructions queue behind its MFMA, and
//     a second wave on the SIMD issues nothing while the first issues MFMAs back to back (same 157 TFLOP/s peak for both:
//     the fp32 MFMA runs on the vector lanes).  VALU work inside the multiply loop is therefore ADDED to the MFMA time;
//   * the L1 takes one (instruction, 128-byte line) pair per ~4 cycles: gathering 6x6 patches of 8-channel (32-byte)
//     pixels costs 2304 such pairs per chunk and 64 tiles -- twice the MFMA time of the chunk.
// So the input transform is NOT fused: wino43_input_kernel (HBM-bound, full-line loads and stores) writes V = B^T d B once,
// already in the LDS image order of the multiply kernel ([tile block][chunk][36 pos][2 channel quads][64 tiles][4]), and
// wino43_kernel is a batched GEMM whose loop holds no vector arithmetic at all: per 8-channel chunk a workgroup copies one
// contiguous 72 KB block into LDS (9 x 16 bytes per thread, double-buffered, one barrier per chunk), every wave multiplies
// 9 of the 72 (position, tile half) units with A = U fragment (32 output channels x 2 reduce channels, straight from the
// packed filter image into registers, reloaded in place one chunk ahead) and B = V fragment (one ds_read_b128).
// An accumulator lane is a TILE and its 16 registers are output channels: the epilogue moves 8 output channels at a time
// through LDS (double-buffered, one barrier per pass); each thread applies A^T . A to one (tile, channel), adds the bias,
// applies the activation and stores 16 pixels.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "pack_device.h"

namespace srgan {

// B^T of F(4,3) on a 6-vector, in place (12 operations per component):
//   [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void bt6(f32x4& x0, f32x4& x1, f32x4& x2, f32x4& x3, f32x4& x4, f32x4& x5) {
  const f32x4 a = x4 - 4.f * x2, b = x3 - 4.f * x1, c = x4 - x2, e = x3 - x1;
  const f32x4 n0 = 4.f * x0 - 5.f * x2 + x4, n5 = 4.f * x1 - 5.f * x3 + x5;
  x0 = n0; x1 = a + b; x2 = a - b; x3 = c + 2.f * e; x4 = c - 2.f * e; x5 = n5;
}

// A^T of F(4,3): [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float* y) {
  const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
  y[0] = m0 + s1 + s2;
  y[1] = d1 + 2.f * d2;
  y[2] = s1 + 4.f * s2;
  y[3] = d1 + 8.f * d2 + m5;
}

__device__ __forceinline__ auto uniform_rsrc43(const float* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  float* q = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

constexpr int W4T = 64;           // output tiles (4x4 pixels each) per workgroup
constexpr int W4N = 32;           // output channels per workgroup
constexpr int W4C = 8;            // reduce channels per chunk
constexpr int W4BLK = 36 * 512;   // floats of one (tile block, chunk) image: [36 pos][2 quads][64 tiles][4]

// ---- input transform: V[m_tile][chunk][pos][quad][tile][4] = B^T d B, one workgroup = 64 tiles x 32 channels ----
// thread = (tile, 4 channels): a pixel's 32 channels are one 128-byte line read by 8 lanes, and the 8 tiles of a wave
// write 128 contiguous bytes per (position, quad): every load and store instruction moves whole lines.
__global__ __launch_bounds__(512) void wino43_input_kernel(WinoParams p, float* vimg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cgs = p.C / 32;
  // The channel groups of one tile block read 128-byte pieces of the SAME 1 KB pixel lines.  Consecutive workgroup ids go
  // round-robin to the 8 XCDs, so with m_tile = id / cgs the pieces of a line would be fetched through 8 different L2s at
  // different times; remapped, all channel groups of tile block m run back to back on XCD m % 8.
  int m_tile, cg;
  if ((p.m_tiles & 7) == 0) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    cg = k % cgs;
    m_tile = (k / cgs) * 8 + xcd;
  } else {
    m_tile = blockIdx.x / cgs;
    cg = blockIdx.x - m_tile * cgs;
  }
  const int quad = lane & 7, tl = wave * 8 + (lane >> 3);
  const int t = m_tile * W4T + tl;
  const bool tv = t < p.T;
  const int tt = tv ? t : 0;
  const int per = p.TH * p.TW;
  const int nb = tt / per;
  const int r0 = tt - nb * per;
  const int ty = r0 / p.TW, tx = r0 - ty * p.TW;
  const int Y = 4 * ty, X = 4 * tx;
  const int RS = p.W * p.C * 4, CS = p.C * 4;      // byte strides of a row / a pixel of the source
  const auto rs_x = uniform_rsrc43(p.src, (unsigned)((size_t)p.NB * p.H * p.W * p.C * 4));
  constexpr unsigned kOutside = 0x80000000u;
  // rows Y-1 .. Y+4: only the first can be above the image, only the last below it (H % 4 == 0); columns alike.
  // The scalar offset of element (r, c) is clamp(r-1, 0, 3) * RS + clamp(c-1, 0, 3) * CS; the class adds the rest, or
  // points past the buffer (the range check then returns the zero padding).
  const unsigned base = (unsigned)(((nb * p.H + Y) * p.W + X) * p.C + cg * 32 + quad * 4) * 4u;
  const bool rok[3] = {Y > 0, true, Y + 4 < p.H};
  const bool cok[3] = {X > 0, true, X + 4 < p.W};
  const int rdl[3] = {-RS, 0, RS}, cdl[3] = {-CS, 0, CS};
  unsigned voff[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) voff[a][b] = (tv && rok[a] && cok[b]) ? (unsigned)((int)base + rdl[a] + cdl[b]) : kOutside;
  f32x4 d[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int rc = r == 0 ? 0 : (r == 5 ? 2 : 1), cc = c == 0 ? 0 : (c == 5 ? 2 : 1);
      const int rs = (r < 1 ? 0 : (r > 4 ? 3 : r - 1)), cs = (c < 1 ? 0 : (c > 4 ? 3 : c - 1));
      d[r][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff[rc][cc], rs * RS + cs * CS, 0));
    }
#pragma unroll
  for (int c = 0; c < 6; ++c) bt6(d[0][c], d[1][c], d[2][c], d[3][c], d[4][c], d[5][c]);
  float* out = vimg + ((size_t)m_tile * p.nchunk + cg * 4 + (quad >> 1)) * W4BLK + (quad & 1) * 256 + tl * 4;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    bt6(d[r][0], d[r][1], d[r][2], d[r][3], d[r][4], d[r][5]);
#pragma unroll
    for (int c = 0; c < 6; ++c) *reinterpret_cast<f32x4*>(out + (r * 6 + c) * 512) = d[r][c];
  }
}

// ---- instance norm + activation + input transform in one pass (32x32 maps: the generator's residual trunk) ----
// SingleResidualBlock (model.py:196-201): conv -> CBIN -> ReLU -> conv.  The normalised activation h between the two
// convolutions has exactly one reader -- the second convolution's input transform -- and the F(4x4,3x3) weight gradient reads
// that transform's V image, not h.  So h is never written: one workgroup holds the (image, 32-channel) slab of the conv output
// in registers like in_fwd_slab (statistics by shuffles + one LDS exchange, exact two-pass variance, same normalise / affine /
// activation expression), parks the normalised slab in LDS (1024 pixels x 36 floats: the 4-float pad spreads the 8 tiles of a
// wave over the banks) and transforms it from there as wino43_input_kernel does: thread = (tile, 4 channels), 36 ds_read_b128
// (out-of-image taps are zeros: the padding applies to h), B^T d B in registers, 36 16-byte stores in the multiply kernel's
// LDS image order.  Reads 1 x and writes 2.25 x the tensor instead of (1 + 1) + (1 + 2.25) x, one launch instead of two.
// 64 tiles per image = one tile block, so m_tile = image; a 16-channel slab = two 8-channel chunks of the V image.
// 16-channel slabs, 256 threads (4 lanes per pixel), 64 KB of LDS: two workgroups share a CU, so one's load / reduce phases
// overlap the other's transform / store phases (a 32-channel, 512-thread, 148 KB version measured the same: ~29 us per launch at
// batch 32 = 3.7 TB/s).  Unpadded 64-byte LDS rows: the 8 tiles of a wave collide on 16 banks, ~1 us of the launch.
template <int NW>
__device__ __forceinline__ f32x4 slab_sum43_q4(f32x4 v, f32x4 (*sh)[4], int q, int wave) {
#pragma unroll
  for (int o = 4; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) < 4) sh[wave][q] = v;
  __syncthreads();
  f32x4 t = sh[0][q];
#pragma unroll
  for (int w = 1; w < NW; ++w) t += sh[w][q];
  return t;
}

__global__ __launch_bounds__(256, 2) void in_fwd_slab_v16_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, float* __restrict__ mean,
                                                                 float* __restrict__ rstd, float* __restrict__ vimg, int N, int C,
                                                                 float eps, int act, float slope) {
  constexpr int HW = 1024, R = 16, LD = 16;
  __shared__ __attribute__((aligned(16))) float hb[HW * LD];
  __shared__ f32x4 sh[4][4];
  const int tid = threadIdx.x, q = tid & 3, ty = tid >> 2, wave = tid >> 6, lane = tid & 63;
  const int nslab = C / 16;
  int slab, n;
  {
    const int L = blockIdx.x;
    if ((N & 7) == 0) { const int xcd = L & 7, k = L >> 3; slab = k % nslab; n = (k / nslab) * 8 + xcd; }
    else { slab = L % nslab; n = L / nslab; }
  }
  const int c = slab * 16 + q * 4;
  const size_t base = (size_t)n * HW * C + c;
  f32x4 v[R];
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < R; ++j) {
    v[j] = *reinterpret_cast<const f32x4*>(x + base + (size_t)(ty + 64 * j) * C);
    s += v[j];
  }
  const float inv = 1.f / (float)HW;
  const f32x4 mu = slab_sum43_q4<4>(s, sh, q, wave) * inv;
  f32x4 m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const f32x4 d = v[j] - mu;
    m2 += d * d;
  }
  const f32x4 var = slab_sum43_q4<4>(m2, sh, q, wave) * inv;
  f32x4 rs;
#pragma unroll
  for (int e = 0; e < 4; ++e) rs[e] = 1.0f / sqrtf(var[e] + eps);
  const int nc = n * C + c;
  if (ty == 0) {
    *reinterpret_cast<f32x4*>(mean + nc) = mu;
    *reinterpret_cast<f32x4*>(rstd + nc) = rs;
  }
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
  if (scale) {
    sc = *reinterpret_cast<const f32x4*>(scale + nc);
    sf = *reinterpret_cast<const f32x4*>(shift + nc);
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    f32x4 o = ((v[j] - mu) * rs) * sc + sf;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = apply_act(o[e], act, slope);
    *reinterpret_cast<f32x4*>(&hb[(ty + 64 * j) * LD + q * 4]) = o;
  }
  __syncthreads();
  // transform role: thread = (tile, channel quad of the 16): 64 tiles x 4 quads
  const int quad = lane & 3, tl = wave * 16 + (lane >> 2);
  const int Y = 4 * (tl >> 3), X = 4 * (tl & 7);
  f32x4 d[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) {
      const int yy = Y - 1 + r, xx = X - 1 + cc;
      const bool ok = (unsigned)yy < 32u && (unsigned)xx < 32u;
      d[r][cc] = ok ? *reinterpret_cast<const f32x4*>(&hb[(yy * 32 + xx) * LD + quad * 4]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
  for (int cc = 0; cc < 6; ++cc) bt6(d[0][cc], d[1][cc], d[2][cc], d[3][cc], d[4][cc], d[5][cc]);
  const int nchunk = C / W4C;
  float* out = vimg + ((size_t)n * nchunk + slab * 2 + (quad >> 1)) * W4BLK + (quad & 1) * 256 + tl * 4;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    bt6(d[r][0], d[r][1], d[r][2], d[r][3], d[r][4], d[r][5]);
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) *reinterpret_cast<f32x4*>(out + (r * 6 + cc) * 512) = d[r][cc];
  }
}

int in_fwd_slab_v_launch(const float* x, const float* scale, const float* shift, float* mean, float* rstd, float* vimg, int N,
                         int C, float eps, int act, float slope, hipStream_t st) {
  // algorithmic bytes: the 32 x 32 x C tensor read once, its transform image (2.25x) written once
  ProfToken tok = prof_begin(38, 3.25 * 4.0 * N * 1024.0 * C, st);
  hipLaunchKernelGGL(in_fwd_slab_v16_kernel, dim3((unsigned)(N * (C / 16))), dim3(256), 0, st, x, scale, shift, mean, rstd, vimg, N,
                       C, eps, act, slope);
  prof_end(tok, st);
  return check_launch("in_fwd_slab_v_kernel");
}

// ---- instance-norm BACKWARD writing both transforms of its result (32x32 maps, residual trunk) ----
// dy = gradient w.r.t. the output of a convolution c that is followed by (CB)IN (+ activation).  dy has exactly two readers: c's
// input-gradient kernel (through B^T dy B, the V image of the transposed problem) and c's F(4x4,3x3) weight gradient (through
// Z = A dy A^T).  So dy is never written: the slab kernel of in_bwd_slab (x-hat and the masked upstream gradient in registers,
// two plane sums, dy = rstd * scale * (g - mean(g) - x-hat * mean(g x-hat))) parks dy in LDS and writes BOTH images from there,
// in the layouts wino43_input_kernel and wino43_dy_kernel produce.  Reads x, g (2x) and writes 2.25x + 2.25x the tensor instead
// of [2 + 1] + [1 + 2.25] + [1 + 2.25]; one launch instead of three.
__global__ __launch_bounds__(256, 2) void in_bwd_slab_vz_kernel(const float* __restrict__ x, const float* __restrict__ gup,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                float* __restrict__ dscale, float* __restrict__ dshift,
                                                                float* __restrict__ vimg, float* __restrict__ zimg, int N, int C,
                                                                int act, float slope) {
  constexpr int HW = 1024, R = 16, LD = 16;
  __shared__ __attribute__((aligned(16))) float hb[HW * LD];
  __shared__ f32x4 sh[4][4];
  const int tid = threadIdx.x, q = tid & 3, ty = tid >> 2, wave = tid >> 6, lane = tid & 63;
  const int nslab = C / 16;
  int slab, n;
  {
    const int L = blockIdx.x;
    if ((N & 7) == 0) { const int xcd = L & 7, k = L >> 3; slab = k % nslab; n = (k / nslab) * 8 + xcd; }
    else { slab = L % nslab; n = L / nslab; }
  }
  const int c = slab * 16 + q * 4;
  const int nc = n * C + c;
  const size_t base = (size_t)n * HW * C + c;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + nc), rs = *reinterpret_cast<const f32x4*>(rstd + nc);
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
  if (scale) {
    sc = *reinterpret_cast<const f32x4*>(scale + nc);
    sf = *reinterpret_cast<const f32x4*>(shift + nc);
  }
  {
    f32x4 xh[R], g[R];
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int r = ty + 64 * j;
      xh[j] = (*reinterpret_cast<const f32x4*>(x + base + (size_t)r * C) - mu) * rs;
      g[j] = *reinterpret_cast<const f32x4*>(gup + base + (size_t)r * C);
      const f32x4 z = xh[j] * sc + sf;
#pragma unroll
      for (int e = 0; e < 4; ++e) g[j][e] *= act_grad(z[e], act, slope);
      a += g[j];
      b += g[j] * xh[j];
    }
    a = slab_sum43_q4<4>(a, sh, q, wave);
    b = slab_sum43_q4<4>(b, sh, q, wave);
    if (ty == 0) {
      *reinterpret_cast<f32x4*>(dshift + nc) = a;
      *reinterpret_cast<f32x4*>(dscale + nc) = b;
    }
    const float inv_hw = 1.f / (float)HW;
    const f32x4 mg = a * inv_hw, mgx = b * inv_hw, k = rs * sc;
#pragma unroll
    for (int j = 0; j < R; ++j)
      *reinterpret_cast<f32x4*>(&hb[(ty + 64 * j) * LD + q * 4]) = k * (g[j] - mg - xh[j] * mgx);     // = in_bwd_slab's dx
  }
  __syncthreads();
  // (1) V image of the transposed problem (the input-gradient kernel's B operand): as in_fwd_slab_v16_kernel
  {
    const int quad = lane & 3, tl = wave * 16 + (lane >> 2);
    const int Y = 4 * (tl >> 3), X = 4 * (tl & 7);
    f32x4 d[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) {
        const int yy = Y - 1 + r, xx = X - 1 + cc;
        const bool ok = (unsigned)yy < 32u && (unsigned)xx < 32u;
        d[r][cc] = ok ? *reinterpret_cast<const f32x4*>(&hb[(yy * 32 + xx) * LD + quad * 4]) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) bt6(d[0][cc], d[1][cc], d[2][cc], d[3][cc], d[4][cc], d[5][cc]);
    const int nchunk = C / W4C;
    float* out = vimg + ((size_t)n * nchunk + slab * 2 + (quad >> 1)) * W4BLK + (quad & 1) * 256 + tl * 4;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      bt6(d[r][0], d[r][1], d[r][2], d[r][3], d[r][4], d[r][5]);
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) *reinterpret_cast<f32x4*>(out + (r * 6 + cc) * 512) = d[r][cc];
    }
  }
  // (2) Z = A dy A^T in the A-operand register image of wino43_wgrad_kernel: [32-o block][8-tile chunk][36][lane = half * 32 + o][4 tiles]
  //     thread = (channel o16 of the slab, tile half lh, tile chunk tc of the image): 4 tiles each
  {
    const int o16 = tid & 15, lh = (tid >> 4) & 1, tc = tid >> 5;
    float z[36][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = tc * 8 + lh * 4 + j;
      const int Y = 4 * (t >> 3), X = 4 * (t & 7);
      float e[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) e[a][b] = hb[((Y + a) * 32 + X + b) * LD + o16];
      float h[6][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float v0 = e[0][b], v1 = e[1][b], v2 = e[2][b], v3 = e[3][b];
        const float s02 = v0 + v2, s13 = v1 + v3, q02 = v0 + 4.f * v2, q13 = 2.f * v1 + 8.f * v3;
        h[0][b] = v0; h[1][b] = s02 + s13; h[2][b] = s02 - s13; h[3][b] = q02 + q13; h[4][b] = q02 - q13; h[5][b] = v3;
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const float v0 = h[r][0], v1 = h[r][1], v2 = h[r][2], v3 = h[r][3];
        const float s02 = v0 + v2, s13 = v1 + v3, q02 = v0 + 4.f * v2, q13 = 2.f * v1 + 8.f * v3;
        z[r * 6 + 0][j] = v0; z[r * 6 + 1][j] = s02 + s13; z[r * 6 + 2][j] = s02 - s13;
        z[r * 6 + 3][j] = q02 + q13; z[r * 6 + 4][j] = q02 - q13; z[r * 6 + 5][j] = v3;
      }
    }
    const int ntc = N * 8;
    const int zlane = lh * 32 + (slab & 1) * 16 + o16;
    float* out = zimg + ((size_t)((slab >> 1) * ntc + n * 8 + tc) * 36) * 256 + zlane * 4;
#pragma unroll
    for (int k = 0; k < 36; ++k) {
      const f32x4 v = {z[k][0], z[k][1], z[k][2], z[k][3]};
      *reinterpret_cast<f32x4*>(out + k * 256) = v;
    }
  }
}

int in_bwd_slab_vz_launch(const float* x, const float* gup, const float* scale, const float* shift, const float* mean,
                          const float* rstd, float* dscale, float* dshift, float* vimg, float* zimg, int N, int C, int act,
                          float slope, hipStream_t st) {
  // algorithmic bytes: x and the upstream gradient read once, both transform images (2.25x each) written once
  ProfToken tok = prof_begin(39, 6.5 * 4.0 * N * 1024.0 * C, st);
  hipLaunchKernelGGL(in_bwd_slab_vz_kernel, dim3((unsigned)(N * (C / 16))), dim3(256), 0, st, x, gup, scale, shift, mean, rstd, dscale,
                     dshift, vimg, zimg, N, C, act, slope);
  prof_end(tok, st);
  return check_launch("in_bwd_slab_vz_kernel");
}

// ---- multiply + output transform ----
// WinoParams as in conv_wino.hip with TH = Ho / 4, TW = Wo / 4, n_tiles = Cd / 32, nchunk = C / 8, pad = 1;
// u = [n_tiles][nchunk][36 pos][64 lanes][4]: lane (lr, lh) holds output channel lr, reduce channels 4 lh .. 4 lh + 3.
// RES: the epilogue adds p.res (input gradient of a residual block's first convolution: the skip path's gradient rides in
// here).  A separate instantiation: a run-time branch in the epilogue cost every launch ~6 us (146 -> 154 us average).
template <bool RES>
__global__ __launch_bounds__(512) void wino43_kernel(WinoParams p, const float* vimg) {
  // V image, double-buffered: [buf][36 pos][2 channel quads][64 tiles x 4 ch + 16 pad]: a lane's MFMA fragment is one 16-B
  // slot, a 16-lane read group covers 256 contiguous bytes; the copy writes 1 KB per wave instruction.
  constexpr int VH = 64 * 4 + 16, VP = 2 * VH, VSZ = 36 * VP;
  constexpr int XT = 36, XP = 16 * XT;             // epilogue image: [36 pos][16 tiles][32 channels + 4 pad]
  static_assert(36 * XP <= 2 * VSZ, "epilogue image must fit the V buffers");
  __shared__ __attribute__((aligned(16))) float lds[2 * VSZ];
  __shared__ int tile_o[W4T];                      // destination pixel index of the tile's first output, or -1

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);      // XCD-aware: see wino_kernel
  const int m_tile = bid / p.n_tiles, n_tile = bid - m_tile * p.n_tiles;
  const int nk = p.nchunk;

  const auto rs_v = uniform_rsrc43(vimg + (size_t)m_tile * p.nchunk * W4BLK, (unsigned)p.nchunk * (W4BLK * 4u));
  const auto rs_u = uniform_rsrc43(p.u + (size_t)n_tile * p.nchunk * (36 * 256), (unsigned)p.nchunk * (36u * 1024u));

  if (tid < W4T) {
    const int t = m_tile * W4T + tid;
    int o = -1;
    if (t < p.T) {
      const int per = p.TH * p.TW;
      const int nb = t / per;
      const int r0 = t - nb * per;
      const int ty = r0 / p.TW, tx = r0 - ty * p.TW;
      o = (nb * p.Ho + 4 * ty) * p.Wo + 4 * tx;
    }
    tile_o[tid] = o;
  }
  // copy role: float4 j * 512 + tid of the chunk image -> position 4 j + tid / 128, quad (tid / 64) & 1, slot tid & 63
  f32x4 stage[9];
  const int vst = (tid >> 7) * VP + ((tid >> 6) & 1) * VH + (tid & 63) * 4;
  auto load_v1 = [&](int kc, int j) __attribute__((always_inline)) {
    stage[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, tid * 16, kc * (W4BLK * 4) + j * 8192, 0));
  };
  auto store_v1 = [&](int buf, int j) __attribute__((always_inline)) {
    *reinterpret_cast<f32x4*>(lds + buf * VSZ + vst + j * 4 * VP) = stage[j];
  };
  auto load_v = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 9; ++j) load_v1(kc, j);
  };
  auto store_v = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 9; ++j) store_v1(buf, j);
  };

  // multiply role: units 9 wave .. 9 wave + 8 of the 72 (position, tile half) pairs
  auto role = [&](auto odd_c) __attribute__((always_inline)) {
    constexpr int ODD = decltype(odd_c)::value;   // parity of the first unit = first tile half
    const int pbase = (9 * wave) >> 1;             // first of the 5 positions this wave touches
    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x4 ufr[5];
    const unsigned ulane = (unsigned)(pbase * 256 + lane * 4) * 4u;
    auto load_u = [&](int a, int kc) __attribute__((always_inline)) {
      // position offset in the SCALAR offset: one address register for the five fragments
      ufr[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, ulane, kc * (36 * 1024) + a * 1024, 0));
    };
    const int vrd = pbase * VP + lh * VH + lr * 4;
    // The copy of the next chunk is spread over the nine units (one 16-byte slice each: LDS store of the slice loaded a
    // chunk ago, then the load of the slice two chunks ahead): LDS and buffer instructions issue in the shadow of the
    // wave's own MFMAs, whereas a separate copy phase would stall behind the MFMAs of the other wave of the SIMD.
    auto mma_chunk = [&](int kc, auto reload, auto st_c, auto ld_c) __attribute__((always_inline)) {
      constexpr bool RL = decltype(reload)::value, ST = decltype(st_c)::value, LD = decltype(ld_c)::value;
      const float* V = lds + (kc & 1) * VSZ + vrd;
      f32x4 vf[2];
      vf[0] = *reinterpret_cast<const f32x4*>(V + (ODD >> 1) * VP + (ODD & 1) * 128);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int g = i + ODD, a = g >> 1, h = g & 1;
        // fragment of the next unit first: its LDS latency hides behind this unit's four MFMAs
        if (i + 1 < 9) vf[(i + 1) & 1] = *reinterpret_cast<const f32x4*>(V + ((g + 1) >> 1) * VP + ((g + 1) & 1) * 128);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[a][s], vf[i & 1][s], acc[i], 0, 0, 0);
        // last unit of position a: reload its filter fragment in place for the next chunk (it lands a whole period later)
        if (RL && (h == 1 || i == 8)) load_u(a, kc + 1);
        if constexpr (ST) store_v1((kc + 1) & 1, i);
        if constexpr (LD) load_v1(kc + 2, i);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    using T = std::true_type;
    using F = std::false_type;
    // prologue: chunk 0 in LDS, chunk 1 in flight; filter loads AFTER the image loads, as in the loop, so that the
    // vmcnt state the compiler merges at the loop header lets the copy wait for the image only
    load_v(0);
    store_v(0);
    load_v(1);                                     // nk >= 2
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 5; ++a) load_u(a, 0);
    __syncthreads();
    int kc = 0;
#ifdef W43_DIAG
    long long t_cp = 0, t_mm = 0, t_bar = 0;
    const long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    for (; kc + 2 < nk; ++kc) {
#ifdef W43_DIAG
      __builtin_amdgcn_sched_barrier(0);
      const long long q0 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef W43_DIAG
      const long long q1 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
#endif
      mma_chunk(kc, T{}, T{}, T{});
#ifdef W43_DIAG
      __builtin_amdgcn_sched_barrier(0);
      const long long q2 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      const long long q3 = __builtin_amdgcn_s_memtime();
      t_cp += q1 - q0; t_mm += q2 - q1; t_bar += q3 - q2;
#else
      __syncthreads();
#endif
    }
#ifdef W43_DIAG
    {
      const long long t_loop = __builtin_amdgcn_s_memtime() - t_begin;
      if (blockIdx.x == 17 && lane == 0) {
        float* o = p.dst + wave * 8;
        o[0] = (float)t_loop; o[1] = (float)t_cp; o[2] = (float)t_mm; o[3] = (float)t_bar; o[4] = (float)(nk - 2);
      }
      float keep = 0.f;
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) keep += acc[i][e];
      p.dst[4096 + (size_t)blockIdx.x * 512 + tid] = keep;
      return;
    }
#endif
    mma_chunk(kc, T{}, T{}, F{});
    __syncthreads();
    mma_chunk(kc + 1, F{}, F{}, F{});
    __syncthreads();                               // every wave is done with V: the epilogue image may overwrite it

    // ---- epilogue: 16 tiles per pass meet in LDS as X[pos][tile][32 channels + 4 pad]; thread = (tile, channel): the 32
    // lanes of a tile store one whole 128-byte line per pixel ----
    const int ec = lane & 31, et = wave + 8 * (lane >> 5);      // tiles et, et + 8 of a wave: 288 floats apart, other banks
    const int n = n_tile * W4N + ec;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q > 0) __syncthreads();                  // the previous pass has been read
      // RES: this pass's 16 values of the skip gradient are requested FIRST, so that their latency hides behind the LDS
      // exchange and the output transform (loaded next to the stores they took 32 us per launch at batch 32)
      float rv[16];
      if constexpr (RES) {
        const int eo_r = tile_o[16 * q + et];
        const float* rp = p.res + (size_t)(eo_r >= 0 ? eo_r : 0) * p.Cd + n;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) rv[a * 4 + b] = eo_r >= 0 ? rp[(size_t)(a * p.Wo + b) * p.Cd] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int g = i + ODD, a = g >> 1, h = g & 1;
        // accumulator lane = tile 32 h + lr, register e = output channel 8 * (e / 4) + 4 * lh + e % 4
        if (h == (q >> 1) && (lr >> 4) == (q & 1)) {
          float* X = lds + (pbase + a) * XP + (lr & 15) * XT + lh * 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x4 v = {acc[i][4 * k + 0], acc[i][4 * k + 1], acc[i][4 * k + 2], acc[i][4 * k + 3]};
            *reinterpret_cast<f32x4*>(X + 8 * k) = v;
          }
        }
      }
      __syncthreads();
      const float* M = lds + et * XT + ec;
      float hh[4][6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        float y[4];
        at6(M[(0 * 6 + c) * XP], M[(1 * 6 + c) * XP], M[(2 * 6 + c) * XP], M[(3 * 6 + c) * XP], M[(4 * 6 + c) * XP],
            M[(5 * 6 + c) * XP], y);
#pragma unroll
        for (int a = 0; a < 4; ++a) hh[a][c] = y[a];
      }
      const int eo = tile_o[16 * q + et];
      if (eo >= 0) {
        float* dp = p.dst + (size_t)eo * p.Cd + n;
        if constexpr (RES) {
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            float y[4];
            at6(hh[a][0], hh[a][1], hh[a][2], hh[a][3], hh[a][4], hh[a][5], y);
#pragma unroll
            for (int b = 0; b < 4; ++b) dp[(size_t)(a * p.Wo + b) * p.Cd] = apply_act(y[b] + bv, p.act, p.slope) + rv[a * 4 + b];
          }
        } else {
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            float y[4];
            at6(hh[a][0], hh[a][1], hh[a][2], hh[a][3], hh[a][4], hh[a][5], y);
#pragma unroll
            for (int b = 0; b < 4; ++b) dp[(size_t)(a * p.Wo + b) * p.Cd] = apply_act(y[b] + bv, p.act, p.slope);
          }
        }
      }
    }
  };

  if (wave & 1) role(std::integral_constant<int, 1>{});
  else role(std::integral_constant<int, 0>{});
}

// ---- weight gradient on the SAME transformed input:  dU[pos][o][c] = sum_tiles Z[pos][tile][o] * V[pos][tile][c],
//      Z = A dY A^T (6x6 from the 4x4 output-gradient tile),  dW = G^T dU G  ----
// V is the image the forward pass wrote (kept by the caller instead of being scratch), so the weight gradient costs one
// transform pass (dy -> Z) instead of two and 1.78x fewer MFMA FLOPs than the F(2x2,3x3) kernel.
// wino43_dy_kernel writes Z as the A-operand register image [32-o block][8-tile chunk][36 pos][lane = half * 32 + o][4 tiles];
// wino43_wgrad_kernel is wino43_kernel with the reduce index = tiles: workgroup = 32 o x 64 c x 36 positions over a split of
// the tile chunks; the B operand comes from the forward image [tile block][c chunk][pos][quad][64 tiles][4 c] -- 576 pieces of
// 128 bytes per 8-tile chunk, copied to LDS as [pos][c quad 16][tile ^ (quad & 7)][4 c] (the XOR keeps the four ds_read_b32 of
// a fragment off each other's banks); epilogue: 8 output channels per pass meet in LDS, thread = (o, c) applies G^T . G and
// writes nine taps into the [split][O][9 C] slab of the split-K reduce.
struct Wino43WgradParams {
  const float* vimg;
  const float* zimg;
  float* slab;
  int C, O, cchunks, ntc, chunks_per_split, splits, c_blocks, o_blocks;
  int split_minor;     // 1: the split index is the FAST index of blockIdx.x (see the kernel)
};

struct Wino43DyParams {
  const float* dy;    // [NB][H][W][O]
  float* zimg;
  int NB, H, W, O, TH, TW, T, ntc, o_blocks;
};

__global__ __launch_bounds__(512) void wino43_dy_kernel(Wino43DyParams p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long unit = (long long)blockIdx.x * 8 + wave;      // one wave = one (tile chunk, o block)
  if (unit >= (long long)p.ntc * p.o_blocks) return;
  const int o_blk = (int)(unit % p.o_blocks), tchunk = (int)(unit / p.o_blocks);
  const int o = o_blk * 32 + (lane & 31), lh = lane >> 5;
  const auto rs = uniform_rsrc43(p.dy, (unsigned)((size_t)p.NB * p.H * p.W * p.O * 4));
  float z[36][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = tchunk * 8 + lh * 4 + j;
    const bool tv = t < p.T;
    const int tt = tv ? t : 0;
    const int per = p.TH * p.TW;
    const int nb = tt / per;
    const int r0 = tt - nb * per;
    const int ty = r0 / p.TW, tx = r0 - ty * p.TW;
    const unsigned base = tv ? (unsigned)(((nb * p.H + 4 * ty) * p.W + 4 * tx) * p.O + o) * 4u : 0x80000000u;
    float e[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        e[a][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, base, (a * p.W + b) * p.O * 4, 0));
    // A = [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 2 4 8; 1 -2 4 -8; 0 0 0 1]: columns first, then rows
    float h[6][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float v0 = e[0][b], v1 = e[1][b], v2 = e[2][b], v3 = e[3][b];
      const float s02 = v0 + v2, s13 = v1 + v3, q02 = v0 + 4.f * v2, q13 = 2.f * v1 + 8.f * v3;
      h[0][b] = v0; h[1][b] = s02 + s13; h[2][b] = s02 - s13; h[3][b] = q02 + q13; h[4][b] = q02 - q13; h[5][b] = v3;
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const float v0 = h[r][0], v1 = h[r][1], v2 = h[r][2], v3 = h[r][3];
      const float s02 = v0 + v2, s13 = v1 + v3, q02 = v0 + 4.f * v2, q13 = 2.f * v1 + 8.f * v3;
      z[r * 6 + 0][j] = v0; z[r * 6 + 1][j] = s02 + s13; z[r * 6 + 2][j] = s02 - s13;
      z[r * 6 + 3][j] = q02 + q13; z[r * 6 + 4][j] = q02 - q13; z[r * 6 + 5][j] = v3;
    }
  }
  float* out = p.zimg + ((size_t)(o_blk * p.ntc + tchunk) * 36) * 256 + lane * 4;
#pragma unroll
  for (int k = 0; k < 36; ++k) {
    const f32x4 v = {z[k][0], z[k][1], z[k][2], z[k][3]};
    *reinterpret_cast<f32x4*>(out + k * 256) = v;
  }
}

__global__ __launch_bounds__(512) void wino43_wgrad_kernel(Wino43WgradParams p) {
  constexpr int BSZ = 36 * 512;                    // floats of one chunk image in LDS: [36 pos][16 c quads][8 tiles][4 c]
  __shared__ __attribute__((aligned(16))) float lds[2 * BSZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  // Workgroup b runs on XCD b % 8.  All (o block, c block) workgroups of one split read the SAME 1/splits of the V and Z images
  // (V once per o block, Z once per c block), so the split index is the fast index: with 8 splits (the 256-channel trunk) every
  // split's 32 workgroups share one XCD and its L2 -- each line of the two images is fetched from the fabric once instead of
  // through two (V) / four (Z) different L2s.
  int bid = blockIdx.x;
  const int per_split = p.o_blocks * p.c_blocks;
  int split;
  if (p.split_minor) { split = bid % p.splits; bid /= p.splits; }
  else { split = bid / per_split; bid -= split * per_split; }
  const int o_blk = bid / p.c_blocks, cb = bid - o_blk * p.c_blocks;
  const int tc0 = split * p.chunks_per_split;
  const int nk = min(p.chunks_per_split, p.ntc - tc0);      // >= 2 (wino43_wgrad_plan)

  const auto rs_v = uniform_rsrc43(p.vimg, (unsigned)((size_t)(p.ntc >> 3) * p.cchunks * W4BLK * 4));
  const auto rs_u = uniform_rsrc43(p.zimg + ((size_t)o_blk * p.ntc + tc0) * (36 * 256), (unsigned)nk * (36u * 1024u));

  // copy role: float4 j * 512 + tid of the chunk -> position 4 j + tid / 128, c quad (tid / 8) & 15, tile tid & 7
  f32x4 stage[9];
  const int cq = (tid >> 3) & 15;
  const unsigned vsrc = (unsigned)((cq >> 1) * W4BLK + (tid >> 7) * 512 + (cq & 1) * 256 + (tid & 7) * 4) * 4u;
  const int vst = (tid >> 7) * 512 + cq * 32 + (((tid & 7) ^ (cq & 7))) * 4;
  auto load_v1 = [&](int kc, int j) __attribute__((always_inline)) {
    const int tc = tc0 + kc;
    const unsigned so = (unsigned)(((tc >> 3) * p.cchunks + cb * 8) * W4BLK + (tc & 7) * 32 + j * 2048) * 4u;
    stage[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_v, vsrc, so, 0));
  };
  auto store_v1 = [&](int buf, int j) __attribute__((always_inline)) {
    *reinterpret_cast<f32x4*>(lds + buf * BSZ + vst + j * 2048) = stage[j];
  };

  auto role = [&](auto odd_c) __attribute__((always_inline)) {
    constexpr int ODD = decltype(odd_c)::value;
    const int pbase = (9 * wave) >> 1;
    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x4 ufr[5];
    const unsigned ulane = (unsigned)(pbase * 256 + lane * 4) * 4u;
    auto load_u = [&](int a, int kc) __attribute__((always_inline)) {
      ufr[a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, ulane, kc * (36 * 1024) + a * 1024, 0));
    };
    // B fragment of unit (position a, c half h), step s: tile 4 lh + s, channel 32 h + lr
    const int g4 = lr >> 2;
    const int vrd = pbase * 512 + g4 * 32 + (lr & 3);
    int tsl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) tsl[s] = ((4 * lh + s) ^ g4) * 4;
    auto read_b = [&](const float* V, int g, float* dst) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 4; ++s) dst[s] = V[(g >> 1) * 512 + (g & 1) * 256 + tsl[s]];
    };
    auto mma_chunk = [&](int kc, auto reload, auto st_c, auto ld_c) __attribute__((always_inline)) {
      constexpr bool RL = decltype(reload)::value, ST = decltype(st_c)::value, LD = decltype(ld_c)::value;
      const float* V = lds + (kc & 1) * BSZ + vrd;
      float vf[2][4];
      read_b(V, ODD, vf[0]);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int g = i + ODD, a = g >> 1, h = g & 1;
        if (i + 1 < 9) read_b(V, g + 1, vf[(i + 1) & 1]);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[a][s], vf[i & 1][s], acc[i], 0, 0, 0);
        if (RL && (h == 1 || i == 8)) load_u(a, kc + 1);
        if constexpr (ST) store_v1((kc + 1) & 1, i);
        if constexpr (LD) load_v1(kc + 2, i);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using T = std::true_type;
    using F = std::false_type;
#pragma unroll
    for (int j = 0; j < 9; ++j) load_v1(0, j);
#pragma unroll
    for (int j = 0; j < 9; ++j) store_v1(0, j);
#pragma unroll
    for (int j = 0; j < 9; ++j) load_v1(1, j);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 5; ++a) load_u(a, 0);
    __syncthreads();
    int kc = 0;
    for (; kc + 2 < nk; ++kc) {
      mma_chunk(kc, T{}, T{}, T{});
      __syncthreads();
    }
    mma_chunk(kc, T{}, T{}, F{});
    __syncthreads();
    mma_chunk(kc + 1, F{}, F{}, F{});

    // ---- epilogue: 8 output channels per pass meet in LDS as X[pos][o][64 c] (two buffers); thread = (o, c) ----
    const int eo = tid >> 6, ec = tid & 63;
    float* slab = p.slab + ((size_t)split * p.O + o_blk * 32) * (9 * p.C) + cb * 64 + ec;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      __syncthreads();                             // pass k - 2 has been read (k = 0: the V buffers are free)
      float* X = lds + (k & 1) * BSZ;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int g = i + ODD, a = g >> 1, h = g & 1;
        // accumulator lane = channel 32 h + lr, register e = output channel 8 * (e / 4) + 4 * lh + e % 4
#pragma unroll
        for (int j = 0; j < 4; ++j) X[(pbase + a) * 512 + (4 * lh + j) * 64 + h * 32 + lr] = acc[i][4 * k + j];
      }
      __syncthreads();
      const float* M = X + eo * 64 + ec;
      // t[a][c] = sum_r G[r][a] M[r][c],  G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
      float t[3][6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const float m0 = M[(0 * 6 + c) * 512], m1 = M[(1 * 6 + c) * 512], m2 = M[(2 * 6 + c) * 512], m3 = M[(3 * 6 + c) * 512],
                    m4 = M[(4 * 6 + c) * 512], m5 = M[(5 * 6 + c) * 512];
        const float s12 = m1 + m2, d12 = m2 - m1, s34 = m3 + m4, d34 = m3 - m4;
        t[0][c] = 0.25f * m0 - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
        t[1][c] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
        t[2][c] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + m5;
      }
      float* dst = slab + (size_t)(8 * k + eo) * (9 * p.C);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float s12 = t[a][1] + t[a][2], d12 = t[a][2] - t[a][1], s34 = t[a][3] + t[a][4], d34 = t[a][3] - t[a][4];
        dst[(a * 3 + 0) * p.C] = 0.25f * t[a][0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
        dst[(a * 3 + 1) * p.C] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
        dst[(a * 3 + 2) * p.C] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + t[a][5];
      }
    }
  };
  if (wave & 1) role(std::integral_constant<int, 1>{});
  else role(std::integral_constant<int, 0>{});
}

size_t wino43_scratch_floats(long long T, int C) { return (size_t)ceil_div(T, (long long)W4T) * (C / W4C) * W4BLK; }

// `flops`: the layer's ALGORITHMIC FLOPs, booked on the multiply kernel; the input transform is HBM-bound and has its own slot
int wino43_launch(const WinoParams& p, float* vimg, long long grid, double flops, hipStream_t st, bool v_ready) {
  // SRGAN_W43_ONLY=1 / 2 (timing experiments only): launch just the input transform / just the multiply kernel
  static const int only = SRGAN_AB_INT("SRGAN_W43_ONLY", 0);
  if (only != 2 && !v_ready) {      // v_ready: the caller's V image already holds B^T d B (in_fwd_slab_v_kernel wrote it)
    ProfToken tok = prof_begin(19, 0.0, st);
    hipLaunchKernelGGL(wino43_input_kernel, dim3((unsigned)(p.m_tiles * (p.C / 32))), dim3(512), 0, st, p, vimg);
    prof_end(tok, st);
  }
  if (only != 1) {
    ProfToken tok = prof_begin(18, flops, st);
    if (p.res) hipLaunchKernelGGL(wino43_kernel<true>, dim3((unsigned)grid), dim3(512), 0, st, p, (const float*)vimg);
    else hipLaunchKernelGGL(wino43_kernel<false>, dim3((unsigned)grid), dim3(512), 0, st, p, (const float*)vimg);
    prof_end(tok, st);
  }
  return 0;
}

// Weight gradient from the forward's V image.  g: geometry from wino43_wgrad_geometry (conv_wino.hip); zimg: Z scratch.
int wino43_wgrad_launch(const Wino43WgradGeom& g, const float* vimg, const float* dy, float* zimg, float* slab, double flops,
                        hipStream_t st, bool z_ready) {
  Wino43DyParams q{};
  q.dy = dy; q.zimg = zimg; q.NB = g.NB; q.H = g.H; q.W = g.W; q.O = g.O; q.TH = g.H / 4; q.TW = g.W / 4;
  q.T = g.NB * q.TH * q.TW; q.ntc = g.ntc; q.o_blocks = g.O / 32;
  if (!z_ready) {      // z_ready: in_bwd_slab_vz_kernel already wrote Z
    ProfToken tok = prof_begin(23, 0.0, st);
    hipLaunchKernelGGL(wino43_dy_kernel, dim3((unsigned)ceil_div((long long)q.ntc * q.o_blocks, 8)), dim3(512), 0, st, q);
    prof_end(tok, st);
  }
  Wino43WgradParams p{};
  p.vimg = vimg; p.zimg = zimg; p.slab = slab; p.C = g.C; p.O = g.O; p.cchunks = g.C / 8; p.ntc = g.ntc;
  p.chunks_per_split = g.chunks_per_split; p.splits = g.splits; p.c_blocks = g.C / 64; p.o_blocks = g.O / 32;
  p.split_minor = 1;
  ProfToken tok = prof_begin(22, flops, st);
  hipLaunchKernelGGL(wino43_wgrad_kernel, dim3((unsigned)(p.o_blocks * p.c_blocks * p.splits)), dim3(512), 0, st, p);
  prof_end(tok, st);
  return 0;
}

}  // namespace srgan

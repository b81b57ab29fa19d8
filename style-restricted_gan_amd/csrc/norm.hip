// Instance normalisation (+ per-(n,c) affine, activation, residual) and the central-biasing
// affine of CBIN, forward and backward, on NHWC fp32 tensors.  HBM-bound kernels: every pass
// reads each element once with channel-contiguous (coalesced) accesses; statistics are reduced
// per (n, c) column with a split over the H*W rows and a deterministic second stage.
//
// Reference: _CBINorm.forward (pyfiles/model.py:54-67), nn.InstanceNorm2d (model.py:178),
// formulas of SURVEY.md Appendix F.1.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "common.h"

namespace srgan {

constexpr int NORM_CH = 32;    // channels per block
constexpr int NORM_ROWS = 8;   // row groups per block (256 threads)

// 4 consecutive channels of an fp32 or bf16 tensor <-> f32x4 (bf16 tensors: the activations the bf16 mode keeps in 16 bits, ops.py)
template <bool B16>
__device__ __forceinline__ f32x4 ld4(const void* base, size_t idx) {
  if constexpr (B16) return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(base) + idx), f32x4);
  else return *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + idx);
}
template <bool B16>
__device__ __forceinline__ void st4(void* base, size_t idx, f32x4 v) {
  if constexpr (B16) *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(base) + idx) = __builtin_convertvector(v, bf16x4);
  else *reinterpret_cast<f32x4*>(static_cast<float*>(base) + idx) = v;
}


// 8 consecutive channels of a lane: one 16-byte access of a bf16 tensor, two of an fp32 one
template <bool B16>
__device__ __forceinline__ void ld8(const void* base, size_t idx, f32x4& lo, f32x4& hi) {
  if constexpr (B16) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(static_cast<const __bf16*>(base) + idx);
    lo = __builtin_convertvector(__builtin_shufflevector(v, v, 0, 1, 2, 3), f32x4);
    hi = __builtin_convertvector(__builtin_shufflevector(v, v, 4, 5, 6, 7), f32x4);
  } else {
    lo = *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + idx);
    hi = *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + idx + 4);
  }
}
template <bool B16>
__device__ __forceinline__ void st8(void* base, size_t idx, f32x4 lo, f32x4 hi) {
  if constexpr (B16) {
    const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
    *reinterpret_cast<bf16x8*>(static_cast<__bf16*>(base) + idx) = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  } else {
    *reinterpret_cast<f32x4*>(static_cast<float*>(base) + idx) = lo;
    *reinterpret_cast<f32x4*>(static_cast<float*>(base) + idx + 4) = hi;
  }
}

// partial[(n*S + s)*C + c] = {sum(x - x0), sum((x - x0)^2)} over the split's rows, x0 = x[n][0][c]
__global__ __launch_bounds__(256) void in_stats_partial(const float* __restrict__ x, float2* __restrict__ part,
                                                        int HW, int C, int S, int rows_per_split) {
  const int tx = threadIdx.x & (NORM_CH - 1), ty = threadIdx.x / NORM_CH;
  const int c = blockIdx.x * NORM_CH + tx;
  const int s = blockIdx.y, n = blockIdx.z;
  __shared__ float sh[2][NORM_ROWS][NORM_CH];
  float a = 0.f, b = 0.f;
  if (c < C) {
    const float* xp = x + (size_t)n * HW * C + c;
    const float x0 = xp[0];
    const int r0 = s * rows_per_split, r1 = min(HW, r0 + rows_per_split);
    for (int r = r0 + ty; r < r1; r += NORM_ROWS) {
      const float v = xp[(size_t)r * C] - x0;
      a += v;
      b += v * v;
    }
  }
  sh[0][ty][tx] = a;
  sh[1][ty][tx] = b;
  __syncthreads();
  if (ty == 0 && c < C) {
#pragma unroll
    for (int i = 1; i < NORM_ROWS; ++i) { a += sh[0][i][tx]; b += sh[1][i][tx]; }
    part[((size_t)n * S + s) * C + c] = make_float2(a, b);
  }
}

// float4 variants (C % 4 == 0): 8 lanes cover the block's 32 channels, 32 row lanes stride over the rows
template <bool X16 = false>
__global__ __launch_bounds__(256) void in_stats_partial_v4(const void* __restrict__ x, float2* __restrict__ part,
                                                           int HW, int C, int S, int rows_per_split) {
  const int q = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c = blockIdx.x * NORM_CH + q * 4;
  const int s = blockIdx.y, n = blockIdx.z;
  __shared__ f32x4 sh[2][32][8];
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const size_t xb = (size_t)n * HW * C + c;
    const f32x4 x0 = ld4<X16>(x, xb);
    const int r0 = s * rows_per_split, r1 = min(HW, r0 + rows_per_split);
    int r = r0 + ty;
    // four rows in flight per thread (round 4): with one dependent load per iteration a CU had ~32 KB outstanding and the pass ran
    // at 4.2 TB/s; the sums keep the row order r, r + 32, ... (same values as the plain loop)
    for (; r + 96 < r1; r += 128) {
      const f32x4 v0 = ld4<X16>(x, xb + (size_t)r * C);
      const f32x4 v1 = ld4<X16>(x, xb + (size_t)(r + 32) * C);
      const f32x4 v2 = ld4<X16>(x, xb + (size_t)(r + 64) * C);
      const f32x4 v3 = ld4<X16>(x, xb + (size_t)(r + 96) * C);
      const f32x4 w0 = v0 - x0, w1 = v1 - x0, w2 = v2 - x0, w3 = v3 - x0;
      a += w0; b += w0 * w0;
      a += w1; b += w1 * w1;
      a += w2; b += w2 * w2;
      a += w3; b += w3 * w3;
    }
    for (; r < r1; r += 32) {
      const f32x4 v = ld4<X16>(x, xb + (size_t)r * C) - x0;
      a += v;
      b += v * v;
    }
  }
  sh[0][ty][q] = a;
  sh[1][ty][q] = b;
  __syncthreads();
  if (threadIdx.x < 32) {              // thread = one channel of the block: sum the 32 row lanes
    const int cc = threadIdx.x, qq = cc >> 2, e = cc & 3;
    float sa = 0.f, sb = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) { sa += sh[0][i][qq][e]; sb += sh[1][i][qq][e]; }
    const int ch = blockIdx.x * NORM_CH + cc;
    if (ch < C) part[((size_t)n * S + s) * C + ch] = make_float2(sa, sb);
  }
}

template <bool X16 = false, bool G16 = false>
__global__ __launch_bounds__(256) void in_bwd_partial_v4(const void* __restrict__ x, const void* __restrict__ dy,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float2* __restrict__ part, int HW, int C, int S, int rows_per_split,
                                                         int act, float slope) {
  const int q = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c = blockIdx.x * NORM_CH + q * 4;
  const int s = blockIdx.y, n = blockIdx.z;
  __shared__ f32x4 sh[2][32][8];
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const int nc = n * C + c;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + nc), rs = *reinterpret_cast<const f32x4*>(rstd + nc);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
    if (scale) {
      sc = *reinterpret_cast<const f32x4*>(scale + nc);
      sf = *reinterpret_cast<const f32x4*>(shift + nc);
    }
    const size_t base = (size_t)n * HW * C + c;
    const int r0 = s * rows_per_split, r1 = min(HW, r0 + rows_per_split);
    auto term = [&](f32x4 xv, f32x4 g) __attribute__((always_inline)) {
      const f32x4 xh = (xv - mu) * rs;
      const f32x4 z = xh * sc + sf;
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] *= act_grad(z[e], act, slope);
      a += g;
      b += g * xh;
    };
    int r = r0 + ty;
    for (; r + 96 < r1; r += 128) {      // four rows of both tensors in flight (see in_stats_partial_v4); row order kept
      const size_t o = base + (size_t)r * C;
      const f32x4 x0v = ld4<X16>(x, o), g0 = ld4<G16>(dy, o);
      const f32x4 x1v = ld4<X16>(x, o + (size_t)32 * C), g1 = ld4<G16>(dy, o + (size_t)32 * C);
      const f32x4 x2v = ld4<X16>(x, o + (size_t)64 * C), g2 = ld4<G16>(dy, o + (size_t)64 * C);
      const f32x4 x3v = ld4<X16>(x, o + (size_t)96 * C), g3 = ld4<G16>(dy, o + (size_t)96 * C);
      term(x0v, g0); term(x1v, g1); term(x2v, g2); term(x3v, g3);
    }
    for (; r < r1; r += 32) {
      const size_t o = base + (size_t)r * C;
      term(ld4<X16>(x, o), ld4<G16>(dy, o));
    }
  }
  sh[0][ty][q] = a;
  sh[1][ty][q] = b;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int cc = threadIdx.x, qq = cc >> 2, e = cc & 3;
    float sa = 0.f, sb = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) { sa += sh[0][i][qq][e]; sb += sh[1][i][qq][e]; }
    const int ch = blockIdx.x * NORM_CH + cc;
    if (ch < C) part[((size_t)n * S + s) * C + ch] = make_float2(sa, sb);
  }
}

// bf16 x (round 4): 8 channels per lane, 64-channel blocks.  With 4 channels per lane a 32-channel block reads 64-byte pieces
// of the bf16 pixel rows -- half of every 128-byte line, the other half going to the neighbouring block -- and the pass took as
// long as the fp32 one (14.3 against 15.6 us for half the bytes).  Same sums in the same row order per channel.
__global__ __launch_bounds__(256) void in_stats_partial_v8(const void* __restrict__ x, float2* __restrict__ part,
                                                           int HW, int C, int S, int rows_per_split) {
  const int q = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c = blockIdx.x * 64 + q * 8;
  const int s = blockIdx.y, n = blockIdx.z;
  __shared__ f32x4 sh[2][32][16];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 a0 = zero, a1 = zero, b0 = zero, b1 = zero;
  if (c < C) {
    const size_t xb = (size_t)n * HW * C + c;
    f32x4 x00, x01;
    ld8<true>(x, xb, x00, x01);
    const int r0 = s * rows_per_split, r1 = min(HW, r0 + rows_per_split);
    int r = r0 + ty;
    for (; r + 96 < r1; r += 128) {
      f32x4 v[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) ld8<true>(x, xb + (size_t)(r + 32 * u) * C, v[u][0], v[u][1]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 w0 = v[u][0] - x00, w1 = v[u][1] - x01;
        a0 += w0; b0 += w0 * w0;
        a1 += w1; b1 += w1 * w1;
      }
    }
    for (; r < r1; r += 32) {
      f32x4 v0, v1;
      ld8<true>(x, xb + (size_t)r * C, v0, v1);
      v0 -= x00; v1 -= x01;
      a0 += v0; b0 += v0 * v0;
      a1 += v1; b1 += v1 * v1;
    }
  }
  sh[0][ty][2 * q] = a0; sh[0][ty][2 * q + 1] = a1;
  sh[1][ty][2 * q] = b0; sh[1][ty][2 * q + 1] = b1;
  __syncthreads();
  if (threadIdx.x < 64) {              // thread = one channel of the block: sum the 32 row lanes
    const int cc = threadIdx.x, qq = cc >> 2, e = cc & 3;
    float sa = 0.f, sb = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) { sa += sh[0][i][qq][e]; sb += sh[1][i][qq][e]; }
    const int ch = blockIdx.x * 64 + cc;
    if (ch < C) part[((size_t)n * S + s) * C + ch] = make_float2(sa, sb);
  }
}

template <bool G16>
__global__ __launch_bounds__(256) void in_bwd_partial_v8(const void* __restrict__ x, const void* __restrict__ dy,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float2* __restrict__ part, int HW, int C, int S, int rows_per_split,
                                                         int act, float slope) {
  const int q = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c = blockIdx.x * 64 + q * 8;
  const int s = blockIdx.y, n = blockIdx.z;
  __shared__ f32x4 sh[2][32][16];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 a[2] = {zero, zero}, b[2] = {zero, zero};
  if (c < C) {
    const int nc = n * C + c;
    f32x4 mu[2], rs[2], sc[2], sf[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      mu[h] = *reinterpret_cast<const f32x4*>(mean + nc + 4 * h);
      rs[h] = *reinterpret_cast<const f32x4*>(rstd + nc + 4 * h);
      sc[h] = f32x4{1.f, 1.f, 1.f, 1.f}; sf[h] = zero;
      if (scale) {
        sc[h] = *reinterpret_cast<const f32x4*>(scale + nc + 4 * h);
        sf[h] = *reinterpret_cast<const f32x4*>(shift + nc + 4 * h);
      }
    }
    const size_t base = (size_t)n * HW * C + c;
    const int r0 = s * rows_per_split, r1 = min(HW, r0 + rows_per_split);
    auto term = [&](const f32x4 (&xv)[2], f32x4 (&g)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 xh = (xv[h] - mu[h]) * rs[h];
        const f32x4 z = xh * sc[h] + sf[h];
#pragma unroll
        for (int e = 0; e < 4; ++e) g[h][e] *= act_grad(z[e], act, slope);
        a[h] += g[h];
        b[h] += g[h] * xh;
      }
    };
    int r = r0 + ty;
    for (; r + 96 < r1; r += 128) {
      f32x4 xv[4][2], g[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t o = base + (size_t)(r + 32 * u) * C;
        ld8<true>(x, o, xv[u][0], xv[u][1]);
        ld8<G16>(dy, o, g[u][0], g[u][1]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) term(xv[u], g[u]);
    }
    for (; r < r1; r += 32) {
      const size_t o = base + (size_t)r * C;
      f32x4 xv[2], g[2];
      ld8<true>(x, o, xv[0], xv[1]);
      ld8<G16>(dy, o, g[0], g[1]);
      term(xv, g);
    }
  }
  sh[0][ty][2 * q] = a[0]; sh[0][ty][2 * q + 1] = a[1];
  sh[1][ty][2 * q] = b[0]; sh[1][ty][2 * q + 1] = b[1];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = threadIdx.x, qq = cc >> 2, e = cc & 3;
    float sa = 0.f, sb = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) { sa += sh[0][i][qq][e]; sb += sh[1][i][qq][e]; }
    const int ch = blockIdx.x * 64 + cc;
    if (ch < C) part[((size_t)n * S + s) * C + ch] = make_float2(sa, sb);
  }
}

// {sum (x - x0), sum (x - x0)^2} over the image -> mean, 1 / sqrt(var + eps); one definition for the stand-alone final kernel
// and for the apply kernels that finish the statistics themselves (same expression, same contraction, same bits)
__device__ __forceinline__ void stats_finish(float a, float b, float x0, int HW, float eps, float* mu, float* rs) {
  const float inv = 1.f / (float)HW;
  const float dm = a * inv;
  float var = b * inv - dm * dm;
  var = var < 0.f ? 0.f : var;
  *mu = x0 + dm;
  *rs = 1.0f / sqrtf(var + eps);
}

__global__ void in_stats_final(const float* __restrict__ x, const float2* __restrict__ part, float* __restrict__ mean,
                               float* __restrict__ rstd, int N, int HW, int C, int S, float eps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * C) return;
  const int n = idx / C, c = idx - n * C;
  float a = 0.f, b = 0.f;
  for (int s = 0; s < S; ++s) {
    const float2 p = part[((size_t)n * S + s) * C + c];
    a += p.x;
    b += p.y;
  }
  stats_finish(a, b, x[(size_t)n * HW * C + c], HW, eps, &mean[idx], &rstd[idx]);
}

// the S partial pairs of this thread's four channels, summed in split order (what in_stats_final / in_bwd_final do per channel)
__device__ __forceinline__ void sum_partials4(const float2* __restrict__ part, int n, int S, int C, int c, f32x4* a, f32x4* b) {
  f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < S; ++s) {
    const f32x4* q = reinterpret_cast<const f32x4*>(part + ((size_t)n * S + s) * C + c);      // (x0 y0 x1 y1), (x2 y2 x3 y3)
    const f32x4 lo = q[0], hi = q[1];
    sa[0] += lo[0]; sb[0] += lo[1]; sa[1] += lo[2]; sb[1] += lo[3];
    sa[2] += hi[0]; sb[2] += hi[1]; sa[3] += hi[2]; sb[3] += hi[3];
  }
  *a = sa;
  *b = sb;
}

template <bool V4>
__global__ void in_apply(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                         const float* __restrict__ res, const float* __restrict__ mean, const float* __restrict__ rstd,
                         float* __restrict__ y, long long total, int HWC, int C, int act, float slope) {
  constexpr int W = V4 ? 4 : 1;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * W; i < total;
       i += (long long)gridDim.x * blockDim.x * W) {
    const int n = (int)(i / HWC);
    const int c = (int)(i % C);
    float xv[W], rv[W], o[W];
    if constexpr (V4) {
      *reinterpret_cast<f32x4*>(xv) = *reinterpret_cast<const f32x4*>(x + i);
      if (res) *reinterpret_cast<f32x4*>(rv) = *reinterpret_cast<const f32x4*>(res + i);
    } else {
      xv[0] = x[i];
      if (res) rv[0] = res[i];
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const int nc = n * C + c + k;
      float v = (xv[k] - mean[nc]) * rstd[nc];
      if (scale) v = v * scale[nc] + shift[nc];
      v = apply_act(v, act, slope);
      if (res) v += rv[k];
      o[k] = v;
    }
    if constexpr (V4) *reinterpret_cast<f32x4*>(y + i) = *reinterpret_cast<const f32x4*>(o);
    else y[i] = o[0];
  }
}

// Fast apply passes for power-of-two channel counts (C | 1024): with a float4 stride of 256*G per image a thread
// always lands on the same 4 channels, so mean / rstd / scale / shift live in registers and the streaming loop has
// no index arithmetic (grid = (G, N)).
// `part` != null (round 3): the statistics arrive as the S partial pairs of in_stats_partial and every thread finishes its own
// four channels (a few KB from L2) -- the 5.6 us in_stats_final launch between the two passes is gone; workgroup x = 0 of an
// image stores mean / rstd for the backward pass.
template <bool X16 = false, bool Y16 = false>
__global__ __launch_bounds__(256) void in_apply_pow2(const void* __restrict__ x, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const float* __restrict__ res,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     void* __restrict__ y, int HWC4, int C, int act, float slope,
                                                     const float2* __restrict__ part, int S, int HW, float eps) {
  const int n = blockIdx.y;
  const int c = (threadIdx.x * 4) % C;
  const int nc = n * C + c;
  f32x4 mu, rs;
  if (part) {
    f32x4 a, b;
    sum_partials4(part, n, S, C, c, &a, &b);
    const f32x4 x0 = ld4<X16>(x, (size_t)n * HWC4 * 4 + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float m1, r1;
      stats_finish(a[e], b[e], x0[e], HW, eps, &m1, &r1);
      mu[e] = m1;
      rs[e] = r1;
    }
    if (blockIdx.x == 0 && threadIdx.x * 4 < C) {
      *reinterpret_cast<f32x4*>(mean + nc) = mu;
      *reinterpret_cast<f32x4*>(rstd + nc) = rs;
    }
  } else {
    mu = *reinterpret_cast<const f32x4*>(mean + nc);
    rs = *reinterpret_cast<const f32x4*>(rstd + nc);
  }
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
  if (scale) {
    sc = *reinterpret_cast<const f32x4*>(scale + nc);
    sf = *reinterpret_cast<const f32x4*>(shift + nc);
  }
  const size_t base = (size_t)n * HWC4;
  const f32x4* rp = res ? reinterpret_cast<const f32x4*>(res) + base : nullptr;
  auto xat = [&](int j) __attribute__((always_inline)) { return ld4<X16>(x, (base + j) * 4); };
  auto yput = [&](int j, f32x4 v) __attribute__((always_inline)) { st4<Y16>(y, (base + j) * 4, v); };
  auto one = [&](f32x4 xv, f32x4 rv) __attribute__((always_inline)) {
    f32x4 v = ((xv - mu) * rs) * sc + sf;         // same expression as the backward's mask recomputation
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k], act, slope);
    return rp ? v + rv : v;
  };
  const int step = gridDim.x * 256;
  int j = blockIdx.x * 256 + threadIdx.x;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (; j + 3 * step < HWC4; j += 4 * step) {     // four float4 of each tensor in flight per thread (round 4)
    const f32x4 x0v = xat(j), x1v = xat(j + step), x2v = xat(j + 2 * step), x3v = xat(j + 3 * step);
    f32x4 r0 = zero, r1 = zero, r2 = zero, r3 = zero;
    if (rp) { r0 = rp[j]; r1 = rp[j + step]; r2 = rp[j + 2 * step]; r3 = rp[j + 3 * step]; }
    yput(j, one(x0v, r0));
    yput(j + step, one(x1v, r1));
    yput(j + 2 * step, one(x2v, r2));
    yput(j + 3 * step, one(x3v, r3));
  }
  for (; j < HWC4; j += step) yput(j, one(xat(j), rp ? rp[j] : zero));
}

template <bool X16 = false, bool G16 = false, bool D16 = false>
__global__ __launch_bounds__(256) void in_bwd_apply_pow2(const void* __restrict__ x, const void* __restrict__ dy,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ dshift, float* __restrict__ dscale,
                                                         void* __restrict__ dx, int HWC4, int C, float inv_hw, int act,
                                                         float slope, const float2* __restrict__ part, int S) {
  const int n = blockIdx.y;
  const int c = (threadIdx.x * 4) % C;
  const int nc = n * C + c;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + nc);
  const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + nc);
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
  if (scale) {
    sc = *reinterpret_cast<const f32x4*>(scale + nc);
    sf = *reinterpret_cast<const f32x4*>(shift + nc);
  }
  f32x4 ds, dsc;
  if (part) {              // the sums of in_bwd_partial finished here instead of by an in_bwd_final launch (see in_apply_pow2)
    sum_partials4(part, n, S, C, c, &ds, &dsc);
    if (blockIdx.x == 0 && threadIdx.x * 4 < C) {
      *reinterpret_cast<f32x4*>(dshift + nc) = ds;
      *reinterpret_cast<f32x4*>(dscale + nc) = dsc;
    }
  } else {
    ds = *reinterpret_cast<const f32x4*>(dshift + nc);
    dsc = *reinterpret_cast<const f32x4*>(dscale + nc);
  }
  const f32x4 mg = ds * inv_hw;
  const f32x4 mgx = dsc * inv_hw;
  const f32x4 k = rs * sc;
  const size_t base = (size_t)n * HWC4;
  auto xat = [&](int j) __attribute__((always_inline)) { return ld4<X16>(x, (base + j) * 4); };
  auto gat = [&](int j) __attribute__((always_inline)) { return ld4<G16>(dy, (base + j) * 4); };
  auto oput = [&](int j, f32x4 v) __attribute__((always_inline)) { st4<D16>(dx, (base + j) * 4, v); };
  auto one = [&](f32x4 xv, f32x4 g) __attribute__((always_inline)) {
    const f32x4 xh = (xv - mu) * rs;
    const f32x4 z = xh * sc + sf;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] *= act_grad(z[e], act, slope);
    return k * (g - mg - xh * mgx);
  };
  const int step = gridDim.x * 256;
  int j = blockIdx.x * 256 + threadIdx.x;
  for (; j + 3 * step < HWC4; j += 4 * step) {     // four float4 of both tensors in flight per thread (round 4)
    const f32x4 x0v = xat(j), x1v = xat(j + step), x2v = xat(j + 2 * step), x3v = xat(j + 3 * step);
    const f32x4 g0 = gat(j), g1 = gat(j + step), g2 = gat(j + 2 * step), g3 = gat(j + 3 * step);
    oput(j, one(x0v, g0));
    oput(j + step, one(x1v, g1));
    oput(j + 2 * step, one(x2v, g2));
    oput(j + 3 * step, one(x3v, g3));
  }
  for (; j < HWC4; j += step) oput(j, one(xat(j), gat(j)));
}

// backward partial: {sum g, sum g*xh}, g = dy * act'(xh*scale+shift)
__global__ __launch_bounds__(256) void in_bwd_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      float2* __restrict__ part, int HW, int C, int S, int rows_per_split,
                                                      int act, float slope) {
  const int tx = threadIdx.x & (NORM_CH - 1), ty = threadIdx.x / NORM_CH;
  const int c = blockIdx.x * NORM_CH + tx;
  const int s = blockIdx.y, n = blockIdx.z;
  __shared__ float sh[2][NORM_ROWS][NORM_CH];
  float a = 0.f, b = 0.f;
  if (c < C) {
    const int nc = n * C + c;
    const float mu = mean[nc], rs = rstd[nc];
    const float sc = scale ? scale[nc] : 1.f, sf = scale ? shift[nc] : 0.f;
    const size_t base = (size_t)n * HW * C + c;
    const int r0 = s * rows_per_split, r1 = min(HW, r0 + rows_per_split);
    for (int r = r0 + ty; r < r1; r += NORM_ROWS) {
      const size_t o = base + (size_t)r * C;
      const float xh = (x[o] - mu) * rs;
      const float g = dy[o] * act_grad(xh * sc + sf, act, slope);
      a += g;
      b += g * xh;
    }
  }
  sh[0][ty][tx] = a;
  sh[1][ty][tx] = b;
  __syncthreads();
  if (ty == 0 && c < C) {
#pragma unroll
    for (int i = 1; i < NORM_ROWS; ++i) { a += sh[0][i][tx]; b += sh[1][i][tx]; }
    part[((size_t)n * S + s) * C + c] = make_float2(a, b);
  }
}

__global__ void in_bwd_final(const float2* __restrict__ part, float* __restrict__ dshift, float* __restrict__ dscale,
                             int NC, int C, int S) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NC) return;
  const int n = idx / C, c = idx - n * C;
  float a = 0.f, b = 0.f;
  for (int s = 0; s < S; ++s) {
    const float2 p = part[((size_t)n * S + s) * C + c];
    a += p.x;
    b += p.y;
  }
  dshift[idx] = a;
  dscale[idx] = b;
}

template <bool V4>
__global__ void in_bwd_apply(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ scale,
                             const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
                             const float* __restrict__ dshift, const float* __restrict__ dscale, float* __restrict__ dx,
                             long long total, int HWC, int C, float inv_hw, int act, float slope) {
  constexpr int W = V4 ? 4 : 1;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * W; i < total;
       i += (long long)gridDim.x * blockDim.x * W) {
    const int n = (int)(i / HWC);
    const int c = (int)(i % C);
    float xv[W], gv[W], o[W];
    if constexpr (V4) {
      *reinterpret_cast<f32x4*>(xv) = *reinterpret_cast<const f32x4*>(x + i);
      *reinterpret_cast<f32x4*>(gv) = *reinterpret_cast<const f32x4*>(dy + i);
    } else {
      xv[0] = x[i];
      gv[0] = dy[i];
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const int nc = n * C + c + k;
      const float rs = rstd[nc];
      const float sc = scale ? scale[nc] : 1.f, sf = scale ? shift[nc] : 0.f;
      const float xh = (xv[k] - mean[nc]) * rs;
      const float g = gv[k] * act_grad(xh * sc + sf, act, slope);
      o[k] = rs * sc * (g - dshift[nc] * inv_hw - xh * dscale[nc] * inv_hw);
    }
    if constexpr (V4) *reinterpret_cast<f32x4*>(dx + i) = *reinterpret_cast<const f32x4*>(o);
    else dx[i] = o[0];
  }
}

// ---- CBIN affine --------------------------------------------------------------------------
__global__ void cbin_affine_fwd_kernel(const float* c, const float* W, const float* b, const float* gamma,
                                       const float* beta, float* t, float* scale, float* shift, int N, int C, int nc) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * C) return;
  const int n = idx / C, ch = idx - n * C;
  float a = b[ch];
  for (int j = 0; j < nc; ++j) a += c[n * nc + j] * W[ch * nc + j];
  const float tv = tanhf(a);
  t[idx] = tv;
  scale[idx] = gamma[ch];
  shift[idx] = tv * gamma[ch] + beta[ch];
}

// one WAVE per channel: lanes stride over the batch, 64-lane shuffle reductions for dgamma, dbeta, db and
// the num_con entries of the dW row; also writes da[N][C] for the dc pass.
__global__ __launch_bounds__(256) void cbin_affine_bwd_ch(const float* c, const float* gamma, const float* t,
                                                          const float* dscale, const float* dshift, float* dgamma,
                                                          float* dbeta, float* dW, float* db, float* da, int N, int C,
                                                          int nc) {
  const int ch = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (ch >= C) return;
  float dg = 0.f, dbt = 0.f, dbb = 0.f;
  float dw[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) dw[j] = 0.f;
  const float g = gamma[ch];
  for (int n = lane; n < N; n += 64) {
    const int i = n * C + ch;
    const float tv = t[i], ds = dshift[i];
    dg += dscale[i] + ds * tv;
    dbt += ds;
    const float a = g * ds * (1.f - tv * tv);
    da[i] = a;
    dbb += a;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nc) dw[j] += a * c[n * nc + j];
  }
  dg = wave_sum(dg);
  dbt = wave_sum(dbt);
  dbb = wave_sum(dbb);
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < nc) dw[j] = wave_sum(dw[j]);
  if (lane == 0) {
    dgamma[ch] = dg;
    dbeta[ch] = dbt;
    db[ch] = dbb;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < nc && lane == j) dW[ch * nc + j] = dw[j];
}

// one WAVE per sample: dc[n][j] = sum_ch da[n][ch] * W[ch][j]
__global__ __launch_bounds__(256) void cbin_affine_bwd_c(const float* W, const float* da, float* dc, int N, int C, int nc) {
  const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  for (int ch = lane; ch < C; ch += 64) {
    const float a = da[n * C + ch];
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nc) acc[j] += a * W[ch * nc + j];
  }
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < nc) {
      const float v = wave_sum(acc[j]);
      if (lane == j) dc[n * nc + j] = v;
    }
}


// ---- the same three kernels over ALL central-biasing layers of a network in one launch each (the affine depends on the
// style code and the layer's parameters only, never on activations: the generator's 15 layers cost 45 launches of ~5 us per
// pass one by one).  Device table of per-layer records; outputs of layer l are dense [N][C_l] blocks. ----
struct CbinRec {
  const float *W, *b, *gamma, *beta;     // forward: parameters.  backward: W, gamma = the forward-time copy (row 0 of scale)
  float *t, *scale, *shift;              // forward outputs / backward inputs (t)
  const float *dscale, *dshift;          // backward inputs ([N][C], never null: the host substitutes zeros)
  float *dgamma, *dbeta, *dW, *db, *da;  // backward outputs; da = [N][C] scratch for the dc pass
  int C, accumulate;                     // accumulate: the backward ADDS to dgamma / dbeta / dW / db (srgan_cbin_rec_set_accumulate)
};

__global__ void cbin_affine_multi_fwd_kernel(const float* c, const CbinRec* tab, int N, int nc) {
  const CbinRec r = tab[blockIdx.y];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * r.C) return;
  const int n = idx / r.C, ch = idx - n * r.C;
  float a = r.b[ch];
  for (int j = 0; j < nc; ++j) a += c[n * nc + j] * r.W[ch * nc + j];
  const float tv = tanhf(a);
  r.t[idx] = tv;
  r.scale[idx] = r.gamma[ch];
  r.shift[idx] = tv * r.gamma[ch] + r.beta[ch];
}

__global__ __launch_bounds__(256) void cbin_affine_multi_bwd_ch(const float* c, const CbinRec* tab, int N, int nc) {
  const CbinRec r = tab[blockIdx.y];
  const int ch = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (ch >= r.C) return;
  const int C = r.C;
  float dg = 0.f, dbt = 0.f, dbb = 0.f;
  float dw[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) dw[j] = 0.f;
  const float g = r.gamma[ch];
  for (int n = lane; n < N; n += 64) {
    const int i = n * C + ch;
    const float tv = r.t[i], ds = r.dshift[i];
    dg += r.dscale[i] + ds * tv;
    dbt += ds;
    const float a = g * ds * (1.f - tv * tv);
    r.da[i] = a;
    dbb += a;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nc) dw[j] += a * c[n * nc + j];
  }
  dg = wave_sum(dg);
  dbt = wave_sum(dbt);
  dbb = wave_sum(dbb);
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < nc) dw[j] = wave_sum(dw[j]);
  if (lane == 0) {
    r.dgamma[ch] = r.accumulate ? r.dgamma[ch] + dg : dg;
    r.dbeta[ch] = r.accumulate ? r.dbeta[ch] + dbt : dbt;
    r.db[ch] = r.accumulate ? r.db[ch] + dbb : dbb;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < nc && lane == j) r.dW[ch * nc + j] = r.accumulate ? r.dW[ch * nc + j] + dw[j] : dw[j];
}

// one WORKGROUP per sample, one wave per layer (<= 16 layers per pass): dc[n][j] = sum over layers and channels of
// da[n][ch] * W[ch][j].  The per-layer wave sums meet in LDS and are added in table order -- the order autograd uses -- so the
// result does not depend on the schedule (a wave per sample walking all 15 layers serially took 113 us for 45 KFLOP).
__global__ __launch_bounds__(1024) void cbin_affine_multi_bwd_c(const CbinRec* tab, int n_layers, float* dc, int N, int nc) {
  __shared__ float part[16][16];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float tot = 0.f;                                   // lane j < nc of wave 0 owns dc[n][j]
  for (int l0 = 0; l0 < n_layers; l0 += 16) {
    const int l = l0 + wave;
    if (l < n_layers) {
      const CbinRec r = tab[l];
      float acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.f;
      for (int ch = lane; ch < r.C; ch += 64) {
        const float a = r.da[n * r.C + ch];
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (j < nc) acc[j] += a * r.W[ch * nc + j];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < nc) { const float v = wave_sum(acc[j]); if (lane == 0) part[wave][j] = v; }
    }
    __syncthreads();
    if (wave == 0 && lane < nc)
      for (int w = 0; w < 16 && l0 + w < n_layers; ++w) tot += part[w][lane];
    __syncthreads();
  }
  if (wave == 0 && lane < nc) dc[n * nc + lane] = tot;
}

// ---- single-pass variants: the (image, 32-channel) slab lives in registers --------------------------------------------
// For maps of <= 1024 pixels (the generator's 32x32x256 trunk: 17 of its ~22 norm layers, and the deep encoder /
// discriminator maps) one workgroup holds its whole slab -- HW x 32 channels, <= 128 KB -- in VGPRs: the statistics, the
// normalisation and the activation (forward), or the two gradient sums and dx (backward) come out of ONE read of the
// tensor(s) instead of two, and of one launch instead of three.  Lanes 0-7 of a row group cover the 32 channels of one
// pixel (128 contiguous bytes); the row groups of a wave are summed with shuffles, the waves through LDS.
template <int NW, int QL = 8>
__device__ __forceinline__ f32x4 slab_sum(f32x4 v, f32x4 (*sh)[8], int q, int wave) {
#pragma unroll
  for (int o = QL; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], o, 64);
  __syncthreads();                       // previous use of sh is over
  if ((threadIdx.x & 63) < QL) sh[wave][q] = v;
  __syncthreads();
  f32x4 t = sh[0][q];
#pragma unroll
  for (int w = 1; w < NW; ++w) t += sh[w][q];
  return t;
}

// Slab kernels: (slab, image) of a workgroup.  The slabs of ONE image are 128-byte (or 64-byte) pieces of the same 1 KB pixel
// lines; the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, so with the plain (x = slab, y = image) grid
// the pieces of a line are fetched by 8 different L2s at 8 different times.  Remapped, all slabs of image n run on XCD n % 8,
// back to back: the line's pieces arrive at one L2 within a short window (SRGAN_NORM_PLAIN_GRID=1: the old mapping).
__device__ __forceinline__ void slab_coords(int nslab, int N, int remap, int& slab, int& n) {
  const int L = blockIdx.y * gridDim.x + blockIdx.x;
  if (remap && (N & 7) == 0) {
    const int xcd = L & 7, k = L >> 3;
    slab = k % nslab;
    n = (k / nslab) * 8 + xcd;
  } else {
    slab = L % nslab;
    n = L / nslab;
  }
}

// forward: 512 threads over HW x 32 channels (8 lanes per pixel: whole 128-byte lines; 16-channel slabs with 256 threads were
// measured slower here, 45.6 vs 34.7 us on the 32x32x256 trunk, while they help the backward below)
template <int R, bool X16 = false, bool Y16 = false>
__global__ __launch_bounds__(512) void in_fwd_slab(const void* __restrict__ x, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, const float* __restrict__ res,
                                                   void* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                   int HW, int C, float eps, int act, float slope, int remap) {
  __shared__ f32x4 sh[8][8];
  const int q = threadIdx.x & 7, ty = threadIdx.x >> 3, wave = threadIdx.x >> 6;
  int slab, n;
  slab_coords(gridDim.x, gridDim.y, remap, slab, n);
  const int c = slab * 32 + q * 4;
  const size_t base = (size_t)n * HW * C + c;
  f32x4 v[R];
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int r = ty + 64 * j;
    v[j] = r < HW ? ld4<X16>(x, base + (size_t)r * C) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += v[j];
  }
  const float inv = 1.f / (float)HW;
  const f32x4 mu = slab_sum<8>(s, sh, q, wave) * inv;
  f32x4 m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < R; ++j)
    if (ty + 64 * j < HW) {
      const f32x4 d = v[j] - mu;
      m2 += d * d;
    }
  const f32x4 var = slab_sum<8>(m2, sh, q, wave) * inv;     // exact two-pass variance (biased), as instance_norm
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
    const int r = ty + 64 * j;
    if (r < HW) {
      f32x4 o = ((v[j] - mu) * rs) * sc + sf;               // same expression as in_apply / the backward's mask recomputation
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = apply_act(o[e], act, slope);
      if (res) o += *reinterpret_cast<const f32x4*>(res + base + (size_t)r * C);
      st4<Y16>(y, base + (size_t)r * C, o);
    }
  }
}

// backward: x-hat and the masked dy both stay in registers (2 x R float4 per thread); 256-thread workgroups over HW x 16
// channels (4 lanes per pixel), two resident per CU so one's reduction / store phase overlaps the other's loads (38 us
// against 43 us with one 1024-thread workgroup per CU).
template <int R, bool X16 = false, bool G16 = false, bool D16 = false>
__global__ __launch_bounds__(256) void in_bwd_slab(const void* __restrict__ x, const void* __restrict__ dy,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    void* __restrict__ dx, float* __restrict__ dscale,
                                                    float* __restrict__ dshift, int HW, int C, int act, float slope, int remap) {
  __shared__ f32x4 sh[4][8];
  const int q = threadIdx.x & 3, ty = threadIdx.x >> 2, wave = threadIdx.x >> 6;
  int slab, n;
  slab_coords(gridDim.x, gridDim.y, remap, slab, n);
  const int c = slab * 16 + q * 4;
  const int nc = n * C + c;
  const size_t base = (size_t)n * HW * C + c;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + nc), rs = *reinterpret_cast<const f32x4*>(rstd + nc);
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
  if (scale) {
    sc = *reinterpret_cast<const f32x4*>(scale + nc);
    sf = *reinterpret_cast<const f32x4*>(shift + nc);
  }
  f32x4 xh[R], g[R];
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int r = ty + 64 * j;
    if (r < HW) {
      xh[j] = (ld4<X16>(x, base + (size_t)r * C) - mu) * rs;
      g[j] = ld4<G16>(dy, base + (size_t)r * C);
      const f32x4 z = xh[j] * sc + sf;
#pragma unroll
      for (int e = 0; e < 4; ++e) g[j][e] *= act_grad(z[e], act, slope);
      a += g[j];
      b += g[j] * xh[j];
    } else {
      xh[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      g[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  a = slab_sum<4, 4>(a, sh, q, wave);
  b = slab_sum<4, 4>(b, sh, q, wave);
  if (ty == 0) {
    *reinterpret_cast<f32x4*>(dshift + nc) = a;
    *reinterpret_cast<f32x4*>(dscale + nc) = b;
  }
  const float inv_hw = 1.f / (float)HW;
  const f32x4 mg = a * inv_hw, mgx = b * inv_hw, k = rs * sc;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int r = ty + 64 * j;
    if (r < HW) st4<D16>(dx, base + (size_t)r * C, k * (g[j] - mg - xh[j] * mgx));
  }
}

// ---- the same slabs with 16-bit tensors on a side: 8 channels per lane --------------------------------------------------
// With 4 channels per lane a bf16 tensor is read in 8-byte pieces: the same number of load instructions as for fp32, each
// moving half the bytes (measured on the 32 x 32 x 256 trunk: bf16 -> bf16 15.0 us against fp32 -> fp32 16.1 us, 2.2 against
// 4.2 TB/s).  Here a lane owns 8 channels of a pixel (one 16-byte load of a bf16 tensor, two of an fp32 one), 4 lanes cover
// the slab's 32 channels, 512 threads stride 128 pixels per pass: half the instructions per byte (13.4 us).  Same arithmetic as
// in_fwd_slab (the per-lane partial sums cover different pixels, so the statistics differ by fp32 rounding).
// both halves of a lane's 8 channels summed over the pixel lanes of a wave (shuffles) and the NW waves (LDS)
template <int NW>
__device__ __forceinline__ void slab_sum8(f32x4& lo, f32x4& hi, f32x4 (*sh)[8], int q, int wave) {
#pragma unroll
  for (int o = 4; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo[e] += __shfl_xor(lo[e], o, 64);
      hi[e] += __shfl_xor(hi[e], o, 64);
    }
  __syncthreads();                       // previous use of sh is over
  if ((threadIdx.x & 63) < 4) { sh[wave][2 * q] = lo; sh[wave][2 * q + 1] = hi; }
  __syncthreads();
  lo = sh[0][2 * q]; hi = sh[0][2 * q + 1];
#pragma unroll
  for (int w = 1; w < NW; ++w) { lo += sh[w][2 * q]; hi += sh[w][2 * q + 1]; }
}

template <int R, bool X16, bool Y16>
__global__ __launch_bounds__(512) void in_fwd_slab8(const void* __restrict__ x, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const float* __restrict__ res,
                                                    void* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                    int HW, int C, float eps, int act, float slope, int remap) {
  __shared__ f32x4 sh[8][8];
  const int q = threadIdx.x & 3, ty = threadIdx.x >> 2, wave = threadIdx.x >> 6;
  int slab, n;
  slab_coords(gridDim.x, gridDim.y, remap, slab, n);
  const int c = slab * 32 + q * 8;
  const size_t base = (size_t)n * HW * C + c;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 v0[R], v1[R];
  f32x4 s0 = zero, s1 = zero;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int r = ty + 128 * j;
    if (r < HW) ld8<X16>(x, base + (size_t)r * C, v0[j], v1[j]);
    else { v0[j] = zero; v1[j] = zero; }
    s0 += v0[j]; s1 += v1[j];
  }
  const float inv = 1.f / (float)HW;
  slab_sum8<8>(s0, s1, sh, q, wave);
  const f32x4 mu0 = s0 * inv, mu1 = s1 * inv;
  f32x4 m0 = zero, m1 = zero;
#pragma unroll
  for (int j = 0; j < R; ++j)
    if (ty + 128 * j < HW) {
      const f32x4 d0 = v0[j] - mu0, d1 = v1[j] - mu1;
      m0 += d0 * d0; m1 += d1 * d1;
    }
  slab_sum8<8>(m0, m1, sh, q, wave);                       // exact two-pass variance (biased), as instance_norm
  f32x4 rs0, rs1;
#pragma unroll
  for (int e = 0; e < 4; ++e) { rs0[e] = 1.0f / sqrtf(m0[e] * inv + eps); rs1[e] = 1.0f / sqrtf(m1[e] * inv + eps); }
  const int nc = n * C + c;
  if (ty == 0) {
    *reinterpret_cast<f32x4*>(mean + nc) = mu0; *reinterpret_cast<f32x4*>(mean + nc + 4) = mu1;
    *reinterpret_cast<f32x4*>(rstd + nc) = rs0; *reinterpret_cast<f32x4*>(rstd + nc + 4) = rs1;
  }
  f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sf0 = zero, sf1 = zero;
  if (scale) {
    sc0 = *reinterpret_cast<const f32x4*>(scale + nc); sc1 = *reinterpret_cast<const f32x4*>(scale + nc + 4);
    sf0 = *reinterpret_cast<const f32x4*>(shift + nc); sf1 = *reinterpret_cast<const f32x4*>(shift + nc + 4);
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int r = ty + 128 * j;
    if (r < HW) {
      f32x4 o0 = ((v0[j] - mu0) * rs0) * sc0 + sf0, o1 = ((v1[j] - mu1) * rs1) * sc1 + sf1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o0[e] = apply_act(o0[e], act, slope); o1[e] = apply_act(o1[e], act, slope); }
      if (res) {
        o0 += *reinterpret_cast<const f32x4*>(res + base + (size_t)r * C);
        o1 += *reinterpret_cast<const f32x4*>(res + base + (size_t)r * C + 4);
      }
      st8<Y16>(y, base + (size_t)r * C, o0, o1);
    }
  }
}

namespace {
// C divides 1024 (so 256 threads * 4 floats wrap onto the same channels) and every float4 stays inside one pixel
bool pow2_fast(int C, int HW) { return C >= 4 && (1024 % C) == 0 && (long long)HW * C / 4 < (1LL << 30); }
int apply_grid(int hwc4, int N) {
  long long per_image = ceil_div(hwc4, 256);                 // blocks if one float4 per thread
  long long want = std::max<long long>(1, 2048 / std::max(1, N));
  return (int)std::max<long long>(1, std::min(per_image, want));
}

int slab_remap() { return 1; }

// single-pass kernels: whole 32-channel groups, a slab of <= 1024 pixels, enough workgroups to cover the device
bool slab_fast(int N, int HW, int C) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_NORM_SLAB");
  return !off && (C % 32) == 0 && HW <= 1024 && (long long)N * (C / 32) >= 128;
}

void plan_split(int N, int HW, int C, int& S, int& rps) {
  const int chunks = (C + NORM_CH - 1) / NORM_CH;
  long long blocks = (long long)N * chunks;
  S = 1;
  while (blocks * S < 1024 && HW / (S * 2) >= 64) S *= 2;
  rps = (HW + S - 1) / S;
}
}  // namespace
}  // namespace srgan

using namespace srgan;

extern "C" size_t srgan_instnorm_workspace(int N, int HW, int C) {
  int S, rps;
  plan_split(N, HW, C, S, rps);
  return (size_t)N * S * C * sizeof(float2);
}

extern "C" int srgan_instnorm_fwd(const float* x, const float* scale, const float* shift, const float* res, float* y,
                                  float* mean, float* rstd, int N, int HW, int C, float eps, int act, float slope,
                                  void* ws, size_t ws_bytes, void* stream) {
  SRGAN_REQUIRE(x && y && mean && rstd, "instnorm_fwd: null pointer");
  SRGAN_REQUIRE(N > 0 && HW > 0 && C > 0, "instnorm_fwd: bad shape");
  SRGAN_REQUIRE((scale == nullptr) == (shift == nullptr), "instnorm_fwd: scale and shift go together");
  hipStream_t st = as_stream(stream);
  int S, rps;
  plan_split(N, HW, C, S, rps);
  SRGAN_REQUIRE(ws && ws_bytes >= (size_t)N * S * C * sizeof(float2), "instnorm_fwd: workspace too small");
  const double tbytes = (double)N * HW * C * sizeof(float);      // bench.py's roofline_hbm: bytes the passes must move
  if (slab_fast(N, HW, C)) {
    const dim3 gs((unsigned)(C / 32), (unsigned)N);
    const int rows = (HW + 63) / 64;
#define SRGAN_FWD_SLAB(R) hipLaunchKernelGGL(in_fwd_slab<R>, gs, dim3(512), 0, st, x, scale, shift, res, y, mean, rstd, HW, C, eps, act, slope, slab_remap())
    ProfToken tok = prof_begin(34, (res ? 3.0 : 2.0) * tbytes, st);
    if (rows <= 1) SRGAN_FWD_SLAB(1);
    else if (rows <= 2) SRGAN_FWD_SLAB(2);
    else if (rows <= 4) SRGAN_FWD_SLAB(4);
    else if (rows <= 8) SRGAN_FWD_SLAB(8);
    else SRGAN_FWD_SLAB(16);
#undef SRGAN_FWD_SLAB
    prof_end(tok, st);
    return check_launch("instnorm_fwd (slab)");
  }
  float2* part = reinterpret_cast<float2*>(ws);
  dim3 g((C + NORM_CH - 1) / NORM_CH, S, N);
  {
    ProfToken tok = prof_begin(30, tbytes, st);
    if ((C & 3) == 0) hipLaunchKernelGGL(in_stats_partial_v4<false>, g, dim3(256), 0, st, (const void*)x, part, HW, C, S, rps);
    else hipLaunchKernelGGL(in_stats_partial, g, dim3(256), 0, st, x, part, HW, C, S, rps);
    prof_end(tok, st);
  }
  const long long total = (long long)N * HW * C;
  // the fused finish re-reads the shift sample x[n][pixel 0][c] in EVERY workgroup of the apply pass while workgroup 0 may
  // already be storing y there: an in-place call (y == x, or the skip tensor == y) takes the separate finalize launch instead
  const bool fuse_final = (const float*)y != x && (const float*)y != res;
  if (pow2_fast(C, HW) && fuse_final) {
    const int hwc4 = HW * C / 4;
    dim3 g2((unsigned)apply_grid(hwc4, N), (unsigned)N);
    ProfToken tok = prof_begin(31, (res ? 3.0 : 2.0) * tbytes, st);
    hipLaunchKernelGGL((in_apply_pow2<false, false>), g2, dim3(256), 0, st, (const void*)x, scale, shift, res, mean, rstd, (void*)y, hwc4, C, act, slope,
                       (const float2*)part, S, HW, eps);
    prof_end(tok, st);
    return check_launch("instnorm_fwd");
  }
  hipLaunchKernelGGL(in_stats_final, dim3((N * C + 255) / 256), dim3(256), 0, st, x, (const float2*)part, mean, rstd, N, HW, C, S, eps);
  if (pow2_fast(C, HW)) {
    const int hwc4 = HW * C / 4;
    dim3 g2((unsigned)apply_grid(hwc4, N), (unsigned)N);
    hipLaunchKernelGGL((in_apply_pow2<false, false>), g2, dim3(256), 0, st, (const void*)x, scale, shift, res, mean, rstd, (void*)y, hwc4, C, act, slope,
                       (const float2*)nullptr, 0, HW, eps);
  } else if ((C & 3) == 0) {
    unsigned blocks = (unsigned)std::min<long long>(ceil_div(total / 4, 256), 8192);
    hipLaunchKernelGGL(in_apply<true>, dim3(blocks), dim3(256), 0, st, x, scale, shift, res, mean, rstd, y, total, HW * C, C, act, slope);
  } else {
    unsigned blocks = (unsigned)std::min<long long>(ceil_div(total, 256), 8192);
    hipLaunchKernelGGL(in_apply<false>, dim3(blocks), dim3(256), 0, st, x, scale, shift, res, mean, rstd, y, total, HW * C, C, act, slope);
  }
  return check_launch("instnorm_fwd");
}

extern "C" int srgan_instnorm_bwd(const float* x, const float* dy, const float* scale, const float* shift,
                                  const float* mean, const float* rstd, float* dx, float* dscale, float* dshift,
                                  int N, int HW, int C, int act, float slope, void* ws, size_t ws_bytes, void* stream) {
  SRGAN_REQUIRE(x && dy && mean && rstd && dx && dscale && dshift, "instnorm_bwd: null pointer");
  SRGAN_REQUIRE((scale == nullptr) == (shift == nullptr), "instnorm_bwd: scale and shift go together");
  hipStream_t st = as_stream(stream);
  int S, rps;
  plan_split(N, HW, C, S, rps);
  SRGAN_REQUIRE(ws && ws_bytes >= (size_t)N * S * C * sizeof(float2), "instnorm_bwd: workspace too small");
  const double tbytes = (double)N * HW * C * sizeof(float);
  if (slab_fast(N, HW, C)) {
    const dim3 gs((unsigned)(C / 16), (unsigned)N);
    const int rows = (HW + 63) / 64;
#define SRGAN_BWD_SLAB(R) hipLaunchKernelGGL(in_bwd_slab<R>, gs, dim3(256), 0, st, x, dy, scale, shift, mean, rstd, dx, dscale, dshift, HW, C, act, slope, slab_remap())
    ProfToken tok = prof_begin(35, 3.0 * tbytes, st);
    if (rows <= 1) SRGAN_BWD_SLAB(1);
    else if (rows <= 2) SRGAN_BWD_SLAB(2);
    else if (rows <= 4) SRGAN_BWD_SLAB(4);
    else if (rows <= 8) SRGAN_BWD_SLAB(8);
    else SRGAN_BWD_SLAB(16);
#undef SRGAN_BWD_SLAB
    prof_end(tok, st);
    return check_launch("instnorm_bwd (slab)");
  }
  float2* part = reinterpret_cast<float2*>(ws);
  dim3 g((C + NORM_CH - 1) / NORM_CH, S, N);
  {
    ProfToken tok = prof_begin(32, 2.0 * tbytes, st);
    if ((C & 3) == 0) hipLaunchKernelGGL((in_bwd_partial_v4<false, false>), g, dim3(256), 0, st, (const void*)x, (const void*)dy, scale, shift, mean, rstd, part, HW, C, S, rps, act, slope);
    else hipLaunchKernelGGL(in_bwd_partial, g, dim3(256), 0, st, x, dy, scale, shift, mean, rstd, part, HW, C, S, rps, act, slope);
    prof_end(tok, st);
  }
  const long long total = (long long)N * HW * C;
  const float inv_hw = 1.f / (float)HW;
  if (pow2_fast(C, HW)) {      // (the backward's fused finish reads the partial sums only: in-place dx is an elementwise update)
    const int hwc4 = HW * C / 4;
    dim3 g2((unsigned)apply_grid(hwc4, N), (unsigned)N);
    ProfToken tok = prof_begin(33, 3.0 * tbytes, st);
    hipLaunchKernelGGL((in_bwd_apply_pow2<false, false, false>), g2, dim3(256), 0, st, (const void*)x, (const void*)dy, scale, shift, mean, rstd, dshift, dscale, (void*)dx, hwc4, C, inv_hw, act,
                       slope, (const float2*)part, S);
    prof_end(tok, st);
    return check_launch("instnorm_bwd");
  }
  hipLaunchKernelGGL(in_bwd_final, dim3((N * C + 255) / 256), dim3(256), 0, st, (const float2*)part, dshift, dscale, N * C, C, S);
  if ((C & 3) == 0) {
    unsigned blocks = (unsigned)std::min<long long>(ceil_div(total / 4, 256), 8192);
    hipLaunchKernelGGL(in_bwd_apply<true>, dim3(blocks), dim3(256), 0, st, x, dy, scale, shift, mean, rstd, dshift, dscale, dx, total, HW * C, C, inv_hw, act, slope);
  } else {
    unsigned blocks = (unsigned)std::min<long long>(ceil_div(total, 256), 8192);
    hipLaunchKernelGGL(in_bwd_apply<false>, dim3(blocks), dim3(256), 0, st, x, dy, scale, shift, mean, rstd, dshift, dscale, dx, total, HW * C, C, inv_hw, act, slope);
  }
  return check_launch("instnorm_bwd");
}

// ---- single-pass slab kernels with bf16 tensors on either side (statistics, scale / shift, sums in fp32) ----
extern "C" int srgan_instnorm_slab_applicable(int N, int HW, int C) { return slab_fast(N, HW, C) ? 1 : 0; }

extern "C" int srgan_instnorm_slab_fwd_io(const void* x, int x_bf16, const float* scale, const float* shift, const float* res,
                                          void* y, int y_bf16, float* mean, float* rstd, int N, int HW, int C, float eps, int act,
                                          float slope, void* stream) {
  SRGAN_REQUIRE(x && y && mean && rstd, "instnorm_slab_fwd_io: null pointer");
  SRGAN_REQUIRE((scale == nullptr) == (shift == nullptr), "instnorm_slab_fwd_io: scale and shift go together");
  SRGAN_REQUIRE(slab_fast(N, HW, C), "instnorm_slab_fwd_io: shape not served by the slab kernels (srgan_instnorm_slab_applicable)");
  SRGAN_REQUIRE(!(res && y_bf16), "instnorm_slab_fwd_io: the skip tensor is added to an fp32 result only");
  hipStream_t st = as_stream(stream);
  const dim3 gs((unsigned)(C / 32), (unsigned)N);
  const double ebytes = (double)N * HW * C;
  ProfToken tok = prof_begin(34, ebytes * ((x_bf16 ? 2 : 4) + (y_bf16 ? 2 : 4) + (res ? 4 : 0)), st);
  if (x_bf16 || y_bf16) {
    // 16-bit tensor on a side: 8 channels per lane, 128 pixels per pass
    const int rows = (HW + 127) / 128;
#define SRGAN_FWD_IO8(R, A, B) hipLaunchKernelGGL((in_fwd_slab8<R, A, B>), gs, dim3(512), 0, st, x, scale, shift, res, y, mean, rstd, HW, C, eps, act, slope, slab_remap())
#define SRGAN_FWD_IO8_R(A, B)                  \
  do {                                         \
    if (rows <= 1) SRGAN_FWD_IO8(1, A, B);     \
    else if (rows <= 2) SRGAN_FWD_IO8(2, A, B);\
    else if (rows <= 4) SRGAN_FWD_IO8(4, A, B);\
    else SRGAN_FWD_IO8(8, A, B);               \
  } while (0)
    if (x_bf16 && y_bf16) SRGAN_FWD_IO8_R(true, true);
    else if (x_bf16) SRGAN_FWD_IO8_R(true, false);
    else SRGAN_FWD_IO8_R(false, true);
#undef SRGAN_FWD_IO8_R
#undef SRGAN_FWD_IO8
  } else {
    const int rows = (HW + 63) / 64;
#define SRGAN_FWD_IO(R) hipLaunchKernelGGL((in_fwd_slab<R, false, false>), gs, dim3(512), 0, st, x, scale, shift, res, y, mean, rstd, HW, C, eps, act, slope, slab_remap())
    if (rows <= 1) SRGAN_FWD_IO(1);
    else if (rows <= 2) SRGAN_FWD_IO(2);
    else if (rows <= 4) SRGAN_FWD_IO(4);
    else if (rows <= 8) SRGAN_FWD_IO(8);
    else SRGAN_FWD_IO(16);
#undef SRGAN_FWD_IO
  }
  prof_end(tok, st);
  return check_launch("instnorm_slab_fwd_io");
}

extern "C" int srgan_instnorm_slab_bwd_io(const void* x, int x_bf16, const void* dy, int dy_bf16, const float* scale,
                                          const float* shift, const float* mean, const float* rstd, void* dx, int dx_bf16,
                                          float* dscale, float* dshift, int N, int HW, int C, int act, float slope, void* stream) {
  SRGAN_REQUIRE(x && dy && mean && rstd && dx && dscale && dshift, "instnorm_slab_bwd_io: null pointer");
  SRGAN_REQUIRE((scale == nullptr) == (shift == nullptr), "instnorm_slab_bwd_io: scale and shift go together");
  SRGAN_REQUIRE(slab_fast(N, HW, C), "instnorm_slab_bwd_io: shape not served by the slab kernels (srgan_instnorm_slab_applicable)");
  SRGAN_REQUIRE((x_bf16 != 0) == (dx_bf16 != 0), "instnorm_slab_bwd_io: the input gradient has the input's type");
  hipStream_t st = as_stream(stream);
  const double ebytes = (double)N * HW * C;
  ProfToken tok = prof_begin(35, ebytes * (2 * (x_bf16 ? 2 : 4) + (dy_bf16 ? 2 : 4)), st);
  // (8 channels per lane with 32-channel slabs, as in the forward, was measured slower here: 23.4 against 15.7 us on the
  // 32 x 32 x 256 trunk -- the backward lives on two 256-thread workgroups per CU overlapping each other's phases)
  const dim3 gs((unsigned)(C / 16), (unsigned)N);
  const int rows = (HW + 63) / 64;
#define SRGAN_BWD_IO(R, X, G) hipLaunchKernelGGL((in_bwd_slab<R, X, G, X>), gs, dim3(256), 0, st, x, dy, scale, shift, mean, rstd, dx, dscale, dshift, HW, C, act, slope, slab_remap())
#define SRGAN_BWD_IO_R(X, G)                  \
  do {                                        \
    if (rows <= 1) SRGAN_BWD_IO(1, X, G);     \
    else if (rows <= 2) SRGAN_BWD_IO(2, X, G);\
    else if (rows <= 4) SRGAN_BWD_IO(4, X, G);\
    else if (rows <= 8) SRGAN_BWD_IO(8, X, G);\
    else SRGAN_BWD_IO(16, X, G);              \
  } while (0)
  if (x_bf16 && dy_bf16) SRGAN_BWD_IO_R(true, true);
  else if (x_bf16) SRGAN_BWD_IO_R(true, false);
  else if (dy_bf16) SRGAN_BWD_IO_R(false, true);
  else SRGAN_BWD_IO_R(false, false);
#undef SRGAN_BWD_IO_R
#undef SRGAN_BWD_IO
  prof_end(tok, st);
  return check_launch("instnorm_slab_bwd_io");
}

// ---- instance norm with 16-bit tensors on either side (round 4: bf16 activation storage outside the residual trunk) ----
// Served shapes: the slab kernels' (maps of <= 1024 pixels) and the fast two-pass kernels' (C | 1024, C % 4 == 0); statistics,
// scale / shift and the parameter-gradient sums stay fp32.  Reference: pyfiles/model.py:54-67, 178 (as srgan_instnorm_fwd / _bwd).
extern "C" int srgan_instnorm_io_applicable(int N, int HW, int C) {
  return (N > 0 && HW > 0 && C > 0 && (slab_fast(N, HW, C) || ((C & 3) == 0 && pow2_fast(C, HW)))) ? 1 : 0;
}

extern "C" int srgan_instnorm_fwd_io(const void* x, int x_bf16, const float* scale, const float* shift, const float* res, void* y,
                                     int y_bf16, float* mean, float* rstd, int N, int HW, int C, float eps, int act, float slope,
                                     void* ws, size_t ws_bytes, void* stream) {
  SRGAN_REQUIRE(x && y && mean && rstd, "instnorm_fwd_io: null pointer");
  SRGAN_REQUIRE(srgan_instnorm_io_applicable(N, HW, C), "instnorm_fwd_io: shape not served (srgan_instnorm_io_applicable)");
  SRGAN_REQUIRE(x != y && (const void*)res != (const void*)y, "instnorm_fwd_io: in-place calls are not supported");
  SRGAN_REQUIRE(!(res && y_bf16), "instnorm_fwd_io: the skip tensor is added to an fp32 result only");
  if (slab_fast(N, HW, C)) return srgan_instnorm_slab_fwd_io(x, x_bf16, scale, shift, res, y, y_bf16, mean, rstd, N, HW, C, eps, act, slope, stream);
  SRGAN_REQUIRE((scale == nullptr) == (shift == nullptr), "instnorm_fwd_io: scale and shift go together");
  hipStream_t st = as_stream(stream);
  int S, rps;
  plan_split(N, HW, C, S, rps);
  SRGAN_REQUIRE(ws && ws_bytes >= (size_t)N * S * C * sizeof(float2), "instnorm_fwd_io: workspace too small (srgan_instnorm_workspace)");
  const double ebytes = (double)N * HW * C;
  float2* part = reinterpret_cast<float2*>(ws);
  dim3 g((C + NORM_CH - 1) / NORM_CH, S, N);
  {
    ProfToken tok = prof_begin(30, ebytes * (x_bf16 ? 2 : 4), st);
    if (x_bf16 && (C & 63) == 0) hipLaunchKernelGGL(in_stats_partial_v8, dim3(C / 64, S, N), dim3(256), 0, st, x, part, HW, C, S, rps);
    else if (x_bf16) hipLaunchKernelGGL(in_stats_partial_v4<true>, g, dim3(256), 0, st, x, part, HW, C, S, rps);
    else hipLaunchKernelGGL(in_stats_partial_v4<false>, g, dim3(256), 0, st, x, part, HW, C, S, rps);
    prof_end(tok, st);
  }
  const int hwc4 = HW * C / 4;
  dim3 g2((unsigned)apply_grid(hwc4, N), (unsigned)N);
  ProfToken tok = prof_begin(31, ebytes * ((x_bf16 ? 2 : 4) + (y_bf16 ? 2 : 4) + (res ? 4 : 0)), st);
#define SRGAN_APPLY_IO(A, B) hipLaunchKernelGGL((in_apply_pow2<A, B>), g2, dim3(256), 0, st, x, scale, shift, res, mean, rstd, y, hwc4, C, act, slope, (const float2*)part, S, HW, eps)
  if (x_bf16 && y_bf16) SRGAN_APPLY_IO(true, true);
  else if (x_bf16) SRGAN_APPLY_IO(true, false);
  else if (y_bf16) SRGAN_APPLY_IO(false, true);
  else SRGAN_APPLY_IO(false, false);
#undef SRGAN_APPLY_IO
  prof_end(tok, st);
  return check_launch("instnorm_fwd_io");
}

extern "C" int srgan_instnorm_bwd_io(const void* x, int x_bf16, const void* dy, int dy_bf16, const float* scale, const float* shift,
                                     const float* mean, const float* rstd, void* dx, int dx_bf16, float* dscale, float* dshift, int N,
                                     int HW, int C, int act, float slope, void* ws, size_t ws_bytes, void* stream) {
  SRGAN_REQUIRE(x && dy && mean && rstd && dx && dscale && dshift, "instnorm_bwd_io: null pointer");
  SRGAN_REQUIRE(srgan_instnorm_io_applicable(N, HW, C), "instnorm_bwd_io: shape not served (srgan_instnorm_io_applicable)");
  SRGAN_REQUIRE((x_bf16 != 0) == (dx_bf16 != 0), "instnorm_bwd_io: the input gradient has the input's type");
  if (slab_fast(N, HW, C)) {
    return srgan_instnorm_slab_bwd_io(x, x_bf16, dy, dy_bf16, scale, shift, mean, rstd, dx, dx_bf16, dscale, dshift, N, HW, C, act, slope, stream);
  }
  SRGAN_REQUIRE((scale == nullptr) == (shift == nullptr), "instnorm_bwd_io: scale and shift go together");
  hipStream_t st = as_stream(stream);
  int S, rps;
  plan_split(N, HW, C, S, rps);
  SRGAN_REQUIRE(ws && ws_bytes >= (size_t)N * S * C * sizeof(float2), "instnorm_bwd_io: workspace too small (srgan_instnorm_workspace)");
  const double ebytes = (double)N * HW * C;
  const int xb = x_bf16 ? 2 : 4, gb = dy_bf16 ? 2 : 4;
  float2* part = reinterpret_cast<float2*>(ws);
  dim3 g((C + NORM_CH - 1) / NORM_CH, S, N);
  {
    ProfToken tok = prof_begin(32, ebytes * (xb + gb), st);
#define SRGAN_BPART_IO(A, B) hipLaunchKernelGGL((in_bwd_partial_v4<A, B>), g, dim3(256), 0, st, x, dy, scale, shift, mean, rstd, part, HW, C, S, rps, act, slope)
    if (x_bf16 && (C & 63) == 0) {
      if (dy_bf16) hipLaunchKernelGGL(in_bwd_partial_v8<true>, dim3(C / 64, S, N), dim3(256), 0, st, x, dy, scale, shift, mean, rstd, part, HW, C, S, rps, act, slope);
      else hipLaunchKernelGGL(in_bwd_partial_v8<false>, dim3(C / 64, S, N), dim3(256), 0, st, x, dy, scale, shift, mean, rstd, part, HW, C, S, rps, act, slope);
    } else if (x_bf16 && dy_bf16) SRGAN_BPART_IO(true, true);
    else if (x_bf16) SRGAN_BPART_IO(true, false);
    else if (dy_bf16) SRGAN_BPART_IO(false, true);
    else SRGAN_BPART_IO(false, false);
#undef SRGAN_BPART_IO
    prof_end(tok, st);
  }
  const int hwc4 = HW * C / 4;
  const float inv_hw = 1.f / (float)HW;
  dim3 g2((unsigned)apply_grid(hwc4, N), (unsigned)N);
  ProfToken tok = prof_begin(33, ebytes * (2 * xb + gb), st);
#define SRGAN_BAPPLY_IO(A, B) hipLaunchKernelGGL((in_bwd_apply_pow2<A, B, A>), g2, dim3(256), 0, st, x, dy, scale, shift, mean, rstd, dshift, dscale, dx, hwc4, C, inv_hw, act, slope, (const float2*)part, S)
  if (x_bf16 && dy_bf16) SRGAN_BAPPLY_IO(true, true);
  else if (x_bf16) SRGAN_BAPPLY_IO(true, false);
  else if (dy_bf16) SRGAN_BAPPLY_IO(false, true);
  else SRGAN_BAPPLY_IO(false, false);
#undef SRGAN_BAPPLY_IO
  prof_end(tok, st);
  return check_launch("instnorm_bwd_io");
}

extern "C" int srgan_cbin_affine_fwd(const float* c, const float* W, const float* b, const float* gamma,
                                     const float* beta, float* t, float* scale, float* shift, int N, int C,
                                     int num_con, void* stream) {
  SRGAN_REQUIRE(c && W && b && gamma && beta && t && scale && shift, "cbin_affine_fwd: null pointer");
  SRGAN_REQUIRE(num_con > 0 && num_con <= 16, "cbin_affine: num_con must be in 1..16");
  hipLaunchKernelGGL(cbin_affine_fwd_kernel, dim3((N * C + 255) / 256), dim3(256), 0, as_stream(stream), c, W, b, gamma, beta, t,
                     scale, shift, N, C, num_con);
  return check_launch("cbin_affine_fwd");
}

extern "C" size_t srgan_cbin_rec_bytes(void) { return sizeof(CbinRec); }

// one host record of the table (the caller copies the array to the device); unused pointers may be null
extern "C" int srgan_cbin_rec_fill(void* rec, const float* W, const float* b, const float* gamma, const float* beta, float* t,
                                   float* scale, float* shift, const float* dscale, const float* dshift, float* dgamma,
                                   float* dbeta, float* dW, float* db, float* da, int C) {
  SRGAN_REQUIRE(rec && C > 0, "cbin_rec_fill: bad argument");
  CbinRec r{W, b, gamma, beta, t, scale, shift, dscale, dshift, dgamma, dbeta, dW, db, da, C, 0};
  std::memcpy(rec, &r, sizeof(r));
  return 0;
}

// The backward pass of this record ADDS its parameter gradients to dgamma / dbeta / dW / db instead of overwriting them (the
// same layer reached a second time in one backward pass: the caller points the record at the first visit's results).
extern "C" int srgan_cbin_rec_set_accumulate(void* rec, int on) {
  SRGAN_REQUIRE(rec, "cbin_rec_set_accumulate: null record");
  CbinRec r;
  std::memcpy(&r, rec, sizeof(r));
  r.accumulate = on != 0;
  std::memcpy(rec, &r, sizeof(r));
  return 0;
}

extern "C" int srgan_cbin_affine_multi_fwd(const float* c, const void* table_dev, int n_layers, int N, int max_C, int num_con,
                                           void* stream) {
  SRGAN_REQUIRE(c && table_dev && n_layers > 0 && N > 0 && max_C > 0, "cbin_affine_multi_fwd: bad argument");
  SRGAN_REQUIRE(num_con > 0 && num_con <= 16, "cbin_affine: num_con must be in 1..16");
  hipLaunchKernelGGL(cbin_affine_multi_fwd_kernel, dim3((N * max_C + 255) / 256, n_layers), dim3(256), 0, as_stream(stream), c,
                     reinterpret_cast<const CbinRec*>(table_dev), N, num_con);
  return check_launch("cbin_affine_multi_fwd");
}

extern "C" int srgan_cbin_affine_multi_bwd(const float* c, const void* table_dev, int n_layers, int N, int max_C, int num_con,
                                           float* dc, void* stream) {
  SRGAN_REQUIRE(c && table_dev && n_layers > 0 && N > 0 && max_C > 0, "cbin_affine_multi_bwd: bad argument");
  SRGAN_REQUIRE(num_con > 0 && num_con <= 16, "cbin_affine: num_con must be in 1..16");
  hipStream_t st = as_stream(stream);
  const CbinRec* tab = reinterpret_cast<const CbinRec*>(table_dev);
  hipLaunchKernelGGL(cbin_affine_multi_bwd_ch, dim3((max_C + 3) / 4, n_layers), dim3(256), 0, st, c, tab, N, num_con);
  if (dc)   // the style code's own gradient: only wanted when the code came out of the encoder (phase 1), not for noise codes
    hipLaunchKernelGGL(cbin_affine_multi_bwd_c, dim3(N), dim3(1024), 0, st, tab, n_layers, dc, N, num_con);
  return check_launch("cbin_affine_multi_bwd");
}

extern "C" int srgan_cbin_affine_bwd(const float* c, const float* W, const float* gamma, const float* t,
                                     const float* dscale, const float* dshift, float* dgamma, float* dbeta, float* dW,
                                     float* db, float* dc, int N, int C, int num_con, void* ws, size_t ws_bytes,
                                     void* stream) {
  SRGAN_REQUIRE(c && W && gamma && t && dscale && dshift && dgamma && dbeta && dW && db && dc, "cbin_affine_bwd: null pointer");
  SRGAN_REQUIRE(num_con > 0 && num_con <= 16, "cbin_affine: num_con must be in 1..16");
  hipStream_t st = as_stream(stream);
  SRGAN_REQUIRE(ws && ws_bytes >= (size_t)N * C * sizeof(float), "cbin_affine_bwd: workspace too small (N*C floats)");
  float* da = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(cbin_affine_bwd_ch, dim3((C + 3) / 4), dim3(256), 0, st, c, gamma, t, dscale, dshift, dgamma, dbeta, dW, db, da, N, C, num_con);
  hipLaunchKernelGGL(cbin_affine_bwd_c, dim3((N + 3) / 4), dim3(256), 0, st, W, (const float*)da, dc, N, C, num_con);
  return check_launch("cbin_affine_bwd");
}

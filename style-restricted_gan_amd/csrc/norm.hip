// Instance normalisation (+ per-(n,c) affine, activation, residual) and the central-biasing
// affine of CBIN, forward and backward, on NHWC fp32 tensors.  HBM-bound kernels: every pass
// reads each element once with channel-contiguous (coalesced) accesses; statistics are reduced
// per (n, c) column with a split over the H*W rows and a deterministic second stage.
//
// Reference: _CBINorm.forward (pyfiles/model.py:54-67), nn.InstanceNorm2d (model.py:178),
// formulas of SURVEY.md Appendix F.1.
#include <algorithm>
#include "common.h"

namespace srgan {

constexpr int NORM_CH = 32;    // channels per block
constexpr int NORM_ROWS = 8;   // row groups per block (256 threads)

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
__global__ __launch_bounds__(256) void in_stats_partial_v4(const float* __restrict__ x, float2* __restrict__ part,
                                                           int HW, int C, int S, int rows_per_split) {
  const int q = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c = blockIdx.x * NORM_CH + q * 4;
  const int s = blockIdx.y, n = blockIdx.z;
  __shared__ f32x4 sh[2][32][8];
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float* xp = x + (size_t)n * HW * C + c;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(xp);
    const int r0 = s * rows_per_split, r1 = min(HW, r0 + rows_per_split);
    for (int r = r0 + ty; r < r1; r += 32) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xp + (size_t)r * C) - x0;
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

__global__ __launch_bounds__(256) void in_bwd_partial_v4(const float* __restrict__ x, const float* __restrict__ dy,
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
    for (int r = r0 + ty; r < r1; r += 32) {
      const size_t o = base + (size_t)r * C;
      const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + o) - mu) * rs;
      f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
      const f32x4 z = xh * sc + sf;
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] *= act_grad(z[e], act, slope);
      a += g;
      b += g * xh;
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
  const float inv = 1.f / (float)HW;
  const float dm = a * inv;
  float var = b * inv - dm * dm;
  var = var < 0.f ? 0.f : var;
  mean[idx] = x[(size_t)n * HW * C + c] + dm;
  rstd[idx] = 1.0f / sqrtf(var + eps);
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
__global__ __launch_bounds__(256) void in_apply_pow2(const float* __restrict__ x, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const float* __restrict__ res,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     float* __restrict__ y, int HWC4, int C, int act, float slope) {
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
  const size_t base = (size_t)n * HWC4;
  const f32x4* xp = reinterpret_cast<const f32x4*>(x) + base;
  const f32x4* rp = res ? reinterpret_cast<const f32x4*>(res) + base : nullptr;
  f32x4* yp = reinterpret_cast<f32x4*>(y) + base;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < HWC4; j += gridDim.x * 256) {
    f32x4 v = ((xp[j] - mu) * rs) * sc + sf;      // same expression as the backward's mask recomputation
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = apply_act(v[k], act, slope);
    if (rp) v += rp[j];
    yp[j] = v;
  }
}

__global__ __launch_bounds__(256) void in_bwd_apply_pow2(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ dshift, const float* __restrict__ dscale,
                                                         float* __restrict__ dx, int HWC4, int C, float inv_hw, int act,
                                                         float slope) {
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
  const f32x4 mg = *reinterpret_cast<const f32x4*>(dshift + nc) * inv_hw;
  const f32x4 mgx = *reinterpret_cast<const f32x4*>(dscale + nc) * inv_hw;
  const f32x4 k = rs * sc;
  const size_t base = (size_t)n * HWC4;
  const f32x4* xp = reinterpret_cast<const f32x4*>(x) + base;
  const f32x4* gp = reinterpret_cast<const f32x4*>(dy) + base;
  f32x4* op = reinterpret_cast<f32x4*>(dx) + base;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < HWC4; j += gridDim.x * 256) {
    const f32x4 xh = (xp[j] - mu) * rs;
    f32x4 g = gp[j];
    const f32x4 z = xh * sc + sf;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] *= act_grad(z[e], act, slope);
    op[j] = k * (g - mg - xh * mgx);
  }
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

namespace {
// C divides 1024 (so 256 threads * 4 floats wrap onto the same channels) and every float4 stays inside one pixel
bool pow2_fast(int C, int HW) { return C >= 4 && (1024 % C) == 0 && (long long)HW * C / 4 < (1LL << 30); }
int apply_grid(int hwc4, int N) {
  long long per_image = ceil_div(hwc4, 256);                 // blocks if one float4 per thread
  long long want = std::max<long long>(1, 2048 / std::max(1, N));
  return (int)std::max<long long>(1, std::min(per_image, want));
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
  float2* part = reinterpret_cast<float2*>(ws);
  dim3 g((C + NORM_CH - 1) / NORM_CH, S, N);
  if ((C & 3) == 0) hipLaunchKernelGGL(in_stats_partial_v4, g, dim3(256), 0, st, x, part, HW, C, S, rps);
  else hipLaunchKernelGGL(in_stats_partial, g, dim3(256), 0, st, x, part, HW, C, S, rps);
  hipLaunchKernelGGL(in_stats_final, dim3((N * C + 255) / 256), dim3(256), 0, st, x, (const float2*)part, mean, rstd, N, HW, C, S, eps);
  const long long total = (long long)N * HW * C;
  if (pow2_fast(C, HW)) {
    const int hwc4 = HW * C / 4;
    dim3 g2((unsigned)apply_grid(hwc4, N), (unsigned)N);
    hipLaunchKernelGGL(in_apply_pow2, g2, dim3(256), 0, st, x, scale, shift, res, mean, rstd, y, hwc4, C, act, slope);
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
  float2* part = reinterpret_cast<float2*>(ws);
  dim3 g((C + NORM_CH - 1) / NORM_CH, S, N);
  if ((C & 3) == 0) hipLaunchKernelGGL(in_bwd_partial_v4, g, dim3(256), 0, st, x, dy, scale, shift, mean, rstd, part, HW, C, S, rps, act, slope);
  else hipLaunchKernelGGL(in_bwd_partial, g, dim3(256), 0, st, x, dy, scale, shift, mean, rstd, part, HW, C, S, rps, act, slope);
  hipLaunchKernelGGL(in_bwd_final, dim3((N * C + 255) / 256), dim3(256), 0, st, (const float2*)part, dshift, dscale, N * C, C, S);
  const long long total = (long long)N * HW * C;
  const float inv_hw = 1.f / (float)HW;
  if (pow2_fast(C, HW)) {
    const int hwc4 = HW * C / 4;
    dim3 g2((unsigned)apply_grid(hwc4, N), (unsigned)N);
    hipLaunchKernelGGL(in_bwd_apply_pow2, g2, dim3(256), 0, st, x, dy, scale, shift, mean, rstd, dshift, dscale, dx, hwc4, C, inv_hw, act, slope);
  } else if ((C & 3) == 0) {
    unsigned blocks = (unsigned)std::min<long long>(ceil_div(total / 4, 256), 8192);
    hipLaunchKernelGGL(in_bwd_apply<true>, dim3(blocks), dim3(256), 0, st, x, dy, scale, shift, mean, rstd, dshift, dscale, dx, total, HW * C, C, inv_hw, act, slope);
  } else {
    unsigned blocks = (unsigned)std::min<long long>(ceil_div(total, 256), 8192);
    hipLaunchKernelGGL(in_bwd_apply<false>, dim3(blocks), dim3(256), 0, st, x, dy, scale, shift, mean, rstd, dshift, dscale, dx, total, HW * C, C, inv_hw, act, slope);
  }
  return check_launch("instnorm_bwd");
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

// Small HBM-bound kernels of the train step: activations, pools, layout repacks, Linear heads,
// fused Adam.  NHWC fp32.  Reference call sites are listed in include/srgan_hip.h.
#include <algorithm>
#include <cmath>
#include <cstring>
#include "common.h"

namespace srgan {

static inline unsigned grid_for(long long n, int per_thread = 1) {
  long long b = ceil_div(ceil_div(n, per_thread), 256);
  return (unsigned)std::max<long long>(1, std::min<long long>(b, 8192));
}

#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

__global__ void act_fwd_kernel(const float* x, float* y, long long n, int act, float slope) {
  GRID_STRIDE(i, n) y[i] = apply_act(x[i], act, slope);
}
__global__ void act_bwd_kernel(const float* y, const float* dy, float* dx, long long n, int act, float slope) {
  GRID_STRIDE(i, n) dx[i] = dy[i] * act_grad(y[i], act, slope);
}
// 16-bit I/O (round 6: the discriminator trunks' bf16 activations): 8 elements per thread, any mix of fp32 / bf16 tensors
template <bool Y16, bool D16, bool X16>
__global__ __launch_bounds__(256) void act_bwd_io_kernel(const void* y, const void* dy, void* dx, long long n8, int act, float slope) {
  const float neg = act_neg_slope(act, slope);
  GRID_STRIDE(i, n8) {
    float yv[8], dv[8];
    if constexpr (Y16) {
      const bf16x8 t = reinterpret_cast<const bf16x8*>(y)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) yv[e] = (float)t[e];
    } else {
      const f32x4 a = reinterpret_cast<const f32x4*>(y)[2 * i], b = reinterpret_cast<const f32x4*>(y)[2 * i + 1];
#pragma unroll
      for (int e = 0; e < 4; ++e) { yv[e] = a[e]; yv[4 + e] = b[e]; }
    }
    if constexpr (D16) {
      const bf16x8 t = reinterpret_cast<const bf16x8*>(dy)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) dv[e] = (float)t[e];
    } else {
      const f32x4 a = reinterpret_cast<const f32x4*>(dy)[2 * i], b = reinterpret_cast<const f32x4*>(dy)[2 * i + 1];
#pragma unroll
      for (int e = 0; e < 4; ++e) { dv[e] = a[e]; dv[4 + e] = b[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dv[e] = dv[e] * (yv[e] > 0.f ? 1.f : neg);
    if constexpr (X16) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)dv[e];
      reinterpret_cast<bf16x8*>(dx)[i] = o;
    } else {
      reinterpret_cast<f32x4*>(dx)[2 * i] = f32x4{dv[0], dv[1], dv[2], dv[3]};
      reinterpret_cast<f32x4*>(dx)[2 * i + 1] = f32x4{dv[4], dv[5], dv[6], dv[7]};
    }
  }
}
__global__ void tanh_fwd_kernel(const float* x, float* y, long long n) {
  GRID_STRIDE(i, n) y[i] = tanhf(x[i]);
}
__global__ void tanh_bwd_kernel(const float* y, const float* dy, float* dx, long long n) {
  GRID_STRIDE(i, n) { const float t = y[i]; dx[i] = dy[i] * (1.f - t * t); }
}
__global__ void add_kernel(const float* a, const float* b, float* y, long long n) {
  GRID_STRIDE(i, n) y[i] = a[i] + b[i];
}

// AvgPool2d(3, s2, p1, count_include_pad=False): Ho = (H+2-3)/2+1
__global__ void avgpool3s2_fwd_kernel(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo) {
  const long long total = (long long)N * Ho * Wo * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float s = 0.f;
    int cnt = 0;
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = oy * 2 + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = ox * 2 + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        s += x[((size_t)(n * H + yy) * W + xx) * C + c];
        ++cnt;
      }
    }
    y[i] = s / (float)cnt;
  }
}
__device__ __forceinline__ int valid3(int o, int L) {  // taps of window o that fall inside [0,L)
  int c = 0;
  for (int d = -1; d <= 1; ++d) c += ((unsigned)(o * 2 + d) < (unsigned)L);
  return c;
}
__global__ void avgpool3s2_bwd_kernel(const float* dy, float* dx, int N, int H, int W, int C, int Ho, int Wo) {
  const long long total = (long long)N * H * W * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    float s = 0.f;
    // windows oy with |2*oy - y| <= 1
    for (int oy = (y - 1 + 1) / 2 - 1; oy <= (y + 1) / 2; ++oy) {
      if (oy < 0 || oy >= Ho || abs(2 * oy - y) > 1) continue;
      const int cy = valid3(oy, H);
      for (int ox = (x - 1 + 1) / 2 - 1; ox <= (x + 1) / 2; ++ox) {
        if (ox < 0 || ox >= Wo || abs(2 * ox - x) > 1) continue;
        const int cx = valid3(ox, W);
        s += dy[((size_t)(n * Ho + oy) * Wo + ox) * C + c] / (float)(cy * cx);
      }
    }
    dx[i] = s;
  }
}

__global__ void avgpool2_fwd_kernel(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo) {
  const long long total = (long long)N * Ho * Wo * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const size_t b = ((size_t)(n * H + oy * 2) * W + ox * 2) * C + c;
    y[i] = 0.25f * (x[b] + x[b + C] + x[b + (size_t)W * C] + x[b + (size_t)W * C + C]);
  }
}
__global__ void avgpool2_bwd_kernel(const float* dy, float* dx, int N, int H, int W, int C, int Ho, int Wo) {
  const long long total = (long long)N * H * W * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    const int oy = y >> 1, ox = x >> 1;
    dx[i] = (oy < Ho && ox < Wo) ? 0.25f * dy[((size_t)(n * Ho + oy) * Wo + ox) * C + c] : 0.f;
  }
}

// AvgPool2d(2, 2) with an fp32 or bf16 tensor on either side (the bf16 mode's 16-bit activations around the style encoder's
// convolutions, pyfiles/model.py:409-411): 4 channels per thread (C % 4 == 0), the mean in fp32.
template <bool B16>
__device__ __forceinline__ f32x4 ld4_io(const void* base, size_t idx) {
  if constexpr (B16) return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(base) + idx), f32x4);
  else return *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + idx);
}
template <bool B16>
__device__ __forceinline__ void st4_io(void* base, size_t idx, f32x4 v) {
  if constexpr (B16) *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(base) + idx) = __builtin_convertvector(v, bf16x4);
  else *reinterpret_cast<f32x4*>(static_cast<float*>(base) + idx) = v;
}
template <bool X16, bool Y16>
__global__ void avgpool2_fwd_io_kernel(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo) {
  const int C4 = C >> 2;
  const long long total = (long long)N * Ho * Wo * C4;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C4) * 4;
    long long r = i / C4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const size_t b = ((size_t)(n * H + oy * 2) * W + ox * 2) * C + c;
    const f32x4 v = 0.25f * (ld4_io<X16>(x, b) + ld4_io<X16>(x, b + C) + ld4_io<X16>(x, b + (size_t)W * C) +
                             ld4_io<X16>(x, b + (size_t)W * C + C));
    st4_io<Y16>(y, (size_t)i * 4, v);
  }
}
template <bool G16, bool D16>
__global__ void avgpool2_bwd_io_kernel(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo) {
  const int C4 = C >> 2;
  const long long total = (long long)N * H * W * C4;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C4) * 4;
    long long r = i / C4;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    const int oy = y >> 1, ox = x >> 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (oy < Ho && ox < Wo) v = 0.25f * ld4_io<G16>(dy, ((size_t)(n * Ho + oy) * Wo + ox) * C + c);
    st4_io<D16>(dx, (size_t)i * 4, v);
  }
}

__global__ void lrelu_gap_fwd_kernel(const float* x, float* y, int N, int HW, int C, float slope) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * C) return;
  const int n = idx / C, c = idx - n * C;
  float s = 0.f;
  for (int r = 0; r < HW; ++r) {
    const float v = x[((size_t)n * HW + r) * C + c];
    s += v > 0.f ? v : v * slope;
  }
  y[idx] = s / (float)HW;
}
__global__ void lrelu_gap_bwd_kernel(const float* x, const float* dy, float* dx, int N, int HW, int C, float slope) {
  const long long total = (long long)N * HW * C;
  const float inv = 1.f / (float)HW;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    const int n = (int)(i / ((long long)HW * C));
    dx[i] = dy[n * C + c] * inv * (x[i] > 0.f ? 1.f : slope);
  }
}

// Linear heads: one wave per output element (M, N small; K = 1024)
__global__ void linear_fwd_kernel(const float* x, const float* W, const float* b, float* y, int M, int N, int K) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= M * N) return;
  const int m = wave / N, n = wave - m * N;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += x[(size_t)m * K + k] * W[(size_t)n * K + k];
  s = wave_sum(s);
  if (lane == 0) y[wave] = s + (b ? b[n] : 0.f);
}
__global__ void linear_bwd_dx_kernel(const float* W, const float* dy, float* dx, int M, int N, int K) {
  const long long total = (long long)M * K;
  GRID_STRIDE(i, total) {
    const int k = (int)(i % K), m = (int)(i / K);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dy[m * N + n] * W[(size_t)n * K + k];
    dx[i] = s;
  }
}
__global__ void linear_bwd_dw_kernel(const float* x, const float* dy, float* dW, float* db, int M, int N, int K) {
  const long long total = (long long)N * K;
  GRID_STRIDE(i, total) {
    const int k = (int)(i % K), n = (int)(i / K);
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dy[m * N + n] * x[(size_t)m * K + k];
    dW[i] = s;
    if (k == 0 && db) {
      float t = 0.f;
      for (int m = 0; m < M; ++m) t += dy[m * N + n];
      db[n] = t;
    }
  }
}

// layout repack through a 32x33 LDS tile: [N][C][P] <-> [N][P][C], P = H*W
__global__ void transpose_cp_kernel(const float* x, float* y, int rows, int cols) {
  // x: [n][rows][cols] -> y: [n][cols][rows]
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const float* xp = x + (size_t)n * rows * cols;
  float* yp = y + (size_t)n * rows * cols;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) tile[j][tx] = xp[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) yp[(size_t)c * rows + r] = tile[tx][j];
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float step_size, float beta1, float beta2, float eps,
                            float inv_sqrt_bc2) {
  GRID_STRIDE(i, n) {
    const float gi = g[i];
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
}

// multi-tensor Adam: one launch for a whole parameter list.  `table` holds n_tensors records of 5 x 8 bytes:
// {p, g, m, v, numel}; blockIdx.y selects the tensor, blockIdx.x grid-strides over its elements.
__global__ __launch_bounds__(256) void adam_multi_kernel(const unsigned long long* __restrict__ table, float step_size,
                                                         float beta1, float beta2, float eps, float inv_sqrt_bc2) {
  const unsigned long long* rec = table + (size_t)blockIdx.y * 5;
  float* __restrict__ p = reinterpret_cast<float*>(rec[0]);
  const float* __restrict__ g = reinterpret_cast<const float*>(rec[1]);
  float* __restrict__ m = reinterpret_cast<float*>(rec[2]);
  float* __restrict__ v = reinterpret_cast<float*>(rec[3]);
  const long long n = (long long)rec[4];
  // a block owns chunks of 4096 elements (blockIdx.x, then a grid stride); blocks past a small tensor's end leave at once
  const bool vec = ((rec[0] | rec[1] | rec[2] | rec[3]) & 15) == 0;
  for (long long c0 = (long long)blockIdx.x * 4096; c0 < n; c0 += (long long)gridDim.x * 4096) {
    if (vec && c0 + 4096 <= n) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long i = c0 + (j * 256 + threadIdx.x) * 4;
        const f32x4 gi = *reinterpret_cast<const f32x4*>(g + i);
        const f32x4 mi = beta1 * *reinterpret_cast<const f32x4*>(m + i) + (1.f - beta1) * gi;
        const f32x4 vi = beta2 * *reinterpret_cast<const f32x4*>(v + i) + (1.f - beta2) * gi * gi;
        f32x4 pi = *reinterpret_cast<const f32x4*>(p + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) pi[e] = pi[e] - step_size * (mi[e] / (sqrtf(vi[e]) * inv_sqrt_bc2 + eps));
        *reinterpret_cast<f32x4*>(m + i) = mi;
        *reinterpret_cast<f32x4*>(v + i) = vi;
        *reinterpret_cast<f32x4*>(p + i) = pi;
      }
    } else {
      const long long end = c0 + 4096 < n ? c0 + 4096 : n;
      for (long long i = c0 + threadIdx.x; i < end; i += 256) {
        const float gi = g[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
      }
    }
  }
}


// ---- small host -> device uploads carried in the kernel ARGUMENTS -------------------------------------------------
// Pointer tables (multi-tensor Adam, the central-biasing layer records, the weight-repack entries) are a few KB that the host
// knows at launch time.  A pinned staging buffer + hipMemcpyAsync needs the staging block kept alive until the copy has run
// and, captured into a hipGraph, would re-read that host address on every replay; a kernel whose by-value argument IS the
// data has neither problem: the bytes travel in the launch packet (and are stored in the graph node).
constexpr int kUploadChunk = 2048;
struct UploadBlob { unsigned int w[kUploadChunk / 4]; };
__global__ __launch_bounds__(256) void upload_small_kernel(unsigned int* __restrict__ dst, UploadBlob blob, int nwords) {
  for (int i = threadIdx.x; i < nwords; i += 256) dst[i] = blob.w[i];
}

// ---- device-resident Adam state ---------------------------------------------------------------------------------------
// {step, lr, beta1, beta2, eps, step_size, inv_sqrt_bc2}: the step counter lives on the device so that a captured train step
// (hipGraph) advances it on every replay; the host mirrors the count for checkpoints only.
struct AdamState { int step; float lr, beta1, beta2, eps, step_size, inv_sqrt_bc2; int pad; };
static_assert(sizeof(AdamState) == 32, "AdamState layout");
__global__ void adam_state_init_kernel(AdamState* s, int step, float lr, float beta1, float beta2, float eps) {
  s->step = step; s->lr = lr; s->beta1 = beta1; s->beta2 = beta2; s->eps = eps; s->step_size = 0.f; s->inv_sqrt_bc2 = 0.f; s->pad = 0;
}
__global__ void adam_state_lr_kernel(AdamState* s, float lr) { s->lr = lr; }
// torch 1.4: p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)   (bias corrections in double, as the host path did)
__global__ void adam_tick_kernel(AdamState* s) {
  const int t = s->step + 1;
  s->step = t;
  const double bc1 = 1.0 - pow((double)s->beta1, (double)t);
  const double bc2 = 1.0 - pow((double)s->beta2, (double)t);
  s->step_size = (float)((double)s->lr / bc1);
  s->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
}
__global__ __launch_bounds__(256) void adam_multi_dev_kernel(const unsigned long long* __restrict__ table,
                                                             const AdamState* __restrict__ st) {
  const float step_size = st->step_size, beta1 = st->beta1, beta2 = st->beta2, eps = st->eps, inv_sqrt_bc2 = st->inv_sqrt_bc2;
  const unsigned long long* rec = table + (size_t)blockIdx.y * 5;
  float* __restrict__ p = reinterpret_cast<float*>(rec[0]);
  const float* __restrict__ g = reinterpret_cast<const float*>(rec[1]);
  float* __restrict__ m = reinterpret_cast<float*>(rec[2]);
  float* __restrict__ v = reinterpret_cast<float*>(rec[3]);
  const long long n = (long long)rec[4];
  const bool vec = ((rec[0] | rec[1] | rec[2] | rec[3]) & 15) == 0;
  for (long long c0 = (long long)blockIdx.x * 4096; c0 < n; c0 += (long long)gridDim.x * 4096) {
    if (vec && c0 + 4096 <= n) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long long i = c0 + (j * 256 + threadIdx.x) * 4;
        const f32x4 gi = *reinterpret_cast<const f32x4*>(g + i);
        const f32x4 mi = beta1 * *reinterpret_cast<const f32x4*>(m + i) + (1.f - beta1) * gi;
        const f32x4 vi = beta2 * *reinterpret_cast<const f32x4*>(v + i) + (1.f - beta2) * gi * gi;
        f32x4 pi = *reinterpret_cast<const f32x4*>(p + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) pi[e] = pi[e] - step_size * (mi[e] / (sqrtf(vi[e]) * inv_sqrt_bc2 + eps));
        *reinterpret_cast<f32x4*>(m + i) = mi;
        *reinterpret_cast<f32x4*>(v + i) = vi;
        *reinterpret_cast<f32x4*>(p + i) = pi;
      }
    } else {
      const long long end = c0 + 4096 < n ? c0 + 4096 : n;
      for (long long i = c0 + threadIdx.x; i < end; i += 256) {
        const float gi = g[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
      }
    }
  }
}

}  // namespace srgan

using namespace srgan;

#define LAUNCH1D(kernel, n, st, ...)                                                        \
  do {                                                                                      \
    hipLaunchKernelGGL(kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(st), __VA_ARGS__); \
    return check_launch(#kernel);                                                           \
  } while (0)

extern "C" int srgan_act_fwd(const float* x, float* y, long long n, int act, float slope, void* stream) {
  SRGAN_REQUIRE(x && y && n >= 0, "act_fwd: bad argument");
  if (n == 0) return 0;
  LAUNCH1D(act_fwd_kernel, n, stream, x, y, n, act, slope);
}
extern "C" int srgan_act_bwd(const float* y, const float* dy, float* dx, long long n, int act, float slope, void* stream) {
  SRGAN_REQUIRE(y && dy && dx && n >= 0, "act_bwd: bad argument");
  if (n == 0) return 0;
  LAUNCH1D(act_bwd_kernel, n, stream, y, dy, dx, n, act, slope);
}
extern "C" int srgan_act_bwd_io(const void* y, int y_bf16, const void* dy, int dy_bf16, void* dx, int dx_bf16, long long n, int act,
                                float slope, void* stream) {
  SRGAN_REQUIRE(y && dy && dx && n >= 0, "act_bwd_io: bad argument");
  SRGAN_REQUIRE((n & 7) == 0, "act_bwd_io: n %% 8 != 0");
  if (n == 0) return 0;
  const long long n8 = n / 8;
  const int sel = (y_bf16 ? 4 : 0) | (dy_bf16 ? 2 : 0) | (dx_bf16 ? 1 : 0);
  switch (sel) {
    case 0: LAUNCH1D((act_bwd_io_kernel<false, false, false>), n8, stream, y, dy, dx, n8, act, slope);
    case 1: LAUNCH1D((act_bwd_io_kernel<false, false, true>), n8, stream, y, dy, dx, n8, act, slope);
    case 2: LAUNCH1D((act_bwd_io_kernel<false, true, false>), n8, stream, y, dy, dx, n8, act, slope);
    case 3: LAUNCH1D((act_bwd_io_kernel<false, true, true>), n8, stream, y, dy, dx, n8, act, slope);
    case 4: LAUNCH1D((act_bwd_io_kernel<true, false, false>), n8, stream, y, dy, dx, n8, act, slope);
    case 5: LAUNCH1D((act_bwd_io_kernel<true, false, true>), n8, stream, y, dy, dx, n8, act, slope);
    case 6: LAUNCH1D((act_bwd_io_kernel<true, true, false>), n8, stream, y, dy, dx, n8, act, slope);
    default: LAUNCH1D((act_bwd_io_kernel<true, true, true>), n8, stream, y, dy, dx, n8, act, slope);
  }
}
extern "C" int srgan_tanh_fwd(const float* x, float* y, long long n, void* stream) {
  SRGAN_REQUIRE(x && y && n >= 0, "tanh_fwd: bad argument");
  if (n == 0) return 0;
  LAUNCH1D(tanh_fwd_kernel, n, stream, x, y, n);
}
extern "C" int srgan_tanh_bwd(const float* y, const float* dy, float* dx, long long n, void* stream) {
  SRGAN_REQUIRE(y && dy && dx && n >= 0, "tanh_bwd: bad argument");
  if (n == 0) return 0;
  LAUNCH1D(tanh_bwd_kernel, n, stream, y, dy, dx, n);
}
extern "C" int srgan_add(const float* a, const float* b, float* y, long long n, void* stream) {
  SRGAN_REQUIRE(a && b && y && n >= 0, "add: bad argument");
  if (n == 0) return 0;
  LAUNCH1D(add_kernel, n, stream, a, b, y, n);
}

extern "C" int srgan_avgpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  SRGAN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, "avgpool3s2_fwd: bad argument");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  LAUNCH1D(avgpool3s2_fwd_kernel, (long long)N * Ho * Wo * C, stream, x, y, N, H, W, C, Ho, Wo);
}
extern "C" int srgan_avgpool3s2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
  SRGAN_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "avgpool3s2_bwd: bad argument");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  LAUNCH1D(avgpool3s2_bwd_kernel, (long long)N * H * W * C, stream, dy, dx, N, H, W, C, Ho, Wo);
}
extern "C" int srgan_avgpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  SRGAN_REQUIRE(x && y && N > 0 && H > 1 && W > 1 && C > 0, "avgpool2_fwd: bad argument");
  LAUNCH1D(avgpool2_fwd_kernel, (long long)N * (H / 2) * (W / 2) * C, stream, x, y, N, H, W, C, H / 2, W / 2);
}
extern "C" int srgan_avgpool2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
  SRGAN_REQUIRE(dy && dx && N > 0 && H > 1 && W > 1 && C > 0, "avgpool2_bwd: bad argument");
  LAUNCH1D(avgpool2_bwd_kernel, (long long)N * H * W * C, stream, dy, dx, N, H, W, C, H / 2, W / 2);
}
extern "C" int srgan_avgpool2_fwd_io(const void* x, int x_bf16, void* y, int y_bf16, int N, int H, int W, int C, void* stream) {
  SRGAN_REQUIRE(x && y && N > 0 && H > 1 && W > 1 && C > 0 && (C & 3) == 0, "avgpool2_fwd_io: bad argument (C % 4 == 0)");
  const long long n = (long long)N * (H / 2) * (W / 2) * (C / 4);
  if (x_bf16 && y_bf16) { LAUNCH1D((avgpool2_fwd_io_kernel<true, true>), n, stream, x, y, N, H, W, C, H / 2, W / 2); }
  if (x_bf16) { LAUNCH1D((avgpool2_fwd_io_kernel<true, false>), n, stream, x, y, N, H, W, C, H / 2, W / 2); }
  if (y_bf16) { LAUNCH1D((avgpool2_fwd_io_kernel<false, true>), n, stream, x, y, N, H, W, C, H / 2, W / 2); }
  LAUNCH1D((avgpool2_fwd_io_kernel<false, false>), n, stream, x, y, N, H, W, C, H / 2, W / 2);
}
extern "C" int srgan_avgpool2_bwd_io(const void* dy, int dy_bf16, void* dx, int dx_bf16, int N, int H, int W, int C, void* stream) {
  SRGAN_REQUIRE(dy && dx && N > 0 && H > 1 && W > 1 && C > 0 && (C & 3) == 0, "avgpool2_bwd_io: bad argument (C % 4 == 0)");
  const long long n = (long long)N * H * W * (C / 4);
  if (dy_bf16 && dx_bf16) { LAUNCH1D((avgpool2_bwd_io_kernel<true, true>), n, stream, dy, dx, N, H, W, C, H / 2, W / 2); }
  if (dy_bf16) { LAUNCH1D((avgpool2_bwd_io_kernel<true, false>), n, stream, dy, dx, N, H, W, C, H / 2, W / 2); }
  if (dx_bf16) { LAUNCH1D((avgpool2_bwd_io_kernel<false, true>), n, stream, dy, dx, N, H, W, C, H / 2, W / 2); }
  LAUNCH1D((avgpool2_bwd_io_kernel<false, false>), n, stream, dy, dx, N, H, W, C, H / 2, W / 2);
}
extern "C" int srgan_lrelu_gap_fwd(const float* x, float* y, int N, int HW, int C, float slope, void* stream) {
  SRGAN_REQUIRE(x && y && N > 0 && HW > 0 && C > 0, "lrelu_gap_fwd: bad argument");
  LAUNCH1D(lrelu_gap_fwd_kernel, (long long)N * C, stream, x, y, N, HW, C, slope);
}
extern "C" int srgan_lrelu_gap_bwd(const float* x, const float* dy, float* dx, int N, int HW, int C, float slope, void* stream) {
  SRGAN_REQUIRE(x && dy && dx && N > 0 && HW > 0 && C > 0, "lrelu_gap_bwd: bad argument");
  LAUNCH1D(lrelu_gap_bwd_kernel, (long long)N * HW * C, stream, x, dy, dx, N, HW, C, slope);
}

// Encoder.reparametrize (pyfiles/model.py:459-463): c = eps * exp(logvar / 2) + mu on [B, ndim].  The reference is four
// elementwise launches forward and three backward on 256 floats; here one each.  Every product and sum is rounded on its own
// (no FMA contraction): the same values as the chain of separate elementwise passes.
namespace srgan {
__global__ void reparam_fwd_kernel(const float* mu, const float* logvar, const float* eps, float* out, float* stdv, long long n) {
  GRID_STRIDE(i, n) {
    const float s = expf(__fmul_rn(0.5f, logvar[i]));
    stdv[i] = s;
    out[i] = __fadd_rn(__fmul_rn(eps[i], s), mu[i]);
  }
}
__global__ void reparam_bwd_kernel(const float* g, const float* eps, const float* stdv, float* dlogvar, long long n) {
  GRID_STRIDE(i, n) dlogvar[i] = __fmul_rn(__fmul_rn(__fmul_rn(g[i], eps[i]), stdv[i]), 0.5f);
}
}  // namespace srgan
extern "C" int srgan_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* out, float* stdv, long long n,
                                 void* stream) {
  SRGAN_REQUIRE(mu && logvar && eps && out && stdv && n > 0, "reparam_fwd: bad argument");
  LAUNCH1D(srgan::reparam_fwd_kernel, n, stream, mu, logvar, eps, out, stdv, n);
}
extern "C" int srgan_reparam_bwd(const float* g, const float* eps, const float* stdv, float* dlogvar, long long n, void* stream) {
  SRGAN_REQUIRE(g && eps && stdv && dlogvar && n > 0, "reparam_bwd: bad argument");
  LAUNCH1D(srgan::reparam_bwd_kernel, n, stream, g, eps, stdv, dlogvar, n);
}

extern "C" int srgan_linear_fwd(const float* x, const float* W, const float* b, float* y, int M, int N, int K, void* stream) {
  SRGAN_REQUIRE(x && W && y && M > 0 && N > 0 && K > 0, "linear_fwd: bad argument");
  const long long threads = (long long)M * N * 64;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, as_stream(stream), x, W, b, y, M, N, K);
  return check_launch("linear_fwd_kernel");
}
extern "C" int srgan_linear_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db,
                                int M, int N, int K, void* stream) {
  SRGAN_REQUIRE(x && W && dy && M > 0 && N > 0 && K > 0, "linear_bwd: bad argument");
  hipStream_t st = as_stream(stream);
  if (dx) hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(grid_for((long long)M * K)), dim3(256), 0, st, W, dy, dx, M, N, K);
  if (dW) hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(grid_for((long long)N * K)), dim3(256), 0, st, x, dy, dW, db, M, N, K);
  return check_launch("linear_bwd");
}

extern "C" int srgan_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream) {
  SRGAN_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad argument");
  const int P = H * W;  // [N][C][P] -> [N][P][C]
  dim3 g((P + 31) / 32, (C + 31) / 32, N);
  hipLaunchKernelGGL(transpose_cp_kernel, g, dim3(256), 0, as_stream(stream), x, y, C, P);
  return check_launch("transpose_cp_kernel");
}
extern "C" int srgan_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream) {
  SRGAN_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad argument");
  const int P = H * W;  // [N][P][C] -> [N][C][P]
  dim3 g((C + 31) / 32, (P + 31) / 32, N);
  hipLaunchKernelGGL(transpose_cp_kernel, g, dim3(256), 0, as_stream(stream), x, y, P, C);
  return check_launch("transpose_cp_kernel");
}

extern "C" int srgan_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                               float beta2, float eps, int step_count, void* stream) {
  SRGAN_REQUIRE(p && g && m && v && n >= 0 && step_count >= 1, "adam_step: bad argument");
  if (n == 0) return 0;
  // torch 1.4: p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
  const double bc1 = 1.0 - std::pow((double)beta1, step_count);
  const double bc2 = 1.0 - std::pow((double)beta2, step_count);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
  LAUNCH1D(adam_kernel, n, stream, p, g, m, v, n, step_size, beta1, beta2, eps, inv_sqrt_bc2);
}

extern "C" int srgan_adam_multi(const void* table, int n_tensors, long long max_numel, float lr, float beta1, float beta2,
                                float eps, int step_count, void* stream) {
  SRGAN_REQUIRE(table && n_tensors > 0 && max_numel > 0 && step_count >= 1, "adam_multi: bad argument");
  const double bc1 = 1.0 - std::pow((double)beta1, step_count);
  const double bc2 = 1.0 - std::pow((double)beta2, step_count);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
  const unsigned bx = (unsigned)std::max<long long>(1, std::min<long long>(ceil_div(max_numel, 4096), 2048));
  hipLaunchKernelGGL(adam_multi_kernel, dim3(bx, (unsigned)n_tensors), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const unsigned long long*>(table), step_size, beta1, beta2, eps, inv_sqrt_bc2);
  return check_launch("adam_multi_kernel");
}

extern "C" int srgan_upload_small(void* dst_dev, const void* src_host, size_t nbytes, void* stream) {
  SRGAN_REQUIRE(dst_dev && src_host && nbytes % 4 == 0, "upload_small: bad argument (bytes must be a multiple of 4)");
  SRGAN_REQUIRE(nbytes <= (1u << 20), "upload_small: meant for tables of a few KB (%zu bytes asked)", nbytes);
  const unsigned char* src = static_cast<const unsigned char*>(src_host);
  unsigned char* dst = static_cast<unsigned char*>(dst_dev);
  for (size_t off = 0; off < nbytes; off += srgan::kUploadChunk) {
    const size_t n = std::min<size_t>(srgan::kUploadChunk, nbytes - off);
    srgan::UploadBlob blob;
    std::memcpy(blob.w, src + off, n);
    hipLaunchKernelGGL(srgan::upload_small_kernel, dim3(1), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<unsigned int*>(dst + off), blob, (int)(n / 4));
  }
  return check_launch("upload_small_kernel");
}

extern "C" size_t srgan_adam_state_bytes(void) { return sizeof(srgan::AdamState); }

extern "C" int srgan_adam_state_init(void* state, float lr, float beta1, float beta2, float eps, int steps_done, void* stream) {
  SRGAN_REQUIRE(state && steps_done >= 0, "adam_state_init: bad argument");
  hipLaunchKernelGGL(srgan::adam_state_init_kernel, dim3(1), dim3(1), 0, as_stream(stream),
                     static_cast<srgan::AdamState*>(state), steps_done, lr, beta1, beta2, eps);
  return check_launch("adam_state_init_kernel");
}

extern "C" int srgan_adam_state_set_lr(void* state, float lr, void* stream) {
  SRGAN_REQUIRE(state, "adam_state_set_lr: bad argument");
  hipLaunchKernelGGL(srgan::adam_state_lr_kernel, dim3(1), dim3(1), 0, as_stream(stream), static_cast<srgan::AdamState*>(state), lr);
  return check_launch("adam_state_lr_kernel");
}

extern "C" int srgan_adam_multi_dev(const void* table, int n_tensors, long long max_numel, void* state, void* stream) {
  SRGAN_REQUIRE(table && state && n_tensors > 0 && max_numel > 0, "adam_multi_dev: bad argument");
  hipLaunchKernelGGL(srgan::adam_tick_kernel, dim3(1), dim3(1), 0, as_stream(stream), static_cast<srgan::AdamState*>(state));
  const unsigned bx = (unsigned)std::max<long long>(1, std::min<long long>(ceil_div(max_numel, 4096), 2048));
  hipLaunchKernelGGL(srgan::adam_multi_dev_kernel, dim3(bx, (unsigned)n_tensors), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const unsigned long long*>(table), static_cast<const srgan::AdamState*>(state));
  return check_launch("adam_multi_dev_kernel");
}

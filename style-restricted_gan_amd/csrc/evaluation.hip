// Evaluation path of the reference (pyfiles/evaluation.py:13-110): the pieces of the VGG19-bn feature
// extractor that the train-step kernels do not already provide (2x2 max pooling) and PRDC -- precision / recall / density /
// coverage of Naeem et al. 2020 as computed by prdc==0.2's compute_prdc (Docker/requirements.txt:13; the package is a
// third-party dependency absent from /root/reference: its published algorithm is restated in oracle/evaluation.py).
// Everything here is HBM / LDS bound fp32 work; the distance matrix is accumulated as sum (x - y)^2 (not |x|^2 + |y|^2 - 2xy:
// no cancellation for near neighbours, self-distances exactly 0), the set statistics are integer counts.
#include <algorithm>
#include "common.h"

namespace srgan {

// ---- MaxPool2d(2, 2), NHWC, 4 channels per thread ----
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C4, int Ho, int Wo) {
  const long long total = (long long)N * Ho * Wo * C4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long long r = i / C4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const f32x4* b = reinterpret_cast<const f32x4*>(x) + ((size_t)(n * H + oy * 2) * W + ox * 2) * C4 + c;
    const f32x4 v0 = b[0], v1 = b[C4], v2 = b[(size_t)W * C4], v3 = b[(size_t)W * C4 + C4];
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(v0[e], v1[e]), fmaxf(v2[e], v3[e]));
    reinterpret_cast<f32x4*>(y)[i] = m;
  }
}

// ---- pairwise Euclidean distances: dist[i][j] = sqrt(sum_d (x[i][d] - y[j][d])^2) ----
// workgroup = 64 x 64 tile of the matrix, thread = 4 x 4; the feature axis goes through LDS in chunks of 16, stored
// transposed ([k][row], row stride 68 floats: the float4 a thread reads for its 4 rows / columns is one LDS access).
constexpr int PD_T = 64, PD_K = 16, PD_LD = PD_T + 4;
__global__ __launch_bounds__(256) void pairwise_dist_kernel(const float* __restrict__ x, int N, const float* __restrict__ y, int M,
                                                            int D, float* __restrict__ dist) {
  __shared__ __attribute__((aligned(16))) float xs[PD_K][PD_LD], ys[PD_K][PD_LD];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int i0 = blockIdx.y * PD_T, j0 = blockIdx.x * PD_T;
  const int lr = tid >> 2, lk = (tid & 3) * 4;                // load role: row lr of the tile, features lk .. lk + 3 of the chunk
  const bool xv = i0 + lr < N, yv = j0 + lr < M;
  const float* xp = x + (size_t)(xv ? i0 + lr : 0) * D + lk;
  const float* yp = y + (size_t)(yv ? j0 + lr : 0) * D + lk;
  // sum over the feature axis: 16 terms per chunk in a fresh register, the chunk sums added with Kahan compensation -- the
  // package computes its distances in float64 and rounds once; a plain fp32 running sum over 4096 terms is ~2e-6 off, this
  // stays within ~2 ulp, so that "d < radius" decisions agree except on genuine fp32 ties
  float acc[4][4], comp[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) { acc[a][b] = 0.f; comp[a][b] = 0.f; }
  for (int k0 = 0; k0 < D; k0 += PD_K) {
    float xa[4], ya[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool kv = k0 + lk + e < D;
      xa[e] = (xv && kv) ? xp[k0 + e] : 0.f;
      ya[e] = (yv && kv) ? yp[k0 + e] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) { xs[lk + e][lr] = xa[e]; ys[lk + e][lr] = ya[e]; }
    __syncthreads();
    float part[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) part[p][q] = 0.f;
#pragma unroll
    for (int k = 0; k < PD_K; ++k) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&xs[k][ty * 4]);
      const f32x4 b = *reinterpret_cast<const f32x4*>(&ys[k][tx * 4]);
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float d = a[p] - b[q]; part[p][q] = fmaf(d, d, part[p][q]); }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float yk = part[p][q] - comp[p][q];
        const float t = acc[p][q] + yk;
        comp[p][q] = (t - acc[p][q]) - yk;
        acc[p][q] = t;
      }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int i = i0 + ty * 4 + p;
    if (i >= N) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = j0 + tx * 4 + q;
      if (j < M) dist[(size_t)i * M + j] = sqrtf(acc[p][q]);
    }
  }
}

// ---- k-th smallest value of every row (1-based k <= 16), one wave per row: prdc's get_kth_value ----
constexpr int KTH_MAX = 16;
__global__ __launch_bounds__(256) void kth_smallest_rows_kernel(const float* __restrict__ d, int N, int M, int k, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* r = d + (size_t)row * M;
  float best[KTH_MAX];                      // ascending; +inf = empty
#pragma unroll
  for (int e = 0; e < KTH_MAX; ++e) best[e] = __builtin_inff();
  for (int j = lane; j < M; j += 64) {
    float v = r[j];
#pragma unroll
    for (int e = 0; e < KTH_MAX; ++e) {       // insertion into the sorted list (only the first k slots matter)
      if (e < k) { const float lo = fminf(best[e], v); v = fmaxf(best[e], v); best[e] = lo; }
    }
  }
  // k rounds: the wave's minimum over the lanes' heads; the lane that owns it (lowest lane on ties) pops its list
  float kth = __builtin_inff();
  for (int round = 0; round < k; ++round) {
    const float head = best[0];
    const float mn = -wave_max(-head);
    const unsigned long long owners = __ballot(head == mn);
    const int owner = owners ? __ffsll((long long)owners) - 1 : 0;
    if (lane == owner) {
#pragma unroll
      for (int e = 0; e + 1 < KTH_MAX; ++e) best[e] = best[e + 1];
      best[KTH_MAX - 1] = __builtin_inff();
    }
    kth = mn;
  }
  if (lane == 0) out[row] = kth;
}

// ---- PRDC set statistics from dist[real i][fake j] and the two radius vectors ----
// rows: recall flag (any_j d < r_fake[j]) and coverage flag (min_j d < r_real[i]); one wave per real sample
__global__ __launch_bounds__(256) void prdc_rows_kernel(const float* __restrict__ d, int N, int M, const float* __restrict__ r_real,
                                                        const float* __restrict__ r_fake, int* __restrict__ recall, int* __restrict__ cover) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const float* r = d + (size_t)i * M;
  float mn = __builtin_inff();
  int any = 0;
  for (int j = lane; j < M; j += 64) {
    const float v = r[j];
    mn = fminf(mn, v);
    any |= v < r_fake[j];
  }
  mn = -wave_max(-mn);
  const unsigned long long b = __ballot(any != 0);
  if (lane == 0) { recall[i] = b != 0; cover[i] = mn < r_real[i]; }
}
// columns: cnt[j] = #{i : d[i][j] < r_real[i]}; thread = fake sample, rows streamed (coalesced across the wave)
__global__ __launch_bounds__(256) void prdc_cols_kernel(const float* __restrict__ d, int N, int M, const float* __restrict__ r_real,
                                                        int* __restrict__ cnt) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  int c = 0;
  for (int i = 0; i < N; ++i) c += d[(size_t)i * M + j] < r_real[i];
  cnt[j] = c;
}
// out = {precision, recall, density, coverage}: means of the flags / counts in double, one workgroup, fixed order
__global__ __launch_bounds__(256) void prdc_final_kernel(const int* __restrict__ recall, const int* __restrict__ cover,
                                                         const int* __restrict__ cnt, int N, int M, int nearest_k, float* __restrict__ out) {
  __shared__ long long red[4][256];
  long long s_rec = 0, s_cov = 0, s_any = 0, s_cnt = 0;
  for (int i = threadIdx.x; i < N; i += 256) { s_rec += recall[i]; s_cov += cover[i]; }
  for (int j = threadIdx.x; j < M; j += 256) { s_any += cnt[j] > 0; s_cnt += cnt[j]; }
  red[0][threadIdx.x] = s_any; red[1][threadIdx.x] = s_rec; red[2][threadIdx.x] = s_cnt; red[3][threadIdx.x] = s_cov;
  __syncthreads();
  if (threadIdx.x < 4) {
    long long t = 0;
    for (int e = 0; e < 256; ++e) t += red[threadIdx.x][e];
    double v;
    if (threadIdx.x == 0) v = (double)t / M;                                  // precision: mean over the fake samples
    else if (threadIdx.x == 1) v = (double)t / N;                             // recall: mean over the real samples
    else if (threadIdx.x == 2) v = (1.0 / (double)nearest_k) * ((double)t / M);   // density
    else v = (double)t / N;                                                   // coverage
    out[threadIdx.x] = (float)v;
  }
}

}  // namespace srgan

using namespace srgan;

extern "C" int srgan_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  SRGAN_REQUIRE(x && y && N > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0, "maxpool2_fwd: bad argument (channels must be a multiple of 4)");
  const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>(ceil_div(total, 256), 16384));
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, y, N, H, W, C / 4, H / 2, W / 2);
  return check_launch("maxpool2_fwd_kernel");
}

extern "C" int srgan_pairwise_dist(const float* x, int N, const float* y, int M, int D, float* dist, void* stream) {
  SRGAN_REQUIRE(x && y && dist && N > 0 && M > 0 && D > 0, "pairwise_dist: bad argument");
  hipLaunchKernelGGL(pairwise_dist_kernel, dim3((unsigned)ceil_div(M, PD_T), (unsigned)ceil_div(N, PD_T)), dim3(256), 0,
                     as_stream(stream), x, N, y, M, D, dist);
  return check_launch("pairwise_dist_kernel");
}

extern "C" int srgan_kth_smallest_rows(const float* dist, int N, int M, int k, float* out, void* stream) {
  SRGAN_REQUIRE(dist && out && N > 0 && M > 0, "kth_smallest_rows: bad argument");
  SRGAN_REQUIRE(k >= 1 && k <= KTH_MAX && k <= M, "kth_smallest_rows: k = %d must lie in 1..min(%d, row length %d)", k, KTH_MAX, M);
  hipLaunchKernelGGL(kth_smallest_rows_kernel, dim3((unsigned)ceil_div(N, 4)), dim3(256), 0, as_stream(stream), dist, N, M, k, out);
  return check_launch("kth_smallest_rows_kernel");
}

extern "C" size_t srgan_prdc_workspace(int N, int M) { return (size_t)(2 * (size_t)std::max(N, 0) + (size_t)std::max(M, 0)) * sizeof(int); }

extern "C" int srgan_prdc_from_dist(const float* dist, int N, int M, const float* r_real, const float* r_fake, int nearest_k,
                                    float* out4, void* ws, size_t ws_bytes, void* stream) {
  SRGAN_REQUIRE(dist && r_real && r_fake && out4 && ws && N > 0 && M > 0 && nearest_k >= 1, "prdc_from_dist: bad argument");
  SRGAN_REQUIRE(ws_bytes >= srgan_prdc_workspace(N, M), "prdc_from_dist: workspace too small");
  int* recall = static_cast<int*>(ws);
  int* cover = recall + N;
  int* cnt = cover + N;
  hipLaunchKernelGGL(prdc_rows_kernel, dim3((unsigned)ceil_div(N, 4)), dim3(256), 0, as_stream(stream), dist, N, M, r_real, r_fake, recall, cover);
  hipLaunchKernelGGL(prdc_cols_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, as_stream(stream), dist, N, M, r_real, cnt);
  hipLaunchKernelGGL(prdc_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), recall, cover, cnt, N, M, nearest_k, out4);
  return check_launch("prdc kernels");
}

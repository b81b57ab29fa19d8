// Fused loss reductions of the SRGAN train step: every kernel produces the loss value AND the
// gradient w.r.t. its input in one launch (wavefront-shuffle reductions, 64-lane waves).
//
//   mse_const       get_loss_D  (pyfiles/util.py:457-462, nn.MSELoss against a constant)
//   softmax_mse     nn.Softmax(dim=1) + get_domainloss_D (model.py:333-346, util.py:464-468)
//   l1_mean         torch.mean(torch.abs(a-b)) (util_notebook.py:625,639,676,686)
//   latent_losses   batch-KL (util_notebook.py:644-650), correlation (util.py:470-517),
//                   histogram imitation (util.py:521-553); closed forms: SURVEY.md Appendix F.2-F.4
#include <algorithm>
#include <cmath>
#include "common.h"

namespace srgan {

__global__ __launch_bounds__(256) void mse_const_kernel(const float* o, long long n, float target, float weight,
                                                        float* loss, float* d_o) {
  __shared__ float red[16];
  float s = 0.f;
  const float gscale = 2.f * weight / (float)n;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const float d = o[i] - target;
    s += d * d;
    if (d_o) d_o[i] = gscale * d;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = weight * s / (float)n;
}

__global__ __launch_bounds__(256) void softmax_mse_kernel(const float* z, const long long* label, int B, int nc,
                                                          float weight, float* q, float* loss, float* dz) {
  __shared__ float red[16];
  float s = 0.f;
  const float gscale = 2.f * weight / (float)(B * nc);
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float zz[16], qq[16];
    float mx = -INFINITY;
    for (int j = 0; j < nc; ++j) { zz[j] = z[b * nc + j]; mx = fmaxf(mx, zz[j]); }
    float den = 0.f;
    for (int j = 0; j < nc; ++j) { qq[j] = expf(zz[j] - mx); den += qq[j]; }
    const int lab = (int)label[b];
    float dot = 0.f;
    float dq[16];
    for (int j = 0; j < nc; ++j) {
      qq[j] /= den;
      const float d = qq[j] - (j == lab ? 1.f : 0.f);
      s += d * d;
      dq[j] = gscale * d;
      dot += qq[j] * dq[j];
    }
    for (int j = 0; j < nc; ++j) {
      if (q) q[b * nc + j] = qq[j];
      if (dz) dz[b * nc + j] = qq[j] * (dq[j] - dot);
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = weight * s / (float)(B * nc);
}

// All losses of one discriminator evaluation in ONE launch (K12 of SURVEY.md 2.2: util.py:457-468 applied by
// util_notebook.py:582-590 and :622-624): per scale s an LSGAN map o_s [rows][per_s] and class logits z_s [rows][nc]; the
// first `rows_first` rows are compared with the constant t_first and carry the class loss against `label`, the remaining rows
// (the translated half of a real | fake batch) with t_rest.  vals = {lsgan_first, class, lsgan_rest, total}, each the MEAN over
// scales of the per-scale nn.MSELoss; total = lsgan_first + w_class * class + lsgan_rest; d_o_s / dz_s = d total / d (.).
struct DLossParams {
  const float* o[4];
  const float* z[4];
  float* d_o[4];
  float* dz[4];
  long long per[4];
  const long long* label;
  float* vals;
  int n_scales, rows, rows_first, nc;
  float t_first, t_rest, w_class;
};

__global__ __launch_bounds__(256) void d_losses_kernel(DLossParams p) {
  __shared__ float red[16];
  const float ws = 1.f / (float)p.n_scales;
  float first = 0.f, rest = 0.f, cls = 0.f;
  for (int s = 0; s < p.n_scales; ++s) {
    const long long n1 = (long long)p.rows_first * p.per[s], n2 = (long long)(p.rows - p.rows_first) * p.per[s];
    const float g1 = n1 > 0 ? 2.f * ws / (float)n1 : 0.f, g2 = n2 > 0 ? 2.f * ws / (float)n2 : 0.f;
    float s1 = 0.f, s2 = 0.f;
    for (long long i = threadIdx.x; i < n1 + n2; i += blockDim.x) {
      const bool f = i < n1;
      const float d = p.o[s][i] - (f ? p.t_first : p.t_rest);
      if (f) s1 += d * d; else s2 += d * d;
      p.d_o[s][i] = (f ? g1 : g2) * d;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (n1 > 0) first += ws * s1 / (float)n1;
    if (n2 > 0) rest += ws * s2 / (float)n2;
    if (p.z[s]) {
      const int nc = p.nc;
      const float gs = 2.f * ws * p.w_class / (float)(p.rows_first * nc);
      float sc = 0.f;
      for (int b = threadIdx.x; b < p.rows; b += blockDim.x) {
        if (b >= p.rows_first) {
          for (int j = 0; j < nc; ++j) p.dz[s][b * nc + j] = 0.f;
          continue;
        }
        float qq[16], dq[16];
        float mx = -INFINITY;
        for (int j = 0; j < nc; ++j) mx = fmaxf(mx, p.z[s][b * nc + j]);
        float den = 0.f;
        for (int j = 0; j < nc; ++j) { qq[j] = expf(p.z[s][b * nc + j] - mx); den += qq[j]; }
        const int lab = (int)p.label[b];
        float dot = 0.f;
        for (int j = 0; j < nc; ++j) {
          qq[j] /= den;
          const float d = qq[j] - (j == lab ? 1.f : 0.f);
          sc += d * d;
          dq[j] = gs * d;
          dot += qq[j] * dq[j];
        }
        for (int j = 0; j < nc; ++j) p.dz[s][b * nc + j] = qq[j] * (dq[j] - dot);
      }
      sc = block_sum(sc, red);
      cls += ws * sc / (float)(p.rows_first * nc);
    }
  }
  if (threadIdx.x == 0) {
    p.vals[0] = first;
    p.vals[1] = cls;
    p.vals[2] = rest;
    p.vals[3] = first + cls * p.w_class + rest;
  }
}

// out = sum_i w_i * x_i over n <= 16 scalars living anywhere on the device (the loss terms of a phase); dx_i = w_i * g in the
// backward (one launch each way instead of one per python-level `+` / `*`).
struct LincombParams {
  const float* x[16];
  float w[16];
  int n;
};
__global__ void lincomb_fwd_kernel(LincombParams p, float* out) {
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < p.n; ++i) s += p.w[i] * p.x[i][0];
    out[0] = s;
  }
}
__global__ void lincomb_bwd_kernel(LincombParams p, const float* g, float* dx /* [n] */) {
  if ((int)threadIdx.x < p.n) dx[threadIdx.x] = p.w[threadIdx.x] * g[0];
}

// nn.CrossEntropyLoss (mean): loss = mean_b( logsumexp(z_b) - z_b[label_b] ), dz = (softmax(z) - onehot) * weight / B
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* z, const long long* label, int B, int nc,
                                                           float weight, float* loss, float* dz) {
  __shared__ float red[16];
  float s = 0.f;
  const float gscale = weight / (float)B;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float mx = -INFINITY;
    for (int j = 0; j < nc; ++j) mx = fmaxf(mx, z[b * nc + j]);
    float den = 0.f;
    for (int j = 0; j < nc; ++j) den += expf(z[b * nc + j] - mx);
    const int lab = (int)label[b];
    s += logf(den) + mx - z[b * nc + lab];
    if (dz)
      for (int j = 0; j < nc; ++j) dz[b * nc + j] = gscale * (expf(z[b * nc + j] - mx) / den - (j == lab ? 1.f : 0.f));
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = weight * s / (float)B;
}

__global__ __launch_bounds__(256) void l1_partial_kernel(const float* a, const float* b, long long n, float gscale,
                                                         float* part, float* da, float* db) {
  __shared__ float red[16];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    s += fabsf(d);
    const float g = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
    if (da) da[i] = g;
    if (db) db[i] = -g;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void l1_final_kernel(const float* part, int nparts, float scale, float* loss) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = s * scale;
}

// mean((a-b)^2) * weight between two tensors (generic nn.MSELoss(a, b)); single block (small inputs)
__global__ __launch_bounds__(256) void mse_pair_kernel(const float* a, const float* b, long long n, float weight,
                                                       float* loss, float* da, float* db) {
  __shared__ float red[16];
  float s = 0.f;
  const float gscale = 2.f * weight / (float)n;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const float d = a[i] - b[i];
    s += d * d;
    if (da) da[i] = gscale * d;
    if (db) db[i] = -gscale * d;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = weight * s / (float)n;
}

// Conventional KL of N(mu, exp(logvar)) against N(0, I), a SUM over all rows and dimensions (util_notebook.py:630-634, :302):
//   L = -0.5 * sum(1 + logvar - mu^2 - exp(logvar));  dL/dmu = mu;  dL/dlogvar = 0.5 * (exp(logvar) - 1)   (times weight)
// value + both gradients in one launch; single block (B x ndim values)
__global__ __launch_bounds__(256) void kl_normal_kernel(const float* mu, const float* logvar, long long n, float weight,
                                                        float* loss, float* dmu, float* dlogvar) {
  __shared__ float red[16];
  float s = 0.f;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const float m = mu[i], lv = logvar[i], ev = expf(lv);
    s += 1.f + lv - m * m - ev;
    if (dmu) dmu[i] = weight * m;
    if (dlogvar) dlogvar[i] = weight * 0.5f * (ev - 1.f);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = -0.5f * weight * s;
}

// GaussianHistogram.forward (util.py:532-537) for a long 1-D sample: part[block][bin], then summed.
__global__ __launch_bounds__(256) void soft_hist_partial_kernel(const float* x, long long n, int bins, float lo,
                                                                float delta, float sigma, float* part) {
  __shared__ float red[16];
  const float knorm = delta / (sigma * 2.5066282746310002f);
  const float inv2s2 = 0.5f / (sigma * sigma);
  for (int k = 0; k < bins; ++k) {
    const float ck = lo + delta * ((float)k + 0.5f);
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
      const float d = x[i] - ck;
      s += expf(-d * d * inv2s2);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * bins + k] = s * knorm;
  }
}
__global__ void soft_hist_final_kernel(const float* part, int nparts, int bins, float* h) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= bins) return;
  float s = 0.f;
  for (int i = 0; i < nparts; ++i) s += part[(size_t)i * bins + k];
  h[k] = s;
}
// dx_i = sum_k g_k * dh_k/dx_i
__global__ void soft_hist_bwd_kernel(const float* x, const float* g, long long n, int bins, float lo, float delta,
                                     float sigma, float* dx) {
  const float knorm = delta / (sigma * 2.5066282746310002f);
  const float inv2s2 = 0.5f / (sigma * sigma), invs2 = 1.f / (sigma * sigma);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < bins; ++k) {
      const float d = x[i] - (lo + delta * ((float)k + 0.5f));
      s += g[k] * (-knorm * expf(-d * d * inv2s2) * d * invs2);
    }
    dx[i] = s;
  }
}

// ---- latent losses: one workgroup, mu staged in LDS ---------------------------------------
constexpr int LAT_MAX_D = 16, LAT_MAX_BINS = 64, LAT_MAX_ELEMS = 16384;

__global__ __launch_bounds__(256) void latent_losses_kernel(const float* mu_g, int B, int d, float n_batch,
                                                            const float* target, int bins, float range_max, float sigma,
                                                            float w_bkl, float w_corr, float w_hist, float* vals,
                                                            float* dmu, float* corr_out) {
  extern __shared__ float smem[];
  float* mu = smem;                        // [B][d]
  float* mean = mu + B * d;                // [d]
  float* var_u = mean + LAT_MAX_D;         // [d] unbiased variance
  float* cov = var_u + LAT_MAX_D;          // [d][d]
  float* dC = cov + LAT_MAX_D * LAT_MAX_D; // [d][d] dL/dC (symmetrised use)
  float* h = dC + LAT_MAX_D * LAT_MAX_D;   // [d][bins]
  float* dh = h + LAT_MAX_D * LAT_MAX_BINS;// [d][bins]
  float* scal = dh + LAT_MAX_D * LAT_MAX_BINS;  // misc scalars: [0]=bkl [1]=corr [2]=hist, [8+j]=S_j
  const int tid = threadIdx.x, nt = blockDim.x;

  for (int i = tid; i < B * d; i += nt) mu[i] = mu_g[i];
  if (tid < 32) scal[tid] = 0.f;
  __syncthreads();
  // means
  if (tid < d) {
    float s = 0.f;
    for (int i = 0; i < B; ++i) s += mu[i * d + tid];
    mean[tid] = s / (float)B;
  }
  __syncthreads();
  // covariance (unnormalised by B-1 applied here)
  for (int p = tid; p < d * d; p += nt) {
    const int a = p / d, b = p - a * d;
    float s = 0.f;
    for (int i = 0; i < B; ++i) s += (mu[i * d + a] - mean[a]) * (mu[i * d + b] - mean[b]);
    cov[a * LAT_MAX_D + b] = s / (float)(B - 1);
  }
  __syncthreads();
  if (tid < d) var_u[tid] = cov[tid * LAT_MAX_D + tid];
  __syncthreads();
  const float sfac = n_batch / (n_batch - 1.f);
  // batch-KL value
  if (tid == 0) {
    float L = 0.f;
    for (int j = 0; j < d; ++j) {
      const float v = sfac * var_u[j];
      L += 1.f + logf(v) - mean[j] * mean[j] - v;
    }
    scal[0] = -0.5f * L;
  }
  // correlation: R, G = sign(R - I)/(d(d-1)) off-diagonal (0 where clamp active), dL/dC
  if (tid == 64) {
    float R[LAT_MAX_D][LAT_MAX_D], G[LAT_MAX_D][LAT_MAX_D], sd[LAT_MAX_D];
    const float norm = 1.f / (float)(d * (d - 1));
    for (int a = 0; a < d; ++a) sd[a] = sqrtf(var_u[a]);
    float L = 0.f;
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) {
        float r = cov[a * LAT_MAX_D + b] / sd[b] / sd[a];
        const bool clamped = r > 1.f || r < -1.f;
        r = fminf(1.f, fmaxf(-1.f, r));
        R[a][b] = r;
        if (corr_out) corr_out[a * d + b] = r;
        const float e = r - (a == b ? 1.f : 0.f);
        L += fabsf(e);
        float g = e > 0.f ? norm : (e < 0.f ? -norm : 0.f);
        if (clamped || a == b) g = 0.f;   // diagonal is identically 1: analytic derivative 0
        G[a][b] = g;
      }
    scal[1] = L * norm;
    for (int a = 0; a < d; ++a) {
      float rs = 0.f, cs = 0.f;
      for (int b = 0; b < d; ++b) { rs += G[a][b] * R[a][b]; cs += G[b][a] * R[b][a]; }
      for (int b = 0; b < d; ++b) {
        float v = G[a][b] / (sd[a] * sd[b]);
        if (a == b) v -= (rs + cs) / (2.f * var_u[a]);
        dC[a * LAT_MAX_D + b] = v;
      }
    }
  }
  // histograms h[j][k]
  const float delta = 2.f * range_max / (float)bins;
  const float knorm = delta / (sigma * 2.5066282746310002f);
  const float inv2s2 = 0.5f / (sigma * sigma);
  for (int p = tid; p < d * bins; p += nt) {
    const int j = p / bins, k = p - j * bins;
    const float ck = -range_max + delta * ((float)k + 0.5f);
    float s = 0.f;
    for (int i = 0; i < B; ++i) {
      const float dd = mu[i * d + j] - ck;
      s += expf(-dd * dd * inv2s2);
    }
    h[j * LAT_MAX_BINS + k] = s * knorm;
  }
  __syncthreads();
  // per-dim normaliser, KL value and dL/dh
  if (tid < d) {
    const int j = tid;
    float S = 0.f;
    for (int k = 0; k < bins; ++k) S += h[j * LAT_MAX_BINS + k];
    float L = 0.f, corr = 0.f;
    for (int k = 0; k < bins; ++k) {
      const float p = h[j * LAT_MAX_BINS + k] / S + 1e-8f;
      const float t = target[k];
      L += t * (logf(t) - logf(p));
      corr += t * h[j * LAT_MAX_BINS + k] / (p * S * S);
    }
    for (int k = 0; k < bins; ++k) {
      const float p = h[j * LAT_MAX_BINS + k] / S + 1e-8f;
      dh[j * LAT_MAX_BINS + k] = -target[k] / (p * S) + corr;
    }
    scal[8 + j] = L;
  }
  __syncthreads();
  if (tid == 0) {
    float L = 0.f;
    for (int j = 0; j < d; ++j) L += scal[8 + j];
    vals[0] = scal[0]; vals[1] = scal[1]; vals[2] = L;
    vals[3] = w_bkl * scal[0] + w_corr * scal[1] + w_hist * L;
  }
  if (!dmu) return;
  // gradient per element
  for (int p = tid; p < B * d; p += nt) {
    const int i = p / d, j = p - i * d;
    const float x = mu[p];
    // batch-KL: m_j/B - (1/v_j - 1) * s * (x - m_j)/(B-1)
    const float v = sfac * var_u[j];
    const float gb = mean[j] / (float)B - (1.f / v - 1.f) * sfac * (x - mean[j]) / (float)(B - 1);
    // correlation: dL/dxm_{j,i} = sum_b (dC[j][b] + dC[b][j]) xm_{b,i} / (B-1); the row mean of this
    // over i is zero (sum_i xm = 0), so the mean-removal Jacobian changes nothing.
    float gc = 0.f;
    for (int b = 0; b < d; ++b) gc += (dC[j * LAT_MAX_D + b] + dC[b * LAT_MAX_D + j]) * (mu[i * d + b] - mean[b]);
    gc /= (float)(B - 1);
    // histogram: sum_k dh_k * (-kappa_ki * d_ki / sigma^2)
    float gh = 0.f;
    for (int k = 0; k < bins; ++k) {
      const float ck = -range_max + delta * ((float)k + 0.5f);
      const float dd = x - ck;
      gh += dh[j * LAT_MAX_BINS + k] * (-knorm * expf(-dd * dd * inv2s2) * dd / (sigma * sigma));
    }
    dmu[p] = w_bkl * gb + w_corr * gc + w_hist * gh;
  }
}

}  // namespace srgan

using namespace srgan;

extern "C" int srgan_mse_const(const float* o, long long n, float target, float weight, float* loss, float* d_o, void* stream) {
  SRGAN_REQUIRE(o && loss && n > 0, "mse_const: bad argument");
  hipLaunchKernelGGL(mse_const_kernel, dim3(1), dim3(256), 0, as_stream(stream), o, n, target, weight, loss, d_o);
  return check_launch("mse_const_kernel");
}

extern "C" int srgan_softmax_mse(const float* z, const long long* label, int B, int n_class, float weight, float* q,
                                 float* loss, float* dz, void* stream) {
  SRGAN_REQUIRE(z && label && loss && B > 0 && n_class > 0 && n_class <= 16, "softmax_mse: bad argument (n_class<=16)");
  hipLaunchKernelGGL(softmax_mse_kernel, dim3(1), dim3(256), 0, as_stream(stream), z, label, B, n_class, weight, q, loss, dz);
  return check_launch("softmax_mse_kernel");
}

extern "C" size_t srgan_l1_workspace(long long n) {
  (void)n;
  return 1024 * sizeof(float);
}

extern "C" int srgan_l1_mean(const float* a, const float* b, long long n, float weight, float* loss, float* da, float* db,
                             void* ws, size_t ws_bytes, void* stream) {
  SRGAN_REQUIRE(a && b && loss && n > 0, "l1_mean: bad argument");
  SRGAN_REQUIRE(ws && ws_bytes >= 1024 * sizeof(float), "l1_mean: workspace too small");
  hipStream_t st = as_stream(stream);
  const int blocks = (int)std::max<long long>(1, std::min<long long>(1024, ceil_div(n, 256 * 8)));
  float* part = (float*)ws;
  hipLaunchKernelGGL(l1_partial_kernel, dim3(blocks), dim3(256), 0, st, a, b, n, weight / (float)n, part, da, db);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, st, (const float*)part, blocks, weight / (float)n, loss);
  return check_launch("l1_mean");
}

extern "C" int srgan_latent_losses(const float* mu, int B, int d, float n_batch, const float* hist_target, int bins,
                                   float range_max, float sigma, float w_bkl, float w_corr, float w_hist, float* vals,
                                   float* dmu, float* corr_out, void* stream) {
  SRGAN_REQUIRE(mu && hist_target && vals, "latent_losses: null pointer");
  SRGAN_REQUIRE(B >= 2 && d >= 2 && d <= LAT_MAX_D && bins >= 1 && bins <= LAT_MAX_BINS && (long long)B * d <= LAT_MAX_ELEMS,
                "latent_losses: need 2<=B, 2<=d<=16, bins<=64, B*d<=16384");
  const size_t shmem = ((size_t)B * d + 2 * LAT_MAX_D + 2 * LAT_MAX_D * LAT_MAX_D + 2 * LAT_MAX_D * LAT_MAX_BINS + 64) * sizeof(float);
  hipLaunchKernelGGL(latent_losses_kernel, dim3(1), dim3(256), shmem, as_stream(stream), mu, B, d, n_batch, hist_target, bins,
                     range_max, sigma, w_bkl, w_corr, w_hist, vals, dmu, corr_out);
  return check_launch("latent_losses_kernel");
}

extern "C" int srgan_mse_pair(const float* a, const float* b, long long n, float weight, float* loss, float* da, float* db,
                              void* stream) {
  SRGAN_REQUIRE(a && b && loss && n > 0, "mse_pair: bad argument");
  hipLaunchKernelGGL(mse_pair_kernel, dim3(1), dim3(256), 0, as_stream(stream), a, b, n, weight, loss, da, db);
  return check_launch("mse_pair_kernel");
}

extern "C" int srgan_d_losses(const float* const* o, const long long* per_row, const float* const* z, int n_scales, int rows,
                              int rows_first, int n_class, const long long* label, float t_first, float t_rest, float w_class,
                              float* vals, float* const* d_o, float* const* dz, void* stream) {
  SRGAN_REQUIRE(o && per_row && d_o && vals && n_scales >= 1 && n_scales <= 4, "d_losses: 1..4 scales");
  SRGAN_REQUIRE(rows > 0 && rows_first >= 0 && rows_first <= rows, "d_losses: bad row split");
  DLossParams p{};
  p.n_scales = n_scales; p.rows = rows; p.rows_first = rows_first; p.nc = n_class;
  p.t_first = t_first; p.t_rest = t_rest; p.w_class = w_class; p.label = label; p.vals = vals;
  for (int s = 0; s < n_scales; ++s) {
    SRGAN_REQUIRE(o[s] && d_o[s] && per_row[s] > 0, "d_losses: null scale");
    p.o[s] = o[s]; p.d_o[s] = d_o[s]; p.per[s] = per_row[s];
    p.z[s] = z ? z[s] : nullptr;
    p.dz[s] = dz ? dz[s] : nullptr;
    SRGAN_REQUIRE(!p.z[s] || (p.dz[s] && label && rows_first > 0 && n_class > 0 && n_class <= 16), "d_losses: class head needs dz, labels, <= 16 classes");
  }
  hipLaunchKernelGGL(d_losses_kernel, dim3(1), dim3(256), 0, as_stream(stream), p);
  return check_launch("d_losses_kernel");
}

extern "C" int srgan_lincomb(const float* const* x, const float* w, int n, float* out, void* stream) {
  SRGAN_REQUIRE(x && w && out && n >= 1 && n <= 16, "lincomb: 1..16 terms");
  LincombParams p{};
  p.n = n;
  for (int i = 0; i < n; ++i) { SRGAN_REQUIRE(x[i], "lincomb: null term"); p.x[i] = x[i]; p.w[i] = w[i]; }
  hipLaunchKernelGGL(lincomb_fwd_kernel, dim3(1), dim3(64), 0, as_stream(stream), p, out);
  return check_launch("lincomb_fwd_kernel");
}

extern "C" int srgan_lincomb_bwd(const float* w, int n, const float* g, float* dx, void* stream) {
  SRGAN_REQUIRE(w && g && dx && n >= 1 && n <= 16, "lincomb_bwd: 1..16 terms");
  LincombParams p{};
  p.n = n;
  for (int i = 0; i < n; ++i) p.w[i] = w[i];
  hipLaunchKernelGGL(lincomb_bwd_kernel, dim3(1), dim3(64), 0, as_stream(stream), p, g, dx);
  return check_launch("lincomb_bwd_kernel");
}

extern "C" int srgan_kl_normal(const float* mu, const float* logvar, long long n, float weight, float* loss, float* dmu,
                               float* dlogvar, void* stream) {
  SRGAN_REQUIRE(mu && logvar && loss && n > 0, "kl_normal: bad argument");
  hipLaunchKernelGGL(kl_normal_kernel, dim3(1), dim3(256), 0, as_stream(stream), mu, logvar, n, weight, loss, dmu, dlogvar);
  return check_launch("kl_normal_kernel");
}

extern "C" size_t srgan_soft_histogram_workspace(long long n, int bins) {
  (void)n;
  return (size_t)256 * bins * sizeof(float);
}

extern "C" int srgan_soft_histogram_fwd(const float* x, long long n, int bins, float lo, float hi, float sigma, float* h,
                                        void* ws, size_t ws_bytes, void* stream) {
  SRGAN_REQUIRE(x && h && n > 0 && bins > 0 && hi > lo && sigma > 0.f, "soft_histogram_fwd: bad argument");
  SRGAN_REQUIRE(ws && ws_bytes >= (size_t)256 * bins * sizeof(float), "soft_histogram_fwd: workspace too small");
  hipStream_t st = as_stream(stream);
  const int blocks = (int)std::max<long long>(1, std::min<long long>(256, ceil_div(n, 256)));
  const float delta = (hi - lo) / (float)bins;
  hipLaunchKernelGGL(soft_hist_partial_kernel, dim3(blocks), dim3(256), 0, st, x, n, bins, lo, delta, sigma, (float*)ws);
  hipLaunchKernelGGL(soft_hist_final_kernel, dim3((bins + 63) / 64), dim3(64), 0, st, (const float*)ws, blocks, bins, h);
  return check_launch("soft_histogram_fwd");
}

extern "C" int srgan_soft_histogram_bwd(const float* x, const float* g, long long n, int bins, float lo, float hi, float sigma,
                                        float* dx, void* stream) {
  SRGAN_REQUIRE(x && g && dx && n > 0 && bins > 0 && hi > lo && sigma > 0.f, "soft_histogram_bwd: bad argument");
  const float delta = (hi - lo) / (float)bins;
  const int blocks = (int)std::max<long long>(1, std::min<long long>(1024, ceil_div(n, 256)));
  hipLaunchKernelGGL(soft_hist_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, g, n, bins, lo, delta, sigma, dx);
  return check_launch("soft_histogram_bwd");
}

extern "C" int srgan_softmax_xent(const float* z, const long long* label, int B, int n_class, float weight, float* loss,
                                  float* dz, void* stream) {
  SRGAN_REQUIRE(z && label && loss && B > 0 && n_class > 0 && n_class <= 1024, "softmax_xent: bad argument");
  hipLaunchKernelGGL(softmax_xent_kernel, dim3(1), dim3(256), 0, as_stream(stream), z, label, B, n_class, weight, loss, dz);
  return check_launch("softmax_xent_kernel");
}

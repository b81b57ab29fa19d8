// Weight-repack bodies shared by the single-layer pack kernels and the multi-layer pack launch (one launch repacks every
// cached operand of a network after its optimiser step).
#pragma once
#include "common.h"

namespace srgan {

// dst[phase][n][ (ty*Tx+tx)*Cs + c ], zero padded to [Npad][Kpad].
//   mode 0: n = O index, c = I index, (ky,kx) = (ty,tx)
//   mode 1: n = I index, c = O index, ky = (py+pad)%s + s*ty (zero if >= kh)
struct PackParams {
  const float* w;
  float* dst;
  long long sO, sI, sH, sW;
  int O, I, kh, kw, mode, stride, pad, Ty, Tx, Cs, N, K, Kpad, Npad, phases;
  int out16;           // 1: the operand is written as bf16 (RNE) at the same element index (igemm16_kernel's layers, bf16 mode)
  int regimg;          // round 6, with out16: the register image of halo16e_kernel instead of [n][k] (3x3 taps, Cs % 64 == 0,
                       // N % 64 == 0, one phase): element (n, k = tap * Cs + c) -> [K step (c / 64 * 9 + tap) * 4 + c / 16 % 4]
                       // [64-channel block n / 64][32-channel half][lane = 32 (c / 8 % 2) + n % 32][c % 8]
};

__device__ __forceinline__ void pack_weights_store(const PackParams& p, long long idx, float v) {
  if (p.regimg) {
    const int k = (int)(idx % p.Kpad), n = (int)(idx / p.Kpad);
    const int t = k / p.Cs, c = k - t * p.Cs;
    const long long ks = (long long)((c >> 6) * 9 + t) * 4 + ((c >> 4) & 3);
    const long long e = (((ks * (p.Npad >> 6) + (n >> 6)) * 2 + ((n >> 5) & 1)) * 64 + ((c >> 3) & 1) * 32 + (n & 31)) * 8 + (c & 7);
    reinterpret_cast<__bf16*>(p.dst)[e] = (__bf16)v;
    return;
  }
  if (p.out16) reinterpret_cast<__bf16*>(p.dst)[idx] = (__bf16)v;
  else p.dst[idx] = v;
}

__device__ __forceinline__ long long pack_weights_total(const PackParams& p) { return (long long)p.phases * p.Npad * p.Kpad; }

__device__ __forceinline__ void pack_weights_item(const PackParams& p, long long idx) {
  const int k = (int)(idx % p.Kpad);
  const long long rest = idx / p.Kpad;
  const int n = (int)(rest % p.Npad);
  const int phase = (int)(rest / p.Npad);
  float v = 0.f;
  if (n < p.N && k < p.K) {
    const int t = k / p.Cs, c = k - t * p.Cs;
    const int ty = t / p.Tx, tx = t - ty * p.Tx;
    if (p.mode == 0) {
      v = p.w[n * p.sO + c * p.sI + ty * p.sH + tx * p.sW];
    } else {
      const int py = phase / p.stride, px = phase % p.stride;
      const int ky = (py + p.pad) % p.stride + p.stride * ty;
      const int kx = (px + p.pad) % p.stride + p.stride * tx;
      if (ky < p.kh && kx < p.kw) v = p.w[c * p.sO + n * p.sI + ky * p.sH + kx * p.sW];
    }
  }
  pack_weights_store(p, idx, v);
}

// The same rows for one WORKGROUP of 256 threads: one (phase, n) row of K = taps x Cs entries at a time.  With OIHW weights the
// row's source is Cs runs of `taps` contiguous floats (mode 0: ONE contiguous run of Cs * taps floats), read tap-fastest --
// lanes walk along memory -- into LDS [tap][Cs + 1] and written back channel-fastest, the destination order: both sides
// coalesced, where the one-thread-per-entry loop reads 4 bytes every taps * 4 bytes (36-64 B apart) and walks the weight tensor
// `taps` times.  Pure data movement: identical bytes.  PACK_LDS_FLOATS bounds taps * (Cs + 1).
constexpr int PACK_LDS_FLOATS = 12288;
__device__ __host__ __forceinline__ bool pack_weights_block_ok(const PackParams& p) {
  return (long long)p.Ty * p.Tx * (p.Cs + 1) <= PACK_LDS_FLOATS && p.Cs >= 8;
}
__device__ __forceinline__ void pack_weights_rows(const PackParams& p, float* tile) {
  const int taps = p.Ty * p.Tx, LD = p.Cs + 1;
  const long long rows = (long long)p.phases * p.Npad;
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = (int)(row % p.Npad), phase = (int)(row / p.Npad);
    const long long dst0 = row * p.Kpad;
    if (n >= p.N) {
      for (int k = threadIdx.x; k < p.Kpad; k += blockDim.x) pack_weights_store(p, dst0 + k, 0.f);
      continue;
    }
    __syncthreads();                      // the previous row has left the tile
    for (int i = threadIdx.x; i < taps * p.Cs; i += blockDim.x) {
      const int c = i / taps, t = i - c * taps;
      const int ty = t / p.Tx, tx = t - ty * p.Tx;
      float v = 0.f;
      if (p.mode == 0) {
        v = p.w[n * p.sO + c * p.sI + ty * p.sH + tx * p.sW];
      } else {
        const int py = phase / p.stride, px = phase % p.stride;
        const int ky = (py + p.pad) % p.stride + p.stride * ty;
        const int kx = (px + p.pad) % p.stride + p.stride * tx;
        if (ky < p.kh && kx < p.kw) v = p.w[c * p.sO + n * p.sI + ky * p.sH + kx * p.sW];
      }
      tile[t * LD + c] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < p.Kpad; k += blockDim.x) {
      float v = 0.f;
      if (k < p.K) {
        const int t = k / p.Cs, c = k - t * p.Cs;
        v = tile[t * LD + c];
      }
      pack_weights_store(p, dst0 + k, v);
    }
  }
}

struct WinoPackParams {
  const float* w;
  float* dst;
  long long sO, sI, sH, sW;
  int N, C, kind, nchunk, n_tiles, variant, phases;
  int no_block;        // 1: never take the workgroup-cooperative path (SRGAN_PACK_ITEM_PATH, A/B)
};


constexpr int WP_WNB = 64, WP_WC = 8;
__device__ __host__ __forceinline__ long long wino_pack_total(const WinoPackParams& p) {
  if (p.variant == 5 || p.variant == 6) return (long long)p.N * (p.C / 8);         // one item = 8 reduce channels of one output channel, 16 taps
  if (p.variant == 4) return (long long)p.nchunk * p.N * 4;                      // one item = 8 reduce channels of one cout, 9 taps
  if (p.variant == 3) return (long long)p.n_tiles * 32 * p.nchunk * 2;          // one item = 4 channels of one cout
  if (p.variant == 7) return (long long)p.phases * p.n_tiles * p.nchunk * 256;  // one item = 4 channels of one cout
  return (long long)p.n_tiles * WP_WNB * p.nchunk * WP_WC * p.phases;
}

// variant 3, F(4x4,3x3):  U = G g G^T (6x6), laid out [n_tile (32 couts)][chunk][36 pos][lane = half * 32 + cout][4 ch]:
// the A-operand register image of wino43_kernel.  kind 0: g = w[n][c][ky][kx];  kind 1: g = w[c][n][2-ky][2-kx].
__device__ __forceinline__ void wino43_pack_item(const WinoPackParams& p, long long idx) {
  // item = (cout nl, channel quad lh, chunk, cout tile): the lane's 4 channels are 16 contiguous bytes of every position
  // plane, so the 36 results go out as 16-byte stores (a quarter of the store instructions of a one-channel item)
  long long r = idx;
  const int nl = (int)(r % 32); r /= 32;
  const int lh = (int)(r % 2); r /= 2;
  const int chunk = (int)(r % p.nchunk);
  const int ntile = (int)(r / p.nchunk);
  const int n = ntile * 32 + nl;
  f32x4 u[36];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = chunk * WP_WC + lh * 4 + j;
    const bool ok = n < p.N && c < p.C;
    float g[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        float v = 0.f;
        if (ok) v = p.kind == 0 ? p.w[n * p.sO + c * p.sI + ky * p.sH + kx * p.sW]
                                : p.w[c * p.sO + n * p.sI + (2 - ky) * p.sH + (2 - kx) * p.sW];
        g[ky][kx] = v;
      }
    // G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
    float h[6][3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float a = g[0][kx], b = g[1][kx], cc = g[2][kx];
      h[0][kx] = 0.25f * a;
      h[1][kx] = (-1.f / 6.f) * (a + b + cc);
      h[2][kx] = (-1.f / 6.f) * (a - b + cc);
      h[3][kx] = (1.f / 24.f) * a + (1.f / 12.f) * b + (1.f / 6.f) * cc;
      h[4][kx] = (1.f / 24.f) * a - (1.f / 12.f) * b + (1.f / 6.f) * cc;
      h[5][kx] = cc;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const float x = h[a][0], y = h[a][1], z = h[a][2];
      u[a * 6 + 0][j] = 0.25f * x;
      u[a * 6 + 1][j] = (-1.f / 6.f) * (x + y + z);
      u[a * 6 + 2][j] = (-1.f / 6.f) * (x - y + z);
      u[a * 6 + 3][j] = (1.f / 24.f) * x + (1.f / 12.f) * y + (1.f / 6.f) * z;
      u[a * 6 + 4][j] = (1.f / 24.f) * x - (1.f / 12.f) * y + (1.f / 6.f) * z;
      u[a * 6 + 5][j] = z;
    }
  }
  float* out = p.dst + (((size_t)ntile * p.nchunk + chunk) * 36) * 256 + lh * 128 + nl * 4;
#pragma unroll
  for (int k = 0; k < 36; ++k) *reinterpret_cast<f32x4*>(out + k * 256) = u[k];
}

// The same items for one WORKGROUP of 256 threads = 256 consecutive items = 32 couts x 32 reduce channels of one cout tile (four
// consecutive chunks): with dense OIHW weights that block's source is 32 rows of 288 contiguous floats (kind 0: row = cout, the
// 32 channels x 9 taps behind it; kind 1: row = reduce channel (the O index of w), the 32 couts (its I index) x 9 taps), so it is
// read with nine coalesced 16-byte loads per thread into LDS (row stride 289: the item's reads are conflict-free) instead of 36
// dword loads per thread that each touch 64 different lines; results leave per filter row (24 registers in flight, no spill).
// Measured on the generator's repack (24 F(4x4,3x3) images per optimiser step): see DESIGN.md section 4.
constexpr int WP43_ROW = 289;
__device__ __host__ __forceinline__ bool wino43_pack_block_ok(const WinoPackParams& p) {
  return p.variant == 3 && !p.no_block && p.sW == 1 && p.sH == 3 && p.sI == 9 && p.sO % 4 == 0 && p.nchunk % 4 == 0 && p.N % 32 == 0 && p.C % 32 == 0 &&
         (reinterpret_cast<unsigned long long>(p.w) & 15) == 0;
}
__device__ __forceinline__ void wino43_pack_block(const WinoPackParams& p, long long base, float* tile /* [32][WP43_ROW] */) {
  const int t = threadIdx.x;
  long long r = base / 64;                              // (chunk, ntile) of the block's first item
  const int chunk0 = (int)(r % p.nchunk);
  const int ntile = (int)(r / p.nchunk);
  const int n0 = ntile * 32, c0 = chunk0 * WP_WC;
  __syncthreads();                                      // the previous block of this workgroup is done with the tile
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int f = t + 256 * i, row = f / 72, q = f - row * 72;
    const float* src = p.kind == 0 ? p.w + (long long)(n0 + row) * p.sO + c0 * 9 : p.w + (long long)(c0 + row) * p.sO + n0 * 9;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + q * 4);
    float* dst = tile + row * WP43_ROW + q * 4;
    dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
  }
  __syncthreads();
  const int nl = t & 31, lh = (t >> 5) & 1, cl = t >> 6;          // item = (cout nl, channel quad lh, chunk chunk0 + cl)
  float h[4][6][3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = cl * WP_WC + lh * 4 + j;                         // channel inside the block
    const float* g = p.kind == 0 ? tile + nl * WP43_ROW + c * 9 : tile + c * WP43_ROW + nl * 9;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      // kind 1: the filter rotated by 180 degrees
      const float a = p.kind == 0 ? g[0 * 3 + kx] : g[8 - (0 * 3 + kx)], b = p.kind == 0 ? g[1 * 3 + kx] : g[8 - (1 * 3 + kx)],
                  cc = p.kind == 0 ? g[2 * 3 + kx] : g[8 - (2 * 3 + kx)];
      h[j][0][kx] = 0.25f * a;
      h[j][1][kx] = (-1.f / 6.f) * (a + b + cc);
      h[j][2][kx] = (-1.f / 6.f) * (a - b + cc);
      h[j][3][kx] = (1.f / 24.f) * a + (1.f / 12.f) * b + (1.f / 6.f) * cc;
      h[j][4][kx] = (1.f / 24.f) * a - (1.f / 12.f) * b + (1.f / 6.f) * cc;
      h[j][5][kx] = cc;
    }
  }
  float* out = p.dst + (((size_t)ntile * p.nchunk + chunk0 + cl) * 36) * 256 + lh * 128 + nl * 4;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    f32x4 u[6];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = h[j][a][0], y = h[j][a][1], z = h[j][a][2];
      u[0][j] = 0.25f * x;
      u[1][j] = (-1.f / 6.f) * (x + y + z);
      u[2][j] = (-1.f / 6.f) * (x - y + z);
      u[3][j] = (1.f / 24.f) * x + (1.f / 12.f) * y + (1.f / 6.f) * z;
      u[4][j] = (1.f / 24.f) * x - (1.f / 12.f) * y + (1.f / 6.f) * z;
      u[5][j] = z;
    }
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<f32x4*>(out + (a * 6 + b) * 256) = u[b];
  }
}

// variant 4 (conv_halo16.hip, bf16 mode -- no transform): bf16 [64-channel quarter][tap][32-chunk of the quarter][N][32], rounded
// to nearest even (K tile kt = (quarter * 9 + tap) * 2 + chunk & 1); N = 256: the register image of halo16r_kernel (below).
// kind 0: B[n][k] = w[n][k][ky][kx];  kind 1: B[n][k] = w[k][n][2-ky][2-kx].
// One item = (output channel n, 8 reduce channels) for ALL NINE taps: with dense OIHW weights its 72 source values are 288
// contiguous bytes (kind 0) or eight 36-byte runs (kind 1); an item per tap would walk the whole weight tensor nine times with
// a 36-byte stride (measured: 145 us per repack of the generator, 12 layers x 2 kinds).
__device__ __forceinline__ void halo16_pack_item(const WinoPackParams& p, long long idx) {
  long long r = idx;
  const int part = (int)(r % 4); r /= 4;
  const int n = (int)(r % p.N);
  const int chunk = (int)(r / p.N);                     // 0 .. nchunk - 1 (32 reduce channels each)
  const int quarter = chunk >> 1;
  float v[8][9];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = chunk * 32 + part * 8 + j;
    const float* src = p.kind == 0 ? p.w + n * p.sO + k * p.sI : p.w + k * p.sO + n * p.sI;
#pragma unroll
    for (int t = 0; t < 9; ++t) v[j][t] = src[(t / 3) * p.sH + (t % 3) * p.sW];
  }
  unsigned short* dst = reinterpret_cast<unsigned short*>(p.dst);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int t = p.kind == 0 ? tap : 8 - tap;          // kind 1: taps rotated by 180 degrees
    const f32x4 lo = {v[0][t], v[1][t], v[2][t], v[3][t]}, hi = {v[4][t], v[5][t], v[6][t], v[7][t]};
    const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
    bf16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    if (p.N == 256) {
      // round 6, halo16r_kernel (the 256-channel trunk): the REGISTER image of the filter operand -- one 16-byte piece per lane
      // and B fragment, [K step ks = (quarter * 9 + tap) * 4 + s (16 reduce channels)][wave = n / 64][n block of 32][lane = 32 h
      // + n % 32][8 channels 16 s + 8 h ..], so that a wave's fragment load is 1 KB of contiguous global memory
      const int s4 = (chunk & 1) * 2 + (part >> 1), h = part & 1;
      const size_t ks = (size_t)(quarter * 9 + tap) * 4 + s4;
      *reinterpret_cast<bf16x8*>(dst + ((((ks * 4 + (n >> 6)) * 2 + ((n >> 5) & 1)) * 64 + h * 32 + (n & 31)) * 8)) = o;
      continue;
    }
    const int kt = (quarter * 9 + tap) * 2 + (chunk & 1);
    *reinterpret_cast<bf16x8*>(dst + ((size_t)kt * p.N + n) * 32 + part * 8) = o;
  }
}

// variant 5 (conv_halo16.hip, transposed 4x4 / stride-2 form, kind 1 of a strided layer w[O][I][4][4]): bf16
// [phase (r,s)][tap (a,b)][chunk][N = I][HK = 8192 / N reduce channels o],  B[n][k] = w[k][n][3 - 2a - r][3 - 2b - s].
// One item = (n, 8 reduce channels) for all 16 (phase, tap) pairs = all 16 filter taps: eight 64-byte source runs.
__device__ __forceinline__ void halo16t_pack_item(const WinoPackParams& p, long long idx) {
  const int hk = 8192 / p.N, pcs = hk / 8, nch = p.C / hk;
  long long r = idx;
  const int part = (int)(r % pcs); r /= pcs;
  const int n = (int)(r % p.N);
  const int chunk = (int)(r / p.N);
  float v[8][16];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = chunk * hk + part * 8 + j;
    const float* src = p.w + k * p.sO + n * p.sI;
#pragma unroll
    for (int t = 0; t < 16; ++t) v[j][t] = src[(t >> 2) * p.sH + (t & 3) * p.sW];
  }
  unsigned short* dst = reinterpret_cast<unsigned short*>(p.dst);
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
      const int ky = 3 - 2 * (tap >> 1) - (ph >> 1), kx = 3 - 2 * (tap & 1) - (ph & 1), t = ky * 4 + kx;
      const f32x4 lo = {v[0][t], v[1][t], v[2][t], v[3][t]}, hi = {v[4][t], v[5][t], v[6][t], v[7][t]};
      const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
      bf16x8 o;
      o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
      const int kt = (ph * 4 + tap) * nch + chunk;
      *reinterpret_cast<bf16x8*>(dst + ((size_t)kt * p.N + n) * hk + part * 8) = o;
    }
}

// variant 6 (conv_halo16.hip, strided 4x4 / stride-2 form, kind 0 of w[O][I][4][4]): bf16
// [64-channel half][tap (ky, kx)][chunk of the half][N = O][HK = 8192 / N reduce channels c],  B[n][k] = w[n][k][ky][kx].
// One item = (n, 8 reduce channels) for all 16 taps: 512 contiguous source bytes with dense OIHW weights.
__device__ __forceinline__ void halo16s_pack_item(const WinoPackParams& p, long long idx) {
  const int hk = 8192 / p.N, pcs = hk / 8, nchk = 64 / hk;
  long long r = idx;
  const int part = (int)(r % pcs); r /= pcs;
  const int n = (int)(r % p.N);
  const int chunkg = (int)(r / p.N);                    // chunk of hk channels over all of C
  const int half = chunkg / nchk, chunk = chunkg - half * nchk;
  float v[8][16];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = chunkg * hk + part * 8 + j;
    const float* src = p.w + n * p.sO + k * p.sI;
#pragma unroll
    for (int t = 0; t < 16; ++t) v[j][t] = src[(t >> 2) * p.sH + (t & 3) * p.sW];
  }
  unsigned short* dst = reinterpret_cast<unsigned short*>(p.dst);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const f32x4 lo = {v[0][t], v[1][t], v[2][t], v[3][t]}, hi = {v[4][t], v[5][t], v[6][t], v[7][t]};
    const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
    bf16x8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    const int kt = (half * 16 + t) * nchk + chunk;
    *reinterpret_cast<bf16x8*>(dst + ((size_t)kt * p.N + n) * hk + part * 8) = o;
  }
}

// variant 7, F(4x4,2x2) (conv_wino42.hip):  U = G g G^T (5x5 from the 2x2 taps of a phase), G = [1/2 0; -1/2 -1/2; -1/6 1/6;
// 1/6 1/3; 0 1], laid out [out phase][n_tile (64 couts)][chunk (16 ch)][25 pos][8-ch half][cout half][lane = lh * 32 + cout][4 ch]:
// the A-operand register image of wino42_kernel (position 24 in the image of the 16x16x4 MFMA, see the end of the function).  kind 0 (strided form): reduce index k = (input phase pq, c), g[a][b] =
// w[n][c][2a+p][2b+q];  kind 1 (transposed form): one image per OUTPUT phase rs, g[a][b] = w[c][n][3-2a-r][3-2b-s].
__device__ __forceinline__ void wino42_pack_item(const WinoPackParams& p, long long idx) {
  long long r = idx;
  const int nl = (int)(r % 32); r /= 32;
  const int lh = (int)(r % 2); r /= 2;
  const int h = (int)(r % 2); r /= 2;
  const int hb = (int)(r % 2); r /= 2;
  const int chunk = (int)(r % p.nchunk); r /= p.nchunk;
  const int ntile = (int)(r % p.n_tiles);
  const int ophase = (int)(r / p.n_tiles);
  const int n = ntile * 64 + h * 32 + nl;
  f32x4 u[25];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = chunk * 16 + hb * 8 + lh * 4 + j;
    int c, pp, qq;
    if (p.kind == 0) { const int ph = k / p.C; c = k - ph * p.C; pp = ph >> 1; qq = ph & 1; }
    else { c = k; pp = ophase >> 1; qq = ophase & 1; }
    const bool ok = n < p.N && c < p.C && (p.kind == 1 || k < 4 * p.C);
    float g[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float v = 0.f;
        if (ok) v = p.kind == 0 ? p.w[n * p.sO + c * p.sI + (2 * a + pp) * p.sH + (2 * b + qq) * p.sW]
                                : p.w[c * p.sO + n * p.sI + (3 - 2 * a - pp) * p.sH + (3 - 2 * b - qq) * p.sW];
        g[a][b] = v;
      }
    float hm[5][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float x = g[0][b], y = g[1][b];
      hm[0][b] = 0.5f * x;
      hm[1][b] = -0.5f * (x + y);
      hm[2][b] = (1.f / 6.f) * (y - x);
      hm[3][b] = (1.f / 6.f) * x + (1.f / 3.f) * y;
      hm[4][b] = y;
    }
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const float x = hm[a][0], y = hm[a][1];
      u[a * 5 + 0][j] = 0.5f * x;
      u[a * 5 + 1][j] = -0.5f * (x + y);
      u[a * 5 + 2][j] = (1.f / 6.f) * (y - x);
      u[a * 5 + 3][j] = (1.f / 6.f) * x + (1.f / 3.f) * y;
      u[a * 5 + 4][j] = y;
    }
  }
  float* img = p.dst + ((((size_t)ophase * p.n_tiles + ntile) * p.nchunk + chunk) * 25) * 1024;
  float* out = img + hb * 512 + h * 256 + lh * 128 + nl * 4;
#pragma unroll
  for (int k = 0; k < 24; ++k) *reinterpret_cast<f32x4*>(out + k * 1024) = u[k];
  // position 24 is multiplied on the 16x16x4 MFMA: [16-channel block of the 64][lane = (channel & 3) * 16 + cout & 15][channel quad]
  float* o24 = img + 24 * 1024 + (2 * h + (nl >> 4)) * 256 + (nl & 15) * 4 + (2 * hb + lh);
#pragma unroll
  for (int j = 0; j < 4; ++j) o24[j * 64] = u[24][j];
}

__device__ __forceinline__ void wino_pack_item(const WinoPackParams& p, long long idx) {
  constexpr int WNB = WP_WNB, WC = WP_WC;
  if (p.variant == 7) { wino42_pack_item(p, idx); return; }
  if (p.variant == 6) { halo16s_pack_item(p, idx); return; }
  if (p.variant == 5) { halo16t_pack_item(p, idx); return; }
  if (p.variant == 4) { halo16_pack_item(p, idx); return; }
  if (p.variant == 3) { wino43_pack_item(p, idx); return; }
    // lanes run over (channel & 3, cout, channel half): the 16 stores of a wave-instruction are 256 contiguous bytes each
    const int c4 = (int)(idx % 4);
    long long r = idx / 4;
    const int nl = (int)(r % WNB); r /= WNB;
    const int cc = (int)(r % 2) * 4 + c4; r /= 2;
    const int chunk = (int)(r % p.nchunk); r /= p.nchunk;
    const int ntile = (int)(r % p.n_tiles);
    const int ophase = (int)(r / p.n_tiles);
    const int n = ntile * WNB + nl;
    float u[4][4];
    if (p.variant == 1) {
      const int c = chunk * WC + cc;
      float g[3][3];
      const bool ok = n < p.N && c < p.C;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float v = 0.f;
          if (ok) v = p.kind == 0 ? p.w[n * p.sO + c * p.sI + ky * p.sH + kx * p.sW]
                                  : p.w[c * p.sO + n * p.sI + (2 - ky) * p.sH + (2 - kx) * p.sW];
          g[ky][kx] = v;
        }
      // G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
      float h[4][3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        h[0][kx] = g[0][kx];
        h[1][kx] = 0.5f * (g[0][kx] + g[1][kx] + g[2][kx]);
        h[2][kx] = 0.5f * (g[0][kx] - g[1][kx] + g[2][kx]);
        h[3][kx] = g[2][kx];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        u[a][0] = h[a][0];
        u[a][1] = 0.5f * (h[a][0] + h[a][1] + h[a][2]);
        u[a][2] = 0.5f * (h[a][0] - h[a][1] + h[a][2]);
        u[a][3] = h[a][2];
      }
    } else {
      const int k = chunk * WC + cc;
      int c, pp, qq;
      if (p.kind == 0) { const int ph = k / p.C; c = k - ph * p.C; pp = ph >> 1; qq = ph & 1; }
      else { c = k; pp = ophase >> 1; qq = ophase & 1; }
      const bool ok = n < p.N && c < p.C && (p.kind == 1 || k < 4 * p.C);
      float g[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float v = 0.f;
          if (ok) v = p.kind == 0 ? p.w[n * p.sO + c * p.sI + (2 * a + pp) * p.sH + (2 * b + qq) * p.sW]
                                  : p.w[c * p.sO + n * p.sI + (3 - 2 * a - pp) * p.sH + (3 - 2 * b - qq) * p.sW];
          g[a][b] = v;
        }
      // A g A^T, A = [1 0; 1 1; 1 -1; 0 -1]
      float h[4][2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        h[0][b] = g[0][b];
        h[1][b] = g[0][b] + g[1][b];
        h[2][b] = g[0][b] - g[1][b];
        h[3][b] = -g[1][b];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        u[a][0] = h[a][0];
        u[a][1] = h[a][0] + h[a][1];
        u[a][2] = h[a][0] - h[a][1];
        u[a][3] = -h[a][1];
      }
    }
    // within a position: [channel half][cout][4 channels] -- the register image of the kernel
    float* out = p.dst + ((((size_t)ophase * p.n_tiles + ntile) * p.nchunk + chunk) * 16) * (WNB * WC) + (cc >> 2) * (WNB * 4) +
                 nl * 4 + (cc & 3);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) out[(a * 4 + b) * (WNB * WC)] = u[a][b];
}


// one record of the multi-layer launch (host-built, copied to the device as an array)
struct PackEntry {
  int type;            // 0: implicit-GEMM operand, 1: Winograd filter transform
  int reserved;
  PackParams ig;
  WinoPackParams wn;
};

// conv_wino.hip
void wino_pack_params(const srgan_conv_desc* d, int kind, const float* w, float* dst, WinoPackParams* out);

}  // namespace srgan

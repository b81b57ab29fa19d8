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
//   workgroup (8 waves) = 16 rows x 64 columns of one image; lane = pixel column, wave w = rows 2w, 2w + 1 (2 accumulators of 4
//   registers);
//   the reduce channels are walked in chunks of 16: the (16 + 6) x (64 + 6) halo of a chunk sits in LDS as four channel-quad
//   planes of 16 bytes per pixel (98.6 KB; consecutive lanes read consecutive 16-byte slots: conflict-free ds_read_b128), so the A
//   operands of the four instructions (pixel, tap, c .. c + 3) are one read, and a pixel's 64 contiguous bytes are fetched by four
//   adjacent lanes (a first version with one-quad chunks asked the texture path for 64 different lines per load instruction --
//   16 bytes of each -- and spent as long on its halo as on its products: 194 -> 147 us without those loads);
//   the whole packed filter (7 x 7 x C x 4 floats = 50 KB at C = 64) sits in LDS as [ky][quad][kx][cout][4 c]; an item =
//   (quad, kx): seven filter reads + eight halo-row reads feed 56 instructions, ONE read of halo row r serving the (output row,
//   ky) pairs with row + ky = r; two register sets, the next item's reads fenced in front of the current item's products;
//   the next chunk's halo is loaded into registers (buffer loads: the range check supplies the zero padding) under the current
//   chunk's 1568 instructions per wave and stored between two barriers (single LDS buffer).
// 128x128 maps at batch 32 are 512 workgroups: two rounds.  Measured (scratch/rgbout_phases.py, -DRGBOUT_EXP=8 stamps): prologue
// 5-6.6 us, 18 us per chunk against 14 for the instructions alone at the clock the chip holds under them; forward at batch 32
// 172 us against 237 for the row convolution + shift-add, +0.6 % on the step.
#include <algorithm>
#include <cstdlib>
#include "common.h"

#ifndef RGBOUT_EXP
#define RGBOUT_EXP 0
#endif

namespace srgan {
namespace {

constexpr int RO_TR = 16, RO_TC = 64, RO_K = 7, RO_PAD = 3, RO_T = 2;   // tile, taps, output rows per wave
constexpr int RO_HR = RO_TR + RO_K - 1, RO_HC = RO_TC + RO_K - 1;      // 22 x 70 halo pixels
constexpr int RO_HALO = RO_HR * RO_HC;                                  // 1540 pixels = float4 slots per quad plane
constexpr int RO_CQ = 4;                                                // channel quads per chunk (16 channels)
constexpr int RO_LOADS = (RO_HALO * RO_CQ + 511) / 512;                 // (pixel, quad) items per thread and chunk (13)
constexpr int RO_MAXQ = 16;                                             // up to 64 reduce channels (the filter must fit LDS)

struct RgboutParams {
  const float* x;      // [NB][H][W][C]
  const float* wp;     // [ky][quad][kx][4 couts][4 c]
  const float* bias;   // [O] or null
  float* y;            // [NB][Ho][Wo][O]
  int NB, H, W, C, Ho, Wo, O, ncq, tiles_x, tiles_y;
};

constexpr unsigned kRoOutside = 0x80000000u;

__device__ __forceinline__ auto ro_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

__global__ __launch_bounds__(512) void rgbout_conv_kernel(RgboutParams p) {
  __shared__ f32x4 halo[RO_CQ * RO_HALO];                               // [4 quads][22 x 70 pixels] x 16 bytes = 98.6 KB
  __shared__ __attribute__((aligned(16))) float wl[RO_K * RO_MAXQ * RO_K * 16];      // [ky][quad][kx][4 couts][4 c]: 50 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int n = b / p.tiles_y;
  const int X0 = tx * RO_TC, Y0 = ty * RO_TR;

#if RGBOUT_EXP & 8      // phase stamps (100 MHz wall clock) instead of the output: scratch/rgbout_phases.py
  const long long t_start = wall_clock64();
  long long t_pro = 0, t_chunk[4] = {0, 0, 0, 0};
#endif
  // the packed filter: all of a thread's loads are issued (with the first halo chunk's, below) before the first LDS store -- a
  // load -> store loop paid one memory latency per trip and the prologue took 8 us of a workgroup's 78
  const int nw = RO_K * p.ncq * RO_K * 4;                               // float4 count of the packed filter
  constexpr int RO_WLOADS = (RO_K * RO_MAXQ * RO_K * 4 + 511) / 512;
  const auto rs_w = ro_rsrc(p.wp, (unsigned)(nw * sizeof(f32x4)));
  f32x4 wst[RO_WLOADS];
#pragma unroll
  for (int j = 0; j < RO_WLOADS; ++j)
    wst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)((tid + 512 * j) * sizeof(f32x4)), 0, 0));

  // this thread's (halo pixel, quad) items: element offsets of chunk 0, or -1 outside the image (zero padding); the four quads
  // of a pixel are four adjacent lanes = 64 contiguous bytes
  // byte offsets for buffer loads (rgbout_applicable: the input is under 2^31 bytes); kRoOutside is beyond every range the
  // resource describes, so the hardware returns the zero padding and the loads need neither pointers nor branches
  const auto rs_x = ro_rsrc(p.x, (unsigned)((size_t)p.NB * p.H * p.W * p.C * sizeof(float)));
  unsigned hoff[RO_LOADS];
#pragma unroll
  for (int j = 0; j < RO_LOADS; ++j) {
    const int it = tid + 512 * j;
    const int hp = it >> 2, q = it & 3;
    const int r = hp / RO_HC, c = hp - r * RO_HC;
    const int gy = Y0 - RO_PAD + r, gx = X0 - RO_PAD + c;
    const bool ok = hp < RO_HALO && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    hoff[j] = ok ? (unsigned)((((n * p.H + gy) * p.W + gx) * p.C + q * 4) * (int)sizeof(float)) : kRoOutside;
  }
  f32x4 stage[RO_LOADS];
  auto load_chunk = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < RO_LOADS; ++j)
      stage[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, hoff[j], ch * (RO_CQ * 4 * (int)sizeof(float)), 0));
  };
  auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < RO_LOADS; ++j) {
      const int it = tid + 512 * j;
      if ((it >> 2) < RO_HALO) halo[(it & 3) * RO_HALO + (it >> 2)] = stage[j];
    }
  };

  // one accumulator per (output row, channel of the quad): 8 independent chains, so an instruction never waits for the result of
  // one issued fewer than 8 instructions earlier (with one accumulator per row the two rows' chains alternated: 147 -> 169 us)
  f32x4 acc[RO_T][4];
#pragma unroll
  for (int t = 0; t < RO_T; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunk = p.ncq / RO_CQ;
  load_chunk(0);
#pragma unroll
  for (int j = 0; j < RO_WLOADS; ++j)
    if (tid + 512 * j < nw) reinterpret_cast<f32x4*>(wl)[tid + 512 * j] = wst[j];
  store_chunk();
  __syncthreads();
  const int co4 = (lane & 3) * 4;
#if RGBOUT_EXP & 8
  t_pro = wall_clock64();
#endif
  for (int ch = 0; ch < nchunk; ++ch) {
#if !(RGBOUT_EXP & 1)   // -DRGBOUT_EXP=<bits>: timing ablations (wrong results): 1 no in-loop halo loads, 2 no in-loop LDS reads, 4 no stores/barriers
    if (ch + 1 < nchunk) load_chunk(ch + 1);
#endif
    // (quad, kx) items of the chunk, two register sets: the 15 reads of item i + 1 are issued before the 56 instructions of item i
    // and nothing may be scheduled across the fence between them (left alone, the compiler sinks each read to just before its
    // first use and the wave waits out the LDS latency fifteen times per item: 170 us instead of ~110)
    const f32x4* Hw = halo + (RO_T * wave) * RO_HC + lane;
    const float* wq = wl + (size_t)ch * RO_CQ * RO_K * 16 + co4;
    auto fetch_b = [&](f32x4 (&B)[RO_K], int it) __attribute__((always_inline)) {
      const int q = it / RO_K, kx = it - q * RO_K;
#pragma unroll
      for (int ky = 0; ky < RO_K; ++ky) B[ky] = *reinterpret_cast<const f32x4*>(wq + ((ky * p.ncq + q) * RO_K + kx) * 16);
    };
    auto fetch_a = [&](f32x4 (&A)[RO_T + RO_K - 1], int it) __attribute__((always_inline)) {
      const int q = it / RO_K, kx = it - q * RO_K;
#pragma unroll
      for (int r = 0; r < RO_T + RO_K - 1; ++r) A[r] = Hw[q * RO_HALO + r * RO_HC + kx];
    };
    auto products = [&](const f32x4 (&B)[RO_K], const f32x4 (&A)[RO_T + RO_K - 1], int r0, int r1) __attribute__((always_inline)) {
#pragma unroll
      for (int r = r0; r < r1; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ky = 0; ky < RO_K; ++ky) {
            const int t = r - ky;
            if (t >= 0 && t < RO_T) acc[t][e] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[r][e], B[ky][e], acc[t][e], 0, 0, 0);
          }
    };
    // the next item's 7 filter reads go out before the first half of this item's products and its 8 halo reads before the second
    // half, so every wait is for reads a half-item old (s_waitcnt counts at most 15 and a merge point takes the strictest count:
    // no conditional fetch -- the last item is simply fetched twice)
    constexpr int NIT = RO_CQ * RO_K, RH = (RO_T + RO_K - 1) / 2, RE = RO_T + RO_K - 1;
    f32x4 B0[RO_K], A0[RE], B1[RO_K], A1[RE];
    fetch_b(B0, 0);
    fetch_a(A0, 0);
#pragma unroll 1
    for (int it = 0; it < NIT; it += 2) {
#if RGBOUT_EXP & 2
      if (it == 0) { fetch_b(B1, 1); fetch_a(A1, 1); }
#else
      fetch_b(B1, it + 1);
#endif
      __builtin_amdgcn_sched_barrier(0);
      products(B0, A0, 0, RH);
      __builtin_amdgcn_sched_barrier(0);
#if !(RGBOUT_EXP & 2)
      fetch_a(A1, it + 1);
#endif
      __builtin_amdgcn_sched_barrier(0);
      products(B0, A0, RH, RE);
      __builtin_amdgcn_sched_barrier(0);
      const int nx = it + 2 < NIT ? it + 2 : NIT - 1;
#if !(RGBOUT_EXP & 2)
      fetch_b(B0, nx);
#endif
      __builtin_amdgcn_sched_barrier(0);
      products(B1, A1, 0, RH);
      __builtin_amdgcn_sched_barrier(0);
#if !(RGBOUT_EXP & 2)
      fetch_a(A0, nx);
#endif
      __builtin_amdgcn_sched_barrier(0);
      products(B1, A1, RH, RE);
      __builtin_amdgcn_sched_barrier(0);
    }
#if !(RGBOUT_EXP & 4)
    __syncthreads();                       // every wave is done with this chunk's halo
    if (ch + 1 < nchunk) store_chunk();
    __syncthreads();
#endif
#if RGBOUT_EXP & 8
    if (ch < 4) t_chunk[ch] = wall_clock64();
#endif
  }
#if RGBOUT_EXP & 8
#pragma unroll
  for (int t = 0; t < RO_T; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(acc[t][e]));
  if (tid == 0) {
    long long* o = reinterpret_cast<long long*>(p.y) + (size_t)blockIdx.x * 8;
    o[0] = t_start; o[1] = t_pro; o[2] = t_chunk[0]; o[3] = t_chunk[1]; o[4] = t_chunk[2]; o[5] = t_chunk[3]; o[6] = wall_clock64();
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    o[7] = ((long long)xcc << 32) | hw;
  }
  if (p.O >= 0) return;
#endif

  // D[pixel 4b + i][cout j] sits in register i of lane 4b + j: lane = (pixel group, cout)
  const int co = lane & 3, pg = lane >> 2;
  if (co < p.O) {
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int t = 0; t < RO_T; ++t) {
      const int oy = Y0 + RO_T * wave + t;
      if (oy >= p.Ho) continue;
      float* row = p.y + (((size_t)n * p.Ho + oy) * p.Wo) * p.O + co;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ox = X0 + 4 * pg + i;
        if (ox < p.Wo) row[(size_t)ox * p.O] = ((acc[t][0][i] + acc[t][1][i]) + (acc[t][2][i] + acc[t][3][i])) + bv;
      }
    }
  }
}

// ---- bf16 compute mode: the 64 -> 3 channel head on v_mfma_f32_32x32x16_bf16 ----
// Round 4 put this layer on the bf16 instruction as D[32 pixels][3 live of 32 columns], one 8 x 32 tile per workgroup with its
// (8 + 6) x (32 + 6) halo parked once: 196 K steps per 32 pixels (~45 us of matrix time per launch at batch 32) BEHIND a halo read
// that fetched every fp32 pixel 2.1 times in one exposed round trip per tile (109 us per launch, 1.08 ms per train step).
// Round 6, two changes:
//   * the filter ROW is a column index: D[32 pixels of halo row g][(cout, ky)] = sum over (kx, c) halo[g][x + kx][c] w[cout][c][ky][kx]
//     -- 21 live columns, K = 7 x 64 = 28 steps instead of 196 -- and the output row y is the sum of seven such partial rows,
//     out[y][x][o] = sum_ky D[g = y + ky - 3][x][(o, ky)], taken from a 16-row ring in LDS in fixed order (deterministic);
//   * a workgroup walks DOWN a 32-column strip in blocks of 8 halo rows (one row per wave): every halo row is fetched once per strip
//     (vertical over-fetch 6 rows per strip instead of 6 per 8), and block b + 1 is in flight (10 x 16 bytes per thread) under the
//     products of block b.  The 28 B fragments (the whole filter) stay in registers for the life of the workgroup.
constexpr int RO16_TC = 32, RO16_BR = 8, RO16_HC = RO16_TC + 6, RO16_PS = 64 * 2 + 16;
constexpr int RO16_KSTEPS = 7 * 4;          // (kx, 16-channel chunk)
constexpr int RO16_PLD = 25;                // floats per pixel of a partial row (21 live columns; odd: conflict-free both ways)

struct Rgbout16Params {
  const void* x;             // [NB][H][W][64] fp32, or bf16 (IN16)
  const unsigned short* wp;  // [28 K steps][32 columns (cout * 7 + ky; zero from 21)][16] bf16
  const float* bias;         // [O] or null
  float* y;                  // [NB][H][W][O]
  int NB, H, W, O, strips_x, vsplit, rows_per;
  int exp;
};

// IN16: the 64-channel source is bf16 (16-bit activation storage: the generator's last norm writes it; as the input gradient of
// the RGB input layer, its first norm's backward does) -- a pixel is 8 pieces of 16 bytes that go to LDS as they are.
template <bool IN16>
__global__ __launch_bounds__(512) void rgbout16_conv_kernel(Rgbout16Params p) {
  __shared__ __attribute__((aligned(16))) unsigned char halo[RO16_BR * RO16_HC * RO16_PS];
  __shared__ float ring[16 * RO16_TC * RO16_PLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  int b = blockIdx.x;
  const int vs = b % p.vsplit; b /= p.vsplit;
  const int tx = b % p.strips_x;
  const int n = b / p.strips_x;
  const int X0 = tx * RO16_TC, Ys = vs * p.rows_per, Ye = min(Ys + p.rows_per, p.H);
  const int n_it = (Ye - Ys + 6 + RO16_BR - 1) / RO16_BR;

  // the filter: B fragment of K step ks = column lr, k = 8 lh .. 8 lh + 7
  bf16x8 fb[RO16_KSTEPS];
  {
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wp);
#pragma unroll
    for (int ks = 0; ks < RO16_KSTEPS; ++ks) fb[ks] = __builtin_bit_cast(bf16x8, wsrc[(ks * 32 + lr) * 2 + lh]);
  }

  // Halo block: thread = (pixel of the pass, 16-byte piece of its 256-byte channel row): a wave instruction covers 4 pixels x 256
  // contiguous bytes.  UNCONDITIONAL loads: an out-of-image pixel reads the image's first pixel (in bounds) and is zeroed when it
  // is parked (a load under a branch, merged with a zero, made the compiler wait for every load in turn: round 4).
  constexpr int PIECES = IN16 ? 8 : 16;               // 16-byte pieces of a pixel's 64 channels
  constexpr int NPX = RO16_BR * RO16_HC, PPP = 512 / PIECES, NPASS = (NPX + PPP - 1) / PPP;
  const int piece = tid & (PIECES - 1), hpl = tid / PIECES;
  const unsigned char* img = static_cast<const unsigned char*>(p.x) + ((size_t)n * p.H * p.W * 64) * (IN16 ? 2 : 4) + piece * 16;
  f32x4 v[NPASS];
  unsigned okm = 0;
  auto issue = [&](int G0) __attribute__((always_inline)) {
    okm = 0;
#pragma unroll
    for (int g = 0; g < NPASS; ++g) {
      const int hp = g * PPP + hpl;
      const int hr = hp / RO16_HC, hc = hp - hr * RO16_HC;
      const int gy = G0 + hr, gx = X0 - 3 + hc;
      const bool ok = hp < NPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      okm |= ok ? (1u << g) : 0u;
      v[g] = *reinterpret_cast<const f32x4*>(img + (ok ? ((size_t)gy * p.W + gx) * (IN16 ? 128 : 256) : 0));
    }
  };
  auto park = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NPASS; ++g) {
      const int hp = g * PPP + hpl;
      if (hp < NPX) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (IN16) *reinterpret_cast<f32x4*>(&halo[hp * RO16_PS + piece * 16]) = ((okm >> g) & 1u) ? v[g] : z4;
        else *reinterpret_cast<bf16x4*>(&halo[hp * RO16_PS + piece * 8]) = __builtin_convertvector(((okm >> g) & 1u) ? v[g] : z4, bf16x4);
      }
    }
  };

  // A: lane (lr, lh) = pixel column lr of the wave's halo row, k = 8 lh .. 8 lh + 7 of the step's 16 channels
  const unsigned char* a_lane = halo + (wave * RO16_HC + lr) * RO16_PS + lh * 16;
  const float bv0 = (p.bias && lh < p.O) ? p.bias[lh] : 0.f, bv2 = (p.bias && 2 < p.O) ? p.bias[2] : 0.f;

  issue(Ys - 3);
  for (int it = 0; it < n_it; ++it) {
    const int G0 = Ys - 3 + RO16_BR * it;
    park();
    __syncthreads();                       // the block is in LDS; every wave is done with the ring rows of the previous block
    if (it + 1 < n_it) issue(G0 + RO16_BR);

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    bf16x8 fa[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) fa[0][q] = *reinterpret_cast<const bf16x8*>(a_lane + q * 32);
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      if (kx + 1 < 7) {
#pragma unroll
        for (int q = 0; q < 4; ++q) fa[(kx + 1) & 1][q] = *reinterpret_cast<const bf16x8*>(a_lane + (kx + 1) * RO16_PS + q * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#ifdef SRGAN_EXPERIMENTS
        if (p.exp & 4) continue;
#endif
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kx & 1][q], fb[kx * 4 + q], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // D[pixel][column]: lane = column lr (21 live), register e = pixel (e % 4) + 8 (e / 4) + 4 lh of halo row G0 + wave
    if (lr < 21) {
      float* pr = ring + (((G0 + wave) & 15) * RO16_TC) * RO16_PLD + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) pr[((e & 3) + 8 * (e >> 2) + 4 * lh) * RO16_PLD] = acc[e];
    }
    __syncthreads();                       // the partial rows are in the ring; everyone is done with the halo block

    // output row y = G0 - 3 + wave: its seven partial rows g = y - 3 .. y + 3 are all in the ring (g <= G0 + 7)
    const int y = G0 - 3 + wave;
    if (y >= Ys && y < Ye) {
      float s01 = bv0, s2 = bv2;           // lane (x = lr, lh): cout lh, and cout 2 on the lh == 0 half
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const float* pr = ring + (((y + ky - 3) & 15) * RO16_TC + lr) * RO16_PLD;
        s01 += pr[lh * 7 + ky];
        s2 += pr[14 + ky];
      }
      const int ox = X0 + lr;
      if (ox < p.W) {
        float* dst = p.y + (((size_t)n * p.H + y) * p.W + ox) * p.O;
        if (lh < p.O) dst[lh] = s01;
        if (lh == 0 && 2 < p.O) dst[2] = s2;
      }
    }
  }
}

// bf16 packed filter [K step = kx * 4 + chunk][column = cout * 7 + ky (zero from 21 / for cout >= O)][16] = w[cout][16 chunk + k][ky][kx]
__global__ void rgbout16_pack_kernel(const float* w, unsigned short* dst, long long sO, long long sI, long long sH, long long sW, int O) {
  const int total = RO16_KSTEPS * 32 * 16;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx & 15, col = (idx >> 4) & 31, step = idx >> 9;
    const int q = step & 3, kx = step >> 2, co = col / 7, ky = col - 7 * co;
    const float v = (col < 21 && co < O) ? w[co * sO + (q * 16 + k) * sI + ky * sH + kx * sW] : 0.f;
    reinterpret_cast<__bf16*>(dst)[idx] = (__bf16)v;
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
  static const bool off = SRGAN_AB_SET("SRGAN_NO_RGBOUT");
  if (off) return false;
  return d->O >= 1 && d->O <= 4 && d->kh == RO_K && d->kw == RO_K && d->stride == 1 && d->pad == RO_PAD && d->pad_mode == SRGAN_PAD_ZERO &&
         d->I % 16 == 0 && d->I >= 16 && d->I <= 4 * RO_MAXQ && d->Wo >= 64 && d->Ho >= 32 && d->Hi == d->Ho &&
         d->Wi == d->Wo && (long long)d->N * d->Hi * d->Wi * d->I * 4 < (1LL << 31);
}

size_t rgbout_packed_elems(const srgan_conv_desc* d) { return (size_t)RO_K * (d->I / 4) * RO_K * 16; }

// bf16 mode: the 64 -> (<= 3) channel head on rgbout16_conv_kernel (its 28 KB packed filter fits the fp32 kernel's 50 KB allocation)
static bool rgbout16_mode(const srgan_conv_desc* d) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_RGBOUT16");
  return !off && compute_bf16() && d->I == 64 && d->O <= 3;
}

int rgbout_pack(const srgan_conv_desc* d, const float* w, float* dst, hipStream_t st) {
  if (rgbout16_mode(d)) {
    hipLaunchKernelGGL(rgbout16_pack_kernel, dim3(56), dim3(256), 0, st, w, reinterpret_cast<unsigned short*>(dst), d->sO, d->sI, d->sH,
                       d->sW, d->O);
    return check_launch("rgbout16_pack_kernel");
  }
  const int total = (int)rgbout_packed_elems(d);
  hipLaunchKernelGGL(rgbout_pack_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 256)), dim3(256), 0, st, w, dst, d->sO, d->sI, d->sH,
                     d->sW, d->O, d->I / 4);
  return check_launch("rgbout_pack_kernel");
}

bool rgbout16_served(const srgan_conv_desc* d) { return rgbout_applicable(d) && rgbout16_mode(d); }

int rgbout_run(const srgan_conv_desc* d, const void* xv, const float* packed, const float* bias, float* y, hipStream_t st, bool src16) {
  SRGAN_REQUIRE(rgbout_applicable(d), "rgb-output conv: layer not applicable");
  SRGAN_REQUIRE(!src16 || rgbout16_mode(d), "rgb-output conv: a bf16 source needs the bf16 compute mode's kernel");
  const float* x = static_cast<const float*>(xv);
  RgboutParams p{};
  p.x = x; p.wp = packed; p.bias = bias; p.y = y;
  p.NB = d->N; p.H = d->Hi; p.W = d->Wi; p.C = d->I; p.Ho = d->Ho; p.Wo = d->Wo; p.O = d->O; p.ncq = d->I / 4;
  p.tiles_x = (int)ceil_div(d->Wo, RO_TC); p.tiles_y = (int)ceil_div(d->Ho, RO_TR);
  const long long grid = (long long)p.tiles_x * p.tiles_y * d->N;
  SRGAN_REQUIRE(grid < (1LL << 31), "rgb-output conv: grid too large");
  ProfToken tok = prof_begin(28, 2.0 * d->N * d->Ho * d->Wo * (double)d->O * d->kh * d->kw * d->I, st);
  if (rgbout16_mode(d)) {
    Rgbout16Params q{};
    q.x = xv; q.wp = reinterpret_cast<const unsigned short*>(packed); q.bias = bias; q.y = y;
    q.NB = d->N; q.H = d->Hi; q.W = d->Wi; q.O = d->O;
    q.strips_x = (int)ceil_div(d->Wo, RO16_TC);
    // strips are cut across until the grid covers the chip (>= 256 workgroups) or a piece would fall under 16 rows
    q.vsplit = 1;
    while ((long long)d->N * q.strips_x * q.vsplit < 256 && ceil_div(d->Hi, q.vsplit * 2) >= 16) q.vsplit *= 2;
    q.rows_per = (int)ceil_div(d->Hi, q.vsplit);
    q.vsplit = (int)ceil_div(d->Hi, q.rows_per);
    q.exp = (int)SRGAN_AB_INT("SRGAN_RGBOUT16_EXP", 0);
    if (src16) hipLaunchKernelGGL(rgbout16_conv_kernel<true>, dim3((unsigned)(q.strips_x * q.vsplit * d->N)), dim3(512), 0, st, q);
    else hipLaunchKernelGGL(rgbout16_conv_kernel<false>, dim3((unsigned)(q.strips_x * q.vsplit * d->N)), dim3(512), 0, st, q);
    prof_end(tok, st);
    return check_launch("rgbout16_conv_kernel");
  }
  hipLaunchKernelGGL(rgbout_conv_kernel, dim3((unsigned)grid), dim3(512), 0, st, p);
  prof_end(tok, st);
  return check_launch("rgbout_conv_kernel");
}

}  // namespace srgan

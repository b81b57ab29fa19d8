// bf16 compute mode (BASELINE configs [2]-[4]): the 3x3 / stride-1 / zero-pad-1 convolutions of the generator's residual trunk
// (reference pyfiles/model.py:188-201) -- forward and input gradient -- as a direct GEMM on v_mfma_f32_32x32x16_bf16 whose
// activation operand lives in LDS for the whole kernel.
//
// The bf16 matrix pipe retires a 32x32x16 product in 32 cycles, 16x the fp32 rate: an implicit GEMM that re-stages an
// activation tile per (tap, channel chunk) moves 9x the activation through L2 -> registers -> convert -> LDS and spends 4x more
// time on that than on the products (igemm_kernel<256,128,4,2,BF>: 0.16 of the bf16 peak).  Here a workgroup owns a 4 x 32 pixel
// patch of one image and ALL reduce channels:
//   * the 6 x 34 pixel halo of the patch (zero outside the image) is read ONCE as fp32 (buffer loads, range-checked), rounded
//     to bf16 (nearest even, as every bf16-mode conv) and parked in LDS as [halo pixel][C + 8 pad] -- 105 KB at C = 256; the
//     16-byte pad puts 8 consecutive pixels on 8 different bank groups, so a fragment read (32 pixels x 16 bytes) is
//     conflict-free.  Only the first 64-channel quarter is loaded in the prologue: the K order is (quarter, tap, 32-chunk) and
//     quarter q + 1 streams in under the products of quarter q (one 32-pixel pass per tap);
//   * loop over C/64 quarters x 9 taps x 2 chunks: the A fragment of (tap, chunk) is ONE ds_read_b128 at
//     halo[(row + ty) * 34 + col + tx][chunk * 32 ...] -- no staging, no conversion, no masks in the loop; only the weight tile
//     [256 output channels][32 k] (16 KB of bf16, pre-packed in exactly that order, XOR-swizzled 16-byte pieces instead of
//     padding: ds_read_b128 serves lanes in groups made of aligned 4-lane blocks, so the swizzle key is the row's 4-block)
//     streams global -> registers (two tiles in flight) -> LDS through THREE buffers, so tile kt + 1 is readable while tile
//     kt is multiplied: the fragments of every 16-deep K step are requested one step ahead, one barrier per tile.  Every
//     workgroup streams the whole 9 C N filter image from L2 (16 KB per 16 MFMAs per wave): that stream, not LDS or the
//     matrix pipe, bounds the loop (measured: 172 us at batch 128 without the epilogue, 134 with the stream removed, 117 with
//     the fragment reads removed as well);
//   * 4 waves, one per SIMD, each 64 pixels (two image rows) x BN/2 output channels: 16 MFMAs per wave and K tile (BN = 256)
//     against 4 + 8 fragment reads;
//   * epilogue: a lane of the accumulator is an output channel, so every store instruction writes whole 128-byte lines of the
//     NHWC result; bias / activation / the residual-gradient add (`res`, input gradient of a block's first conv) ride here.
// HBM tensors stay fp32 (the same contract as the other bf16-mode kernels); the packed filter image is bf16
// [n tile][quarter][tap][chunk of the quarter][BN][32], kind 1 = taps rotated by 180 degrees with O and I swapped (conv_wino.hip: variant 4).
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "pack_device.h"

namespace srgan {

namespace {

constexpr int HK = 32;                 // reduce channels per K tile
constexpr int HPX = 6 * 34;            // halo pixels of a 4 x 32 patch

struct Halo16Params {
  const void* src;             // [NB][H][W][C] fp32, or bf16 (IN16)
  const unsigned short* wp;    // packed bf16 filters
  const float* bias;           // [N] or null
  const float* res;            // [NB][H][W][N] fp32 or null
  void* dst;                   // [NB][H][W][N] fp32, or bf16 (OUT16)
  int NB, H, W, N, tiles_y, tiles_x, n_tiles, act;
  float slope;
};

__device__ __forceinline__ auto uniform_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// IN16 / OUT16: the source / destination tensor is bf16 in HBM (the intermediates of the fused residual block, ops._ResBlockBf16Fn:
// conv outputs, the normalised activation, the gradients between the norm and conv backward kernels); fp32 otherwise.
template <int C, int BN, bool RES, bool IN16, bool OUT16>
__global__ __launch_bounds__(256) void halo16_kernel(Halo16Params p) {
  static_assert(!(RES && OUT16), "the skip gradient is added to an fp32 result");
  constexpr int ISZ = IN16 ? 2 : 4;            // bytes per source element
  constexpr int PS = C * 2 + 16;               // bytes per halo pixel
  constexpr int NQ = C / 64;                   // 64-channel quarters of the reduce dimension (K order: quarter, tap, 32-chunk)
  constexpr int NK = 9 * NQ * 2;               // K tiles
  constexpr int TN = BN / 64;                  // 32-wide output-channel blocks per wave (2 waves across N)
  constexpr int WTILE = BN * 64;               // bytes of one weight tile in LDS: [BN rows][4 swizzled 16-byte pieces]
  constexpr int WLD = BN * 4 / 256;            // 16-byte pieces of a weight tile per thread
  constexpr int HP = 7;                        // passes of 32 pixels over the 204 halo pixels of one quarter
  static_assert(BN == 64 || BN == 128 || BN == 256, "BN");
  __shared__ __attribute__((aligned(16))) unsigned char halo[HPX * PS];
  __shared__ __attribute__((aligned(16))) unsigned char wt[3 * WTILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);      // neighbouring patches (shared halo rows) on one XCD
  const int nt = bid % p.n_tiles;
  int r = bid / p.n_tiles;
  const int tx = r % p.tiles_x; r /= p.tiles_x;
  const int ty = r % p.tiles_y;
  const int nb = r / p.tiles_y;
  const int Y0 = ty * 4, X0 = tx * 32;

  const auto rs_x = uniform_rsrc(p.src, (unsigned)((size_t)p.NB * p.H * p.W * C * ISZ));
  const auto rs_w = uniform_rsrc(p.wp + (size_t)nt * NK * (BN * HK), (unsigned)(NK * BN * HK * 2));

  // ---- weight tiles: global -> registers -> LDS, three buffers (tile kt + 1 is readable while tile kt is multiplied) ----
  f32x4 wreg[WLD], wreg2[WLD];
  auto load_w = [&](f32x4* dst, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WLD; ++i)
      dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, tid * 16 + i * 4096, kt * (BN * HK * 2), 0));
  };
  auto store_w = [&](const f32x4* src, int boff) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WLD; ++i) {
      const int q = tid + 256 * i;               // piece q: row q / 4, 16-byte piece q % 4 (swizzled by the row's 4-block)
      const int n = q >> 2;
      *reinterpret_cast<f32x4*>(&wt[boff + n * 64 + (((q & 3) ^ ((n >> 2) & 3)) << 4)]) = src[i];
    }
  };
  load_w(wreg, 0);
  load_w(wreg2, 1);

  // ---- halo: one 64-channel quarter at a time; thread = (pixel of the pass, 8 channels), 32 pixels per pass ----
  const int hcg = tid & 7, hpl = tid >> 3;
  constexpr unsigned kOutside = 0x80000000u;
  auto halo_off = [&](int quarter, int pass) __attribute__((always_inline)) -> unsigned {
    const int hp = pass * 32 + hpl;
    const int hr = hp / 34, hc = hp - hr * 34;
    const int y = Y0 - 1 + hr, x = X0 - 1 + hc;
    const bool ok = hp < HPX && y >= 0 && y < p.H && x >= 0 && x < p.W;
    return ok ? (unsigned)((((nb * p.H + y) * p.W + x) * C + quarter * 64 + hcg * 8) * ISZ) : kOutside;
  };
  auto halo_put = [&](int quarter, int pass, f32x4 lo, f32x4 hi) __attribute__((always_inline)) {
    const int hp = pass * 32 + hpl;
    if (hp < HPX) {
      if constexpr (IN16) {                      // lo already holds the 8 bf16 channels
        *reinterpret_cast<f32x4*>(&halo[hp * PS + quarter * 128 + hcg * 16]) = lo;
      } else {
        const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
        bf16x8 v;
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        *reinterpret_cast<bf16x8*>(&halo[hp * PS + quarter * 128 + hcg * 16]) = v;
      }
    }
  };
  {
    f32x4 lo[HP], hi[HP];
#pragma unroll
    for (int g = 0; g < HP; ++g) {
      const unsigned off = halo_off(0, g);
      lo[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
      if constexpr (!IN16) hi[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 16, 0));
      else hi[g] = lo[g];
    }
    store_w(wreg, 0);
    store_w(wreg2, WTILE);
    load_w(wreg, 2);             // NK >= 18
    load_w(wreg2, 3);
#pragma unroll
    for (int g = 0; g < HP; ++g) halo_put(0, g, lo[g], hi[g]);
  }
  __syncthreads();

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // lane's A address: pixel (row 2 wm, column lr) of the patch at tap (0,0) = halo pixel (2 wm) * 34 + lr, k half lh
  const unsigned char* a_lane = halo + ((2 * wm) * 34 + lr) * PS + lh * 16;
  // lane's B offsets inside a weight tile for the two K steps of 16: row wn * BN/2 + lr (+ 32 j), piece (2 s + lh) ^ swizzle
  const int b_row = (wn * (BN / 2) + lr) * 64, b_sw = (lr >> 2) & 3;
  const int b_lane0 = b_row + (((0 + lh) ^ b_sw) << 4), b_lane1 = b_row + (((2 + lh) ^ b_sw) << 4);

  int kt = 0;
  bf16x8 fa[2][2], fb[2][TN];
  auto read_frags = [&](int slot, const unsigned char* a, int woff, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[slot][i] = *reinterpret_cast<const bf16x8*>(a + i * 34 * PS + s * 32);
    const unsigned char* B = wt + woff + (s ? b_lane1 : b_lane0);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const bf16x8*>(B + j * 32 * 64);
  };
  auto mma = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot][i], fb[slot][j], acc[i][j], 0, 0, 0);
  };

  int w_cur = 0, w_nxt = WTILE, w_nn = 2 * WTILE;       // LDS offsets of the tiles kt, kt + 1, kt + 2
  read_frags(0, a_lane, w_cur, 0);
  f32x4 hlo = {0.f, 0.f, 0.f, 0.f}, hhi = {0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < NQ; ++q) {
    for (int tap = 0; tap < 9; ++tap) {
      const int t3 = tap / 3;
      const unsigned char* a_tap = a_lane + (t3 * 34 + (tap - 3 * t3)) * PS + q * 128;
      // A address of the tile after this tap's two: next tap of the quarter, or tap 0 of the next quarter
      const int ntap = tap == 8 ? 0 : tap + 1, nq = tap == 8 ? q + 1 : q, n3 = ntap / 3;
      const unsigned char* a_next = a_lane + (n3 * 34 + (ntap - 3 * n3)) * PS + nq * 128;
      // the next quarter of the halo streams in under this quarter's products: pass `tap` is requested here and parked in
      // LDS one tap later (two barriers before its first reader)
      if (q + 1 < NQ) {
        if (tap >= 1 && tap <= HP) halo_put(q + 1, tap - 1, hlo, hhi);
        if (tap < HP) {
          const unsigned off = halo_off(q + 1, tap);
          hlo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
          if constexpr (!IN16) hhi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 16, 0));
        }
      }
#pragma unroll
      for (int kc2 = 0; kc2 < 2; ++kc2, ++kt) {
        const unsigned char* a_cur = a_tap + kc2 * 64;
        // K step 0 of tile kt: its fragments are in slot 0; request step 1 first
        read_frags(1, a_cur, w_cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(0);
        __builtin_amdgcn_sched_barrier(0);
        // K step 1: request step 0 of tile kt + 1 (already visible: stored an iteration ago)
        if (kt + 1 < NK) read_frags(0, kc2 == 0 ? a_tap + 64 : a_next, w_nxt, 0);
        // tile kt + 2 (requested two tiles ago into this parity's registers) -> LDS; request tile kt + 4
        if (kt + 2 < NK) store_w(kc2 == 0 ? wreg : wreg2, w_nn);
        if (kt + 4 < NK) load_w(kc2 == 0 ? wreg : wreg2, kt + 4);
        __builtin_amdgcn_sched_barrier(0);
        mma(1);
        __syncthreads();
        const int t = w_cur; w_cur = w_nxt; w_nxt = w_nn; w_nn = t;
      }
    }
  }

  // ---- epilogue: lane = output channel; register e of acc[i][j] = pixel column (e % 4) + 8 (e / 4) + 4 lh of image row
  // Y0 + 2 wm + i; the 32 lanes of a half-wave store one whole 128-byte line ----
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const size_t row = ((size_t)(nb * p.H + Y0 + 2 * wm + i) * p.W + X0) * p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nt * BN + wn * (BN / 2) + j * 32 + lr;
      const float bv = p.bias ? p.bias[n] : 0.f;
      float rv[16];
      if constexpr (RES) {
#pragma unroll
        for (int e = 0; e < 16; ++e) rv[e] = p.res[row + (size_t)((e & 3) + 8 * (e >> 2) + 4 * lh) * p.N + n];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = apply_act(acc[i][j][e] + bv, p.act, p.slope);
        if constexpr (RES) v += rv[e];
        const size_t o = row + (size_t)((e & 3) + 8 * (e >> 2) + 4 * lh) * p.N + n;
        if constexpr (OUT16) static_cast<__bf16*>(p.dst)[o] = (__bf16)v;
        else static_cast<float*>(p.dst)[o] = v;
      }
    }
  }
}

// ---- round 6: the 256-channel trunk layers with the FILTER operand fed global -> registers (VERDICT r5 item 1 i) ----
// halo16_kernel above moves every 16 KB filter tile global -> registers -> LDS -> registers and synchronises its four waves once
// per tile: with the stream and the fragment reads removed its loop still ran at half the matrix rate (profiles/LOG.md, round 2:
// 29 us of skeleton per round against 15.4 us of MFMAs) -- one wave per SIMD, in-order issue, a barrier every 16 MFMAs.  Here:
//   * the four waves split the OUTPUT CHANNELS (wave w: channels 64 w .. 64 w + 63, all 128 pixels of the patch: 4 pixel rows x
//     2 channel blocks = 8 MFMAs per 16-deep K step, 128 accumulator registers), so a B (filter) fragment belongs to ONE wave:
//     it is loaded straight from the packed register image (pack_device.h, variant 4 with N = 256: 1 KB contiguous per wave
//     and fragment) into a ring of registers five K steps ahead -- no LDS stores, no LDS reads, no swizzle for the filter
//     operand, and the kernel has one wave per SIMD and 512 registers to spend on the ring;
//   * the A (activation) fragments come from the LDS-resident halo exactly as before, 4 ds_read_b128 per 8 MFMAs = 512 bytes of
//     LDS reads per MFMA (was 768 + the tile stores);
//   * nothing in the K loop is shared between waves any more, so there is NO barrier inside a 64-channel quarter: the waves
//     drift apart and fill each other's stalls; one barrier per quarter publishes the next quarter of the halo, which streams
//     in under the products as before.
// Every workgroup still reads the whole 1.18 MB filter image from L2 (32 bytes per clock and CU at the full matrix rate).
#ifndef H16R_EXP
#define H16R_EXP 0          // timing ablations (wrong results): experiment builds only (common.h refuses the flag otherwise)
#endif
template <bool RES, bool IN16, bool OUT16>
__global__ __launch_bounds__(256) void halo16r_kernel(Halo16Params p) {
  static_assert(!(RES && OUT16), "the skip gradient is added to an fp32 result");
  constexpr int C = 256, N = 256;
  constexpr int ISZ = IN16 ? 2 : 4;
  constexpr int PS = C * 2 + 16;               // bytes per halo pixel (16-byte pad: conflict-free fragment reads)
  constexpr int NQ = C / 64;                   // 64-channel quarters
  constexpr int QS = 36;                       // K steps of 16 per quarter: 9 taps x 4
  constexpr int NKS = NQ * QS;                 // 144
  constexpr int RING = 6, PF = RING - 1;       // B fragments are requested PF K steps ahead (QS % RING == 0)
  constexpr int HP = 7;                        // passes of 32 pixels over the 204 halo pixels of one quarter
  __shared__ __attribute__((aligned(16))) unsigned char halo[HPX * PS];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);      // neighbouring patches (shared halo rows) on one XCD
  int r = bid;
  const int tx = r % p.tiles_x; r /= p.tiles_x;
  const int ty = r % p.tiles_y;
  const int nb = r / p.tiles_y;
  const int Y0 = ty * 4, X0 = tx * 32;

  const auto rs_x = uniform_rsrc(p.src, (unsigned)((size_t)p.NB * p.H * p.W * C * ISZ));
  const auto rs_w = uniform_rsrc(p.wp, (unsigned)(NKS * 4 * 2 * 1024));      // past the image: zeros (the ring's last requests)

  // ---- B ring: fragment (K step kg, channel block j) of this wave = 1 KB at ((kg * 4 + wave) * 2 + j) * 1024 ----
  bf16x8 fb[RING][2];
  const int w_lane = lane * 16, w_wave = wave * 2048;
  auto load_b = [&](int slot, int kg) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      fb[slot][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, kg * 8192 + w_wave + j * 1024, 0));
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) load_b(s, s);

  // ---- halo: one 64-channel quarter at a time; thread = (pixel of the pass, 8 channels), 32 pixels per pass ----
  const int hcg = tid & 7, hpl = tid >> 3;
  constexpr unsigned kOutside = 0x80000000u;
  auto halo_off = [&](int quarter, int pass) __attribute__((always_inline)) -> unsigned {
    const int hp = pass * 32 + hpl;
    const int hr = hp / 34, hc = hp - hr * 34;
    const int y = Y0 - 1 + hr, x = X0 - 1 + hc;
    const bool ok = hp < HPX && y >= 0 && y < p.H && x >= 0 && x < p.W;
    return ok ? (unsigned)((((nb * p.H + y) * p.W + x) * C + quarter * 64 + hcg * 8) * ISZ) : kOutside;
  };
  auto halo_put = [&](int quarter, int pass, f32x4 lo, f32x4 hi) __attribute__((always_inline)) {
    const int hp = pass * 32 + hpl;
    if (hp < HPX) {
      if constexpr (IN16) {                      // lo already holds the 8 bf16 channels
        *reinterpret_cast<f32x4*>(&halo[hp * PS + quarter * 128 + hcg * 16]) = lo;
      } else {
        const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
        bf16x8 v;
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        *reinterpret_cast<bf16x8*>(&halo[hp * PS + quarter * 128 + hcg * 16]) = v;
      }
    }
  };
  {
    f32x4 lo[HP], hi[HP];
#pragma unroll
    for (int g = 0; g < HP; ++g) {
      const unsigned off = halo_off(0, g);
      lo[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
      if constexpr (!IN16) hi[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 16, 0));
      else hi[g] = lo[g];
    }
#pragma unroll
    for (int g = 0; g < HP; ++g) halo_put(0, g, lo[g], hi[g]);
  }
  __syncthreads();

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // lane's A address: pixel (row 0, column lr) of the patch at tap (0, 0) = halo pixel lr, channels 8 lh .. of the K step
  const unsigned char* a_lane = halo + lr * PS + lh * 16;
  bf16x8 fa[2][4];
  auto read_a = [&](int slot, const unsigned char* a) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[slot][i] = *reinterpret_cast<const bf16x8*>(a + i * 34 * PS);
  };
  auto mma = [&](int sa, int sb) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sa][i], fb[sb][j], acc[i][j], 0, 0, 0);
  };

  f32x4 hlo = {0.f, 0.f, 0.f, 0.f}, hhi = {0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < NQ; ++q) {
    const unsigned char* a_q = a_lane + q * 128;
    read_a(0, a_q);                                    // (tap 0, step 0) of this quarter: published by the barrier just passed
#pragma unroll
    for (int t = 0; t < QS; ++t) {
      const int tap = t >> 2, s = t & 3;
      // the next quarter of the halo streams in under this quarter's products: pass `tap` is requested at the tap's first K
      // step and parked in LDS one tap later; the barrier at the end of the quarter publishes it
      if (s == 0 && q + 1 < NQ && !(H16R_EXP & 16)) {
        if (tap >= 1 && tap <= HP) halo_put(q + 1, tap - 1, hlo, hhi);
        if (tap < HP) {
          const unsigned off = halo_off(q + 1, tap);
          hlo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
          if constexpr (!IN16) hhi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 16, 0));
        }
      }
      // request: the A fragments of the next K step of this quarter, the B fragments PF steps ahead (past the image: zeros)
      if (t + 1 < QS && !(H16R_EXP & 4)) {
        const int t2 = t + 1, tap2 = t2 >> 2, s2 = t2 & 3;
        read_a(t2 & 1, a_q + ((tap2 / 3) * 34 + tap2 % 3) * PS + s2 * 32);
      }
      if (!(H16R_EXP & 2)) load_b((t + PF) % RING, q * QS + t + PF);
      __builtin_amdgcn_sched_barrier(0);
      if (!(H16R_EXP & 8)) mma(t & 1, t % RING);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q + 1 < NQ) __syncthreads();
  }

  // ---- epilogue: lane = output channel; register e of acc[i][j] = pixel column (e % 4) + 8 (e / 4) + 4 lh of image row
  // Y0 + i; the 32 lanes of a half-wave store one whole 128-byte line (fp32) ----
  if ((H16R_EXP & 1) && acc[0][0][0] != 12345.678f) return;          // (ablation: no result stores)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const size_t row = ((size_t)(nb * p.H + Y0 + i) * p.W + X0) * N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = wave * 64 + j * 32 + lr;
      const float bv = p.bias ? p.bias[n] : 0.f;
      float rv[16];
      if constexpr (RES) {
#pragma unroll
        for (int e = 0; e < 16; ++e) rv[e] = p.res[row + (size_t)((e & 3) + 8 * (e >> 2) + 4 * lh) * N + n];
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = apply_act(acc[i][j][e] + bv, p.act, p.slope);
        if constexpr (RES) v += rv[e];
        const size_t o = row + (size_t)((e & 3) + 8 * (e >> 2) + 4 * lh) * N + n;
        if constexpr (OUT16) static_cast<__bf16*>(p.dst)[o] = (__bf16)v;
        else static_cast<float*>(p.dst)[o] = v;
      }
    }
  }
}

// ---- weight gradient of the same layers:  dW[o][c][ty][tx] = sum_pixels dy[pixel][o] * x[pixel + (ty-1, tx-1)][c]  ----
// A workgroup owns a 64 (o) x 64 (c) block of ALL NINE taps (9 x 16 accumulator registers per lane: wave (wo, wc) holds the
// 32 x 32 sub-block of every tap) over a range of 4 x 32 pixel patches.  Per patch the 64-channel slices of dy (128 pixels)
// and of the x halo (6 x 34 pixels, zero outside the image) are read once as fp32, rounded to bf16 and parked in LDS as
// [pixel][64 channels + 32 pad] (192-byte rows: the four pixel rows of a transposing read lie 16 banks apart); the reduce
// index of the MFMA is the PIXEL, so both operands come from ds_read_b64_tr_b16 (a 16-lane group reads 4 pixels x 16 channels
// and gets them channel-major): the dy fragments of the 8 K steps (16 pixels each) are read once per patch and reused by the
// nine taps, the x fragment of (tap, K step) is the same read shifted by (ty * 34 + tx) halo pixels -- no staging per tap.
// 72 MFMAs per wave and patch against 84 KB of fp32 input per workgroup; the next patch's 22 loads per thread are in flight
// during the products (double-buffered LDS, one barrier per patch).  Split-K over patch ranges into [split][O][9 C] slabs, summed
// by wgrad_reduce_kernel (conv_igemm.hip) like every other weight-gradient kernel.
struct Halo16WgradParams {
  const void* x;      // [NB][H][W][C] fp32, or bf16 (X16)
  const void* dy;     // [NB][H][W][O] fp32, or bf16 (D16)
  float* slab;        // [splits][O][9 C]
  int NB, H, W, C, O, tiles_y, tiles_x, patches, per_split, splits, o_tiles, c_tiles;
};

constexpr int GPS = 192;               // bytes per pixel row of a 64-channel bf16 slice in LDS
constexpr int GX = HPX * GPS, GD = 128 * GPS;

template <bool X16, bool D16>
__global__ __launch_bounds__(256) void halo16_wgrad_kernel(Halo16WgradParams p) {
  constexpr int XSZ = X16 ? 2 : 4, DSZ = D16 ? 2 : 4;
  typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4;
  __shared__ __attribute__((aligned(16))) unsigned char xs[2 * GX];
  __shared__ __attribute__((aligned(16))) unsigned char ds[2 * GD];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave >> 1, wc = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int li = lane & 15, q4 = li >> 2, pq = li & 3, g1 = (lane >> 4) & 1;

  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);      // the 16 blocks of one patch range share an XCD's L2
  const int tiles = p.o_tiles * p.c_tiles;
  const int split = bid / tiles, tile = bid - split * tiles;
  const int ot = tile / p.c_tiles, ct = tile - ot * p.c_tiles;
  const int p_begin = split * p.per_split, p_end = min(p_begin + p.per_split, p.patches);

  const auto rs_x = uniform_rsrc(p.x, (unsigned)((size_t)p.NB * p.H * p.W * p.C * XSZ));
  const auto rs_d = uniform_rsrc(p.dy, (unsigned)((size_t)p.NB * p.H * p.W * p.O * DSZ));
  constexpr unsigned kOutside = 0x80000000u;
  const int hcg = tid & 7, hpl = tid >> 3;           // 8 channels of one of the 32 pixels of a pass

  f32x4 xl[7], xh[7], dl[4], dh[4];
  auto issue = [&](int patch) __attribute__((always_inline)) {
    int r = patch;
    const int tx = r % p.tiles_x; r /= p.tiles_x;
    const int ty = r % p.tiles_y;
    const int nb = r / p.tiles_y;
    const int Y0 = ty * 4, X0 = tx * 32;
#pragma unroll
    for (int g = 0; g < 7; ++g) {
      const int hp = g * 32 + hpl;
      const int hr = hp / 34, hc = hp - hr * 34;
      const int y = Y0 - 1 + hr, x = X0 - 1 + hc;
      const bool ok = hp < HPX && y >= 0 && y < p.H && x >= 0 && x < p.W;
      const unsigned off = ok ? (unsigned)((((nb * p.H + y) * p.W + x) * p.C + ct * 64 + hcg * 8) * XSZ) : kOutside;
      xl[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
      if constexpr (!X16) xh[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 16, 0));
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int px = g * 32 + hpl;                   // pixel (px / 32, px % 32) of the patch
      const unsigned off = (unsigned)((((nb * p.H + Y0 + (px >> 5)) * p.W + X0 + (px & 31)) * p.O + ot * 64 + hcg * 8) * DSZ);
      dl[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, off, 0, 0));
      if constexpr (!D16) dh[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, off, 16, 0));
    }
  };
  auto pack8 = [](f32x4 lo, f32x4 hi) __attribute__((always_inline)) {
    const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
    bf16x8 v;
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    return v;
  };
  auto park = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 7; ++g) {
      const int hp = g * 32 + hpl;
      if (hp < HPX) {
        if constexpr (X16) *reinterpret_cast<f32x4*>(&xs[buf * GX + hp * GPS + hcg * 16]) = xl[g];
        else *reinterpret_cast<bf16x8*>(&xs[buf * GX + hp * GPS + hcg * 16]) = pack8(xl[g], xh[g]);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if constexpr (D16) *reinterpret_cast<f32x4*>(&ds[buf * GD + (g * 32 + hpl) * GPS + hcg * 16]) = dl[g];
      else *reinterpret_cast<bf16x8*>(&ds[buf * GD + (g * 32 + hpl) * GPS + hcg * 16]) = pack8(dl[g], dh[g]);
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // lane part of a transposing fragment read: pixel 8 lh + q4 of the K step, channels 16 g1 + 4 pq .. + 3 of the wave's 32
  const int d_lane = (8 * lh + q4) * GPS + (wo * 32 + 16 * g1 + 4 * pq) * 2;
  const int x_lane = (8 * lh + q4) * GPS + (wc * 32 + 16 * g1 + 4 * pq) * 2;
  auto frag = [&](const unsigned char* base) __attribute__((always_inline)) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(base));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(base + 4 * GPS));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  if (p_begin < p_end) {
    issue(p_begin);
    park(0);
    __syncthreads();
  }
  for (int pt = p_begin; pt < p_end; ++pt) {
    const int buf = (pt - p_begin) & 1;
    if (pt + 1 < p_end) issue(pt + 1);
    const unsigned char* D = ds + buf * GD + d_lane;
    const unsigned char* X = xs + buf * GX + x_lane;
    bf16x8 df[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) df[ks] = frag(D + ((ks >> 1) * 32 + (ks & 1) * 16) * GPS);
    bf16x8 xf[2];
    xf[0] = frag(X);                                  // tap (0,0), K step 0
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // round 6: the next patch's slices (requested at the top of this patch) are parked in LDS UNDER the last three taps'
      // products instead of behind all of them -- the 11 16-byte stores per thread and their wait no longer sit between the
      // last product of a patch and its barrier (the compiler schedules freely on either side of the fence)
      if (t == 6) {
        __builtin_amdgcn_sched_barrier(0);
        if (pt + 1 < p_end) park(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int g = t * 8 + ks;
        if (g + 1 < 72) {
          const int t2 = (g + 1) >> 3, k2 = (g + 1) & 7;
          xf[(g + 1) & 1] = frag(X + (((k2 >> 1) + t2 / 3) * 34 + (k2 & 1) * 16 + t2 % 3) * GPS);
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[ks], xf[g & 1], acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // slab[split][o][tap * C + c]: lane = c (128-byte lines), register e = o row (e % 4) + 8 (e / 4) + 4 lh
  float* out = p.slab + ((size_t)split * p.O + ot * 64 + wo * 32) * (9 * p.C) + ct * 64 + wc * 32 + lr;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) out[(size_t)((e & 3) + 8 * (e >> 2) + 4 * lh) * (9 * p.C) + t * p.C] = acc[t][e];
}

// ---- weight gradient of the 4x4 / stride-2 / pad-1 layers (round 4; reference pyfiles/model.py:212-215, 227-230, 302-309) ----
//   dW[o][c][ky][kx] = sum_{n, oy, ox} dy[n][oy][ox][o] * x[n][2 oy + ky - 1][2 ox + kx - 1][c]
// The staged GEMM (wgrad_kernel<..., BF>) gives every (o tile, tap pair) workgroup its own pass over dy and x: 8 passes over each
// tensor for the generator's 64 -> 128 layer, ~1 GB per launch through L2 / the Infinity Cache -- it runs at the rate of that
// stream (0.10 of the bf16 matrix peak), not of the matrix pipe.  Here a workgroup owns a 64 (o) x 64 (c) block of ALL SIXTEEN
// taps over a range of 2 x 32 output-pixel patches, as halo16_wgrad_kernel does for the 3x3 layers:
//   * per patch the 64-channel slice of the 6 x 66 pixel region of x that the patch's taps touch (rows 4 py - 1 .. 4 py + 4,
//     columns 64 px - 1 .. 64 px + 64; zero outside the image) and of the 64 dy pixels is read ONCE as fp32 (buffer loads, range-
//     checked), rounded to bf16 and parked in LDS as [pixel][64 channels + pad]: x is read O / 64 times and dy C / 64 times in all;
//   * the reduce index of the MFMA is the pixel, so both operands come from the transposing read ds_read_b64_tr_b16 (a 16-lane
//     group reads 4 pixels x 16 channels and receives them channel-major).  The x fragment of (tap, 16-pixel K step) is that read
//     at region pixel (2 i + ky) * 66 + 2 j + kx: consecutive output pixels lie TWO region pixels apart, and the row stride of
//     160 bytes puts the four pixels of a read 16 banks apart at that distance (2 x 160 B = 80 dwords = 16 mod 64), the second
//     16-channel group 8 banks further: conflict-free without separating the four input phases in LDS;
//   * 8 waves = (o half, c half, upper / lower two filter rows): 8 taps x 16 accumulator registers per lane; the 4 dy fragments
//     of a patch are read once and reused by the eight taps; 32 MFMAs per wave and patch, the next patch's 16 loads per thread
//     in flight under them (double-buffered LDS, one barrier per patch);
//   * split-K over patch ranges into [split][O][16 C] slabs (tap-major columns, as every weight-gradient kernel), summed by the
//     deterministic slab reduce of conv_igemm.hip.
struct Halo16S2WgradParams {
  const void* x;      // [NB][2 Ho][2 Wo][C] fp32, or bf16 (X16)
  const void* dy;     // [NB][Ho][Wo][O] fp32, or bf16 (D16)
  float* slab;        // [splits][O][16 C]
  int NB, Ho, Wo, C, O, tiles_y, tiles_x, patches, per_split, splits, o_tiles, c_tiles;
};

constexpr int S2XS = 160;              // bytes per region pixel in LDS (64 bf16 channels + 32 pad)
constexpr int S2D = 64 * GPS;

// PW: patch width in output pixels (32, 16 or 8; the patch is 64 / PW rows high): the discriminator's 16 x 16 and 8 x 8 output maps
// are covered by 4 x 16 and 8 x 8 patches (regions of 10 x 34 and 18 x 18 pixels).
template <int PW, bool X16, bool D16>
__global__ __launch_bounds__(512) void halo16s2_wgrad_kernel(Halo16S2WgradParams p) {
  constexpr int PH = 64 / PW, RW = 2 * PW + 2, S2R = (2 * PH + 2) * RW, S2X = S2R * S2XS;
  constexpr int XSZ = X16 ? 2 : 4, DSZ = D16 ? 2 : 4;
  typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4;
  __shared__ __attribute__((aligned(16))) unsigned char xs[2 * S2X];
  __shared__ __attribute__((aligned(16))) unsigned char ds[2 * S2D];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave >> 2, wc = (wave >> 1) & 1, th = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int li = lane & 15, q4 = li >> 2, pq = li & 3, g1 = (lane >> 4) & 1;

  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);      // the (o, c) blocks of one patch range share an XCD's L2
  const int tiles = p.o_tiles * p.c_tiles;
  const int split = bid / tiles, tile = bid - split * tiles;
  const int ot = tile / p.c_tiles, ct = tile - ot * p.c_tiles;
  const int p_begin = split * p.per_split, p_end = min(p_begin + p.per_split, p.patches);
  const int Hi = 2 * p.Ho, Wi = 2 * p.Wo;

  const auto rs_x = uniform_rsrc(p.x, (unsigned)((size_t)p.NB * Hi * Wi * p.C * XSZ));
  const auto rs_d = uniform_rsrc(p.dy, (unsigned)((size_t)p.NB * p.Ho * p.Wo * p.O * DSZ));
  constexpr unsigned kOutside = 0x80000000u;
  const int hcg = tid & 7, hpl = tid >> 3;           // 8 channels of one of the 64 pixels of a pass
  constexpr int XP = (S2R + 63) / 64;                // 7 passes over the region

  f32x4 xl[XP], xh[XP], dl, dh;
  auto issue = [&](int patch) __attribute__((always_inline)) {
    int r = patch;
    const int tx = r % p.tiles_x; r /= p.tiles_x;
    const int ty = r % p.tiles_y;
    const int nb = r / p.tiles_y;
    const int Y0 = 2 * PH * ty - 1, X0 = 2 * PW * tx - 1;     // region origin in x
#pragma unroll
    for (int g = 0; g < XP; ++g) {
      const int hp = g * 64 + hpl;
      const int hr = hp / RW, hc = hp - hr * RW;
      const int y = Y0 + hr, x = X0 + hc;
      const bool ok = hp < S2R && y >= 0 && y < Hi && x >= 0 && x < Wi;
      const unsigned off = ok ? (unsigned)((((nb * Hi + y) * Wi + x) * p.C + ct * 64 + hcg * 8) * XSZ) : kOutside;
      xl[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
      if constexpr (!X16) xh[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 16, 0));
    }
    {
      const int oy = PH * ty + hpl / PW, ox = PW * tx + hpl % PW;         // pixel hpl of the patch
      const unsigned off = (unsigned)((((nb * p.Ho + oy) * p.Wo + ox) * p.O + ot * 64 + hcg * 8) * DSZ);
      dl = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, off, 0, 0));
      if constexpr (!D16) dh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, off, 16, 0));
    }
  };
  auto pack8 = [](f32x4 lo, f32x4 hi) __attribute__((always_inline)) {
    const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
    bf16x8 v;
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    return v;
  };
  auto park = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < XP; ++g) {
      const int hp = g * 64 + hpl;
      if (hp < S2R) {
        if constexpr (X16) *reinterpret_cast<f32x4*>(&xs[buf * S2X + hp * S2XS + hcg * 16]) = xl[g];
        else *reinterpret_cast<bf16x8*>(&xs[buf * S2X + hp * S2XS + hcg * 16]) = pack8(xl[g], xh[g]);
      }
    }
    if constexpr (D16) *reinterpret_cast<f32x4*>(&ds[buf * S2D + hpl * GPS + hcg * 16]) = dl;
    else *reinterpret_cast<bf16x8*>(&ds[buf * S2D + hpl * GPS + hcg * 16]) = pack8(dl, dh);
  };

  f32x16 acc[8];      // tap (ky, kx) = (2 th + (t >> 2), t & 3)
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // lane part of a transposing fragment read: pixel 8 lh + q4 (and + 4) of the K step, channels 16 g1 + 4 pq .. + 3 of the wave's 32
  const int d_lane = (8 * lh + q4) * GPS + (wo * 32 + 16 * g1 + 4 * pq) * 2;
  // pixel 16 ks + 8 lh + q4 of the patch = (row, column) = (that / PW, that % PW): the lane's part of its region pixel
  const int x_lane = (2 * th * RW + (PW >= 16 ? 2 * (8 * lh + q4) : 2 * lh * RW + 2 * q4)) * S2XS + (wc * 32 + 16 * g1 + 4 * pq) * 2;
  auto dfrag = [&](const unsigned char* base) __attribute__((always_inline)) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(base));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(base + 4 * GPS));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto xfrag = [&](const unsigned char* base) __attribute__((always_inline)) {      // pixels two region pixels apart
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(base));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(base + 8 * S2XS));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // region-pixel offset of (tap t of this wave's eight, K step ks): row 2 i + ky (the 2 th part rides in x_lane), column 2 j0 + kx
  // with (i, j0) = patch position of the K step's first pixel (its lane part rides in x_lane as well)
  auto x_off = [](int t, int ks) __attribute__((always_inline)) {
    const int i = PW == 32 ? ks >> 1 : PW == 16 ? ks : 2 * ks, j0 = PW == 32 ? 16 * (ks & 1) : 0;
    return ((2 * i + (t >> 2)) * RW + 2 * j0 + (t & 3)) * S2XS;
  };

  if (p_begin < p_end) {
    issue(p_begin);
    park(0);
    __syncthreads();
  }
  for (int pt = p_begin; pt < p_end; ++pt) {
    const int buf = (pt - p_begin) & 1;
    if (pt + 1 < p_end) issue(pt + 1);
    const unsigned char* D = ds + buf * S2D + d_lane;
    const unsigned char* X = xs + buf * S2X + x_lane;
    bf16x8 df[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) df[ks] = dfrag(D + 16 * ks * GPS);
    bf16x8 xf[2];
    xf[0] = xfrag(X + x_off(0, 0));
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      // (round 6: the next patch is parked under the last three taps' products, as above -- in the all-bf16 instantiation; with an
      //  fp32 side the staging registers beside the fence spill at this kernel's 256-register limit)
      if constexpr (X16 && D16) {
        if (t == 5) {
          __builtin_amdgcn_sched_barrier(0);
          if (pt + 1 < p_end) park(buf ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int g = t * 4 + ks;
        if (g + 1 < 32) xf[(g + 1) & 1] = xfrag(X + x_off((g + 1) >> 2, (g + 1) & 3));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[ks], xf[g & 1], acc[t], 0, 0, 0);
      }
    }
    if constexpr (!(X16 && D16)) {
      if (pt + 1 < p_end) park(buf ^ 1);
    }
    __syncthreads();
  }

  // slab[split][o][tap * C + c]: lane = c (128-byte lines), register e = o row (e % 4) + 8 (e / 4) + 4 lh
  float* out = p.slab + ((size_t)split * p.O + ot * 64 + wo * 32) * (16 * p.C) + ct * 64 + wc * 32 + lr;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      out[(size_t)((e & 3) + 8 * (e >> 2) + 4 * lh) * (16 * p.C) + (size_t)(8 * th + t) * p.C] = acc[t][e];
}

// ---- transposed form of the 4x4 / stride-2 / pad-1 layers (input gradient of the generator's down convolutions and of the
// discriminator trunk, forward of the generator's ConvTranspose2d layers; reference pyfiles/model.py:212-215,227-230,302-309) ----
// Output pixel (2u + r, 2v + s) of phase (r, s) is a 2x2 stride-1 correlation of the SOURCE map:
//   out[2u+r][2v+s][n] = sum_{a,b in {0,1}} sum_k src[u + a + r - 1][v + b + s - 1][k] * w[k][n][3 - 2a - r][3 - 2b - s]
// -- in halo coordinates the shifts (a + r, b + s) are four of the nine shifts of the 3x3 kernel above, so the same 6 x 34 halo
// of a 4 x 32 source patch serves all four phases: one GEMM per phase (K = 4 taps x C), run back to back as ONE tile stream
// (phase, tap, chunk) so that the filter tiles keep flowing and the 8 x 64-pixel result of a phase is stored under the
// products of the next.  N = C / 2 output channels: a filter tile is always 16 KB = [BN][HK] with HK = 8192 / BN reduce
// channels (64 at BN = 128, 128 at BN = 64), i.e. 16 MFMAs per wave and tile as above; swizzle key of a row = its index
// shifted so that the 16 rows of a ds_read_b128 lane group hit 16 different 16-byte columns.
struct Halo16TParams {
  const void* src;             // [NB][Hs][Ws][C] fp32, or bf16 (IN16)
  const unsigned short* wp;    // packed bf16 filters [phase][tap][chunk][BN][HK]
  void* dst;                   // [NB][2 Hs][2 Ws][N] fp32, or bf16 (OUT16)
  int NB, Hs, Ws, tiles_y, tiles_x;
};

template <int C, int BN, bool IN16 = false, bool OUT16 = false>
__global__ __launch_bounds__(256) void halo16t_kernel(Halo16TParams p) {
  constexpr int ISZ = IN16 ? 2 : 4;            // bytes per source element
  constexpr int PS = C * 2 + 16;               // bytes per halo pixel
  constexpr int HKT = 8192 / BN;               // reduce channels per tile
  constexpr int NCH = C / HKT;                 // chunks per tap
  constexpr int NKP = 4 * NCH;                 // tiles per phase
  constexpr int NK = 4 * NKP;                  // tiles in all
  constexpr int KS = HKT / 16;                 // 16-deep K steps per tile
  constexpr int TN = BN / 64;                  // 32-wide output-channel blocks per wave
  constexpr int WTILE = 16384;
  constexpr int PCS = HKT / 8;                 // 16-byte pieces per tile row
  constexpr int SWS = PCS == 8 ? 1 : 0;        // swizzle key = (row >> SWS) & (PCS - 1)
  static_assert((BN == 128 && C == 256) || (BN == 64 && C == 128), "instantiated for (C, N) = (256, 128) and (128, 64)");
  static_assert(NKP % 2 == 0, "two tiles per loop body");
  __shared__ __attribute__((aligned(16))) unsigned char halo[HPX * PS];
  __shared__ __attribute__((aligned(16))) unsigned char wt[3 * WTILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  int r0 = bid;
  const int tx = r0 % p.tiles_x; r0 /= p.tiles_x;
  const int ty = r0 % p.tiles_y;
  const int nb = r0 / p.tiles_y;
  const int Y0 = ty * 4, X0 = tx * 32;

  const auto rs_x = uniform_rsrc(p.src, (unsigned)((size_t)p.NB * p.Hs * p.Ws * C * ISZ));
  const auto rs_w = uniform_rsrc(p.wp, (unsigned)(NK * WTILE));

  f32x4 wreg[4], wreg2[4];
  auto load_w = [&](f32x4* dst, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, tid * 16 + i * 4096, kt * WTILE, 0));
  };
  auto store_w = [&](const f32x4* src, int boff) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i;               // piece q: row q / PCS, piece q % PCS
      const int n = q / PCS, pc = q % PCS;
      *reinterpret_cast<f32x4*>(&wt[boff + n * (HKT * 2) + ((pc ^ ((n >> SWS) & (PCS - 1))) << 4)]) = src[i];
    }
  };
  load_w(wreg, 0);
  load_w(wreg2, 1);

  // ---- the whole halo, one 64-channel quarter at a time (7 passes of 32 pixels x 8 channels per thread) ----
  {
    const int hcg = tid & 7, hpl = tid >> 3;
    constexpr unsigned kOutside = 0x80000000u;
#pragma unroll
    for (int quarter = 0; quarter < C / 64; ++quarter) {
      f32x4 lo[7], hi[7];
#pragma unroll
      for (int g = 0; g < 7; ++g) {
        const int hp = g * 32 + hpl;
        const int hr = hp / 34, hc = hp - hr * 34;
        const int y = Y0 - 1 + hr, x = X0 - 1 + hc;
        const bool ok = hp < HPX && y >= 0 && y < p.Hs && x >= 0 && x < p.Ws;
        const unsigned off = ok ? (unsigned)((((nb * p.Hs + y) * p.Ws + x) * C + quarter * 64 + hcg * 8) * ISZ) : kOutside;
        lo[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
        if constexpr (!IN16) hi[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 16, 0));
        else hi[g] = lo[g];
      }
#pragma unroll
      for (int g = 0; g < 7; ++g) {
        const int hp = g * 32 + hpl;
        if (hp < HPX) {
          if constexpr (IN16) {
            *reinterpret_cast<f32x4*>(&halo[hp * PS + quarter * 128 + hcg * 16]) = lo[g];
          } else {
            const bf16x4 a = __builtin_convertvector(lo[g], bf16x4), b = __builtin_convertvector(hi[g], bf16x4);
            bf16x8 v;
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
            *reinterpret_cast<bf16x8*>(&halo[hp * PS + quarter * 128 + hcg * 16]) = v;
          }
        }
      }
    }
  }
  store_w(wreg, 0);
  store_w(wreg2, WTILE);
  load_w(wreg, 2);
  load_w(wreg2, 3);
  __syncthreads();

  f32x16 acc[2][TN];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  zero_acc();

  const unsigned char* a_lane = halo + ((2 * wm) * 34 + lr) * PS + lh * 16;
  const int b_row = (wn * (BN / 2) + lr) * (HKT * 2), b_key = (lr >> SWS) & (PCS - 1);
  auto b_off = [&](int s) __attribute__((always_inline)) { return b_row + (((2 * s + lh) ^ b_key) << 4); };

  bf16x8 fa[2][2], fb[2][TN];
  auto read_frags = [&](int slot, const unsigned char* a, int woff, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[slot][i] = *reinterpret_cast<const bf16x8*>(a + i * 34 * PS + s * 32);
    const unsigned char* B = wt + woff + b_off(s);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const bf16x8*>(B + j * 32 * (HKT * 2));
  };
  auto mma = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot][i], fb[slot][j], acc[i][j], 0, 0, 0);
  };
  // A address of tile (phase, tap, chunk): shift (a + r, b + s) halo pixels, chunk * HKT channels
  auto a_of = [&](int ph, int tap, int ch) __attribute__((always_inline)) {
    return a_lane + (((tap >> 1) + (ph >> 1)) * 34 + (tap & 1) + (ph & 1)) * PS + ch * (HKT * 2);
  };
  auto flush = [&](int ph) __attribute__((always_inline)) {
    const int r = ph >> 1, sx = ph & 1;
    const int Wd = 2 * p.Ws;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const size_t row = ((size_t)(nb * 2 * p.Hs + 2 * (Y0 + 2 * wm + i) + r) * Wd + 2 * X0 + sx) * BN;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = wn * (BN / 2) + j * 32 + lr;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const size_t o = row + (size_t)(2 * ((e & 3) + 8 * (e >> 2) + 4 * lh)) * BN + n;
          if constexpr (OUT16) static_cast<__bf16*>(p.dst)[o] = (__bf16)acc[i][j][e];
          else static_cast<float*>(p.dst)[o] = acc[i][j][e];
        }
      }
    }
  };

  int w_cur = 0, w_nxt = WTILE, w_nn = 2 * WTILE;
  int ph = 0, tap = 0, ch = 0;                 // coordinates of tile kt
  read_frags(0, a_of(0, 0, 0), w_cur, 0);
  for (int kt = 0; kt < NK; kt += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = kt + u;
      const unsigned char* a_cur = a_of(ph, tap, ch);
      // coordinates of the next tile
      int nch = ch + 1, ntap = tap, nph = ph;
      if (nch == NCH) { nch = 0; ntap = tap + 1; if (ntap == 4) { ntap = 0; nph = ph + 1; } }
      const unsigned char* a_nx = a_of(nph, ntap, nch);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        // request the fragments of the next K step (of this tile, or step 0 of the next tile: stored an iteration ago)
        if (s + 1 < KS) read_frags((s + 1) & 1, a_cur, w_cur, s + 1);
        else {
          if (t + 1 < NK) read_frags((s + 1) & 1, a_nx, w_nxt, 0);
          if (t + 2 < NK) store_w(u == 0 ? wreg : wreg2, w_nn);
          if (t + 4 < NK) load_w(u == 0 ? wreg : wreg2, t + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(s & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (nph != ph) {                         // last tile of a phase: its 8 x 64-pixel result goes out under the next phase
        flush(ph);
        zero_acc();
      }
      __syncthreads();
      const int tw = w_cur; w_cur = w_nxt; w_nxt = w_nn; w_nn = tw;
      ph = nph; tap = ntap; ch = nch;
    }
  }
}

// ---- strided form of the 4x4 / stride-2 / pad-1 layers (round 4): forward of the generator's down convolutions and of the
// discriminator trunk, input gradient of the generator's ConvTranspose2d layers (reference pyfiles/model.py:212-215, 227-230,
// 302-309) ----
//   out[oy][ox][n] = sum_{ky, kx, c} x[2 oy + ky - 1][2 ox + kx - 1][c] * w[n][c][ky][kx]
// The staged GEMM (igemm_kernel<256,128,BF>) pushes every input pixel through registers -> convert -> LDS once per tap that uses
// it (4 of the 16) and streams fp32 filter tiles.  Here a workgroup owns a 4 x 32 output patch and all N output channels:
//   * the 10 x 66 pixel region of x behind the patch (zero outside the image) is read ONCE per 64-channel half as fp32, rounded
//     to bf16 and parked in LDS (95 KB).  Output pixels that are neighbours in a row read region pixels TWO apart, so the region
//     is stored with its even and odd columns in separate planes, [row][column parity][33][64 channels + 8 pad]: the A fragment
//     of (tap, K step) is then ONE ds_read_b128 per 32-pixel row at plane (kx & 1), column j + (kx >> 1) -- consecutive lanes
//     144 bytes apart, conflict-free as in halo16_kernel;
//   * K order (half, tap, chunk): the bf16 filter tiles [N][8192 / N reduce channels] (16 KB, 16 MFMAs per wave) stream
//     global -> registers -> LDS through three buffers exactly as in halo16t_kernel; with C = 128 the second half of the region
//     replaces the first between two tiles (exposed: 2 of 66 tile times);
//   * 4 waves, each 2 rows x 32 pixels x N / 2 output channels; bias / activation in the epilogue, whole 128-byte lines per store.
// These layers are short in K (16 or 64 filter tiles per patch): by ablation (make exp builds, profiles/LOG.md) a 94 us launch of
// the 64 -> 128 layer at batch 32 spends 23 us reading regions, 21 us writing results and 13 us in the matrix pipe, one after the
// other -- every CU holds one workgroup (143 KB of LDS) and all of them change phase together.  A persistent variant with loader
// waves and a double-buffered 2 x 32 patch overlapped the phases but halved the work per fragment read and per barrier and came
// out even (92 us); what would pay is bf16 activations in HBM (region straight into LDS, half the bytes).
struct Halo16SParams {
  const void* src;             // [NB][2 Ho][2 Wo][C] fp32, or bf16 (IN16)
  const unsigned short* wp;    // packed bf16 filters [64-channel half][tap][chunk][N][8192 / N]
  const float* bias;           // [N] or null
  void* dst;                   // [NB][Ho][Wo][N] fp32, or bf16 (OUT16)
  int NB, Ho, Wo, tiles_y, tiles_x, act;
  float slope;
};

// IN16 / OUT16: the source / destination tensor is bf16 in HBM (round 4: the activations between the generator's down / up
// convolutions and their norms, ops.py): the region then needs one 16-byte load per 8 channels and no conversion.
template <int C, int BN, bool IN16 = false, bool OUT16 = false>
__global__ __launch_bounds__(256) void halo16s_kernel(Halo16SParams p) {
  constexpr int ISZ = IN16 ? 2 : 4;            // bytes per source element
  constexpr int PS = 64 * 2 + 16;              // bytes per region pixel (the resident 64-channel half)
  constexpr int NH = C / 64;                   // halves
  constexpr int HKT = 8192 / BN;               // reduce channels per filter tile
  constexpr int NCH = 64 / HKT;                // tiles per tap of a half
  constexpr int NKH = 16 * NCH;                // tiles per half
  constexpr int NK = NH * NKH;
  constexpr int KS = HKT / 16;                 // 16-deep K steps per tile
  constexpr int TN = BN / 64;                  // 32-wide output-channel blocks per wave
  constexpr int WTILE = 16384;
  constexpr int PCS = HKT / 8;                 // 16-byte pieces per tile row
  constexpr int SWS = PCS == 16 ? 0 : PCS == 8 ? 1 : 2;      // swizzle key = (row >> SWS) & (PCS - 1): one key per 256 bytes of rows
  constexpr int RPX = 10 * 66;
  static_assert((C == 64 && BN == 128) || (C == 128 && BN == 256), "instantiated for (C, N) = (64, 128) and (128, 256)");
  static_assert(KS % 2 == 0 && NKH % 2 == 0, "fragment slots alternate per K step, two tiles per loop body");
  __shared__ __attribute__((aligned(16))) unsigned char halo[RPX * PS];
  __shared__ __attribute__((aligned(16))) unsigned char wt[3 * WTILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);      // neighbouring patches (shared region rows) on one XCD
  int r0 = bid;
  const int tx = r0 % p.tiles_x; r0 /= p.tiles_x;
  const int ty = r0 % p.tiles_y;
  const int nb = r0 / p.tiles_y;
  const int Hi = 2 * p.Ho, Wi = 2 * p.Wo;
  const int Y0 = 8 * ty - 1, X0 = 64 * tx - 1;          // region origin in x

  const auto rs_x = uniform_rsrc(p.src, (unsigned)((size_t)p.NB * Hi * Wi * C * ISZ));
  const auto rs_w = uniform_rsrc(p.wp, (unsigned)(NK * WTILE));

  f32x4 wreg[4], wreg2[4];
  auto load_w = [&](f32x4* dst, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, tid * 16 + i * 4096, kt * WTILE, 0));
  };
  auto store_w = [&](const f32x4* src, int boff) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i;               // piece q: row q / PCS, piece q % PCS
      const int n = q / PCS, pc = q % PCS;
      *reinterpret_cast<f32x4*>(&wt[boff + n * (HKT * 2) + ((pc ^ ((n >> SWS) & (PCS - 1))) << 4)]) = src[i];
    }
  };
  load_w(wreg, 0);
  load_w(wreg2, 1);

  // ---- one 64-channel half of the region: 21 passes of 32 pixels x 8 channels per thread, seven passes in flight ----
  const int hcg = tid & 7, hpl = tid >> 3;
  constexpr unsigned kOutside = 0x80000000u;
  auto load_region = [&](int half) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      f32x4 lo[7], hi[7];
#pragma unroll
      for (int g = 0; g < 7; ++g) {
        const int hp = (b * 7 + g) * 32 + hpl;
        const int hr = hp / 66, hc = hp - hr * 66;
        const int y = Y0 + hr, x = X0 + hc;
        const bool ok = hp < RPX && y >= 0 && y < Hi && x >= 0 && x < Wi;
        const unsigned off = ok ? (unsigned)((((nb * Hi + y) * Wi + x) * C + half * 64 + hcg * 8) * ISZ) : kOutside;
        lo[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
        if constexpr (!IN16) hi[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 16, 0));
        else hi[g] = lo[g];
      }
#pragma unroll
      for (int g = 0; g < 7; ++g) {
        const int hp = (b * 7 + g) * 32 + hpl;
        if (hp < RPX) {
          const int hr = hp / 66, hc = hp - hr * 66;
          unsigned char* dst = &halo[((hr * 2 + (hc & 1)) * 33 + (hc >> 1)) * PS + hcg * 16];
          if constexpr (IN16) {                  // lo already holds the 8 bf16 channels
            *reinterpret_cast<f32x4*>(dst) = lo[g];
          } else {
            const bf16x4 a = __builtin_convertvector(lo[g], bf16x4), c = __builtin_convertvector(hi[g], bf16x4);
            bf16x8 v;
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = c[0]; v[5] = c[1]; v[6] = c[2]; v[7] = c[3];
            *reinterpret_cast<bf16x8*>(dst) = v;
          }
        }
      }
    }
  };
  load_region(0);
  store_w(wreg, 0);
  store_w(wreg2, WTILE);
  load_w(wreg, 2);
  load_w(wreg2, 3);
  __syncthreads();

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // lane's A address: output pixel (row 2 wm, column lr) at tap (0, 0) = region row 4 wm, plane 0, column lr; K half lh
  const unsigned char* a_lane = halo + (wm * 8 * 33 + lr) * PS + lh * 16;
  const int b_row = (wn * (BN / 2) + lr) * (HKT * 2), b_key = (lr >> SWS) & (PCS - 1);
  auto b_off = [&](int s) __attribute__((always_inline)) { return b_row + (((2 * s + lh) ^ b_key) << 4); };

  bf16x8 fa[2][2], fb[2][TN];
  auto read_a = [&](int slot, const unsigned char* a, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[slot][i] = *reinterpret_cast<const bf16x8*>(a + i * (4 * 33 * PS) + s * 32);
  };
  auto read_b = [&](int slot, int woff, int s) __attribute__((always_inline)) {
    const unsigned char* B = wt + woff + b_off(s);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const bf16x8*>(B + j * 32 * (HKT * 2));
  };
  auto mma = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot][i], fb[slot][j], acc[i][j], 0, 0, 0);
  };
  // A address of tile (tap, chunk): region row + ky, plane kx & 1, column + (kx >> 1); chunk * HKT channels
  auto a_of = [&](int tap, int ch) __attribute__((always_inline)) {
    const int ky = tap >> 2, kx = tap & 3;
    return a_lane + ((ky * 2 + (kx & 1)) * 33 + (kx >> 1)) * PS + ch * (HKT * 2);
  };

  int w_cur = 0, w_nxt = WTILE, w_nn = 2 * WTILE;
  int half = 0, tap = 0, ch = 0;               // coordinates of tile kt
  read_a(0, a_of(0, 0), 0);
  read_b(0, w_cur, 0);
  for (int kt = 0; kt < NK; kt += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = kt + u;
      const unsigned char* a_cur = a_of(tap, ch);
      int nch = ch + 1, ntap = tap, nhalf = half;
      if (nch == NCH) { nch = 0; ntap = tap + 1; if (ntap == 16) { ntap = 0; nhalf = half + 1; } }
      const unsigned char* a_nx = a_of(ntap, nch);
      const bool reload = NH > 1 && nhalf != half && t + 1 < NK;      // the region changes before the next tile
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
          read_a((s + 1) & 1, a_cur, s + 1);
          read_b((s + 1) & 1, w_cur, s + 1);
        } else {
          if (t + 1 < NK) {
            if (!reload) read_a(0, a_nx, 0);
            read_b(0, w_nxt, 0);
          }
          if (t + 2 < NK) store_w(u == 0 ? wreg : wreg2, w_nn);
          if (t + 4 < NK) load_w(u == 0 ? wreg : wreg2, t + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(s & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      if (reload) {                            // (wave-uniform) every wave is done with the old half
        load_region(nhalf);
        __syncthreads();
        read_a(0, a_nx, 0);
      }
      const int tw = w_cur; w_cur = w_nxt; w_nxt = w_nn; w_nn = tw;
      half = nhalf; tap = ntap; ch = nch;
    }
  }

  // ---- epilogue: lane = output channel; register e of acc[i][j] = pixel column (e % 4) + 8 (e / 4) + 4 lh of output row
  // 4 ty + 2 wm + i (16-byte stores through an LDS transpose were measured: no gain -- the write phase is bound by memory) ----
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const size_t row = ((size_t)(nb * p.Ho + 4 * ty + 2 * wm + i) * p.Wo + 32 * tx) * BN;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = wn * (BN / 2) + j * 32 + lr;
      const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = apply_act(acc[i][j][e] + bv, p.act, p.slope);
        const size_t o = row + (size_t)((e & 3) + 8 * (e >> 2) + 4 * lh) * BN + n;
        if constexpr (OUT16) static_cast<__bf16*>(p.dst)[o] = (__bf16)v;
        else static_cast<float*>(p.dst)[o] = v;
      }
    }
  }
}

// C = 256: halo16r_kernel (filter operand from the register image)
int launch_r(const Halo16Params& p, bool in16, bool out16, long long grid, hipStream_t st) {
  const dim3 g((unsigned)grid), b(256);
  if (!in16 && !out16) {
    if (p.res) hipLaunchKernelGGL((halo16r_kernel<true, false, false>), g, b, 0, st, p);
    else hipLaunchKernelGGL((halo16r_kernel<false, false, false>), g, b, 0, st, p);
  } else if (!in16 && out16) {
    hipLaunchKernelGGL((halo16r_kernel<false, false, true>), g, b, 0, st, p);
  } else if (in16 && out16) {
    hipLaunchKernelGGL((halo16r_kernel<false, true, true>), g, b, 0, st, p);
  } else {
    if (p.res) hipLaunchKernelGGL((halo16r_kernel<true, true, false>), g, b, 0, st, p);
    else hipLaunchKernelGGL((halo16r_kernel<false, true, false>), g, b, 0, st, p);
  }
  return 0;
}

template <int C>
int launch_c(const Halo16Params& p, bool in16, bool out16, long long grid, hipStream_t st) {
  const dim3 g((unsigned)grid), b(256);
  if (!in16 && !out16) {
    if (p.res) hipLaunchKernelGGL((halo16_kernel<C, C, true, false, false>), g, b, 0, st, p);
    else hipLaunchKernelGGL((halo16_kernel<C, C, false, false, false>), g, b, 0, st, p);
  } else if (!in16 && out16) {
    hipLaunchKernelGGL((halo16_kernel<C, C, false, false, true>), g, b, 0, st, p);
  } else if (in16 && out16) {
    hipLaunchKernelGGL((halo16_kernel<C, C, false, true, true>), g, b, 0, st, p);
  } else {
    if (p.res) hipLaunchKernelGGL((halo16_kernel<C, C, true, true, false>), g, b, 0, st, p);
    else hipLaunchKernelGGL((halo16_kernel<C, C, false, true, false>), g, b, 0, st, p);
  }
  return 0;
}

}  // namespace

// the layer shapes the kernel is instantiated for: square channel counts 64 / 128 / 256, maps of whole 4 x 32 patches
bool halo16_applicable(const srgan_conv_desc* d, int kind) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_HALO16");
  if (off || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != SRGAN_PAD_ZERO) return false;
  if (d->I != d->O || !(d->I == 64 || d->I == 128 || d->I == 256)) return false;
  if (d->Hi % 4 != 0 || d->Wi % 32 != 0 || d->Ho != d->Hi || d->Wo != d->Wi) return false;
  (void)kind;
  return (long long)d->N * d->Hi * d->Wi * d->I < (1LL << 29);      // 32-bit byte offsets into the source
}

size_t halo16_packed_bytes(const srgan_conv_desc* d) { return (size_t)9 * d->I * d->O * 2; }

// src: x (kind 0) or dy (kind 1), both [N][H][W][C]; dst likewise with the other channel count (equal here)
int halo16_run(const srgan_conv_desc* d, int kind, const void* src, const void* packed, const float* bias, const float* res,
               void* dst, int act, float slope, double flops, hipStream_t st, bool in16, bool out16) {
  SRGAN_REQUIRE(halo16_applicable(d, kind), "halo16: layer not applicable");
  SRGAN_REQUIRE(!(res && out16), "halo16: the skip gradient is added to an fp32 result only");
  Halo16Params p{};
  p.src = src; p.wp = reinterpret_cast<const unsigned short*>(packed); p.bias = bias; p.res = res; p.dst = dst;
  p.NB = d->N; p.H = d->Hi; p.W = d->Wi; p.N = d->I;
  p.tiles_y = d->Hi / 4; p.tiles_x = d->Wi / 32; p.n_tiles = 1; p.act = act; p.slope = slope;
  const long long grid = (long long)p.NB * p.tiles_y * p.tiles_x * p.n_tiles;
  SRGAN_REQUIRE(grid > 0 && grid < (1LL << 31), "halo16: grid");
  ProfToken tok = prof_begin(24, flops, st);
  if (d->I == 256) launch_r(p, in16, out16, grid, st);
  else if (d->I == 128) launch_c<128>(p, in16, out16, grid, st);
  else launch_c<64>(p, in16, out16, grid, st);
  prof_end(tok, st);
  return check_launch("halo16_kernel");
}

// ---- transposed 4x4 / stride-2 form: kind 1 of a strided layer d (its input gradient = a ConvTranspose2d forward) ----
bool halo16t_applicable(const srgan_conv_desc* d) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_HALO16T");
  if (off || d->kh != 4 || d->kw != 4 || d->stride != 2 || d->pad != 1 || d->pad_mode != SRGAN_PAD_ZERO) return false;
  if (!((d->O == 256 && d->I == 128) || (d->O == 128 && d->I == 64))) return false;
  if (d->Hi != 2 * d->Ho || d->Wi != 2 * d->Wo || d->Ho % 4 != 0 || d->Wo % 32 != 0) return false;
  return (long long)d->N * d->Ho * d->Wo * d->O < (1LL << 29) && (long long)d->N * d->Hi * d->Wi * d->I < (1LL << 30);
}

size_t halo16t_packed_bytes(const srgan_conv_desc* d) { return (size_t)16 * d->I * d->O * 2; }

int halo16t_run(const srgan_conv_desc* d, const void* dy, const void* packed, void* dx, double flops, hipStream_t st, bool in16,
                bool out16) {
  SRGAN_REQUIRE(halo16t_applicable(d), "halo16t: layer not applicable");
  Halo16TParams p{};
  p.src = dy; p.wp = reinterpret_cast<const unsigned short*>(packed); p.dst = dx;
  p.NB = d->N; p.Hs = d->Ho; p.Ws = d->Wo; p.tiles_y = d->Ho / 4; p.tiles_x = d->Wo / 32;
  const long long grid = (long long)p.NB * p.tiles_y * p.tiles_x;
  SRGAN_REQUIRE(grid > 0 && grid < (1LL << 31), "halo16t: grid");
  ProfToken tok = prof_begin(27, flops, st);
#define SRGAN_H16T(C_, N_)                                                                                            \
  do {                                                                                                                \
    if (in16 && out16) hipLaunchKernelGGL((halo16t_kernel<C_, N_, true, true>), dim3((unsigned)grid), dim3(256), 0, st, p);        \
    else if (in16) hipLaunchKernelGGL((halo16t_kernel<C_, N_, true, false>), dim3((unsigned)grid), dim3(256), 0, st, p);          \
    else if (out16) hipLaunchKernelGGL((halo16t_kernel<C_, N_, false, true>), dim3((unsigned)grid), dim3(256), 0, st, p);         \
    else hipLaunchKernelGGL((halo16t_kernel<C_, N_, false, false>), dim3((unsigned)grid), dim3(256), 0, st, p);                   \
  } while (0)
  if (d->O == 256) SRGAN_H16T(256, 128);
  else SRGAN_H16T(128, 64);
#undef SRGAN_H16T
  prof_end(tok, st);
  return check_launch("halo16t_kernel");
}

// ---- strided 4x4 / stride-2 form: kind 0 of the layer (variant 6 of conv_wino.hip's slot) ----
bool halo16s_applicable(const srgan_conv_desc* d) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_HALO16S");
  if (off || d->kh != 4 || d->kw != 4 || d->stride != 2 || d->pad != 1 || d->pad_mode != SRGAN_PAD_ZERO) return false;
  if (!((d->I == 64 && d->O == 128) || (d->I == 128 && d->O == 256))) return false;
  if (d->Hi != 2 * d->Ho || d->Wi != 2 * d->Wo || d->Ho % 4 != 0 || d->Wo % 32 != 0) return false;
  return (long long)d->N * d->Hi * d->Wi * d->I < (1LL << 29) && (long long)d->N * d->Ho * d->Wo * d->O < (1LL << 30);
}

size_t halo16s_packed_bytes(const srgan_conv_desc* d) { return (size_t)16 * d->I * d->O * 2; }

int halo16s_run(const srgan_conv_desc* d, const void* x, const void* packed, const float* bias, void* y, int act, float slope,
                double flops, hipStream_t st, bool in16, bool out16) {
  SRGAN_REQUIRE(halo16s_applicable(d), "halo16s: layer not applicable");
  Halo16SParams p{};
  p.src = x; p.wp = reinterpret_cast<const unsigned short*>(packed); p.bias = bias; p.dst = y;
  p.NB = d->N; p.Ho = d->Ho; p.Wo = d->Wo; p.tiles_y = d->Ho / 4; p.tiles_x = d->Wo / 32; p.act = act; p.slope = slope;
  const long long grid = (long long)p.NB * p.tiles_y * p.tiles_x;
  SRGAN_REQUIRE(grid > 0 && grid < (1LL << 31), "halo16s: grid");
  ProfToken tok = prof_begin(29, flops, st);
#define SRGAN_H16S(C_, N_)                                                                                            \
  do {                                                                                                                \
    if (in16 && out16) hipLaunchKernelGGL((halo16s_kernel<C_, N_, true, true>), dim3((unsigned)grid), dim3(256), 0, st, p);        \
    else if (in16) hipLaunchKernelGGL((halo16s_kernel<C_, N_, true, false>), dim3((unsigned)grid), dim3(256), 0, st, p);          \
    else if (out16) hipLaunchKernelGGL((halo16s_kernel<C_, N_, false, true>), dim3((unsigned)grid), dim3(256), 0, st, p);         \
    else hipLaunchKernelGGL((halo16s_kernel<C_, N_, false, false>), dim3((unsigned)grid), dim3(256), 0, st, p);                   \
  } while (0)
  if (d->I == 64) SRGAN_H16S(64, 128);
  else SRGAN_H16S(128, 256);
#undef SRGAN_H16S
  prof_end(tok, st);
  return check_launch("halo16s_kernel");
}

// ---- weight gradient host side (hooked into conv_wino.hip's weight-gradient slot in bf16 mode) ----
static void halo16_wgrad_plan(const srgan_conv_desc* d, Halo16WgradParams* p) {
  p->NB = d->N; p->H = d->Hi; p->W = d->Wi; p->C = d->I; p->O = d->O;
  p->tiles_y = d->Hi / 4; p->tiles_x = d->Wi / 32;
  p->patches = d->N * p->tiles_y * p->tiles_x;
  p->o_tiles = d->O / 64; p->c_tiles = d->I / 64;
  const int tiles = p->o_tiles * p->c_tiles;
  int splits = std::max(1, std::min(p->patches, 256 / tiles));      // one workgroup per CU (127 KB of LDS)
  p->per_split = (p->patches + splits - 1) / splits;
  p->splits = (p->patches + p->per_split - 1) / p->per_split;
}

// 4x4 / stride-2 / pad-1 layers with whole 64-channel blocks on both sides and output maps of whole 2 x 32 patches
static bool halo16s2_wgrad_ok(const srgan_conv_desc* d) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_HALO16_S2_WGRAD");
  if (off || d->kh != 4 || d->kw != 4 || d->stride != 2 || d->pad != 1 || d->pad_mode != SRGAN_PAD_ZERO) return false;
  if (d->I % 64 != 0 || d->O % 64 != 0 || d->Hi != 2 * d->Ho || d->Wi != 2 * d->Wo) return false;
  if (!((d->Wo % 32 == 0 && d->Ho % 2 == 0) || (d->Wo == 16 && d->Ho % 4 == 0) || (d->Wo == 8 && d->Ho % 8 == 0))) return false;
  return (long long)d->N * d->Hi * d->Wi * d->I < (1LL << 29) && (long long)d->N * d->Ho * d->Wo * d->O < (1LL << 29);
}

static void halo16s2_wgrad_plan(const srgan_conv_desc* d, Halo16S2WgradParams* p) {
  p->NB = d->N; p->Ho = d->Ho; p->Wo = d->Wo; p->C = d->I; p->O = d->O;
  const int pw = d->Wo % 32 == 0 ? 32 : d->Wo;
  p->tiles_y = d->Ho / (64 / pw); p->tiles_x = d->Wo / pw;
  p->patches = d->N * p->tiles_y * p->tiles_x;
  p->o_tiles = d->O / 64; p->c_tiles = d->I / 64;
  const int tiles = p->o_tiles * p->c_tiles;
  int splits = std::max(1, std::min(p->patches, 256 / tiles));      // one workgroup per CU (148 KB of LDS)
  p->per_split = (p->patches + splits - 1) / splits;
  p->splits = (p->patches + p->per_split - 1) / p->per_split;
}

bool halo16_wgrad_applicable(const srgan_conv_desc* d) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_HALO16_WGRAD");
  return (!off && halo16_applicable(d, 0)) || halo16s2_wgrad_ok(d);
}

void halo16_wgrad_slab(const srgan_conv_desc* d, int* splits, int* Cdpad, int* NNpad) {
  if (halo16s2_wgrad_ok(d)) {
    Halo16S2WgradParams q{};
    halo16s2_wgrad_plan(d, &q);
    *splits = q.splits; *Cdpad = d->O; *NNpad = 16 * d->I;
    return;
  }
  Halo16WgradParams p{};
  halo16_wgrad_plan(d, &p);
  *splits = p.splits; *Cdpad = d->O; *NNpad = 9 * d->I;
}

int halo16_wgrad_run(const srgan_conv_desc* d, const void* x, const void* dy, float* slab, double flops, hipStream_t st, bool x16,
                     bool d16) {
  SRGAN_REQUIRE(halo16_wgrad_applicable(d), "halo16 wgrad: layer not applicable");
  if (halo16s2_wgrad_ok(d)) {
    Halo16S2WgradParams q{};
    halo16s2_wgrad_plan(d, &q);
    q.x = x; q.dy = dy; q.slab = slab;
    ProfToken tok = prof_begin(26, flops, st);
    const dim3 grid((unsigned)(q.o_tiles * q.c_tiles * q.splits));
#define SRGAN_S2W(PW_)                                                                                             \
  do {                                                                                                                \
    if (x16 && d16) hipLaunchKernelGGL((halo16s2_wgrad_kernel<PW_, true, true>), grid, dim3(512), 0, st, q);          \
    else if (x16) hipLaunchKernelGGL((halo16s2_wgrad_kernel<PW_, true, false>), grid, dim3(512), 0, st, q);           \
    else if (d16) hipLaunchKernelGGL((halo16s2_wgrad_kernel<PW_, false, true>), grid, dim3(512), 0, st, q);           \
    else hipLaunchKernelGGL((halo16s2_wgrad_kernel<PW_, false, false>), grid, dim3(512), 0, st, q);                  \
  } while (0)
    if (d->Wo % 32 == 0) SRGAN_S2W(32);
    else if (d->Wo == 16) SRGAN_S2W(16);
    else SRGAN_S2W(8);
#undef SRGAN_S2W
    prof_end(tok, st);
    return check_launch("halo16s2_wgrad_kernel");
  }
  Halo16WgradParams p{};
  halo16_wgrad_plan(d, &p);
  p.x = x; p.dy = dy; p.slab = slab;
  ProfToken tok = prof_begin(25, flops, st);
  const dim3 grid((unsigned)(p.o_tiles * p.c_tiles * p.splits));
  if (!x16 && !d16) hipLaunchKernelGGL((halo16_wgrad_kernel<false, false>), grid, dim3(256), 0, st, p);
  else if (!x16 && d16) hipLaunchKernelGGL((halo16_wgrad_kernel<false, true>), grid, dim3(256), 0, st, p);
  else if (x16 && d16) hipLaunchKernelGGL((halo16_wgrad_kernel<true, true>), grid, dim3(256), 0, st, p);
  else { set_error("halo16 wgrad: bf16 x with fp32 dy is not instantiated"); return -1; }
  prof_end(tok, st);
  return check_launch("halo16_wgrad_kernel");
}

}  // namespace srgan

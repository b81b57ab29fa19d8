// Winograd F(4x4,2x2) for the 4x4 / stride-2 / pad-1 layers of G and D and their transposed forms (reference
// pyfiles/model.py:212-215, 227-230, 302-309) whose output maps (strided form) / source maps (transposed form) are multiples of
// 4 in both directions -- every such layer of the 128x128 and 256x256 configurations.
//
//   A 4x4 / stride-2 convolution is the sum over the four input phases I_pq[u][v] = in[2u+p-1][2v+q-1] of a 2x2 / stride-1
//   correlation with g_pq[a][b] = w[2a+p][2b+q]; its transpose is one 2x2 / stride-1 correlation per OUTPUT phase.  On a 2x2
//   correlation          Y = A^T [ sum_k (G g G^T) .* (B^T d B) ] A,      4x4 output tile, 5x5 patch, 25 positions
//   (interpolation points 0, 1, -1, 2, inf): 25 multiplies per 16 outputs of a phase instead of the 16 per 9 of F(3x3,2x2)
//   (conv_wino.hip) -- 12 % fewer MFMAs per output -- and 4x4 tiles cover the 64 / 32 / 16-pixel maps exactly where 3x3 tiles
//   compute 66 x 66 outputs for a 64 x 64 map.  Every product is still an exact fp32 product on v_mfma_f32_32x32x2_f32.
//
// One fused kernel per direction (MODE 1: strided form, reduce index = (input phase, channel); MODE 2: transposed form,
// blockIdx.y = output phase, outputs scattered with pixel stride 2), same skeleton as wino_kernel:
//   * a workgroup (8 waves) owns 32 tiles x 64 output channels for all 25 positions (100 accumulator registers per lane; 64
//     tiles would need 200).  The transform of a patch is amortised over 64 output channels as in wino_kernel: what the fp32
//     matrix pipe cannot hide is the vector work per MFMA (the fp32 MFMA runs on the vector lanes, profiles/LOG.md), and per
//     OUTPUT this kernel gathers / stores 0.83x and adds 1.5x of what F(3x3,2x2) does, next to 0.83x the MFMAs;
//   * per 16-channel chunk each thread gathers one (tile, channel) 5x5 patch (buffer loads; out-of-image taps point past the
//     buffer), transforms it in place (10 x 9 operations) and writes the 25 results to LDS [pos][channel quad][tile][4];
//   * wave = (output-channel half, one of four groups of 6 of the first 24 positions): per chunk 2 x 6 units of four MFMAs
//     with A = U fragment (32 output channels x 2 reduce channels, straight from the packed image into registers, a ring of
//     6 fragments reloaded in place half a chunk ahead) and B = V fragment (one ds_read_b128); the 25th position is cut into
//     eight 16 x 16 blocks, one per wave, on v_mfma_f32_16x16x4_f32: every wave runs the same code and every SIMD carries the
//     same 6400 MFMA cycles per chunk (a 7 / 6 / 6 / 6 split of whole positions measured the same time: the chunk takes ~7800
//     cycles = MFMAs + the transform's vector work + barrier skew either way -- kept for the single code path);
//   * the pieces of the next chunk's transform, its LDS stores and the gather of the chunk after it sit between the units of
//     the SAME wave; LDS double-buffered, one barrier per chunk, placed before the last unit so that the first fragment of the
//     next chunk is read under that unit's MFMAs;
//   * epilogue: an accumulator lane is a tile and its registers are output channels; the two output-channel halves meet in LDS
//     one after the other ([pos][tile][32 + 4]), the four waves whose accumulators just left apply A^T . A (thread = (tile, 4
//     channels): 16-byte LDS reads, the 8 lanes of a tile store one 128-byte line per pixel), bias / activation / MASK as in
//     wino_kernel.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "pack_device.h"

#ifndef WINO42_EXP
#define WINO42_EXP 0
#endif

namespace srgan {

constexpr int W2T = 32;                 // tiles (4x4 outputs of a phase image) per workgroup
constexpr int W2N = 64;                 // output channels per workgroup
constexpr int W2C = 16;                 // reduce channels per chunk
constexpr int W2QS = W2T * 4 + 8;       // LDS stride of a channel quad: the four quads a gather wave writes land on different banks
constexpr int W2PS = 4 * W2QS;          // LDS stride of a position
constexpr int W2VSZ = 25 * W2PS;        // one V buffer (floats)
constexpr int W2XT = 36;                // epilogue image: [25 pos][32 tiles][32 channels + 4 pad]
constexpr int W2XP = W2T * W2XT;
constexpr int W2LDS = 25 * W2XP > 2 * W2VSZ ? 25 * W2XP : 2 * W2VSZ;
constexpr int W2UCH = 25 * 2 * 2 * 256; // floats of one chunk of the packed filter image: [25 pos][2 halves of 8 ch][2 cout halves][64 lanes][4]

__device__ __forceinline__ auto uniform_rsrc42(const float* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  float* q = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// B^T of F(4,2) (points 0, 1, -1, 2, inf) on a 5-vector, in place, 9 operations:
//   [2 -1 -2 1 0; 0 -2 -1 1 0; 0 2 -3 1 0; 0 -1 0 1 0; 0 2 -1 -2 1]
__device__ __forceinline__ void bt5(float& x0, float& x1, float& x2, float& x3, float& x4) {
  const float t1 = x3 - x1, t0 = x0 - x2, s = x3 - x2, e = x1 - x2, f = x4 - x2;
  x0 = __builtin_fmaf(2.f, t0, t1);
  x1 = __builtin_fmaf(-2.f, x1, s);
  x2 = __builtin_fmaf(2.f, e, s);
  x3 = t1;
  x4 = __builtin_fmaf(-2.f, t1, f);
}

// A^T of F(4,2): [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 0; 0 1 -1 8 1]
template <typename V>
__device__ __forceinline__ void at5(const V& m0, const V& m1, const V& m2, const V& m3, const V& m4, V* y) {
  const V s = m1 + m2, d = m1 - m2;
  y[0] = m0 + s + m3;
  y[1] = d + 2.f * m3;
  y[2] = s + 4.f * m3;
  y[3] = d + 8.f * m3 + m4;
}

// MODE 1: strided form (kind 0 of a 4x4 / stride-2 / pad-1 layer): tiles over the Ho x Wo output map, K = (input phase, channel).
// MODE 2: transposed form (kind 1): blockIdx.y = output phase (r, s), tiles over the source map (= the phase image of the
//         destination), taps w[3-2a-r][3-2b-s] (packed per phase), outputs at (2u + r, 2v + s).
// MASK (MODE 2): as wino_kernel<2, true> -- the LeakyReLU backward of the previous layer in the epilogue.
template <int MODE, bool MASK = false>
__global__ __launch_bounds__(512) void wino42_kernel(WinoParams p) {
  static_assert(MODE == 1 || MODE == 2, "strided or transposed form");
  static_assert(!MASK || MODE == 2, "the mask epilogue exists for the transposed form only");
  __shared__ __attribute__((aligned(16))) float lds[W2LDS];
  __shared__ int tile_o[W2T];            // destination pixel index of the tile's first output, or -1
  constexpr int PXS = MODE == 2 ? 2 : 1;  // pixel stride of the outputs in the destination
  constexpr int TS = MODE == 1 ? 8 : 4;   // source pixels between tile origins
  constexpr int DS = MODE == 1 ? 2 : 1;   // source pixels between patch elements
  constexpr int NPH = MODE == 2 ? 4 : 1;  // output phases = items per (tile block, channel block) unit

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  // A workgroup walks a contiguous range of items = (tile block, output-channel block, output phase), phase fastest: the items of
  // a range read the same (MODE 2: one-pixel-shifted) patches.  XCD-aware order of the ranges: see wino_kernel.
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int n_items = p.m_tiles * p.n_tiles * NPH;
  const int item0 = bid * p.ipw, item1 = min(item0 + p.ipw, n_items);
  const int nk = p.nchunk;
#if WINO42_EXP & 128
  long long stamp[12];
  auto tick = [&](int k) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    stamp[k] = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
  };
  tick(0);
#else
  auto tick = [&](int) __attribute__((always_inline)) {};
#endif

  // ---- gather role: thread = (tile tl, channel ch of the chunk) ----
  const int tl = tid >> 4, ch = tid & 15;
  constexpr unsigned kOutside = 0x80000000u;
  int g_b = 0, g_ty = 0, g_tx = 0, g_mt = -1, g_to = -1;
  bool g_tv = false;
  int ph_r = 0, ph_s = 0, n_tile = 0;
  // Patch element (i, j) = source pixel (TS ty + DS i + o_y, TS tx + DS j + o_x), o = phase - 1.  Only i, j = 0 can lie above /
  // left of the image and only i, j = 4 below / right of it (maps are multiples of the tile): per row one offset for columns
  // 1..3 (the column goes into the scalar offset), one for column 0 and one for column 4, each the element's byte offset or a
  // value past the buffer (the range check then supplies the zero padding without a select).
  const int CS = DS * p.C * 4;           // bytes between patch columns
  unsigned offm[5], off0[5], off4[5];
  auto set_offsets = [&](int phase) __attribute__((always_inline)) {
    const int oy = (MODE == 1 ? (phase >> 1) : ph_r) - 1, ox = (MODE == 1 ? (phase & 1) : ph_s) - 1;
    const int ix0 = TS * g_tx + ox;
    const bool c0ok = ix0 >= 0, c4ok = ix0 + 4 * DS < p.W;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int iy = TS * g_ty + DS * i + oy;
      const bool rok = g_tv && (unsigned)iy < (unsigned)p.H;
      const unsigned base = (unsigned)(((g_b * p.H + iy) * p.W + ix0 + DS) * p.C + ch) * 4u;      // column 1
      offm[i] = rok ? base : kOutside;
      off0[i] = rok && c0ok ? base - (unsigned)CS : kOutside;
      off4[i] = rok && c4ok ? base + 3u * (unsigned)CS : kOutside;
    }
  };
  int lc = 0, lphase = 0;                 // load cursor: chunks are consumed strictly in order
  // coordinates, gather offsets and load cursor of an item (the tile decode only when the tile block changes)
  auto setup_item = [&](int item) __attribute__((always_inline)) {
    const int unit = MODE == 2 ? item >> 2 : item, oph = MODE == 2 ? item & 3 : 0;
    const int m_tile = unit / p.n_tiles;
    n_tile = unit - m_tile * p.n_tiles;
    ph_r = oph >> 1; ph_s = oph & 1;
    if (m_tile != g_mt) {
      g_mt = m_tile;
      const int t = m_tile * W2T + tl;
      g_tv = t < p.T;
      const int tt = g_tv ? t : 0;
      const int per = p.TH * p.TW;
      g_b = tt / per;
      const int r = tt - g_b * per;
      g_ty = r / p.TW; g_tx = r - g_ty * p.TW;
    }
    g_to = g_tv ? (g_b * p.Ho + PXS * 4 * g_ty + ph_r) * p.Wo + PXS * 4 * g_tx + ph_s : -1;
    lc = 0; lphase = 0;
    set_offsets(0);
  };
  const unsigned src_bytes = (unsigned)((size_t)p.NB * p.H * p.W * p.C * 4);
  const auto rs_x = uniform_rsrc42(p.src, src_bytes);
  // destination (and mask) through buffer instructions too: one 32-bit offset per thread, the pixel of the tile in the scalar
  // offset -- sixteen 64-bit store addresses would cost 32 registers beside the output transform
  const unsigned dst_bytes = (unsigned)((size_t)p.NB * p.Ho * p.Wo * p.Cd * 4);
  const auto rs_y = uniform_rsrc42(p.dst, dst_bytes);
  const auto rs_m = uniform_rsrc42(MASK ? p.mask : p.dst, dst_bytes);

  float d[25];
  auto ldx = [&](unsigned vo, int so) __attribute__((always_inline)) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, vo, so, 0));
  };
  auto load_row = [&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
#if WINO42_EXP & 1
    if (lc + lphase > 0) return;          // ablation: only the first chunk is gathered
#endif
    const int so = lc * (W2C * 4);
    d[i * 5 + 0] = ldx(off0[i], so);
    d[i * 5 + 1] = ldx(offm[i], so);
    d[i * 5 + 2] = ldx(offm[i], so + CS);
    d[i * 5 + 3] = ldx(offm[i], so + 2 * CS);
    d[i * 5 + 4] = ldx(off4[i], so);
    if constexpr (i == 4) {
      ++lc;
      if (MODE == 1 && lc == p.cpp) {     // next input phase: new patch origin and padding (wave-uniform, 3x per item)
        lc = 0;
        ++lphase;
        if (lphase < 4) set_offsets(lphase);
      }
    }
  };
  auto col_pass = [&](int j) __attribute__((always_inline)) {
#if !(WINO42_EXP & 2)
    bt5(d[0 * 5 + j], d[1 * 5 + j], d[2 * 5 + j], d[3 * 5 + j], d[4 * 5 + j]);
#endif
  };
  const int vst = (ch >> 2) * W2QS + tl * 4 + (ch & 3);
  auto row_store = [&](int buf, int r) __attribute__((always_inline)) {
#if !(WINO42_EXP & 2)
    bt5(d[r * 5 + 0], d[r * 5 + 1], d[r * 5 + 2], d[r * 5 + 3], d[r * 5 + 4]);
#endif
#if !(WINO42_EXP & 4)
    float* V = lds + buf * W2VSZ + vst + r * 5 * W2PS;
#pragma unroll
    for (int c = 0; c < 5; ++c) V[c * W2PS] = d[r * 5 + c];
#endif
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;
  auto load_patch = [&]() __attribute__((always_inline)) {
    load_row(I0{}); load_row(I1{}); load_row(I2{}); load_row(I3{}); load_row(I4{});
  };
  // the side work of one iteration in eight pieces: transform + store of chunk kc + 1 (ST), gather of chunk kc + 2 (LD); a
  // row of d is reloaded as soon as its row pass has left for LDS
  auto side = [&](int nbuf, auto piece_c, auto st_c, auto ld_c) __attribute__((always_inline)) {
    constexpr int PC = decltype(piece_c)::value;
    constexpr bool ST = decltype(st_c)::value, LD = decltype(ld_c)::value;
    if constexpr (ST) {
      if constexpr (PC == 0) { col_pass(0); col_pass(1); }
      if constexpr (PC == 1) { col_pass(2); col_pass(3); }
      if constexpr (PC == 2) { col_pass(4); row_store(nbuf, 0); }
      if constexpr (PC >= 3 && PC <= 6) row_store(nbuf, PC - 2);
    }
    if constexpr (LD) {
      if constexpr (PC >= 3 && PC <= 7) load_row(std::integral_constant<int, (PC >= 3 && PC <= 7) ? PC - 3 : 0>{});
    }
  };

  using T = std::true_type;
  using F = std::false_type;
  if (item0 >= item1) return;

  // ---- multiply role: wave = (output-channel half h, position group g of 6 of the first 24 positions) on the 32x32x2 MFMA;
  // the 25th position is cut into eight 16 x 16 (output channel, tile) blocks, one per wave, on v_mfma_f32_16x16x4_f32 (same
  // FLOPs per cycle): every SIMD carries 2 x (12 units x 256 + 128) = 6400 MFMA cycles per chunk -- with positions dealt 7 / 6 /
  // 6 / 6 two SIMDs carried 6656 ----
  {
    constexpr int NP = 6;
    constexpr int NU = 2 * NP;            // units per chunk: (position slot, 8-channel half), half-major
    const int g = wave >> 1, h = wave & 1;
    const int pbase = NP * g;
    const int cb25 = wave >> 1, tb25 = wave & 1;      // block of position 24: output channels 16 cb25 .., tiles 16 tb25 ..
    f32x16 acc[NP];
    f32x4 ufr[NP];
    f32x4 acc25, u25;
    float v25[4];
    const unsigned ulane = (unsigned)((pbase * 4 + h) * 256 + lane * 4) * 4u;
    // filter image of the item: [output phase][n_tile][chunk][25 pos][2 halves of 8 ch][2 cout halves][64 lanes][4]
    auto u_rsrc = [&]() __attribute__((always_inline)) {
      return uniform_rsrc42(p.u + ((size_t)(ph_r * 2 + ph_s) * p.n_tiles + n_tile) * p.nchunk * W2UCH, (unsigned)p.nchunk * (unsigned)(W2UCH * 4));
    };
    auto rs_u = u_rsrc();
    auto load_u = [&](int slot, int hb, int kc) __attribute__((always_inline)) {
#if WINO42_EXP & 8
      if (kc > 0) return;                 // ablation: filter fragments of the first chunk only
#endif
      ufr[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, ulane, kc * (W2UCH * 4) + slot * 4096 + hb * 2048, 0));
    };
    // position 24's filter block: [4 channel blocks][lane = (channel & 3) * 16 + cout][4 channel quads]
    const unsigned ulane25 = (unsigned)(24 * 1024 + cb25 * 256 + lane * 4) * 4u;
    auto load_u25 = [&](int kc) __attribute__((always_inline)) {
#if WINO42_EXP & 8
      if (kc > 0) return;
#endif
      u25 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, ulane25, kc * (W2UCH * 4), 0));
    };
    // its V fragment: lane (tile l % 16, channel l / 16 of a quad) reads that channel of the four quads
    const int vrd25 = 24 * W2PS + (tb25 * 16 + (lane & 15)) * 4 + (lane >> 4);
    auto read_v25 = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 4; ++s) v25[s] = lds[buf * W2VSZ + vrd25 + s * W2QS];
    };
    const int vrd = pbase * W2PS + lh * W2QS + lr * 4;
    f32x4 vf[2];
    auto read_v = [&](int buf, int u) __attribute__((always_inline)) {
      return *reinterpret_cast<const f32x4*>(lds + buf * W2VSZ + vrd + (u % NP) * W2PS + (u / NP) * 2 * W2QS);
    };

    auto iter = [&](int kc, auto st_c, auto ld_c) __attribute__((always_inline)) {
      constexpr bool ST = decltype(st_c)::value;
      const int cur = kc & 1;
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const int slot = i % NP;
        if (i + 1 < NU) vf[(i + 1) & 1] = read_v(cur, i + 1);
        if (i == NU - 1) {
          read_v25(cur);
          // every store of chunk kc + 1 has been issued and every wave holds its last fragments of chunk kc
          __syncthreads();
          if constexpr (ST) vf[0] = read_v(cur ^ 1, 0);
        }
#if WINO42_EXP & 32
        asm volatile("" ::"v"(ufr[slot]), "v"(vf[i & 1]));      // ablation: no MFMAs (operands kept alive)
#else
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[slot] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[slot][s], vf[i & 1][s], acc[slot], 0, 0, 0);
#endif
        // the fragment this slot needs half a chunk from now
        if (i < NP) load_u(slot, 1, kc);
        else if constexpr (ST) load_u(slot, 0, kc + 1);
        if (i == 0) side(cur ^ 1, I0{}, st_c, ld_c);
        if (i == 1) side(cur ^ 1, I1{}, st_c, ld_c);
        if (i == 2) side(cur ^ 1, I2{}, st_c, ld_c);
        if (i == 3) side(cur ^ 1, I3{}, st_c, ld_c);
        if (i == 4) side(cur ^ 1, I4{}, st_c, ld_c);
        if (i == 5) side(cur ^ 1, std::integral_constant<int, 5>{}, st_c, ld_c);
        if (i == 6) side(cur ^ 1, std::integral_constant<int, 6>{}, st_c, ld_c);
        if (i == 7) side(cur ^ 1, std::integral_constant<int, 7>{}, st_c, ld_c);
        __builtin_amdgcn_sched_barrier(0);
      }
#if !(WINO42_EXP & 32)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc25 = __builtin_amdgcn_mfma_f32_16x16x4f32(u25[s], v25[s], acc25, 0, 0, 0);
#endif
      if constexpr (ST) load_u25(kc + 1);
      __builtin_amdgcn_sched_barrier(0);
    };

    // ---- epilogue: the two output-channel halves, one after the other.  The half is a compile-time constant of the code a
    // wave runs, so that its accumulators are provably dead once they have left for LDS (the transform needs ~180 registers) ----
    // e_nt / e_mask0: output-channel block and destination of THIS item (the gather state already belongs to the next one)
    // position 24: lane = tile 16 tb25 + l % 16, registers = output channels 16 cb25 + 4 (l / 16) + r; the waves whose block lies
    // in half hp (cb25 >> 1 == hp) write it while that half's image is open for writing
    auto store25 = [&]() __attribute__((always_inline)) {
      *reinterpret_cast<f32x4*>(lds + 24 * W2XP + (tb25 * 16 + (lane & 15)) * W2XT + (cb25 & 1) * 16 + (lane >> 4) * 4) = acc25;
    };
    auto pass = [&](int hp, int e_nt, auto mid_c) __attribute__((always_inline)) {
      if ((cb25 >> 1) == hp) store25();
      // accumulator lane = tile lr, register e = output channel 8 * (e / 4) + 4 * lh + e % 4 of the half
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        float* X = lds + (pbase + i) * W2XP + lr * W2XT + lh * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4 v = {acc[i][4 * k + 0], acc[i][4 * k + 1], acc[i][4 * k + 2], acc[i][4 * k + 3]};
          *reinterpret_cast<f32x4*>(X + 8 * k) = v;
        }
      }
      tick(3);
      __syncthreads();
      tick(4);
      // (derived from an opaque copy of the lane index: computed here, not carried through the main loop in registers)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int tq = (wave >> 1) * 64 + ln;         // index among the 256 threads that transform a pass
      const int cq = tq & 7, et = tq >> 3;
      const int n = e_nt * W2N + hp * 32 + cq * 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
      const int o = tile_o[et];
      // MASK: row 0 is requested before the LDS reads and the output transform, row a + 1 before row a is computed and stored
      // (all sixteen up front do not fit beside the transform's registers)
      f32x4 mk[4][4];
      const unsigned yoff = o >= 0 ? (unsigned)(o * p.Cd + n) * 4u : kOutside;      // outside: loads return 0, stores are dropped
      auto load_mask = [&](int a) __attribute__((always_inline)) {
#pragma unroll
        for (int b2 = 0; b2 < 4; ++b2)
          mk[a][b2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_m, yoff + (unsigned)((a * p.Wo + b2) * PXS * p.Cd * 4), 0, 0));
      };
      if constexpr (MASK) load_mask(0);
      // Output transform of the thread's four channels as two PAIRS, one after the other: the first pair's 16 results wait in
      // registers while the second pair goes through (82 live registers at the peak instead of 116 for all four at once --
      // with the next item's patch in flight the four-channel form spilled, and spill reloads wait for the stores before them).
      const float* M = lds + et * W2XT + cq * 4;
      f32x2 ya[4][4];
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        f32x2 hh[4][5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          f32x2 y[4];
          at5(*reinterpret_cast<const f32x2*>(M + (0 * 5 + c) * W2XP + 2 * pr), *reinterpret_cast<const f32x2*>(M + (1 * 5 + c) * W2XP + 2 * pr),
              *reinterpret_cast<const f32x2*>(M + (2 * 5 + c) * W2XP + 2 * pr), *reinterpret_cast<const f32x2*>(M + (3 * 5 + c) * W2XP + 2 * pr),
              *reinterpret_cast<const f32x2*>(M + (4 * 5 + c) * W2XP + 2 * pr), y);
#pragma unroll
          for (int a = 0; a < 4; ++a) hh[a][c] = y[a];
          __builtin_amdgcn_sched_barrier(0);          // keeps the LDS reads of later columns / the other pair from being hoisted
        }
        if (pr == 0) {
#pragma unroll
          for (int a = 0; a < 4; ++a) at5(hh[a][0], hh[a][1], hh[a][2], hh[a][3], hh[a][4], ya[a]);
        } else {
          tick(5);
          // pass 0: the image has been read -- the other half's accumulators may come in under this half's row pass and stores
          if constexpr (decltype(mid_c)::value) __syncthreads();
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            if constexpr (MASK) { if (a + 1 < 4) load_mask(a + 1); }
            f32x2 yb[4];
            at5(hh[a][0], hh[a][1], hh[a][2], hh[a][3], hh[a][4], yb);
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) {
              f32x4 v = f32x4{ya[a][b2][0], ya[a][b2][1], yb[b2][0], yb[b2][1]} + bv;
#pragma unroll
              for (int c = 0; c < 4; ++c) v[c] = apply_act(v[c], p.act, p.slope);
              if constexpr (MASK) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] *= act_grad(mk[a][b2][c], SRGAN_ACT_LRELU, p.mask_slope);
              }
              // (the pixel offset rides in the VECTOR offset: with it in the scalar-offset operand this store put dwords 1 and 3
              // of some lanes one pixel off on gfx950 / ROCm 7.2 -- scratch/wino42/check.py, profiles/LOG.md round 5)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), rs_y,
                                                     yoff + (unsigned)((a * p.Wo + b2) * PXS * p.Cd * 4), 0, 0);
            }
          }
        }
      }
    };

    // ---- the items of this workgroup.  At the top of an item: its gather state is set and its first patch is in flight ----
    setup_item(item0);
    load_patch();
    for (int item = item0; item < item1; ++item) {
      tick(7);
      if (ch == 0) tile_o[tl] = g_to;
#pragma unroll
      for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
      acc25 = f32x4{0.f, 0.f, 0.f, 0.f};
      const int e_nt = n_tile;
      rs_u = u_rsrc();
      // prologue: chunk 0 transformed into buffer 0, chunk 1 in flight, the first NP filter fragments requested
#pragma unroll
      for (int j = 0; j < 5; ++j) col_pass(j);
#pragma unroll
      for (int r = 0; r < 5; ++r) row_store(0, r);
      tick(8);
      load_patch();                                 // nk >= 2
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < NP; ++a) load_u(a, 0, 0);
      load_u25(0);
      tick(9);
      __syncthreads();
      tick(1);
      vf[0] = read_v(0, 0);
      int kc = 0;
      for (; kc + 2 < nk; ++kc) iter(kc, T{}, T{});
      iter(kc, T{}, F{});
      iter(kc + 1, F{}, F{});
      tick(2);
      const bool more = item + 1 < item1;

#if WINO42_EXP & 16
      {                                               // ablation: no epilogue (accumulators kept alive)
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) keep += acc[i][e];
        keep += acc25[0] + acc25[1] + acc25[2] + acc25[3];
        if (keep == 12345.678f) p.dst[tid] = keep;
        __syncthreads();
        continue;
      }
#endif
      if (h == 0) {
        pass(0, e_nt, T{});
        if (cb25 >> 1) store25();                     // (waves 4, 6: their block belongs to half 1, open since the mid barrier)
        __syncthreads();                              // pass 1 is in LDS
      } else {
        if (!(cb25 >> 1)) store25();                  // (waves 1, 3: their block belongs to half 0)
        __syncthreads();                              // pass 0 is in LDS
        __syncthreads();                              // pass 0 has been read
        pass(1, e_nt, F{});
      }
      if (more) {
        // (requesting the next item's first patch before this item's epilogue was measured: no gain -- what the gap between two
        // items costs is the ISSUE of 2 x 25 gather instructions per thread with no MFMAs to hide behind, not their latency)
        setup_item(item + 1);
        load_patch();
        __syncthreads();                              // pass 1 and tile_o have been read: the next item may overwrite them
      }
    }
    tick(6);
#if WINO42_EXP & 128
    __builtin_amdgcn_s_waitcnt(0);                  // the stamps land in the words of pixel 0, which this workgroup wrote
    __syncthreads();
    if (item0 == 0 && lane == 0) {
      float* o = p.dst + wave * 12;
#pragma unroll
      for (int k = 1; k < 10; ++k) o[k] = (float)(stamp[k] - stamp[0]);
      o[0] = (float)nk;
    }
#endif
  }

}

// ---- host side (dispatch and geometry live in conv_wino.hip: variant 7) ----
// `units` = tile blocks x output-channel blocks; one workgroup per CU at a time (115 KB of LDS, 512 threads).  Transposed form: the
// items (units x output phases) are dealt in contiguous ranges of ceil(items / CUs) to as many workgroups as that takes -- the
// phases of a unit re-read the same source pixels, and between two items of a workgroup no store has to be acknowledged before
// the next one starts (G.down1 input gradient 185 -> 178 us).  Strided form: one item per workgroup (measured 7 us faster than
// ranges of two on G.down1: a fresh workgroup's prologue is shorter than the gap between two items, see the kernel)
int wino42_launch(const WinoParams& p0, int kind, long long units, bool mask, double flops, hipStream_t st) {
  WinoParams p = p0;
  const long long items = units * (kind == 1 ? 4 : 1);
  static const int cus = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  p.ipw = kind == 1 ? (int)ceil_div(items, cus) : 1;
  const long long grid = ceil_div(items, p.ipw);
  ProfToken tok = prof_begin(37, flops, st);
  if (kind == 0) hipLaunchKernelGGL((wino42_kernel<1>), dim3((unsigned)grid), dim3(512), 0, st, p);
  else if (mask) hipLaunchKernelGGL((wino42_kernel<2, true>), dim3((unsigned)grid), dim3(512), 0, st, p);
  else hipLaunchKernelGGL((wino42_kernel<2>), dim3((unsigned)grid), dim3(512), 0, st, p);
  prof_end(tok, st);
  return check_launch("wino42_kernel");
}

}  // namespace srgan

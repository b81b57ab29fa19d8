// Winograd convolutions on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   F(2x2,3x3) for the 3x3 / stride-1 layers (the generator's residual trunk is 64 % of the step's FLOPs: SURVEY.md
//   section 8a row a2, reference pyfiles/model.py:196-201; the encoder's 3x3 convs, model.py:433-437):
//       Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//   F(3x3,2x2), its transpose, for the 4x4 / stride-2 layers and their transposed convs (model.py:212-215, 227-230,
//   302-309), written as sums over input phases / one problem per output phase (MODE 1 / 2 below):
//       Y = G^T [ sum (A g A^T) .* (B^T d B) ] G        per 3x3 output tile, 4x4 patch of a phase image, 2x2 filter
//   16 multiplies per 4 (9) outputs instead of 36: 2.25x fewer MFMA FLOPs than the implicit GEMM, every product still
//   an exact fp32 product.
//
// One fused kernel per direction:
//   * the filter transform is done once per optimiser step by the pack kernel (pack_device.h), laid out as the register
//     image a wave loads: position p of U is consumed by exactly one wave, so U never touches LDS,
//   * a workgroup (8 waves) owns 64 output tiles x 64 output channels for ALL 16 transform positions; per 8-channel chunk
//     each thread gathers one (tile, channel) 4x4 patch with buffer loads (out-of-image taps point past the buffer: the
//     range check supplies the zero padding), transforms it in registers (32 adds) and writes the 16 results to LDS,
//   * wave w multiplies positions 2w, 2w+1: [64 tiles x 8] x [8 x 64 channels]; LDS double-buffered, one barrier per
//     chunk, global loads a chunk ahead; the two waves of a SIMD run their non-MFMA work at different points of the
//     iteration, in straight-line bodies (see the comment in the kernel),
//   * epilogue: the 16 position accumulators meet in LDS (two halves of 32 channels), each thread applies the output
//     transform and stores its tile x 32 contiguous channels (+ bias, activation).
// wino_kernel<0/1/2>: forward and input gradient; wino_wgrad_kernel<0/1>: weight gradient (both operands transformed).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "pack_device.h"

#ifndef WINO_EXP
#define WINO_EXP 0
#endif

namespace srgan {

constexpr int WT = 64;   // output tiles per workgroup
constexpr int WNB = 64;  // output channels per workgroup
constexpr int WC = 8;    // reduce channels per chunk

// A buffer descriptor whose words are provably wave-uniform (readfirstlane of the base halves and the byte count):
// otherwise the compiler may keep it in VGPRs and wrap every buffer load in a readfirstlane "waterfall" loop.
__device__ __forceinline__ auto uniform_rsrc(const float* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  float* q = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}


// MODE 0: 3x3 stride-1 conv, F(2x2,3x3): 2x2 output tile from a 4x4 patch, Y = A^T M A.
// MODE 1: 4x4 stride-2 pad-1 conv = sum over the 4 input phases I_pq[u][v] = in[2u+p-1][2v+q-1] of a 2x2 stride-1
//         correlation; F(3x3,2x2) -- the transpose of F(2x2,3x3): 3x3 output tile from a 4x4 patch of the phase image,
//         U = A g A^T, V = B^T d B, Y = G^T M G, 16 multiplies per 9 outputs instead of 36; the reduce index is
//         (phase, channel), chunks are phase-major.
// MODE 2: the transposed form (input gradient of MODE 1's conv / ConvTranspose2d forward): output phase (r,s) =
//         blockIdx.y is a 2x2 stride-1 correlation over dy with taps w[3-2a-r][3-2b-s]; F(3x3,2x2) on the phase image,
//         outputs scattered with pixel stride 2.

// MASK (MODE 2 only): the epilogue multiplies every output by 1 or p.mask_slope after the sign of p.mask at the same element --
// the LeakyReLU backward of the PREVIOUS layer, whose activated output is this input gradient's forward tensor: one read of
// that tensor here instead of a three-tensor elementwise pass (act_bwd_kernel) in front of the previous layer's backward.
template <int MODE, bool MASK = false>
__global__ __launch_bounds__(512) void wino_kernel(WinoParams p) {
  static_assert(!MASK || MODE == 2, "the mask epilogue exists for the transposed form only");
  // V image, double-buffered: [buf][16 pos][2 channel halves][64 tiles x 4 ch + 16 pad] (+32 pad per position).
  // A lane's MFMA fragment (4 channels of one tile) is one 16-B slot and a 16-lane read group covers 256 contiguous
  // bytes (conflict-free ds_read_b128); the 16-float pad puts the two halves a gather wave writes on different
  // banks; the position stride 576 % 64 == 0 keeps the paired store (ds_write2st64_b32) usable.
  // The transformed filters never touch LDS: position p of U is used by exactly one wave, which loads its fragments
  // straight from the packed image into registers.  The epilogue reuses the whole array (32768 floats).
  constexpr int VH = 64 * 4 + 16, VP = 2 * VH + 32, VSZ = 16 * VP;
  static_assert(2 * VSZ <= 32768, "V buffers must fit the epilogue image");
  __shared__ __attribute__((aligned(16))) float lds[32768];
  __shared__ int tile_o[WT];      // destination pixel index of the tile's (0,0) output, or -1
  __shared__ int tile_f[WT];      // valid output rows | valid output columns << 4 (of the OT x OT tile)
  constexpr int OT = MODE == 0 ? 2 : 3;            // output tile edge
  constexpr int PS = MODE == 2 ? 2 : 1;            // pixel stride of the outputs in the destination
  const int ph_r = MODE == 2 ? (int)blockIdx.y >> 1 : 0, ph_s = MODE == 2 ? (int)blockIdx.y & 1 : 0;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // XCD-aware order: hardware sends workgroup b to XCD b % 8; give each XCD a contiguous range of tile rows so the
  // n_tiles workgroups that read the same input patch share one L2
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int m_tile = bid / p.n_tiles, n_tile = bid - m_tile * p.n_tiles;

  // ---- gather role: thread = (tile tl, channel ch); 16 byte offsets, out-of-image taps point past the buffer
  // (the buffer load's range check returns 0 for them: zero padding without a select) ----
  const int tl = tid >> 3, ch = tid & 7;
  constexpr unsigned kOutside = 0x80000000u;
  unsigned off[16];
  int g_b, g_ty, g_tx;
  bool g_tv;
  {
    const int t = m_tile * WT + tl;
    g_tv = t < p.T;
    const int tt = g_tv ? t : 0;
    const int per = p.TH * p.TW;
    g_b = tt / per;
    const int r = tt - g_b * per;
    g_ty = r / p.TW; g_tx = r - g_ty * p.TW;
    if (ch == 0) {
      const int oy = PS * OT * g_ty + ph_r, ox = PS * OT * g_tx + ph_s;      // first output pixel of the tile
      const int nr = min(OT, (p.Ho - oy + PS - 1) / PS), nc = min(OT, (p.Wo - ox + PS - 1) / PS);
      tile_o[tl] = (g_tv && nr > 0 && nc > 0) ? (g_b * p.Ho + oy) * p.Wo + ox : -1;
      tile_f[tl] = max(nr, 0) | (max(nc, 0) << 4);
    }
  }
  // patch element (i, j) of this thread's tile for input phase `phase` (MODE 1; ignored otherwise)
  auto set_offsets = [&](int phase) __attribute__((always_inline)) {
    int sy, dy_, oy0, ox0;
    if (MODE == 0) { sy = 2; dy_ = 1; oy0 = -p.pad; ox0 = -p.pad; }
    else if (MODE == 1) { sy = 6; dy_ = 2; oy0 = (phase >> 1) - p.pad; ox0 = (phase & 1) - p.pad; }
    else { sy = 3; dy_ = 1; oy0 = ph_r - 1; ox0 = ph_s - 1; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy = sy * g_ty + dy_ * i + oy0;
      bool yok = g_tv;
      if (MODE == 0 && p.reflect) {
        iy = iy < 0 ? -iy : iy;
        iy = iy >= p.H ? 2 * p.H - 2 - iy : iy;
        iy = min(max(iy, 0), p.H - 1);
      } else {
        yok = yok && (unsigned)iy < (unsigned)p.H;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int ix = sy * g_tx + dy_ * j + ox0;
        bool ok = yok;
        if (MODE == 0 && p.reflect) {
          ix = ix < 0 ? -ix : ix;
          ix = ix >= p.W ? 2 * p.W - 2 - ix : ix;
          ix = min(max(ix, 0), p.W - 1);
        } else {
          ok = ok && (unsigned)ix < (unsigned)p.W;
        }
        off[i * 4 + j] = ok ? (unsigned)(((g_b * p.H + iy) * p.W + ix) * p.C + ch) * 4u : kOutside;
      }
    }
  };
  set_offsets(0);
  // descriptors from wave-uniform values only; the per-chunk advance goes into the scalar offset
  const unsigned src_bytes = (unsigned)((size_t)p.NB * p.H * p.W * p.C * 4);
  const auto rs_x = uniform_rsrc(p.src, src_bytes);
  const auto rs_u = uniform_rsrc(p.u + ((size_t)(MODE == 2 ? blockIdx.y : 0) * p.n_tiles + n_tile) * p.nchunk * 8192,
                                 (unsigned)p.nchunk * 32768u);
  // this lane's fragment of U inside a chunk image [16 pos][2 halves][64 couts][4]: position 2*wave + slot, column
  // tile j -> byte offset ubase + slot * 2048 + j * 512
  const unsigned ubase = (unsigned)((2 * wave) * 512 + lh * 256 + lr * 4) * 4u;

  float d[16];
  f32x4 bfr[2][2][2];      // [chunk parity][slot = position within the wave][column tile]
  int lc8 = 0, lphase = 0;      // load cursor: chunks are consumed strictly in order
  // (issued in four parts between groups of MFMAs: the texture path takes ~20 cycles per gather instruction of a CU, so
  // sixteen back-to-back loads per wave would hold all eight waves at the load with the matrix pipe idle)
  auto load_x_part = [&](auto part_c) __attribute__((always_inline)) {
    constexpr int PART = decltype(part_c)::value;
#if WINO_EXP == 5 || WINO_EXP == 6
    if (lc8 + lphase > 0) return;      // ablation: only the first chunk is gathered
#endif
#pragma unroll
    for (int i = 4 * PART; i < 4 * PART + 4; ++i)
      d[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, off[i], lc8 * (WC * 4), 0));
    if constexpr (PART == 3) {
      ++lc8;
      if (MODE == 1 && lc8 == p.cpp) {      // next input phase: new patch origin and padding mask (wave-uniform, 3x per kernel)
        lc8 = 0;
        ++lphase;
        set_offsets(lphase);
      }
    }
  };
  auto load_x = [&]() __attribute__((always_inline)) {
    load_x_part(std::integral_constant<int, 0>{});
    load_x_part(std::integral_constant<int, 1>{});
    load_x_part(std::integral_constant<int, 2>{});
    load_x_part(std::integral_constant<int, 3>{});
  };
  auto load_u1 = [&](int kc, auto par, int a, int j) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
#if WINO_EXP == 7
    if (kc > 1) return;                // ablation: filter fragments of the first two chunks only
#endif
    bfr[P][a][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_u, ubase + a * 2048 + j * 512, kc * 32768, 0));
  };
  auto load_u = [&](int kc, auto par) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j) load_u1(kc, par, a, j);
  };
  // B^T d B in registers, 16 results to V[pos][half][tile][ch & 3]; three pieces (column pass, rows 0-1, rows 2-3) that the
  // main loop places between groups of MFMAs
  float tv[16];
  auto store_piece = [&](int buf, auto piece_c) __attribute__((always_inline)) {
    constexpr int PIECE = decltype(piece_c)::value;
#if WINO_EXP == 4 || WINO_EXP == 6
    return;
#endif
    float* V = lds + buf * VSZ + (ch >> 2) * VH + tl * 4 + (ch & 3);
    if constexpr (PIECE == 0) {
#if WINO_EXP == 9
      return;
#endif
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        tv[0 * 4 + c] = d[0 * 4 + c] - d[2 * 4 + c];
        tv[1 * 4 + c] = d[1 * 4 + c] + d[2 * 4 + c];
        tv[2 * 4 + c] = d[2 * 4 + c] - d[1 * 4 + c];
        tv[3 * 4 + c] = d[1 * 4 + c] - d[3 * 4 + c];
      }
    } else {
#if WINO_EXP == 8        // ablation: the transform arithmetic without the LDS stores
#pragma unroll
      for (int r = 2 * (PIECE - 1); r < 2 * PIECE; ++r) {
        asm volatile("" ::"v"(tv[r * 4 + 0] - tv[r * 4 + 2]), "v"(tv[r * 4 + 1] + tv[r * 4 + 2]), "v"(tv[r * 4 + 2] - tv[r * 4 + 1]),
                     "v"(tv[r * 4 + 1] - tv[r * 4 + 3]));
      }
      (void)V;
#elif WINO_EXP == 9      // ablation: the LDS stores without the transform arithmetic
#pragma unroll
      for (int r = 2 * (PIECE - 1); r < 2 * PIECE; ++r) {
        V[(r * 4 + 0) * VP] = d[r * 4 + 0];
        V[(r * 4 + 1) * VP] = d[r * 4 + 1];
        V[(r * 4 + 2) * VP] = d[r * 4 + 2];
        V[(r * 4 + 3) * VP] = d[r * 4 + 3];
      }
#else
#pragma unroll
      for (int r = 2 * (PIECE - 1); r < 2 * PIECE; ++r) {
        V[(r * 4 + 0) * VP] = tv[r * 4 + 0] - tv[r * 4 + 2];
        V[(r * 4 + 1) * VP] = tv[r * 4 + 1] + tv[r * 4 + 2];
        V[(r * 4 + 2) * VP] = tv[r * 4 + 2] - tv[r * 4 + 1];
        V[(r * 4 + 3) * VP] = tv[r * 4 + 1] - tv[r * 4 + 3];
      }
#endif
    }
  };
  auto store_v = [&](int buf) __attribute__((always_inline)) {
    store_piece(buf, std::integral_constant<int, 0>{});
    store_piece(buf, std::integral_constant<int, 1>{});
    store_piece(buf, std::integral_constant<int, 2>{});
  };

  // ---- multiply role: wave = positions 2*wave, 2*wave+1 over the whole 64 x 64 tile ----
  f32x16 acc[2][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][i][j][e] = 0.f;

  f32x4 af[2][2];   // [slot][row tile]
  auto read_frags = [&](int buf, int slot) __attribute__((always_inline)) {
    const float* V = lds + buf * VSZ + (2 * wave + slot) * VP + lh * VH;
#pragma unroll
    for (int i = 0; i < 2; ++i) af[slot][i] = *reinterpret_cast<const f32x4*>(V + (i * 32 + lr) * 4);
  };
  auto mfma_steps = [&](auto par, int slot, int e0, int e1) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
#pragma unroll
    for (int e = e0; e < e1; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[slot][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][i][e], bfr[P][slot][j][e], acc[slot][i][j], 0, 0, 0);
  };

  // Chunk kc = 8 groups of four MFMAs (one k-step of one position).  The vector work of the NEXT chunk (patch transform),
  // its LDS stores, the patch loads of the chunk after it and the filter fragments of the next chunk sit between the groups
  // of the SAME wave: LDS and buffer instructions issue in the shadow of the wave's own MFMAs, whereas a separate store
  // phase waits behind the back-to-back MFMAs of the other wave of the SIMD, and a block of gather loads holds every wave at
  // the texture path (~20 cycles per instruction) with the matrix pipe idle (scratch/coissue, scratch/shadow).
  // The LDS / memory counters are in-order and the compiler merges their state at every control-flow join, so the
  // steady-state loop holds one straight-line body per chunk parity; the tail iterations (no store / no load) are
  // separate instantiations.
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  using T = std::true_type;
  using F = std::false_type;
  const int nk = p.nchunk;
  load_x();
  load_u(0, I0{});
  store_v(0);
  __syncthreads();
  read_frags(0, 0);
  if (nk > 1) load_x();
  auto iter = [&](int kc, auto par, auto st, auto ld) __attribute__((always_inline)) {
    constexpr bool ST = decltype(st)::value, LD = decltype(ld)::value;
    constexpr int P = decltype(par)::value;
    using NP = std::integral_constant<int, 1 - P>;
    auto sb = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
    read_frags(P, 1);
    mfma_steps(par, 0, 0, 1);
    if constexpr (ST) store_piece(1 - P, I0{});       // chunk kc+1 (its loads were issued a whole chunk ago); frees d
    sb();
    mfma_steps(par, 0, 1, 2);
    if constexpr (ST) store_piece(1 - P, I1{});
    if constexpr (LD) load_x_part(I0{});
    sb();
    mfma_steps(par, 0, 2, 3);
    if constexpr (ST) store_piece(1 - P, I2{});
    if constexpr (LD) load_x_part(I1{});
    sb();
    mfma_steps(par, 0, 3, 4);
    if constexpr (ST) load_u1(kc + 1, NP{}, 0, 0);     // fragments of chunk kc+1: wanted at the top of the next iteration
    if constexpr (LD) load_x_part(I2{});
    sb();
    mfma_steps(par, 1, 0, 1);
    if constexpr (ST) load_u1(kc + 1, NP{}, 0, 1);
    if constexpr (LD) load_x_part(I3{});
    sb();
    mfma_steps(par, 1, 1, 2);
    if constexpr (ST) { load_u1(kc + 1, NP{}, 1, 0); load_u1(kc + 1, NP{}, 1, 1); }
    sb();
    mfma_steps(par, 1, 2, 3);
    sb();
    __syncthreads();                       // chunk kc+1 visible; every wave holds its last fragments of chunk kc
    if constexpr (ST) read_frags(1 - P, 0);
    mfma_steps(par, 1, 3, 4);
  };
  {
    int kc = 0;
    for (; kc + 3 < nk; kc += 2) {
      iter(kc, I0{}, T{}, T{});
      iter(kc + 1, I1{}, T{}, T{});
    }
    // nchunk is even (wino_applicable): exactly two chunks are left
    iter(kc, I0{}, T{}, F{});
    iter(kc + 1, I1{}, F{}, F{});
  }

  // ---- epilogue: output transform (A^T M A: 2x2, or G^T M G: 3x3), two halves of 32 output channels; thread = (tile, 4
  // channels): 16-byte LDS reads and 16-byte stores, the 8 lanes of a tile write one whole 128-byte line per pixel
  // (a quarter of the store instructions of a one-channel-per-thread layout: the texture path takes ~16 cycles each) ----
  const int cq = tid & 7, et = tid >> 3;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int tile = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          lds[(2 * wave + a) * 2048 + tile * 32 + lr] = acc[a][i][half][e];
        }
    __syncthreads();
    const int n = n_tile * WNB + half * 32 + cq * 4;
    const bool nok = n < p.Cd;                       // Cd % 4 == 0 (wino_variant): the four channels are in or out together
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (nok && p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
    const int o = tile_o[et], f = tile_f[et];
    // MASK: requested before the LDS reads and the output transform, consumed at the stores (loads next to stores are exposed)
    f32x4 mk[OT][OT];
    if constexpr (MASK) {
      if (nok && o >= 0) {
        const float* mp = p.mask + (size_t)o * p.Cd + n;
#pragma unroll
        for (int a = 0; a < OT; ++a)
#pragma unroll
          for (int b2 = 0; b2 < OT; ++b2)
            if (a < (f & 15) && b2 < (f >> 4)) mk[a][b2] = *reinterpret_cast<const f32x4*>(mp + (size_t)(a * p.Wo + b2) * PS * p.Cd);
      }
    }
    f32x4 m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = *reinterpret_cast<const f32x4*>(lds + k * 2048 + et * 32 + cq * 4);
    f32x4 y[OT][OT];
    if constexpr (MODE == 0) {
      f32x4 s0[4], s1[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s0[c] = m[0 * 4 + c] + m[1 * 4 + c] + m[2 * 4 + c];
        s1[c] = m[1 * 4 + c] - m[2 * 4 + c] - m[3 * 4 + c];
      }
      y[0][0] = s0[0] + s0[1] + s0[2]; y[0][1] = s0[1] - s0[2] - s0[3];
      y[1][0] = s1[0] + s1[1] + s1[2]; y[1][1] = s1[1] - s1[2] - s1[3];
    } else {
      f32x4 h[3][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        h[0][c] = m[0 * 4 + c] + 0.5f * (m[1 * 4 + c] + m[2 * 4 + c]);
        h[1][c] = 0.5f * (m[1 * 4 + c] - m[2 * 4 + c]);
        h[2][c] = 0.5f * (m[1 * 4 + c] + m[2 * 4 + c]) + m[3 * 4 + c];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        y[a][0] = h[a][0] + 0.5f * (h[a][1] + h[a][2]);
        y[a][1] = 0.5f * (h[a][1] - h[a][2]);
        y[a][2] = 0.5f * (h[a][1] + h[a][2]) + h[a][3];
      }
    }
    if (nok && o >= 0) {
      const int nr = f & 15, nc = f >> 4;
      float* dp = p.dst + (size_t)o * p.Cd + n;
#pragma unroll
      for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int b2 = 0; b2 < OT; ++b2)
          if (a < nr && b2 < nc) {
            f32x4 v = y[a][b2] + bv;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = apply_act(v[c], p.act, p.slope);
            if constexpr (MASK) {
#pragma unroll
              for (int c = 0; c < 4; ++c) v[c] *= act_grad(mk[a][b2][c], SRGAN_ACT_LRELU, p.mask_slope);
            }
            *reinterpret_cast<f32x4*>(dp + (size_t)(a * p.Wo + b2) * PS * p.Cd) = v;
          }
    }
  }
}

// ---- weight gradient:  dU[pos][o][i] = sum_tiles (A dY A^T)[pos][t][o] * (B^T d B)[pos][t][i],  dW = G^T dU G ----
// Same 2.25x saving; both MFMA operands are transformed data, so both go through LDS.
//   * a workgroup owns 64 output channels x 64 input channels for all 16 positions over a SPLIT of the tiles;
//     a chunk is 8 tiles at the same 8 tile positions of consecutive images, so a thread's gather offsets (and the
//     padding mask folded into them) are constant while the scalar offset walks over the batch,
//   * thread = (tile of the chunk, channel): 16 + 4 buffer loads, two register transforms, 32 LDS stores laid out
//     [pos][tile half][channel][4 tiles] so that a lane's MFMA fragment (4 consecutive K = tiles) is one 16-B slot,
//   * epilogue: positions meet in LDS, each thread applies G^T . G and writes the 3x3 taps of its (o, i) pairs into
//     the split-K slab [split][O][9 * I] that wgrad_reduce_kernel (conv_igemm.hip) sums.
struct WinoWgradParams {
  const float* x;     // [NB][H][W][C]
  const float* dy;    // [NB][Ho][Wo][O]
  float* slab;        // [splits][Opad][9 * C]
  int NB, H, W, C, Ho, Wo, O;
  int pad, reflect;
  int TH, TW;          // tile grid per image
  int groups;          // ceil(TH * TW / 8): 8-tile position groups per image
  int groups_per_split, splits;   // splits = (splits over tile-position groups) x bsplits
  int bsplits, nb_per;             // split of the batch: a workgroup walks images [b0, b0 + nb_per) of its groups
  int o_tiles, i_tiles, Opad;
  int xcd_splits;                  // 1: the workgroups of one split share an XCD (see the kernel); needs splits % 8 == 0
};

// MODE 0: 3x3 stride-1 layer (above).  MODE 1: 4x4 stride-2 pad-1 layer, the transpose of wino_kernel<1>:
//   dU[pos][o][(pq, c)] = sum_tiles (G dY G^T)[pos] * (B^T d_pq B)[pos] with dY the 3x3 output-gradient tile and d_pq the
//   4x4 patch of input phase pq; dW[o][c][2a+p][2b+q] = (A^T dU_pq A)[a][b].  The 64-wide column tile of a workgroup
//   lies inside one input phase (C % 64 == 0).
template <int MODE>
__global__ __launch_bounds__(512) void wino_wgrad_kernel(WinoWgradParams p) {
  // [buf][ Z: 16 pos x 2 tile halves x 64 o x 4 tiles | V: same with 64 i ] = 2 x 64 KB, reused by the epilogue
  __shared__ __attribute__((aligned(16))) float lds[32768];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // Workgroup b runs on XCD b % 8.  The (o tile, i tile) workgroups of one split gather the SAME tile range of x and dy, so a
  // split's workgroups are dealt to ONE XCD (split = the low three bits of b inside a group of 8 splits): each line of the two
  // activations goes through one L2 instead of up to eight.
  int bid = blockIdx.x;
  const int tiles_per_split = p.o_tiles * p.i_tiles;
  int split;
  if (p.xcd_splits) {
    const int grp = bid / (8 * tiles_per_split), r = bid - grp * 8 * tiles_per_split;
    split = grp * 8 + (r & 7);
    bid = r >> 3;
  } else {
    split = bid / tiles_per_split;
    bid -= split * tiles_per_split;
  }
  const int o_tile = bid / p.i_tiles, i_tile = bid - o_tile * p.i_tiles;

  // gather role: tile of the chunk = 4 * (wave >> 2) + (lane & 3), channel = 16 * (wave & 3) + (lane >> 2)
  const int th = wave >> 2, tq = lane & 3;
  const int chl = (wave & 3) * 16 + (lane >> 2);
  constexpr int ND = MODE == 0 ? 4 : 9;            // output-gradient elements per tile (2x2 or 3x3)
  constexpr int OT = MODE == 0 ? 2 : 3;
  const int phase = MODE == 1 ? (i_tile * 64) / p.C : 0;                  // input phase of this column tile
  const int ph_p = phase >> 1, ph_q = phase & 1;
  const int ci = i_tile * 64 - phase * p.C + chl, co = o_tile * 64 + chl;
  constexpr unsigned kOutside = 0x80000000u;
  unsigned offx[16], offd[ND];
  auto set_group = [&](int g) __attribute__((always_inline)) {       // offsets of this thread's tile at tile-position group g (image 0)
    const int q = g * 8 + th * 4 + tq;
    const bool tv = q < p.TH * p.TW && g < p.groups;
    const int qq = tv ? q : 0;
    const int ty = qq / p.TW, tx = qq - ty * p.TW;
    const int sy = MODE == 0 ? 2 : 6, dstep = MODE == 0 ? 1 : 2;
    const int oy0 = (MODE == 0 ? 0 : ph_p) - p.pad, ox0 = (MODE == 0 ? 0 : ph_q) - p.pad;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy = sy * ty + dstep * i + oy0;
      bool yok = tv && ci < p.C;
      if (p.reflect) {
        iy = iy < 0 ? -iy : iy;
        iy = iy >= p.H ? 2 * p.H - 2 - iy : iy;
        iy = min(max(iy, 0), p.H - 1);
      } else {
        yok = yok && (unsigned)iy < (unsigned)p.H;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int ix = sy * tx + dstep * j + ox0;
        bool ok = yok;
        if (p.reflect) {
          ix = ix < 0 ? -ix : ix;
          ix = ix >= p.W ? 2 * p.W - 2 - ix : ix;
          ix = min(max(ix, 0), p.W - 1);
        } else {
          ok = ok && (unsigned)ix < (unsigned)p.W;
        }
        offx[i * 4 + j] = ok ? (unsigned)((iy * p.W + ix) * p.C + ci) * 4u : kOutside;
      }
    }
#pragma unroll
    for (int i = 0; i < OT; ++i)
#pragma unroll
      for (int j = 0; j < OT; ++j) {
        const int oy = OT * ty + i, ox = OT * tx + j;
        const bool ok = tv && co < p.O && oy < p.Ho && ox < p.Wo;
        offd[i * OT + j] = ok ? (unsigned)((oy * p.Wo + ox) * p.O + co) * 4u : kOutside;
      }
  };
  const unsigned x_img = (unsigned)(p.H * p.W * p.C) * 4u, d_img = (unsigned)(p.Ho * p.Wo * p.O) * 4u;
  const auto rs_x = uniform_rsrc(p.x, x_img * (unsigned)p.NB);
  const auto rs_d = uniform_rsrc(p.dy, d_img * (unsigned)p.NB);

  // load cursor (two chunks ahead of the multiply): group lg, image lb
  const int gsplit = split / p.bsplits, bsplit = split - gsplit * p.bsplits;
  const int g0 = gsplit * p.groups_per_split;
  const int g1 = min(g0 + p.groups_per_split, p.groups);
  const int b0 = bsplit * p.nb_per, b1 = min(b0 + p.nb_per, p.NB);
  const int nk = max(g1 - g0, 0) * max(b1 - b0, 0);
  int lg = g0, lb = b0;
  set_group(lg);
  float dx[16], dd[ND];
  // (the loads of a chunk are issued in five parts: the texture path takes ~20 cycles per gather instruction of a CU, so
  // twenty back-to-back loads per wave would hold all eight waves at the load for ~3000 cycles with the matrix pipe idle)
  auto load_part = [&](auto part_c) __attribute__((always_inline)) {
    constexpr int PART = decltype(part_c)::value;
    if constexpr (PART < 4) {
#pragma unroll
      for (int i = 4 * PART; i < 4 * PART + 4; ++i)
        dx[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, offx[i], lb * x_img, 0));
    } else {
#pragma unroll
      for (int i = 0; i < ND; ++i)
        dd[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, offd[i], lb * d_img, 0));
      if (++lb == b1) {        // next tile-position group (wave-uniform, once per batch range)
        lb = b0;
        ++lg;
        set_group(lg);
      }
    }
  };
  auto load_chunk = [&]() __attribute__((always_inline)) {
    load_part(std::integral_constant<int, 0>{});
    load_part(std::integral_constant<int, 1>{});
    load_part(std::integral_constant<int, 2>{});
    load_part(std::integral_constant<int, 3>{});
    load_part(std::integral_constant<int, 4>{});
  };
  const int wofs = th * 256 + (wave & 3) * 64 + lane;     // [half][channel][tile & 3], lane-linear
  // The transform + store of the next chunk is cut into four pieces that the main loop places between groups of four
  // MFMAs: LDS stores and buffer loads issue in the shadow of the wave's own MFMAs, whereas a separate store phase would
  // wait behind the back-to-back MFMAs of the other wave of the SIMD (scratch/coissue).
  float tv[16];
  auto store_piece = [&](int buf, auto piece_c) __attribute__((always_inline)) {
    constexpr int PIECE = decltype(piece_c)::value;
    float* Z = lds + buf * 16384 + wofs;
    float* V = Z + 8192;
    if constexpr (PIECE == 0) {                    // column pass of B^T d B
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        tv[0 * 4 + c] = dx[0 * 4 + c] - dx[2 * 4 + c];
        tv[1 * 4 + c] = dx[1 * 4 + c] + dx[2 * 4 + c];
        tv[2 * 4 + c] = dx[2 * 4 + c] - dx[1 * 4 + c];
        tv[3 * 4 + c] = dx[1 * 4 + c] - dx[3 * 4 + c];
      }
    } else if constexpr (PIECE == 1 || PIECE == 2) {      // row pass, two rows each
#pragma unroll
      for (int r = 2 * (PIECE - 1); r < 2 * PIECE; ++r) {
        V[(r * 4 + 0) * 512] = tv[r * 4 + 0] - tv[r * 4 + 2];
        V[(r * 4 + 1) * 512] = tv[r * 4 + 1] + tv[r * 4 + 2];
        V[(r * 4 + 2) * 512] = tv[r * 4 + 2] - tv[r * 4 + 1];
        V[(r * 4 + 3) * 512] = tv[r * 4 + 1] - tv[r * 4 + 3];
      }
    } else if constexpr (MODE == 0) {
      // A dY A^T, A = [1 0; 1 1; 1 -1; 0 -1]
      const float r0[2] = {dd[0], dd[1]};
      const float r1[2] = {dd[0] + dd[2], dd[1] + dd[3]};
      const float r2[2] = {dd[0] - dd[2], dd[1] - dd[3]};
      const float r3[2] = {-dd[2], -dd[3]};
      const float* rr[4] = {r0, r1, r2, r3};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Z[(r * 4 + 0) * 512] = rr[r][0];
        Z[(r * 4 + 1) * 512] = rr[r][0] + rr[r][1];
        Z[(r * 4 + 2) * 512] = rr[r][0] - rr[r][1];
        Z[(r * 4 + 3) * 512] = -rr[r][1];
      }
    } else {
      // G dY G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
      float zr[4][3];
#pragma unroll
      for (int l = 0; l < 3; ++l) {
        zr[0][l] = dd[0 * 3 + l];
        zr[1][l] = 0.5f * (dd[0 * 3 + l] + dd[1 * 3 + l] + dd[2 * 3 + l]);
        zr[2][l] = 0.5f * (dd[0 * 3 + l] - dd[1 * 3 + l] + dd[2 * 3 + l]);
        zr[3][l] = dd[2 * 3 + l];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Z[(r * 4 + 0) * 512] = zr[r][0];
        Z[(r * 4 + 1) * 512] = 0.5f * (zr[r][0] + zr[r][1] + zr[r][2]);
        Z[(r * 4 + 2) * 512] = 0.5f * (zr[r][0] - zr[r][1] + zr[r][2]);
        Z[(r * 4 + 3) * 512] = zr[r][2];
      }
    }
  };
  auto store_chunk = [&](int buf) __attribute__((always_inline)) {
    store_piece(buf, std::integral_constant<int, 0>{});
    store_piece(buf, std::integral_constant<int, 1>{});
    store_piece(buf, std::integral_constant<int, 2>{});
    store_piece(buf, std::integral_constant<int, 3>{});
  };

  f32x16 acc[2][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][i][j][e] = 0.f;
  f32x4 af[2][2], bf[2][2];
  auto read_frags = [&](int buf, int slot) __attribute__((always_inline)) {
    const float* Z = lds + buf * 16384 + (2 * wave + slot) * 512 + lh * 256;
    const float* V = Z + 8192;
#pragma unroll
    for (int i = 0; i < 2; ++i) af[slot][i] = *reinterpret_cast<const f32x4*>(Z + (i * 32 + lr) * 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[slot][j] = *reinterpret_cast<const f32x4*>(V + (j * 32 + lr) * 4);
  };
  auto mfma_steps = [&](int slot, int e0, int e1) __attribute__((always_inline)) {
#pragma unroll
    for (int e = e0; e < e1; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[slot][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][i][e], bf[slot][j][e], acc[slot][i][j], 0, 0, 0);
  };

  using T = std::true_type;
  using F = std::false_type;
  if (nk > 0) {
    load_chunk();
    store_chunk(0);
    __syncthreads();
    read_frags(0, 0);
    if (nk > 1) load_chunk();
    // chunk kc: 8 groups of four MFMAs (one k-step of one position each); the pieces of the next chunk's transform and
    // the loads of the chunk after it sit between the groups; one barrier, after every store and before the first read
    // of the other buffer
    auto iter = [&](int kc, auto st, auto ld) __attribute__((always_inline)) {
      constexpr bool ST = decltype(st)::value, LD = decltype(ld)::value;
      const int cur = kc & 1;
      auto sb = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
      read_frags(cur, 1);
      mfma_steps(0, 0, 1);
      if constexpr (ST) store_piece(cur ^ 1, std::integral_constant<int, 0>{});      // frees dx
      sb();
      mfma_steps(0, 1, 2);
      if constexpr (ST) store_piece(cur ^ 1, std::integral_constant<int, 1>{});
      if constexpr (LD) load_part(std::integral_constant<int, 0>{});
      sb();
      mfma_steps(0, 2, 3);
      if constexpr (ST) store_piece(cur ^ 1, std::integral_constant<int, 2>{});
      if constexpr (LD) load_part(std::integral_constant<int, 1>{});
      sb();
      mfma_steps(0, 3, 4);
      if constexpr (ST) store_piece(cur ^ 1, std::integral_constant<int, 3>{});      // frees dd
      if constexpr (LD) load_part(std::integral_constant<int, 2>{});
      sb();
      mfma_steps(1, 0, 1);
      if constexpr (LD) load_part(std::integral_constant<int, 3>{});
      sb();
      mfma_steps(1, 1, 2);
      if constexpr (LD) load_part(std::integral_constant<int, 4>{});
      sb();
      mfma_steps(1, 2, 3);
      sb();
      __syncthreads();
      if constexpr (ST) read_frags(cur ^ 1, 0);
      mfma_steps(1, 3, 4);
    };
    int kc = 0;
    for (; kc + 2 < nk; ++kc) iter(kc, T{}, T{});
    if (nk >= 2) iter(nk - 2, T{}, F{});
    iter(nk - 1, F{}, F{});
  }

  // ---- epilogue: G^T dU G (3x3 taps) or A^T dU A (the 2x2 taps of this input phase) per (o, i), two halves of 32
  // input channels; slab[split][o][tap * C + i]; thread = (output channel, 4 input channels): 16-byte reads and stores ----
  const int cq = tid & 7, row = tid >> 3;
  constexpr int NT = MODE == 0 ? 9 : 16;
  float* slab = p.slab + (size_t)split * p.Opad * (NT * p.C);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          lds[(2 * wave + a) * 2048 + r * 32 + lr] = acc[a][i][half][e];
        }
    __syncthreads();
    const int ic = i_tile * 64 - phase * p.C + half * 32 + cq * 4;      // C % 64 == 0: the four channels are in or out together
    const int oc = o_tile * 64 + row;
    f32x4 u[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) u[k] = *reinterpret_cast<const f32x4*>(lds + k * 2048 + row * 32 + cq * 4);
    if (oc < p.O && ic < p.C) {
      float* dst = slab + (size_t)oc * (NT * p.C) + ic;
      if constexpr (MODE == 0) {
        f32x4 h[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          h[0][c] = u[0 * 4 + c] + 0.5f * (u[1 * 4 + c] + u[2 * 4 + c]);
          h[1][c] = 0.5f * (u[1 * 4 + c] - u[2 * 4 + c]);
          h[2][c] = 0.5f * (u[1 * 4 + c] + u[2 * 4 + c]) + u[3 * 4 + c];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          *reinterpret_cast<f32x4*>(dst + (a * 3 + 0) * p.C) = h[a][0] + 0.5f * (h[a][1] + h[a][2]);
          *reinterpret_cast<f32x4*>(dst + (a * 3 + 1) * p.C) = 0.5f * (h[a][1] - h[a][2]);
          *reinterpret_cast<f32x4*>(dst + (a * 3 + 2) * p.C) = 0.5f * (h[a][1] + h[a][2]) + h[a][3];
        }
      } else {
        f32x4 h[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          h[0][c] = u[0 * 4 + c] + u[1 * 4 + c] + u[2 * 4 + c];
          h[1][c] = u[1 * 4 + c] - u[2 * 4 + c] - u[3 * 4 + c];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int ky = 2 * a + ph_p;
          *reinterpret_cast<f32x4*>(dst + (ky * 4 + 0 + ph_q) * p.C) = h[a][0] + h[a][1] + h[a][2];
          *reinterpret_cast<f32x4*>(dst + (ky * 4 + 2 + ph_q) * p.C) = h[a][1] - h[a][2] - h[a][3];
        }
      }
    }
  }
}

// ---- filter transform: U[(out phase)][n_tile][chunk][pos][c/4][64][c%4], one thread per (n, k) pair ----
// variant 1, F(2x2,3x3):  U = G g G^T;  kind 0 (forward): g = w[n][c][ky][kx];  kind 1 (input gradient): g = w[c][n][2-ky][2-kx]
// variant 2, F(3x3,2x2):  U = A g A^T;  kind 0 (4x4 stride-2 conv): reduce index k = (input phase pq, c),
//   g[a][b] = w[n][c][2a+p][2b+q];  kind 1 (its transpose): one image per OUTPUT phase rs, g[a][b] = w[c][n][3-2a-r][3-2b-s]
__global__ __launch_bounds__(256) void wino_pack_kernel(WinoPackParams p) {
  const long long total = wino_pack_total(p);
  if (wino43_pack_block_ok(p)) {       // same code as the multi-layer launch (pack_multi_kernel): identical bytes either way
    __shared__ float tile[32 * WP43_ROW];
    for (long long base = (long long)blockIdx.x * 256; base < total; base += (long long)gridDim.x * 256)
      wino43_pack_block(p, base, tile);
    return;
  }
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x)
    wino_pack_item(p, idx);
}

static double conv_flops_of(const srgan_conv_desc* d) {
  return 2.0 * d->N * d->Ho * d->Wo * (double)d->O * d->kh * d->kw * d->I;
}

// ---- host side ----
// variant 1: kind 0: y = conv(x, w); kind 1: dx = conv(dy, flipped w^T) with pad' = 2 - pad (zero pad) or the full
// correlation onto the reflect-padded image (pad' = 2, output Hi+2p), folded by the caller.
// variant 2: kind 0: the 4x4 stride-2 conv (MODE 1); kind 1: its input gradient / the transposed conv (MODE 2).
static bool wino_disabled() {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_WINOGRAD");
  return off;
}
// Dispatch thresholds scale with this factor (default 1; 0 = take every geometrically valid layer).  Read on every call so
// that the unit tests can drive small shapes through the Winograd kernels and through the default dispatch in one process.
double wino_threshold_scale() {
  const char* e = std::getenv("SRGAN_WINOGRAD_THRESHOLD_SCALE");
  return e ? std::atof(e) : 1.0;
}
static bool wino43_disabled() {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_WINOGRAD43");
  return off;
}
static bool wino_s2_disabled() {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_WINOGRAD_S2");
  return off;
}
static bool wino42_disabled() {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_WINOGRAD42");
  return off;
}

// 0: none, 1: F(2x2,3x3) on a 3x3 stride-1 pad-1 layer, 2: F(3x3,2x2) on a 4x4 stride-2 pad-1 layer,
// 3: F(4x4,3x3) (conv_wino43.hip) on a 3x3 stride-1 zero-pad-1 layer whose map is a multiple of 4 in both directions
// 7: F(4x4,2x2) (conv_wino42.hip) on a 4x4 stride-2 pad-1 layer whose output map is a multiple of 4 in both directions
// 4: bf16 mode only -- NOT a Winograd form: the direct bf16 GEMM of conv_halo16.hip (residual-trunk shapes), which shares this
//    slot's packed-filter / residual-add plumbing
static int wino_variant(const srgan_conv_desc* d, int kind) {
  if (compute_bf16()) {                                            // bf16 mode: the transforms would eat the 8-bit mantissa
    if (halo16_applicable(d, kind)) return 4;
    if (kind == 1 && halo16t_applicable(d)) return 5;               // 5: transposed 4x4 / stride-2 form (conv_halo16.hip)
    return kind == 0 && halo16s_applicable(d) ? 6 : 0;              // 6: strided 4x4 / stride-2 form
  }
  if (wino_disabled()) return 0;

  const int C = kind == 0 ? d->I : d->O, N = kind == 0 ? d->O : d->I;
  if (C % (2 * WC) != 0 || C < 32 || N < 32 || N % 4 != 0) return 0;   // an even number of 8-channel chunks; 16-byte stores
  // the gather's "outside" offset (2 GiB) must lie past the end of the source tensor
  const long long src_elems = (long long)d->N * (kind == 0 ? (long long)d->Hi * d->Wi * d->I : (long long)d->Ho * d->Wo * d->O);
  if (src_elems >= (1LL << 29)) return 0;
  if (d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1) {
    if (d->Hi < 3 || d->Wi < 3) return 0;
    // one workgroup per CU: maps too small to give ~0.4 of a device round (the encoder's 7x7 / 3x3 maps at batch 32) run a long
    // serial chunk loop on a few CUs -- measured 52 TFLOP/s against ~80 on the implicit GEMM
    const int ho = kind == 0 ? d->Ho : d->Hi, wo = kind == 0 ? d->Wo : d->Wi;
    if (!wino43_disabled() && d->pad_mode == SRGAN_PAD_ZERO && ho % 4 == 0 && wo % 4 == 0 && N % 32 == 0 && C % 32 == 0) {
      // 64 tiles x 32 channels per workgroup, one workgroup per CU; a round costs ~2/3 of wino_kernel's and holds half as
      // many workgroups, so the same lower bound (100 of its workgroups = 50 of these) applies
      const long long b43 = ceil_div((long long)d->N * (ho / 4) * (wo / 4), 64) * (N / 32);
      if (b43 >= 50 * wino_threshold_scale()) return 3;
    }
    const long long blocks = ceil_div((long long)d->N * ceil_div(ho, 2) * ceil_div(wo, 2), WT) * ceil_div(N, WNB);
    return blocks >= 100 * wino_threshold_scale() ? 1 : 0;
  }
  if (d->kh == 4 && d->kw == 4 && d->stride == 2 && d->pad == 1 && d->pad_mode == SRGAN_PAD_ZERO && !wino_s2_disabled()) {
    if ((d->Hi & 1) || (d->Wi & 1)) return 0;
    // F(4x4,2x2): exact 4x4 tiling, 32 tiles x 64 channels per workgroup (one workgroup per CU); the same lower bound on the
    // device fill as below
    // (its epilogue addresses the destination through a buffer descriptor with 32-bit byte offsets: < 4 GiB)
    const long long dst_elems = (long long)d->N * (kind == 0 ? (long long)d->Ho * d->Wo * d->O : (long long)d->Hi * d->Wi * d->I);
    // (strided form: at most 128 input channels = 512 reduce terms per position -- with 256 the result of one random geometry
    //  was 2.07e-5 of the tensor's maximum away from the direct convolution, over this library's 2e-5 per-op bound: the points
    //  0, 1, -1, 2 of F(4,2) cost a little accuracy against F(3,2)'s 0, 1, -1; no layer of the networks has more than 128 there)
    if (!wino42_disabled() && d->Ho % 4 == 0 && d->Wo % 4 == 0 && N % 64 == 0 && C % 16 == 0 && dst_elems < (1LL << 30) &&
        (kind == 1 || C <= 128)) {
      const long long b42 = ceil_div((long long)d->N * (d->Ho / 4) * (d->Wo / 4), 32) * (N / 64) * (kind == 1 ? 4 : 1);
      if (b42 >= 120 * wino_threshold_scale()) return 7;
    }
    // 3x3 output tiles over Ho x Wo (= the phase image of the transposed form): skip maps the tiling wastes
    const long long th = ceil_div(d->Ho, 3), tw = ceil_div(d->Wo, 3);
    const double eff = (double)d->Ho * d->Wo / (9.0 * th * tw);
    if (eff * 2.25 < 1.5) return 0;
    // one workgroup per CU: below ~half a device round the long (4 C / 8 chunk) K loop loses to the implicit GEMM (measured:
    // D.c2 at batch 32, 122 workgroups: 94 us against 107 us; D.c4, 16 workgroups: 277 us against 130 us)
    const long long blocks = ceil_div((long long)d->N * th * tw, WT) * ceil_div(N, WNB) * (kind == 1 ? 4 : 1);
    return blocks >= 120 * wino_threshold_scale() ? 2 : 0;
  }
  return 0;
}

bool wino_applicable(const srgan_conv_desc* d, int kind) { return wino_variant(d, kind) != 0; }

static void wino_dims(const srgan_conv_desc* d, int kind, int* C, int* N, int* n_tiles, int* nchunk, int* phases) {
  const int v = wino_variant(d, kind);
  *C = kind == 0 ? d->I : d->O;
  *N = kind == 0 ? d->O : d->I;
  *n_tiles = v == 7 ? *N / 64 : (v >= 4) ? 1 : v == 3 ? *N / 32 : (int)ceil_div(*N, WNB);
  *nchunk = v == 7 ? (kind == 0 ? 4 : 1) * (*C / 16) : (v >= 4) ? *C / 32 : (v == 2 && kind == 0 ? 4 : 1) * (*C / WC);      // MODE 1 reduces over (input phase, channel)
  *phases = ((v == 2 || v == 7) && kind == 1) ? 4 : 1;                  // (variant 5 keeps its four phases inside one packed image)                  // MODE 2: one filter image per output phase
}

size_t wino_packed_bytes(const srgan_conv_desc* d, int kind) {
  int C, N, n_tiles, nchunk, phases;
  wino_dims(d, kind, &C, &N, &n_tiles, &nchunk, &phases);
  if (wino_variant(d, kind) == 4) return halo16_packed_bytes(d);
  if (wino_variant(d, kind) == 6) return halo16s_packed_bytes(d);
  if (wino_variant(d, kind) == 5) return halo16t_packed_bytes(d);
  if (wino_variant(d, kind) == 3) return (size_t)n_tiles * nchunk * (36 * 256) * sizeof(float);
  if (wino_variant(d, kind) == 7) return (size_t)phases * n_tiles * nchunk * (25 * 1024) * sizeof(float);
  return (size_t)phases * n_tiles * nchunk * 8192 * sizeof(float);
}

// scratch the run needs beside the packed filters: the transformed-input image of the F(4x4,3x3) pair of kernels
size_t wino_scratch_bytes(const srgan_conv_desc* d, int kind) {
  if (wino_variant(d, kind) != 3) return 0;
  const int ho = kind == 0 ? d->Ho : d->Hi, wo = kind == 0 ? d->Wo : d->Wi;
  return wino43_scratch_floats((long long)d->N * (ho / 4) * (wo / 4), kind == 0 ? d->I : d->O) * sizeof(float);
}

// Weight gradient of an F(4x4,3x3) layer from the V image its forward wrote (conv_wino43.hip): applicable when the forward runs
// that path, Cin % 64 == 0, Cout % 32 == 0; split-K over 8-tile chunks so that ~256 workgroups run.
bool wino43_wgrad_geometry(const srgan_conv_desc* d, Wino43WgradGeom* g) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_WINOGRAD43_WGRAD");
  if (off || wino_variant(d, 0) != 3 || d->I % 64 != 0 || d->O % 32 != 0) return false;
  const long long T = (long long)d->N * (d->Ho / 4) * (d->Wo / 4);
  const long long ntc = ceil_div(T, 64) * 8;
  const int tiles = (d->O / 32) * (d->I / 64);
  long long splits = std::max(1, 256 / tiles);
  long long cps = std::max<long long>(2, ceil_div(ntc, splits));
  if (ntc % cps == 1) ++cps;                       // the last split needs two chunks too
  splits = ceil_div(ntc, cps);
  if (ntc < 2 || ntc - (splits - 1) * cps < 2) return false;
  g->NB = d->N; g->H = d->Ho; g->W = d->Wo; g->C = d->I; g->O = d->O; g->ntc = (int)ntc;
  g->splits = (int)splits; g->chunks_per_split = (int)cps;
  g->v_bytes = wino_scratch_bytes(d, 0);
  g->z_bytes = (size_t)(d->O / 32) * ntc * 36 * 256 * sizeof(float);
  g->slab_bytes = (size_t)splits * d->O * 9 * d->I * sizeof(float);
  return g->v_bytes < (1ULL << 32) && g->z_bytes < (1ULL << 32);
}

void wino_pack_params(const srgan_conv_desc* d, int kind, const float* w, float* dst, WinoPackParams* out) {
  WinoPackParams q{};
  int C, N;
  wino_dims(d, kind, &C, &N, &q.n_tiles, &q.nchunk, &q.phases);
  q.variant = wino_variant(d, kind);
  q.w = w; q.dst = dst; q.sO = d->sO; q.sI = d->sI; q.sH = d->sH; q.sW = d->sW; q.N = N; q.C = C; q.kind = kind;
  // A/B switch (read per call: the unit test packs both ways in one process): one thread per item, uncoalesced reads
  q.no_block = std::getenv("SRGAN_PACK_ITEM_PATH") != nullptr ? 1 : 0;
  *out = q;
}

int wino_pack(const srgan_conv_desc* d, int kind, const float* w, float* dst, hipStream_t st) {
  WinoPackParams q{};
  wino_pack_params(d, kind, w, dst, &q);
  const long long total = wino_pack_total(q);
  hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0, st, q);
  return check_launch("wino_pack_kernel");
}

// dst geometry: kind 0 -> [N][Ho][Wo][O]; kind 1 -> [N][Hd][Wd][I] with Hd = Hi (zero pad) or Hi + 2 (reflect scratch)
// `scratch`: wino_scratch_bytes(d, kind) bytes (may be null when that is 0)
// `res` (optional): added to the result in the epilogue where the kernel supports it (F(4x4,3x3)); *res_done tells the caller
// whether it was, so that it can add the tensor itself otherwise
bool wino43_fwd_applicable(const srgan_conv_desc* d) { return wino_variant(d, 0) == 3; }
bool wino43_dgrad_applicable(const srgan_conv_desc* d) { return d->pad_mode == SRGAN_PAD_ZERO && wino_variant(d, 1) == 3; }

int wino_run(const srgan_conv_desc* d, int kind, const float* src, const float* packed, const float* bias, float* dst,
             int act, float slope, float* scratch, hipStream_t st, const float* res, bool* res_done, bool v_ready,
             const float* mask, float mask_slope, bool* mask_done) {
  if (res_done) *res_done = false;
  if (mask_done) *mask_done = false;
  WinoParams p{};
  p.act = act; p.slope = slope;
  int C, N, phases;
  wino_dims(d, kind, &C, &N, &p.n_tiles, &p.nchunk, &phases);
  const int variant = wino_variant(d, kind);
  SRGAN_REQUIRE(variant != 0, "winograd: layer not applicable");
  if (variant == 6) {
    SRGAN_REQUIRE(!v_ready && !res && !mask, "halo16s: no prepared image, skip tensor or mask on this path");
    return halo16s_run(d, src, packed, bias, dst, act, slope, conv_flops_of(d), st);
  }
  if (variant == 5) {
    SRGAN_REQUIRE(!v_ready && !bias && act == SRGAN_ACT_NONE, "halo16t: plain transposed product only");
    return halo16t_run(d, src, packed, dst, conv_flops_of(d), st);
  }
  if (variant == 4) {
    SRGAN_REQUIRE(!v_ready, "halo16: no transformed-input image on this path");
    if (res_done) *res_done = res != nullptr;
    return halo16_run(d, kind, src, packed, bias, res, dst, act, slope, conv_flops_of(d), st);
  }
  const bool reflect = d->pad_mode == SRGAN_PAD_REFLECT;
  p.src = src; p.u = packed; p.bias = bias; p.dst = dst;
  p.NB = d->N; p.C = C; p.Cd = N; p.cpp = C / WC;
  int ot = 2;
  if (variant == 1 || variant == 3) {
    if (kind == 0) {
      p.H = d->Hi; p.W = d->Wi; p.Ho = d->Ho; p.Wo = d->Wo; p.pad = d->pad; p.reflect = reflect ? 1 : 0;
    } else {
      p.H = d->Ho; p.W = d->Wo; p.reflect = 0;
      p.pad = reflect ? 2 : 2 - d->pad;
      p.Ho = d->Ho + 2 * p.pad - 2; p.Wo = d->Wo + 2 * p.pad - 2;
    }
    if (variant == 3) { ot = 4; p.TH = p.Ho / 4; p.TW = p.Wo / 4; }
    else { p.TH = (p.Ho + 1) / 2; p.TW = (p.Wo + 1) / 2; }
  } else {
    ot = variant == 7 ? 4 : 3;
    p.pad = d->pad; p.reflect = 0;
    if (kind == 0) { p.H = d->Hi; p.W = d->Wi; p.Ho = d->Ho; p.Wo = d->Wo; }
    else { p.H = d->Ho; p.W = d->Wo; p.Ho = d->Hi; p.Wo = d->Wi; }
    p.TH = (int)ceil_div(d->Ho, ot); p.TW = (int)ceil_div(d->Wo, ot);      // tiles over Ho x Wo (kind 1: the phase image)
    if (variant == 7) p.cpp = C / 16;
  }
  (void)ot;
  const long long T = (long long)p.NB * p.TH * p.TW;
  SRGAN_REQUIRE(T < (1LL << 30), "winograd: too many tiles");
  p.T = (int)T;
  p.m_tiles = (int)ceil_div(T, variant == 7 ? 32 : WT);
  const long long grid = (long long)p.m_tiles * p.n_tiles;
  SRGAN_REQUIRE(grid < (1LL << 31), "winograd: grid too large");
  // ALGORITHMIC FLOPs of the direct convolution; the kernel issues ~2.25x fewer on the matrix pipe
  if (variant == 3) {
    SRGAN_REQUIRE(p.pad == 1 && p.Ho == p.H && p.Wo == p.W && p.nchunk >= 2 && p.Cd % 32 == 0 && p.C % 32 == 0,
                  "winograd F(4,3): geometry");
    SRGAN_REQUIRE(scratch, "winograd F(4,3): no scratch for the transformed input");
    p.res = res;
    if (res_done) *res_done = res != nullptr;
    wino43_launch(p, scratch, grid, conv_flops_of(d), st, v_ready);
    return check_launch("wino43_kernel");
  }
  SRGAN_REQUIRE(!v_ready, "winograd: a prepared V image only serves the F(4x4,3x3) path");
  if (variant == 7) {
    SRGAN_REQUIRE(p.pad == 1 && p.nchunk >= 2 && p.Cd % 64 == 0 && p.C % 16 == 0 && d->Ho % 4 == 0 && d->Wo % 4 == 0, "winograd F(4,2): geometry");
    if (kind == 1 && mask) {
      p.mask = mask; p.mask_slope = mask_slope;
      if (mask_done) *mask_done = true;
    }
    return wino42_launch(p, kind, grid, kind == 1 && mask != nullptr, conv_flops_of(d), st);
  }
  ProfToken tok = prof_begin(variant == 1 ? 14 : 16, conv_flops_of(d), st);
  if (variant == 1) hipLaunchKernelGGL(wino_kernel<0>, dim3((unsigned)grid), dim3(512), 0, st, p);
  else if (kind == 0) hipLaunchKernelGGL(wino_kernel<1>, dim3((unsigned)grid), dim3(512), 0, st, p);
  else if (mask) {
    p.mask = mask; p.mask_slope = mask_slope;
    if (mask_done) *mask_done = true;
    hipLaunchKernelGGL((wino_kernel<2, true>), dim3((unsigned)grid, 4), dim3(512), 0, st, p);
  } else hipLaunchKernelGGL(wino_kernel<2>, dim3((unsigned)grid, 4), dim3(512), 0, st, p);
  prof_end(tok, st);
  return check_launch("wino_kernel");
}

// ---- weight gradient host side ----
// returns 0 (not applicable), 1 (3x3 stride-1) or 2 (4x4 stride-2)
static int wino_wgrad_geometry(const srgan_conv_desc* d, WinoWgradParams* p) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_WINOGRAD_WGRAD");
  if (wino_disabled() || off || compute_bf16()) return 0;
  int variant = 0;
  if (d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1) variant = 1;
  else if (d->kh == 4 && d->kw == 4 && d->stride == 2 && d->pad == 1 && d->pad_mode == SRGAN_PAD_ZERO && !(d->Hi & 1) &&
           !(d->Wi & 1) && !wino_s2_disabled())
    variant = 2;
  if (variant == 0) return 0;
  if (d->I % 64 != 0 || d->O % 64 != 0 || d->Hi < 3 || d->Wi < 3) return 0;
  if ((long long)d->N * d->Hi * d->Wi * d->I >= (1LL << 29) || (long long)d->N * d->Ho * d->Wo * d->O >= (1LL << 29)) return 0;
  const int ot = variant == 1 ? 2 : 3;
  p->NB = d->N; p->H = d->Hi; p->W = d->Wi; p->C = d->I; p->Ho = d->Ho; p->Wo = d->Wo; p->O = d->O;
  p->pad = d->pad; p->reflect = d->pad_mode == SRGAN_PAD_REFLECT ? 1 : 0;
  p->TH = (int)ceil_div(d->Ho, ot); p->TW = (int)ceil_div(d->Wo, ot);
  if (variant == 2 && (double)d->Ho * d->Wo / ((double)ot * ot * p->TH * p->TW) * 2.25 < 1.5) return 0;   // tiling waste
  p->groups = (int)ceil_div((long long)p->TH * p->TW, 8);
  // tile positions are consumed in groups of 8: a 3x3 grid of tiles (9 positions) would run 16 slots
  if (variant == 2 && (double)d->Ho * d->Wo / ((double)ot * ot * 8 * p->groups) * 2.25 < 1.5) return 0;
  p->o_tiles = d->O / 64; p->i_tiles = (variant == 2 ? 4 : 1) * d->I / 64; p->Opad = d->O;
  const int tiles = p->o_tiles * p->i_tiles;
  // one workgroup per CU (128 KB of LDS): fill whole rounds of 256
  int splits = std::max(1, 256 / tiles);
  splits = std::min(splits, p->groups);
  p->groups_per_split = (int)ceil_div(p->groups, splits);
  p->splits = (int)ceil_div(p->groups, p->groups_per_split);
  p->bsplits = 1; p->nb_per = d->N;
  long long blocks = (long long)tiles * p->splits, chunks = (long long)p->groups_per_split * d->N;
  const double ts = wino_threshold_scale();
  // few tile positions (the discriminator's 32x32 / 16x16 maps at batch 64): the groups alone do not fill the device, so the
  // K range (groups x images) is cut along the batch as well: the (group splits, batch splits) pair with the most
  // workgroups within one device round wins, ties go to the longer K range per workgroup, then to the longer batch range
  static const int min_chunks_b = SRGAN_AB_INT("SRGAN_WGRAD_BSPLIT_MIN_CHUNKS", 24);
  if (variant == 2 && min_chunks_b > 0 && blocks < 192) {
    const int S = std::max(1, 256 / tiles);
    long long best_blocks = blocks, best_chunks = chunks;
    int best_gps = p->groups_per_split, best_gs = p->splits, best_bs = 1, best_nb = d->N;
    for (int gs = 1; gs <= std::min(p->groups, S); ++gs) {
      const int gps = (int)ceil_div(p->groups, gs), gse = (int)ceil_div(p->groups, gps);
      const int bs = std::max(1, std::min(d->N, S / gse));
      const int nb = (int)ceil_div(d->N, bs), bse = (int)ceil_div(d->N, nb);
      const long long bl = (long long)tiles * gse * bse, ch = (long long)gps * nb;
      if (ch < min_chunks_b) continue;
      // (ties in both: the longer batch range -- the gather offsets are rebuilt every time a workgroup moves to its next group)
      if (bl > best_blocks || (bl == best_blocks && (ch > best_chunks || (ch == best_chunks && nb > best_nb)))) {
        best_blocks = bl; best_chunks = ch; best_gps = gps; best_gs = gse; best_bs = bse; best_nb = nb;
      }
    }
    if (best_bs > 1) {
      p->groups_per_split = best_gps; p->bsplits = best_bs; p->nb_per = best_nb;
      p->splits = best_gs * best_bs;
      blocks = best_blocks; chunks = best_chunks;
    }
  }
  if (variant == 1) {
    if (blocks < 16 * ts || chunks < 4 * ts) return 0;      // too few workgroups / too short a K range: implicit GEMM instead
  } else {
    static const int min_blocks = SRGAN_AB_INT("SRGAN_WGRAD_S2_MIN_BLOCKS", 192);
    if (blocks < min_blocks * ts || chunks < (p->bsplits > 1 ? min_chunks_b : 48) * ts) return 0;    // measured: the discriminator's smallest maps stay faster on the implicit GEMM
  }
  return variant;
}

// bf16 mode: the slot serves the residual-trunk shapes through conv_halo16.hip's weight-gradient kernel (not a Winograd form)
bool wino_wgrad_applicable(const srgan_conv_desc* d) {
  if (compute_bf16()) return halo16_wgrad_applicable(d);
  WinoWgradParams p{};
  return wino_wgrad_geometry(d, &p) != 0;
}

// slab geometry for wgrad_reduce_kernel: [splits][Cdpad = O][NNpad = kh * kw * I]
void wino_wgrad_slab(const srgan_conv_desc* d, int* splits, int* Cdpad, int* NNpad) {
  if (compute_bf16()) { halo16_wgrad_slab(d, splits, Cdpad, NNpad); return; }
  WinoWgradParams p{};
  wino_wgrad_geometry(d, &p);
  *splits = p.splits; *Cdpad = p.Opad; *NNpad = d->kh * d->kw * d->I;
}

int wino_wgrad_run(const srgan_conv_desc* d, const float* x, const float* dy, float* slab, hipStream_t st) {
  if (compute_bf16()) return halo16_wgrad_run(d, x, dy, slab, conv_flops_of(d), st);
  WinoWgradParams p{};
  const int variant = wino_wgrad_geometry(d, &p);
  SRGAN_REQUIRE(variant != 0, "winograd wgrad: layer not applicable");
  p.x = x; p.dy = dy; p.slab = slab;
  p.xcd_splits = (p.splits % 8 == 0) ? 1 : 0;
  ProfToken tok = prof_begin(variant == 1 ? 15 : 17, conv_flops_of(d), st);   // algorithmic FLOPs, as above
  const dim3 grid((unsigned)(p.o_tiles * p.i_tiles * p.splits));
  if (variant == 1) hipLaunchKernelGGL(wino_wgrad_kernel<0>, grid, dim3(512), 0, st, p);
  else hipLaunchKernelGGL(wino_wgrad_kernel<1>, grid, dim3(512), 0, st, p);
  prof_end(tok, st);
  return check_launch("wino_wgrad_kernel");
}

}  // namespace srgan

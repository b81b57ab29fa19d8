// bf16 compute mode (BASELINE configs [2]-[4]), round 6: the style encoder's 3x3 / stride-1 convolutions on its large maps
// (reference pyfiles/model.py:413-437: `BasicBlock_classification` -- conv1 C -> C and cmp C -> 2C, reflect-padded, on 62 x 62 and
// 31 x 31 maps) with the ACTIVATION operand resident in LDS and the FILTER operand fed global -> registers: halo16r_kernel's shape
// (conv_halo16.hip) generalised to
//   * 64 or 128 reduce channels (one or two 64-channel quarters; the second streams in under the first one's products),
//   * any map size: a workgroup owns a (4 WM) x 32 pixel patch of the DESTINATION map, ragged at the right / bottom edge (masked
//     stores), WM x WN = 4 waves of 128 pixels x 64 channels each (8 MFMAs per 16-deep K step, 128 accumulator registers),
//   * reflect or zero padding (mirrored coordinates only in the halo fill), and a destination that may be LARGER than the source:
//     the input gradient of a reflect-padded layer is the full correlation of dy (zero outside) written as the (H + 2) x (W + 2)
//     gradient of the padded image (folded by reflect_fold_kernel afterwards, as on the implicit GEMM): halo origin and tap order
//     are parameters (`oy0`, `ox0`, `flip`),
//   * fp32 or bf16 tensors on either side.
// The implicit GEMM (igemm16_kernel) re-gathers every activation tile once per tap and spends as long in the prologue and
// epilogue of its 9-18-tile K loops as in the loops; here an activation byte enters LDS once per workgroup.
// Filter operand: the register image written by pack_weights_store (pack_device.h, PackParams::regimg):
// [K step = (quarter * 9 + tap) * 4 + s][64-channel block of N][32-channel half][lane][8 bf16] -- 1 KB contiguous per fragment.
#include <algorithm>
#include "common.h"

namespace srgan {
namespace {

__device__ __forceinline__ auto uniform_rsrc_e(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// <64, 2, 2> (64 -> 128 channels, 49 KB of LDS) is compiled for TWO workgroups per CU (212 registers, no spill): with one
// 64-channel quarter its 36-step K loop is as short as its halo fill and its result stores, and a second resident workgroup is
// what runs under them; the other shapes are bound to one by their LDS or accumulators.
template <int CIN, int WM, int WN, bool IN16, bool OUT16>
__global__ __launch_bounds__(256, (CIN == 64 && WM == 2) ? 2 : 1) void halo16e_kernel(Halo16eParams p) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int ISZ = IN16 ? 2 : 4;
  constexpr int PH = 4 * WM;                   // patch rows
  constexpr int HR = PH + 2, HPX = HR * 34;    // halo rows / pixels
  constexpr int PS = CIN * 2 + 16;             // bytes per halo pixel (16-byte pad: conflict-free fragment reads)
  constexpr int NQ = CIN / 64;
  constexpr int QS = 36;                       // K steps of 16 per quarter: 9 taps x 4
  constexpr int NKS = NQ * QS;
  constexpr int RING = 6, PF = RING - 1;
  constexpr int HP = (HPX + 31) / 32;          // passes of 32 pixels over the halo of one quarter
  constexpr int PPT = (HP + 7) / 8;            // passes per tap while the next quarter streams in (taps 0 .. 7 request, 1 .. 8 park)
  __shared__ __attribute__((aligned(16))) unsigned char halo[HPX * PS];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int lr = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);      // neighbouring patches (shared halo rows) on one XCD
  int r = bid;
  const int nt = r % p.n_tiles; r /= p.n_tiles;
  const int tx = r % p.tiles_x; r /= p.tiles_x;
  const int ty = r % p.tiles_y;
  const int nb = r / p.tiles_y;
  const int Y0 = ty * PH, X0 = tx * 32;        // patch origin in the destination map

  const auto rs_x = uniform_rsrc_e(p.src, (unsigned)((size_t)p.NB * p.Hs * p.Ws * CIN * ISZ));
  const int nblocks = p.N >> 6;                // 64-channel blocks of the whole layer
  const auto rs_w = uniform_rsrc_e(p.wp, (unsigned)((size_t)NKS * nblocks * 2048));      // past the image: zeros

  // ---- B ring: fragment (K step kw, channel half j) of this wave's 64-channel block nt * WN + wn ----
  bf16x8 fb[RING][2];
  const int w_lane = lane * 16, w_blk = (nt * WN + wn) * 2048, w_step = nblocks * 2048;
  // K step kg = q * 36 + tap * 4 + s in HALO tap order; flip: the filter image holds tap 8 - tap there
  auto load_b = [&](int slot, int kg) __attribute__((always_inline)) {
    int kw = kg;
    if (p.flip) {
      const int q = kg / QS, t = kg - q * QS;
      kw = q * QS + (8 - (t >> 2)) * 4 + (t & 3);
    }
    const int off = kg < NKS ? kw * w_step + w_blk : (1 << 30);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      fb[slot][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, off + j * 1024, 0));
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) load_b(s, s);

  // ---- halo: one 64-channel quarter at a time; thread = (pixel of the pass, 8 channels), 32 pixels per pass ----
  const int hcg = tid & 7, hpl = tid >> 3;
  constexpr unsigned kOutside = 0x80000000u;
  auto halo_off = [&](int quarter, int pass) __attribute__((always_inline)) -> unsigned {
    const int hp = pass * 32 + hpl;
    const int hr = hp / 34, hc = hp - hr * 34;
    int y = Y0 + p.oy0 + hr, x = X0 + p.ox0 + hc;
    bool ok = hp < HPX;
    if (p.reflect) {
      // pad 1: -1 -> 1, Hs -> Hs - 2; further out (only under a ragged patch's masked outputs): clamped
      y = y < 0 ? -y : y; y = y >= p.Hs ? 2 * p.Hs - 2 - y : y;
      x = x < 0 ? -x : x; x = x >= p.Ws ? 2 * p.Ws - 2 - x : x;
      y = min(max(y, 0), p.Hs - 1); x = min(max(x, 0), p.Ws - 1);
    } else {
      ok = ok && y >= 0 && y < p.Hs && x >= 0 && x < p.Ws;
    }
    return ok ? (unsigned)((((nb * p.Hs + y) * p.Ws + x) * CIN + quarter * 64 + hcg * 8) * ISZ) : kOutside;
  };
  auto halo_put = [&](int quarter, int pass, f32x4 lo, f32x4 hi) __attribute__((always_inline)) {
    const int hp = pass * 32 + hpl;
    if (hp < HPX) {
      if constexpr (IN16) {
        *reinterpret_cast<f32x4*>(&halo[hp * PS + quarter * 128 + hcg * 16]) = lo;
      } else {
        const bf16x4 a = __builtin_convertvector(lo, bf16x4), b = __builtin_convertvector(hi, bf16x4);
        bf16x8 v;
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        *reinterpret_cast<bf16x8*>(&halo[hp * PS + quarter * 128 + hcg * 16]) = v;
      }
    }
  };
  // quarter 0 in chunks of up to 8 passes (bounded staging registers)
#pragma unroll
  for (int c0 = 0; c0 < HP; c0 += 8) {
    constexpr int CH = 8;
    f32x4 lo[CH], hi[CH];
#pragma unroll
    for (int g = 0; g < CH; ++g) {
      if (c0 + g < HP) {
        const unsigned off = halo_off(0, c0 + g);
        lo[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
        if constexpr (!IN16) hi[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 16, 0));
        else hi[g] = lo[g];
      }
    }
#pragma unroll
    for (int g = 0; g < CH; ++g)
      if (c0 + g < HP) halo_put(0, c0 + g, lo[g], hi[g]);
  }
  __syncthreads();

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // lane's A address: pixel (patch row 4 wm, column lr) at tap (0, 0) = halo pixel (4 wm) * 34 + lr, channels 8 lh .. of the K step
  const unsigned char* a_lane = halo + ((4 * wm) * 34 + lr) * PS + lh * 16;
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

  f32x4 hlo[PPT], hhi[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) { hlo[k] = f32x4{0.f, 0.f, 0.f, 0.f}; hhi[k] = hlo[k]; }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const unsigned char* a_q = a_lane + q * 128;
    read_a(0, a_q);                                    // (tap 0, step 0) of this quarter: published by the barrier just passed
#pragma unroll
    for (int t = 0; t < QS; ++t) {
      const int tap = t >> 2, s = t & 3;
      // the next quarter of the halo streams in under this quarter's products: PPT passes are requested at a tap's first K
      // step and parked in LDS one tap later; the barrier at the end of the quarter publishes them
      if (s == 0 && q + 1 < NQ) {
        if (tap >= 1) {
#pragma unroll
          for (int k = 0; k < PPT; ++k)
            if ((tap - 1) * PPT + k < HP) halo_put(q + 1, (tap - 1) * PPT + k, hlo[k], hhi[k]);
        }
        if (tap < 8) {
#pragma unroll
          for (int k = 0; k < PPT; ++k)
            if (tap * PPT + k < HP) {
              const unsigned off = halo_off(q + 1, tap * PPT + k);
              hlo[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
              if constexpr (!IN16) hhi[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 16, 0));
            }
        }
      }
      if (t + 1 < QS) {
        const int t2 = t + 1, tap2 = t2 >> 2, s2 = t2 & 3;
        read_a(t2 & 1, a_q + ((tap2 / 3) * 34 + tap2 % 3) * PS + s2 * 32);
      }
      load_b((t + PF) % RING, q * QS + t + PF);
      __builtin_amdgcn_sched_barrier(0);
      mma(t & 1, t % RING);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q + 1 < NQ) __syncthreads();
  }

  // ---- epilogue: lane = output channel; register e of acc[i][j] = pixel column (e % 4) + 8 (e / 4) + 4 lh of destination row
  // Y0 + 4 wm + i; stores masked at the map's right / bottom edge.  A bf16 result: the wave's 64 channels of a pixel are ONE
  // 128-byte line, so lanes are paired as in igemm16_kernel -- the even lane of a pair stores both lanes' values of channel block
  // 0 (columns lr, lr + 1), the odd lane those of block 1: 4 bytes per lane, a half-wave writes the whole line, half the store
  // instructions (with a column per lane every store is a 64-byte partial line) ----
  const int nb0 = (nt * WN + wn) * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int y = Y0 + 4 * wm + i;
    if (y >= p.Hd) continue;
    const size_t row = ((size_t)(nb * p.Hd + y) * p.Wd) * p.N;
    if constexpr (OUT16) {
      const int odd = lr & 1;
      const int n = nb0 + (odd ? 31 + lr : lr);                 // first of this lane's two adjacent columns
      const float b0 = p.bias ? p.bias[n] : 0.f, b1 = p.bias ? p.bias[n + 1] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int x = X0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const float own = odd ? acc[i][1][e] : acc[i][0][e], give = odd ? acc[i][0][e] : acc[i][1][e];
        const float got = __shfl_xor(give, 1, 64);
        const float v0 = apply_act((odd ? got : own) + b0, p.act, p.slope), v1 = apply_act((odd ? own : got) + b1, p.act, p.slope);
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 pk = {(__bf16)v0, (__bf16)v1};
        if (x < p.Wd) *reinterpret_cast<bf16x2*>(static_cast<__bf16*>(p.dst) + row + (size_t)x * p.N + n) = pk;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = nb0 + j * 32 + lr;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int x = X0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          if (x < p.Wd) static_cast<float*>(p.dst)[row + (size_t)x * p.N + n] = apply_act(acc[i][j][e] + bv, p.act, p.slope);
        }
      }
    }
  }
}

template <int CIN, int WM, int WN>
void launch_e(const Halo16eParams& p, unsigned grid, hipStream_t st) {
  const dim3 g(grid), b(256);
  if (p.src16 && p.dst16) hipLaunchKernelGGL((halo16e_kernel<CIN, WM, WN, true, true>), g, b, 0, st, p);
  else if (p.src16) hipLaunchKernelGGL((halo16e_kernel<CIN, WM, WN, true, false>), g, b, 0, st, p);
  else if (p.dst16) hipLaunchKernelGGL((halo16e_kernel<CIN, WM, WN, false, true>), g, b, 0, st, p);
  else hipLaunchKernelGGL((halo16e_kernel<CIN, WM, WN, false, false>), g, b, 0, st, p);
}

}  // namespace

// the (reduce channels, output channels) pairs the kernel is instantiated for: 64 -> 64 / 128, 128 -> 128 / 256
bool halo16e_shape_ok(int Cs, int N) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_HALO16E");
  if (off) return false;
  return (Cs == 64 && (N == 64 || N == 128)) || (Cs == 128 && (N == 128 || N == 256));
}

// patch rows of the workgroup that serves (Cs, N): 4 WM
int halo16e_patch_rows(int Cs, int N) {
  if (Cs == 64) return N == 64 ? 16 : 8;
  return N == 128 ? 8 : 4;
}

int halo16e_run(Halo16eParams p, double flops, hipStream_t st) {
  SRGAN_REQUIRE(halo16e_shape_ok(p.Cs, p.N), "halo16e: shape not served");
  const int ph = halo16e_patch_rows(p.Cs, p.N);
  p.tiles_y = (int)ceil_div(p.Hd, ph); p.tiles_x = (int)ceil_div(p.Wd, 32);
  const int nwg = p.Cs == 64 ? (p.N == 64 ? 64 : 128) : (p.N == 128 ? 128 : 256);      // output channels per workgroup
  p.n_tiles = p.N / nwg;
  const long long grid = (long long)p.NB * p.tiles_y * p.tiles_x * p.n_tiles;
  SRGAN_REQUIRE(grid > 0 && grid < (1LL << 31), "halo16e: grid");
  ProfToken tok = prof_begin(36, flops, st);      // (the profile slot of the implicit GEMM it replaces on these layers)
  if (p.Cs == 64 && p.N == 64) launch_e<64, 4, 1>(p, (unsigned)grid, st);
  else if (p.Cs == 64) launch_e<64, 2, 2>(p, (unsigned)grid, st);
  else if (p.N == 128) launch_e<128, 2, 2>(p, (unsigned)grid, st);
  else launch_e<128, 1, 4>(p, (unsigned)grid, st);
  prof_end(tok, st);
  return check_launch("halo16e_kernel");
}

}  // namespace srgan

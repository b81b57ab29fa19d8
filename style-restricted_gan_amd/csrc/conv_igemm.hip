// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
//   out[m][n] = sum_k A[m][k] * B[n][k]
//     m = (image, grid-y, grid-x)       rows of the destination (NHWC pixels)
//     n = destination channel
//     k = (tap-y, tap-x, source channel)  gathered on the fly from the NHWC source tensor
//
// One kernel serves three reference ops (see include/srgan_hip.h):
//   mode 0  nn.Conv2d forward           src = x,  dst = y    in_y = a*stride + ty - pad
//   mode 1  conv input-gradient and     src = dy, dst = dx   in_y = a + floor((py+pad)/s) - ty
//           nn.ConvTranspose2d forward  (one launch z-slice per output parity phase (py,px);
//                                        taps ky = (py+pad)%s + s*ty, zero-weighted if ky>=kh)
// The weight operand is first repacked to [phase][Npad][Kpad] (K contiguous, zero padded) by
// pack_weights_kernel so the main loop needs no masks on B.
//
// Tiling: 256 threads = 4 waves, block tile BM x BN x 32, each wave owns TM x TN MFMA tiles of
// 32x32.  A and B tiles are staged global -> registers -> LDS (rows padded by one 16-B access
// so the ds_read_b128 fragment reads are conflict-free); the next tile's global loads are in
// flight while the current tile is multiplied.  Within a 32-deep K tile lane (r, h) feeds the
// MFMA k-slots {8q+4h+e} so that A and B fragments come from single ds_read_b128's.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "common.h"
#include "pack_device.h"

namespace srgan {

// ---- optional in-process launch timer (bench.py roofline leg) -------------------------------
// When enabled, every implicit-GEMM launch is bracketed by HIP events on ITS stream; durations are
// read back after the timed region.  Off by default: no events, no overhead.
struct ProfSlot { hipEvent_t a, b; int kid; double flops; };
static bool g_prof_on = false;
static std::vector<ProfSlot> g_prof_slots;      // used slots of the current session
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
static size_t g_prof_next = 0;
static const char* const kProfNames[] = {
    "igemm_kernel<128,128,2,2,gen>", "igemm_kernel<128,128,2,2,vec>", "igemm_kernel<128,64,2,2,gen>", "igemm_kernel<128,64,2,2,vec>",
    "igemm_kernel<128,32,4,1,gen>",  "igemm_kernel<128,32,4,1,vec>",  "igemm_kernel<64,64,2,2,gen>",  "igemm_kernel<64,64,2,2,vec>",
    "wgrad_kernel<gen>", "wgrad_kernel<vec>", "igemm_kernel<256,128,4,2,gen>", "igemm_kernel<256,128,4,2,vec>",
    "igemm_kernel<256,64,4,2,gen>", "igemm_kernel<256,64,4,2,vec>", "wino_kernel", "wino_wgrad_kernel", "wino_kernel<4x4s2>", "wino_wgrad_kernel<4x4s2>",
    "wino43_kernel", "wino43_input_kernel", "rgbin_conv_kernel", "rgb_wgrad_kernel", "wino43_wgrad_kernel", "wino43_dy_kernel",
    "halo16_kernel", "halo16_wgrad_kernel", "halo16s2_wgrad_kernel", "halo16t_kernel", "rgbout_conv_kernel", "halo16s_kernel",
    // HBM-bound passes (norm.hip): the "flops" slot of their brackets carries ALGORITHMIC BYTES (tensor bytes each pass must move)
    "in_stats_partial", "in_apply", "in_bwd_partial", "in_bwd_apply", "in_fwd_slab", "in_bwd_slab", "igemm16_kernel",
    "wino42_kernel",
    // the fused norm + Winograd-transform kernels of the trunk (conv_wino43.hip), HBM-bound too: bytes in the "flops" slot
    "in_fwd_slab_v", "in_bwd_slab_vz"};
constexpr int kProfKernels = 40;

struct ProfScope {
  bool on;
  hipStream_t st;
  size_t idx;
  ProfScope(int kid, double flops, hipStream_t s) : on(g_prof_on), st(s), idx(0) {
    if (!on) return;
    if (g_prof_next == g_prof_pool.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
      g_prof_pool.emplace_back(a, b);
    }
    auto& ev = g_prof_pool[g_prof_next++];
    g_prof_slots.push_back({ev.first, ev.second, kid, flops});
    idx = g_prof_slots.size() - 1;
    (void)hipEventRecord(ev.first, st);
  }
  ~ProfScope() {
    if (on) (void)hipEventRecord(g_prof_slots[idx].b, st);
  }
};

// the same bracket for kernels that live in other translation units (conv_wino.hip)
ProfToken prof_begin(int kid, double flops, hipStream_t st) {
  ProfToken t{0, false};
  if (!g_prof_on) return t;
  if (g_prof_next == g_prof_pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return t;
    g_prof_pool.emplace_back(a, b);
  }
  auto& ev = g_prof_pool[g_prof_next++];
  g_prof_slots.push_back({ev.first, ev.second, kid, flops});
  t.idx = g_prof_slots.size() - 1;
  t.on = true;
  (void)hipEventRecord(ev.first, st);
  return t;
}
void prof_end(ProfToken t, hipStream_t st) {
  if (t.on) (void)hipEventRecord(g_prof_slots[t.idx].b, st);
}

// ---- compute mode: 0 = exact fp32 (default), 1 = bf16 MFMA operands with fp32 accumulation ----
static int g_compute_mode = 0;
bool compute_bf16() { return g_compute_mode == 1; }

struct IgemmParams {
  const float* src;
  const float* wp;    // packed weights [phases][Npad][Kpad]
  const float* bias;  // [Cd] or null
  float* dst;
  int NB, Hs, Ws, Cs;  // source tensor
  int Hg, Wg;          // destination grid per phase
  int Hd, Wd, Cd;      // destination tensor
  int mode, stride, pad;
  int xpad_off;        // mode 0: the column origin is b*stride - pad + xpad_off (0, or +pad for the 7x1 row convolution)
  int Ty, Tx;          // taps per phase
  int K, Kpad, Npad;
  int reflect;
  int act;
  float slope;
  int M;               // NB*Hg*Wg
  int m_tiles, n_tiles;
  int ksplit, kt_per_split;      // split-K (blockIdx.z): K tiles [z * kt_per_split, ...) -> partial sums into dst + z * split_stride
  long long split_stride;
  int src16, dst16;              // igemm16_kernel only: src / dst are bf16 tensors (the 16-bit activations of the bf16 mode)
#ifdef SRGAN_EXPERIMENTS
  int exp;                       // timing ablations of igemm16_kernel (wrong results): scratch/README.md
#endif
};

constexpr int BK = 32;
constexpr int LDK = BK + 4;

// defined with the forward / input-gradient dispatch below, needed by the workspace query
static bool rowconv_applicable(const srgan_conv_desc* d);
static size_t rowconv_packed_elems(const srgan_conv_desc* d);
static bool narrow_dgrad_desc(const srgan_conv_desc* d, srgan_conv_desc* f, long long* w_off);

// BF = bf16 MFMA compute (BASELINE configs [2]-[4]): operands are rounded to bf16 (RNE) when the tile is written to LDS and
// multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; HBM tensors, the gather and the epilogue stay fp32.
template <int BM, int BN, int WM, int WN, bool VEC, bool BF = false>
__global__ __launch_bounds__(WM * WN * 64) void igemm_kernel(IgemmParams p) {
  static_assert(!BF || VEC, "the bf16 variant rides on the vector gather");
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int NT = WM * WN * 64;        // threads per workgroup (4 or 8 waves)
  constexpr int RP = NT / 8;              // tile rows staged per pass of float4 loads (8 float4 = 32 floats per row)
  constexpr int RG = NT / 32;             // tile rows staged per pass of the generic element gather
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
  // double-buffered operand tiles: tile t+1 is written while tile t is multiplied (one barrier per K tile)
  __shared__ __attribute__((aligned(16))) float As2[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs2[2][BN * LDK];
  __shared__ int row_n[BM], row_y[BM], row_x[BM], row_o[BM];

  const int tid = threadIdx.x;
  const int mt = blockIdx.x % p.m_tiles, nt = blockIdx.x / p.m_tiles;
  const int phase = blockIdx.y;
  const int s = p.stride;
  int py = 0, px = 0;
  if (p.mode == 1) { py = phase / s; px = phase % s; }
  const int sgn = p.mode == 0 ? 1 : -1;
  const int m0 = mt * BM, n0 = nt * BN;

  for (int r = tid; r < BM; r += NT) {
    int m = m0 + r;
    int n = 0, y0 = -(1 << 28), x0 = -(1 << 28), o = -1;
    if (m < p.M) {
      n = m / (p.Hg * p.Wg);
      int rem = m - n * (p.Hg * p.Wg);
      int a = rem / p.Wg, b = rem - a * p.Wg;
      if (p.mode == 0) {
        y0 = a * s - p.pad; x0 = b * s - p.pad + p.xpad_off;
        o = (n * p.Hd + a) * p.Wd + b;
      } else {
        int oy = a * s + py, ox = b * s + px;
        if (oy < p.Hd && ox < p.Wd) {
          y0 = a + (py + p.pad) / s; x0 = b + (px + p.pad) / s;
          o = (n * p.Hd + oy) * p.Wd + ox;
        }
      }
    }
    row_n[r] = n; row_y[r] = y0; row_x[r] = x0; row_o[r] = o;
  }
  __syncthreads();

  const float* wp = p.wp + (size_t)phase * p.Npad * p.Kpad + (size_t)n0 * p.Kpad;
  // split-K: this workgroup multiplies K tiles [kt0, kt0 + nk) only and writes raw partial sums into its own slab
  const int kt0 = p.ksplit > 1 ? (int)blockIdx.z * p.kt_per_split : 0;
  const int nk = p.ksplit > 1 ? max(min(p.Kpad / BK - kt0, p.kt_per_split), 0) : p.Kpad / BK;

  // staging registers
  constexpr int A_VEC_IT = BM / RP;      // float4 per thread (VEC)
  constexpr int A_SC_IT = BM / RG;       // scalars per thread (generic)
  constexpr int B_IT = BN / RP;
  f32x4 a_reg[VEC ? A_VEC_IT : 1];
  float a_sc[VEC ? 1 : A_SC_IT];
  f32x4 b_reg[B_IT];

  auto src_index = [&](int r, int ty, int tx, bool& ok) -> size_t {
    int y = row_y[r] + sgn * ty, x = row_x[r] + sgn * tx;
    if (p.reflect) {
      if (y < 0) y = -y;
      if (y >= p.Hs) y = 2 * p.Hs - 2 - y;
      if (x < 0) x = -x;
      if (x >= p.Ws) x = 2 * p.Ws - 2 - x;
    }
    ok = (unsigned)y < (unsigned)p.Hs && (unsigned)x < (unsigned)p.Ws;
    return ((size_t)(row_n[r] * p.Hs + y) * p.Ws + x) * p.Cs;
  };

  // VEC path: the BM/32 rows a thread stages are fixed for the whole K loop.  Per row keep ONE 32-bit byte offset
  // and a tap-validity bitmask in registers; the tap / channel position is wave-uniform, so each gather is
  //   address = (scalar base of this tap & channel block) + (row offset or a safe offset)   [saddr + voffset form]
  // and costs a handful of VALU ops instead of a multiply-add chain per load.
  unsigned voff[VEC ? A_VEC_IT : 1];
  unsigned long long vmask[VEC ? A_VEC_IT : 1];
  unsigned mapy[VEC ? A_VEC_IT : 1], mapx[VEC ? A_VEC_IT : 1];   // reflect: mirrored tap displacement per nibble
  float a_msk[VEC ? A_VEC_IT : 1];
  const long long bias = ((long long)max(p.pad, p.Ty) * p.Ws + max(p.pad, p.Tx)) * p.Cs;   // keeps offsets >= 0
  if constexpr (VEC) {
    const unsigned seg16 = (tid & 7) * 16;
#pragma unroll
    for (int i = 0; i < A_VEC_IT; ++i) {
      const int r = (tid >> 3) + RP * i;
      const int y0 = row_y[r], x0 = row_x[r];
      const bool rowok = y0 > -(1 << 27);
      const long long lin = rowok ? ((long long)(row_n[r] * p.Hs + y0) * p.Ws + x0) * p.Cs : 0;
      voff[i] = (unsigned)((lin + bias) * 4) + seg16;
      unsigned long long m = 0;
      unsigned my = 0, mx = 0;
      if (rowok) {
        if (p.reflect) {
          m = ~0ull;
          for (int t = 0; t < p.Ty; ++t) {
            int y = y0 + t;
            y = y < 0 ? -y : y;
            y = y >= p.Hs ? 2 * p.Hs - 2 - y : y;
            my |= (unsigned)(y - y0) << (4 * t);
          }
          for (int t = 0; t < p.Tx; ++t) {
            int x = x0 + t;
            x = x < 0 ? -x : x;
            x = x >= p.Ws ? 2 * p.Ws - 2 - x : x;
            mx |= (unsigned)(x - x0) << (4 * t);
          }
        } else {
          int t = 0;
          for (int ty = 0; ty < p.Ty; ++ty) {
            const bool yok = (unsigned)(y0 + sgn * ty) < (unsigned)p.Hs;
            for (int tx = 0; tx < p.Tx; ++tx, ++t)
              if (yok && (unsigned)(x0 + sgn * tx) < (unsigned)p.Ws) m |= 1ull << t;
          }
        }
      }
      vmask[i] = m; mapy[i] = my; mapx[i] = mx;
    }
  }
  int tap_y = 0, tap_x = 0, tap_c = 0;       // position of the NEXT tile to load (VEC path)
  if (VEC && kt0 > 0) {
    const int k0 = kt0 * BK, t = k0 / p.Cs;
    tap_c = k0 - t * p.Cs;
    tap_y = t / p.Tx;
    tap_x = t - tap_y * p.Tx;
  }
  // generic path: per-row window origin and linear element offset of the thread's BM/RG rows
  int g_y[VEC ? 1 : A_SC_IT], g_x[VEC ? 1 : A_SC_IT], g_lin[VEC ? 1 : A_SC_IT];
  if constexpr (!VEC) {
#pragma unroll
    for (int i = 0; i < A_SC_IT; ++i) {
      const int r = (tid >> 5) + RG * i;
      g_y[i] = row_y[r];
      g_x[i] = row_x[r];
      const bool rowok = row_y[r] > -(1 << 27);
      g_lin[i] = rowok ? ((row_n[r] * p.Hs + row_y[r]) * p.Ws + row_x[r]) * p.Cs : 0;
    }
  }

  auto load_tiles = [&](int kt_rel) {
    const int kt = kt0 + kt_rel;
    const int k0 = kt * BK;
    if constexpr (VEC) {
      const unsigned seg16 = (tid & 7) * 16;
      const unsigned cs4 = p.Cs * 4;
      const int t = tap_y * p.Tx + tap_x;
      const long long tap_lin = p.reflect ? 0 : (long long)sgn * (tap_y * p.Ws + tap_x) * p.Cs;
      const char* sb = reinterpret_cast<const char*>(p.src) + (tap_lin + tap_c - bias) * 4;   // wave-uniform
      const unsigned safe = (unsigned)((bias - tap_lin) * 4) + seg16;                          // -> src + tap_c
#pragma unroll
      for (int i = 0; i < A_VEC_IT; ++i) {
        const bool ok = (vmask[i] >> t) & 1;
        // reflect: mirrored displacement from the nibble maps (all-zero maps otherwise, so no branch is needed)
        const unsigned dyv = (mapy[i] >> (4 * tap_y)) & 15u, dxv = (mapx[i] >> (4 * tap_x)) & 15u;
        unsigned off = voff[i] + (dyv * p.Ws + dxv) * cs4;
        off = ok ? off : safe;
        a_reg[i] = *reinterpret_cast<const f32x4*>(sb + off);
        a_msk[i] = ok ? 1.f : 0.f;          // applied when the tile is written to LDS (keeps the load in flight)
      }
      tap_c += BK;                                  // wave-uniform counters, branch-free wrap
      const int wc = tap_c == p.Cs;
      tap_c = wc ? 0 : tap_c;
      tap_x += wc;
      const int wx = tap_x == p.Tx;
      tap_x = wx ? 0 : tap_x;
      tap_y += wx;
    } else {
      const int kk = tid & 31;
      const int k = k0 + kk;
      const bool kok = k < p.K;
      const int t = kok ? k / p.Cs : 0;
      const int c = k - t * p.Cs;
      const int ty = t / p.Tx, tx = t - ty * p.Tx;
      if (p.reflect) {
#pragma unroll
        for (int i = 0; i < A_SC_IT; ++i) {
          const int r = (tid >> 5) + RG * i;
          bool ok;
          size_t off = src_index(r, ty, tx, ok);
          a_sc[i] = (ok && kok) ? p.src[off + c] : 0.f;
        }
      } else {
        // zero padding: the thread's rows are fixed -> row origin and linear offset live in registers; one tap /
        // channel offset per tile per thread, ~7 VALU ops per gathered element
        const int dy = sgn * ty, dx = sgn * tx;
        const int tapoff = (dy * p.Ws + dx) * p.Cs + c;
#pragma unroll
        for (int i = 0; i < A_SC_IT; ++i) {
          const bool ok = kok && (unsigned)(g_y[i] + dy) < (unsigned)p.Hs && (unsigned)(g_x[i] + dx) < (unsigned)p.Ws;
          const int off = ok ? g_lin[i] + tapoff : 0;
          const float v = p.src[off];
          a_sc[i] = ok ? v : 0.f;
        }
      }
    }
    const int seg = tid & 7;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int r = (tid >> 3) + RP * i;
      b_reg[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)r * p.Kpad + k0 + seg * 4);
    }
  };

  constexpr int LDH = BK + 8;              // bf16 row stride (80 B): 16-lane ds_read_b128 groups land on distinct banks
  auto store_tiles = [&](int buf) {
    float* As = As2[buf];
    float* Bs = Bs2[buf];
    if constexpr (BF) {
      unsigned short* Ah = reinterpret_cast<unsigned short*>(As);
      unsigned short* Bh = reinterpret_cast<unsigned short*>(Bs);
      const int seg = tid & 7;
#pragma unroll
      for (int i = 0; i < A_VEC_IT; ++i) {
        const int r = (tid >> 3) + RP * i;
        *reinterpret_cast<bf16x4*>(&Ah[r * LDH + seg * 4]) = __builtin_convertvector(a_reg[i] * a_msk[i], bf16x4);
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int r = (tid >> 3) + RP * i;
        *reinterpret_cast<bf16x4*>(&Bh[r * LDH + seg * 4]) = __builtin_convertvector(b_reg[i], bf16x4);
      }
      return;
    }
    if constexpr (VEC) {
      const int seg = tid & 7;
#pragma unroll
      for (int i = 0; i < A_VEC_IT; ++i) {
        const int r = (tid >> 3) + RP * i;
        *reinterpret_cast<f32x4*>(&As[r * LDK + seg * 4]) = a_reg[i] * a_msk[i];
      }
    } else {
      const int kk = tid & 31;
#pragma unroll
      for (int i = 0; i < A_SC_IT; ++i) {
        const int r = (tid >> 5) + RG * i;
        As[r * LDK + kk] = a_sc[i];
      }
    }
    const int seg = tid & 7;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int r = (tid >> 3) + RP * i;
      *reinterpret_cast<f32x4*>(&Bs[r * LDK + seg * 4]) = b_reg[i];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Software pipeline over K tiles (one barrier per tile):
  //   global loads of tile t+2 are issued right after the barrier of tile t and written to LDS during tile t+1,
  //   the q=0 fragments of tile t+1 are read right after that barrier, under the last 16 MFMAs of tile t,
  //   fragments of step q+1 are read before the MFMAs of step q.
  f32x4 af[2][TM], bf[2][TN];
  auto read_frags = [&](int buf, int q, int slot) {
    const float* As = As2[buf];
    const float* Bs = Bs2[buf];
#pragma unroll
    for (int i = 0; i < TM; ++i)
      af[slot][i] = *reinterpret_cast<const f32x4*>(&As[(wm * TM * 32 + i * 32 + lr) * LDK + q * 8 + lh * 4]);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      bf[slot][j] = *reinterpret_cast<const f32x4*>(&Bs[(wn * TN * 32 + j * 32 + lr) * LDK + q * 8 + lh * 4]);
  };
  auto mfma_step = [&](int slot) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][i][e], bf[slot][j][e], acc[i][j], 0, 0, 0);
  };

  if (nk > 0) {
  if constexpr (BF) {
    // 8 MFMAs (2 K-steps of 16) per wave and K tile: 256 matrix cycles against ~1000 of loads / LDS / barrier -- the loop
    // is bound by data movement, so it is kept simple: fragments of the whole tile first, then the next tile's stores
    bf16x8 ah[2][TM], bh[2][TN];
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    if (nk > 1) load_tiles(1);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      const unsigned short* Ah = reinterpret_cast<const unsigned short*>(As2[cur]);
      const unsigned short* Bh = reinterpret_cast<const unsigned short*>(Bs2[cur]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          ah[q][i] = *reinterpret_cast<const bf16x8*>(&Ah[(wm * TM * 32 + i * 32 + lr) * LDH + q * 16 + lh * 8]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bh[q][j] = *reinterpret_cast<const bf16x8*>(&Bh[(wn * TN * 32 + j * 32 + lr) * LDH + q * 16 + lh * 8]);
      }
      if (kt + 1 < nk) store_tiles(cur ^ 1);
      if (kt + 2 < nk) load_tiles(kt + 2);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[q][i], bh[q][j], acc[i][j], 0, 0, 0);
      __syncthreads();
    }
  } else {
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  read_frags(0, 0, 0);
  if (nk > 1) load_tiles(1);

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    read_frags(cur, 1, 1);
    mfma_step(0);
    read_frags(cur, 2, 0);
    mfma_step(1);
    if (more) store_tiles(cur ^ 1);          // tile kt+1 (its loads were issued a whole tile ago)
    read_frags(cur, 3, 1);
    mfma_step(0);
    __syncthreads();                         // tile kt+1 visible; everyone has read the last fragments of tile kt
    if (more) {
      read_frags(cur ^ 1, 0, 0);
      if (kt + 2 < nk) load_tiles(kt + 2);
    }
    mfma_step(1);
  }
  }
  }

  // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float* const dst_base = p.ksplit > 1 ? p.dst + (size_t)blockIdx.z * p.split_stride : p.dst;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * TN * 32 + j * 32 + lr;
    const bool nok = n < p.Cd;
    const float bv = (nok && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int o = row_o[r];
        if (nok && o >= 0) {
          float v = acc[i][j][e] + bv;
          v = apply_act(v, p.act, p.slope);
          dst_base[(size_t)o * p.Cd + n] = v;
        }
      }
    }
  }
}

// ---- bf16 mode: implicit GEMM with 64-deep K tiles (round 4) --------------------------------------------------------------
// The layers of the bf16 compute mode that no LDS-resident-patch kernel serves (the style encoder's 3x3 reflect-padded
// convolutions on 62 / 31 / 15 / 7-pixel maps, the discriminators' 4x4 stride-2 layers on 16- and 8-pixel maps, and the input
// gradients of both; pyfiles/model.py:100-121, 128-160) ran on igemm_kernel<.., BF>: a 32-deep K tile per barrier with 2-8 MFMAs
// per wave, ~100 vector instructions of address arithmetic, masking and conversion around them and fp32 weights (158-290
// TFLOP/s).  Here: 128 x BN output tiles, 4 waves of 64 x (BN / 2), K tiles of 64 channels of ONE tap (Cs % 64 == 0), 16 or 32
// MFMAs (32x32x16 bf16) per wave and barrier; the gathered rows are masked by a select on the packed bf16 pairs; fragments of
// K step q + 1 are read under the MFMAs of step q; tile kt + 1 goes to LDS in the middle of tile kt and the loads of tile
// kt + 2 are issued behind it.  Rows are 144 bytes apart in LDS (36 banks: the 16-lane groups of ds_read_b128 and the 8-lane
// row segments of ds_write_b64 land on distinct banks).  Source and destination stay fp32 in HBM and the source is rounded to
// bf16 (RNE) on the way into LDS; the weights are packed [phases][Npad][Kpad] as for igemm_kernel but already as bf16
// (PackParams::out16: the same RNE rounding, done once per optimiser step instead of once per tile), so the two kernels
// produce the same sums up to fp32 summation order.
constexpr int BK16 = 64;
constexpr int LDH16 = BK16 + 8;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// one 16-byte LDS-DMA piece per lane: lane L's bytes land at `lds` (wave-uniform) + 16 L; an out-of-range `voff` (>= bytes)
// writes zeros.  `base` / `bytes` are wave-uniform (made so here: a descriptor the compiler cannot prove uniform gets a
// waterfall loop).  A __device__ helper on raw pointers: with the builtin, or a buffer-resource VALUE, in the __global__
// template itself the HOST pass of hipcc (ROCm 7.2) drops the instantiation without a diagnostic (undefined kernel stub at load).
__device__ __forceinline__ void lds_dma16(const void* base, unsigned bytes, unsigned char* lds, unsigned voff) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
}

// DMA (round 6, VERDICT r5 item 1 ii; bf16 source tensors only): both operand tiles go global -> LDS by LDS-DMA
// (`buffer_load_dwordx4 ... lds`: lane L's 16 bytes land at the wave's LDS base + 16 L), no registers, no ds_write -- the store
// path (~80 B/clk per CU, shared by the two resident workgroups) was what bounded the loop.  The LDS image is then lane-linear:
// rows of 128 bytes WITHOUT padding, so the bank spread comes from an XOR swizzle applied on the SOURCE side -- slot s of row r
// holds the 16-byte piece s ^ ((r >> 1) & 7) (the 16 rows of a ds_read_b128 lane group hit 16 different 16-byte columns of the
// 256-byte bank row).  A masked row (zero padding, rows past M) is an out-of-range buffer offset: the DMA writes zeros
// (scratch/dma_test).  Tile kt + 1 is requested when tile kt - 1's buffer is released, one barrier per tile; the wait for a
// tile's own pieces is a counted vmcnt in front of that barrier (which then publishes every wave's pieces).
template <int BN, bool IN16 = false, bool OUT16 = false, bool DMA = false>
__global__ __launch_bounds__(256, 2) void igemm16_kernel(IgemmParams p) {
  constexpr int BM = 128;
  constexpr int ESZ = IN16 ? 2 : 4;         // bytes per source element
  constexpr int TN = BN / 64;               // 32-wide column blocks per wave (2 waves across N)
  constexpr int A_IT = BM / 32, B_IT = BN / 32;
  static_assert(BN == 128 || BN == 64, "BN");
  static_assert(!DMA || IN16, "LDS-DMA moves bf16 source rows as they are");
  constexpr int LDR = DMA ? BK16 : LDH16;   // bf16 elements per LDS row
  // ONE shared array (a second __shared__ object beside an LDS-DMA target makes hipcc drain vmcnt in front of every fragment read)
  constexpr int A_BYTES = BM * LDR * 2, B_BYTES = BN * LDR * 2, OFF_B = 2 * A_BYTES, OFF_ROW = OFF_B + 2 * B_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_ROW + BM * 4];
  int* const row_o = reinterpret_cast<int*>(smem + OFF_ROW);
  auto Ah_of = [&](int buf) __attribute__((always_inline)) { return reinterpret_cast<unsigned short*>(smem + buf * A_BYTES); };
  auto Bh_of = [&](int buf) __attribute__((always_inline)) { return reinterpret_cast<unsigned short*>(smem + OFF_B + buf * B_BYTES); };

  const int tid = threadIdx.x;
  const int phase = blockIdx.y;
  const int s = p.stride;
  int py = 0, px = 0;
  if (p.mode == 1) { py = phase / s; px = phase % s; }
  const int sgn = p.mode == 0 ? 1 : -1;
  // Persistent over the output tiles (launch_igemm16: at most two workgroups per CU): a workgroup's result stores drain under
  // its next tile's prologue and K loop instead of holding its CU slot until they have landed -- with one tile per workgroup
  // the rounds run in lockstep and the epilogue of a 9-tile K loop (the encoder's 62 x 62 maps) was 40 % of the kernel.
  const int total_tiles = p.m_tiles * p.n_tiles;
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
  int bid = tile;
  if ((total_tiles & 7) == 0) bid = (bid & 7) * (total_tiles >> 3) + (bid >> 3);   // neighbouring pixel tiles (shared image rows) on one XCD
  const int mt = bid % p.m_tiles, nt = bid / p.m_tiles;
  const int m0 = mt * BM, n0 = nt * BN;

  // window origin and destination pixel of GEMM row m.  The two divisions are float multiplications by a reciprocal with one
  // correction step (exact for m < 2^23, igemm16_ok): the prologue runs once per 128 x BN tile and K loops here are as short as
  // 9 tiles, so the ~60-instruction integer divisions and per-tap loops of igemm_kernel's prologue cost as much as the loop
  const int hw_g = p.Hg * p.Wg;
  const float inv_hw = 1.f / (float)hw_g, inv_w = 1.f / (float)p.Wg;
  const int qy = p.mode == 1 ? (py + p.pad) / s : 0, qx = p.mode == 1 ? (px + p.pad) / s : 0;
  auto fdiv = [](int m, int d, float inv, int& q, int& r) __attribute__((always_inline)) {
    q = (int)((float)m * inv);
    r = m - q * d;
    if (r >= d) { ++q; r -= d; }
    if (r < 0) { --q; r += d; }
  };
  auto row_geom = [&](int m, int& n, int& y0, int& x0, int& o) __attribute__((always_inline)) {
    n = 0; y0 = -(1 << 28); x0 = -(1 << 28); o = -1;
    if (m < p.M) {
      int rem, a, b;
      fdiv(m, hw_g, inv_hw, n, rem);
      fdiv(rem, p.Wg, inv_w, a, b);
      if (p.mode == 0) {
        y0 = a * s - p.pad; x0 = b * s - p.pad + p.xpad_off;
        o = (n * p.Hd + a) * p.Wd + b;
      } else {
        const int oy = a * s + py, ox = b * s + px;
        if (oy < p.Hd && ox < p.Wd) {
          y0 = a + qy; x0 = b + qx;
          o = (n * p.Hd + oy) * p.Wd + ox;
        }
      }
    }
  };
  if (tid < BM) {
    int n, y0, x0, o;
    row_geom(m0 + tid, n, y0, x0, o);
    row_o[tid] = o;
  }

  const unsigned short* wp = reinterpret_cast<const unsigned short*>(p.wp) + (size_t)phase * p.Npad * p.Kpad;
  const int kt0 = p.ksplit > 1 ? (int)blockIdx.z * p.kt_per_split : 0;
  int nk = p.ksplit > 1 ? max(min(p.Kpad / BK16 - kt0, p.kt_per_split), 0) : p.Kpad / BK16;
#ifdef SRGAN_EXPERIMENTS
  if (p.exp & 16) nk = 0;
#endif

  // a thread stages 4 pixel rows (8 lanes per row: floats [4 seg, 4 seg + 4) and [32 + 4 seg, ...) of the 64-channel slice);
  // per row ONE 32-bit byte offset, the valid tap rows / tap columns as two 8-bit ranges (zero padding is separable) and, for
  // reflect padding, the mirrored tap displacements as nibbles
  const unsigned seg16 = (tid & 7) * 16;
  unsigned voff[A_IT], vmask[A_IT];          // vmask: bits 0-7 valid tap rows, bits 8-15 valid tap columns
  unsigned mapy[A_IT], mapx[A_IT];
  const long long bias = ((long long)max(p.pad, p.Ty) * p.Ws + max(p.pad, p.Tx)) * p.Cs;   // keeps offsets >= 0
  auto bits = [](int lo, int hi) -> unsigned { return hi >= lo ? ((2u << (hi & 31)) - 1u) & ~((1u << (lo & 31)) - 1u) : 0u; };
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int n, y0, x0, o;
    row_geom(m0 + (tid >> 3) + 32 * i, n, y0, x0, o);
    const bool rowok = y0 > -(1 << 27);
    const long long lin = rowok ? ((long long)(n * p.Hs + y0) * p.Ws + x0) * p.Cs : 0;
    voff[i] = (unsigned)((lin + bias) * ESZ) + seg16;
    unsigned m = 0, my = 0, mx = 0;
    if (p.reflect) {
      if (rowok) {
        m = 0xffffu;
        for (int t = 0; t < p.Ty; ++t) {
          int y = y0 + t;
          y = y < 0 ? -y : y;
          y = y >= p.Hs ? 2 * p.Hs - 2 - y : y;
          my |= (unsigned)(y - y0) << (4 * t);
        }
        for (int t = 0; t < p.Tx; ++t) {
          int x = x0 + t;
          x = x < 0 ? -x : x;
          x = x >= p.Ws ? 2 * p.Ws - 2 - x : x;
          mx |= (unsigned)(x - x0) << (4 * t);
        }
      }
    } else if (rowok) {
      // tap ty reads image row y0 + sgn ty: valid for ty in [lo, hi]
      const unsigned ym = sgn > 0 ? bits(max(0, -y0), min(p.Ty - 1, p.Hs - 1 - y0)) : bits(max(0, y0 - p.Hs + 1), min(p.Ty - 1, y0));
      const unsigned xm = sgn > 0 ? bits(max(0, -x0), min(p.Tx - 1, p.Ws - 1 - x0)) : bits(max(0, x0 - p.Ws + 1), min(p.Tx - 1, x0));
      m = ym | (xm << 8);
    }
    vmask[i] = m; mapy[i] = my; mapx[i] = mx;
  }
  int tap_y = 0, tap_x = 0, tap_c = 0;       // position of the NEXT tile to load (wave-uniform)
  if (kt0 > 0) {
    const int k0 = kt0 * BK16, t = k0 / p.Cs;
    tap_c = k0 - t * p.Cs;
    tap_y = t / p.Tx;
    tap_x = t - tap_y * p.Tx;
  }
  // weight rows of this thread: n0 + (tid >> 3) + 32 i, clamped to the packed rows (columns >= Cd are never stored)
  unsigned woff[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) woff[i] = (unsigned)(((size_t)min(n0 + (tid >> 3) + 32 * i, p.Npad - 1) * p.Kpad) * 2) + seg16;

  f32x4 a_reg[A_IT][2], b_reg[B_IT];          // b_reg: 8 bf16 weights (row, k = 8 seg .. 8 seg + 7)
  bool a_ok[A_IT];
  auto load_tiles = [&](int kt_rel) __attribute__((always_inline)) {
    const int k0 = (kt0 + kt_rel) * BK16;
    const unsigned cs4 = p.Cs * ESZ;
    const long long tap_lin = p.reflect ? 0 : (long long)sgn * (tap_y * p.Ws + tap_x) * p.Cs;
    const char* sb = reinterpret_cast<const char*>(p.src) + (tap_lin + tap_c - bias) * ESZ;   // wave-uniform
    const unsigned safe = (unsigned)((bias - tap_lin) * ESZ) + seg16;                          // -> src + tap_c
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const bool ok = ((vmask[i] >> tap_y) & (vmask[i] >> (8 + tap_x)) & 1u) != 0;
      // reflect: mirrored displacement from the nibble maps (all-zero maps otherwise)
      const unsigned dyv = (mapy[i] >> (4 * tap_y)) & 15u, dxv = (mapx[i] >> (4 * tap_x)) & 15u;
      unsigned off = voff[i] + (dyv * p.Ws + dxv) * cs4;
      off = ok ? off : safe;
#ifdef SRGAN_EXPERIMENTS
      if (p.exp & 1) { a_reg[i][0] = f32x4{1.f, 1.f, 1.f, 1.f}; a_reg[i][1] = a_reg[i][0]; a_ok[i] = ok; continue; }
#endif
      // IN16: the 64-channel slice of a pixel is 128 bytes -- one 16-byte piece per lane, already in LDS order
      a_reg[i][0] = *reinterpret_cast<const f32x4*>(sb + off);
      if constexpr (!IN16) a_reg[i][1] = *reinterpret_cast<const f32x4*>(sb + off + 128);
      a_ok[i] = ok;
    }
    tap_c += BK16;
    const int wc = tap_c == p.Cs;
    tap_c = wc ? 0 : tap_c;
    tap_x += wc;
    const int wx = tap_x == p.Tx;
    tap_x = wx ? 0 : tap_x;
    tap_y += wx;
    const char* wb = reinterpret_cast<const char*>(wp) + (size_t)k0 * 2;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
#ifdef SRGAN_EXPERIMENTS
      if (p.exp & 2) { b_reg[i] = f32x4{1.f, 1.f, 1.f, 1.f}; continue; }
#endif
      b_reg[i] = *reinterpret_cast<const f32x4*>(wb + woff[i]);
    }
  };
  auto store_tiles = [&](int buf) __attribute__((always_inline)) {
    unsigned short* Ah = Ah_of(buf);
    unsigned short* Bh = Bh_of(buf);
    const int seg = tid & 7;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int r = (tid >> 3) + 32 * i;
      if constexpr (IN16) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(&Ah[r * LDH16 + seg * 8]) = a_ok[i] ? a_reg[i][0] : z;
        continue;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x2 v = __builtin_bit_cast(u32x2, __builtin_convertvector(a_reg[i][h], bf16x4));
        v[0] = a_ok[i] ? v[0] : 0u;
        v[1] = a_ok[i] ? v[1] : 0u;
        *reinterpret_cast<u32x2*>(&Ah[r * LDH16 + h * 32 + seg * 4]) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) *reinterpret_cast<f32x4*>(&Bh[((tid >> 3) + 32 * i) * LDH16 + seg * 8]) = b_reg[i];
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  bf16x8 ah[2][2], bh[2][TN];
  auto read_frags = [&](int buf, int q, int slot) __attribute__((always_inline)) {
    const unsigned short* Ah = Ah_of(buf);
    const unsigned short* Bh = Bh_of(buf);
    // DMA image: piece 2 q + lh of a row sits in slot (2 q + lh) ^ ((row >> 1) & 7); the row bases are multiples of 16
    const int col = DMA ? (((2 * q + lh) ^ ((lr >> 1) & 7)) * 8) : (q * 16 + lh * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i) ah[slot][i] = *reinterpret_cast<const bf16x8*>(&Ah[(wm * 64 + i * 32 + lr) * LDR + col]);
#pragma unroll
    for (int j = 0; j < TN; ++j) bh[slot][j] = *reinterpret_cast<const bf16x8*>(&Bh[(wn * (BN / 2) + j * 32 + lr) * LDR + col]);
  };
  auto mma = [&](int slot) __attribute__((always_inline)) {
#ifdef SRGAN_EXPERIMENTS
    if (p.exp & 4) return;
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[slot][i], bh[slot][j], acc[i][j], 0, 0, 0);
  };

  if constexpr (DMA) {
    // ---- LDS-DMA tiles: per thread 4 source rows (A) and B_IT weight rows, one 16-byte piece each, slot = tid & 7 ----
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned key = (tid >> 4) & 7, piece16 = ((tid & 7) ^ key) * 16;       // the piece this lane's slot holds
    const unsigned a_bytes = (unsigned)((size_t)p.NB * p.Hs * p.Ws * p.Cs * 2), b_bytes = (unsigned)((size_t)p.Npad * p.Kpad * 2);
    constexpr unsigned kOutside = 0x80000000u;
    int abase[A_IT];                      // byte offset of the row's window origin (tap (0, 0), channel 0) + this lane's piece
#pragma unroll
    for (int i = 0; i < A_IT; ++i) abase[i] = (int)voff[i] - (int)(bias * ESZ) - (int)seg16 + (int)piece16;
    unsigned wbase[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) wbase[i] = woff[i] - seg16 + piece16;
    auto dma_tile = [&](int buf, int kt_rel) __attribute__((always_inline)) {
      const int k0 = (kt0 + kt_rel) * BK16;
      const int tap_off = ((p.reflect ? 0 : sgn * (tap_y * p.Ws + tap_x)) * p.Cs + tap_c) * 2;    // wave-uniform
      unsigned char* const lds_a = smem + buf * A_BYTES + wave_u * 1024;
      unsigned char* const lds_b = smem + OFF_B + buf * B_BYTES + wave_u * 1024;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const bool ok = ((vmask[i] >> tap_y) & (vmask[i] >> (8 + tap_x)) & 1u) != 0;
        const unsigned dyv = (mapy[i] >> (4 * tap_y)) & 15u, dxv = (mapx[i] >> (4 * tap_x)) & 15u;
        const unsigned off = (unsigned)(abase[i] + tap_off + (int)((dyv * p.Ws + dxv) * p.Cs * 2));
        lds_dma16(p.src, a_bytes, lds_a + i * 4096, ok ? off : kOutside);
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i)
        lds_dma16(wp, b_bytes, lds_b + i * 4096, wbase[i] + (unsigned)(k0 * 2));
      tap_c += BK16;
      const int wc = tap_c == p.Cs;
      tap_c = wc ? 0 : tap_c;
      tap_x += wc;
      const int wx = tap_x == p.Tx;
      tap_x = wx ? 0 : tap_x;
      tap_y += wx;
    };
    if (nk > 0) {
      dma_tile(0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                   // tile 0 and row_o are visible
      if (nk > 1) dma_tile(1, 1);
      for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        read_frags(cur, 0, 0);
        read_frags(cur, 1, 1);
        mma(0);
        read_frags(cur, 2, 0);
        mma(1);
        read_frags(cur, 3, 1);
        mma(0);
        mma(1);
        if (kt + 1 < nk) {
          // this wave's pieces of tile kt + 1 have landed; behind the barrier so have everyone's, and nobody reads buffer
          // `cur` any more: tile kt + 2 may overwrite it
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (kt + 2 < nk) dma_tile(cur, kt + 2);
        }
      }
    }
    __syncthreads();
  } else {
  if (nk > 0) {
    load_tiles(0);
    store_tiles(0);
    if (nk > 1) load_tiles(1);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      read_frags(cur, 0, 0);
      read_frags(cur, 1, 1);
      mma(0);
      read_frags(cur, 2, 0);
      mma(1);
      if (kt + 1 < nk) store_tiles(cur ^ 1);       // tile kt + 1: requested a whole tile ago
      read_frags(cur, 3, 1);
      if (kt + 2 < nk) load_tiles(kt + 2);
      mma(0);
      mma(1);
      __syncthreads();
    }
  } else {
    __syncthreads();
  }
  }

  // epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* const dst_base = p.ksplit > 1 ? p.dst + (size_t)blockIdx.z * p.split_stride : p.dst;
#ifdef SRGAN_EXPERIMENTS
  if ((p.exp & 8) && acc[0][0][0] != 12345.f) { __syncthreads(); continue; }
#endif
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const bool relu = p.act == SRGAN_ACT_RELU;
  const float neg = p.act == SRGAN_ACT_LRELU ? p.slope : 1.f;      // branch-free apply_act (same values)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    i32x4 ro[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ro[g] = *reinterpret_cast<const i32x4*>(&row_o[wm * 64 + i * 32 + 8 * g + 4 * lh]);
    if constexpr (OUT16 && TN == 2) {
      // A bf16 row of this wave's 64 columns is ONE 128-byte line.  With a column per lane a half-wave store covers 64 bytes of
      // it (and the other column block another 64 bytes, 16 instructions later): every store is a partial-line write, and the
      // layer's result took as long to write as the fp32 one (55 us for 63 MB: scratch/io16/bench_io.py).  Lanes are paired
      // instead: the even lane of a pair takes both lanes' values of column block 0 (columns lr, lr + 1), the odd lane those of
      // block 1 (columns 32 + lr - 1, 32 + lr) -- one 4-byte store per lane and row, the 32 lanes of a half-wave write the whole
      // line, half the store instructions.  (a split-K launch writes fp32 slabs: launch_igemm16 takes OUT16 only with ksplit == 1)
      const int odd = lr & 1;
      const int n = n0 + wn * 64 + (odd ? 31 + lr : lr);        // first of this lane's two adjacent columns
      const bool nok = n < p.Cd;                                  // Cd % 4 == 0 (igemm16_io_ok): both columns in or out
      const float b0 = (nok && p.bias) ? p.bias[n] : 0.f, b1 = (nok && p.bias) ? p.bias[n + 1] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int o = ro[e >> 2][e & 3];
        const float own = odd ? acc[i][1][e] : acc[i][0][e], give = odd ? acc[i][0][e] : acc[i][1][e];
        const float got = __shfl_xor(give, 1, 64);
        const float v0 = (odd ? got : own) + b0, v1 = (odd ? own : got) + b1;
        const float r0 = v0 > 0.f ? v0 : (relu ? 0.f : v0 * neg), r1 = v1 > 0.f ? v1 : (relu ? 0.f : v1 * neg);
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 pk = {(__bf16)r0, (__bf16)r1};
        if (nok && o >= 0) *reinterpret_cast<bf16x2*>(reinterpret_cast<__bf16*>(dst_base) + (size_t)o * p.Cd + n) = pk;
      }
    } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 32 + lr;
      const bool nok = n < p.Cd;
      const float bv = (nok && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int o = ro[e >> 2][e & 3];
        const float v = acc[i][j][e] + bv;
        const float r = v > 0.f ? v : (relu ? 0.f : v * neg);
        if constexpr (OUT16) {      // (a split-K launch writes fp32 slabs: launch_igemm16 takes OUT16 only with ksplit == 1)
          if (nok && o >= 0) reinterpret_cast<__bf16*>(dst_base)[(size_t)o * p.Cd + n] = (__bf16)r;
        } else {
          if (nok && o >= 0) dst_base[(size_t)o * p.Cd + n] = r;
        }
      }
    }
    }
  }
  __syncthreads();        // row_o and the operand buffers are rewritten by the next tile
  }
}

// ---- weight repack ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weights_kernel(PackParams p) {
  if (pack_weights_block_ok(p)) {
    __shared__ float tile[PACK_LDS_FLOATS];
    pack_weights_rows(p, tile);
    return;
  }
  const long long total = pack_weights_total(p);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x)
    pack_weights_item(p, idx);
}

// every cached operand of a network in one launch: blockIdx.y = entry, blockIdx.x grid-strides over its elements
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackEntry* __restrict__ entries) {
  const PackEntry& e = entries[blockIdx.y];
  __shared__ float tile[PACK_LDS_FLOATS];
  static_assert(PACK_LDS_FLOATS >= 32 * WP43_ROW, "one LDS tile serves both cooperative paths");
  if (e.type == 0) {
    if (pack_weights_block_ok(e.ig)) { pack_weights_rows(e.ig, tile); return; }
    const long long total = pack_weights_total(e.ig);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x)
      pack_weights_item(e.ig, idx);
  } else if (wino43_pack_block_ok(e.wn)) {
    const long long total = wino_pack_total(e.wn);       // a multiple of 256 (nchunk % 4 == 0)
    for (long long base = (long long)blockIdx.x * 256; base < total; base += (long long)gridDim.x * 256)
      wino43_pack_block(e.wn, base, tile);
  } else {
    const long long total = wino_pack_total(e.wn);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x)
      wino_pack_item(e.wn, idx);
  }
}

// ---- reflect-pad fold (dgrad of a reflect-padded conv) --------------------------------------
// dxp: [N][H+2P][W+2P][C] gradient w.r.t. the padded image; dx[n][y][x][c] sums every padded
// position that mirrors onto (y,x).  P = pad (1 in the reference).
template <bool OUT16 = false>
__global__ void reflect_fold_kernel(const float* dxp, float* dx, int N, int H, int W, int C, int P) {
  const long long total = (long long)N * H * W * C;
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long long r = idx / C;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    // candidate padded rows: y+P (direct), P-y (top mirror, needs 1<=y<=P), 2H-2-y+P (bottom)
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y + P;
    if (y >= 1 && y <= P) ys[ny++] = P - y;
    if (y <= H - 2 && y >= H - 1 - P) ys[ny++] = 2 * H - 2 - y + P;
    xs[nx++] = x + P;
    if (x >= 1 && x <= P) xs[nx++] = P - x;
    if (x <= W - 2 && x >= W - 1 - P) xs[nx++] = 2 * W - 2 - x + P;
    float v = 0.f;
    for (int i = 0; i < ny; ++i)
      for (int j = 0; j < nx; ++j) v += dxp[((size_t)(n * Hp + ys[i]) * Wp + xs[j]) * C + c];
    if constexpr (OUT16) reinterpret_cast<__bf16*>(dx)[idx] = (__bf16)v;
    else dx[idx] = v;
  }
}

// ---- weight gradient ----------------------------------------------------------------------
//   dW[co][nn] = sum_m dy[m][co] * X[m][nn],  nn = (ky*kw+kx)*Cs + ci, X gathered from x.
// Block tile BMc (co) x BNn (nn) over a split of the m range; partial tiles go to
// slab[split][co_pad][NNpad] and are summed by wgrad_reduce_kernel.
struct WgradParams {
  const float* x;
  const float* dy;
  float* slab;
  int NB, Hi, Wi, Cs, Hg, Wg, Cd, stride, pad, kw, reflect;
  int NN, NNpad, Cdpad, M, rows_per_split, co_tiles, nn_tiles;
};

// ROWS: every 32-pixel K tile lies inside one image row (Wg % 32 == 0, VEC only): the tile's (n, y) and the source row
// base are wave-uniform scalars advanced with counters; a thread only adds its constant column offset.
// BF (bf16 compute mode): the 32-pixel tiles are written to LDS as bf16 [pixel][channel] rows (channel-contiguous, as they
// come from NHWC memory) and the MFMA operands -- 8 consecutive PIXELS of one channel per lane -- are fetched with the
// transposing read ds_read_b64_tr_b16 (a 16-lane group reads 4 pixel rows x 16 channels and receives them column-major).
// Row stride = channels * 2 + 64 bytes: the four rows of a block start 16 banks apart, the two blocks of a half 8 banks.
// IO16 (with BF, without ROWS): x and dy are bf16 TENSORS (the 16-bit activations of the bf16 mode outside the patch kernels:
// the style encoder's blocks); a staged piece is 4 channels = 8 bytes and goes to LDS as loaded.
template <int BMc, int BNn, int WM, int WN, bool VEC, bool ROWS = false, bool BF = false, bool IO16 = false>
__global__ __launch_bounds__((WM * WN > 4 ? WM * WN : 4) * 64) void wgrad_kernel(WgradParams p) {
  static_assert(!BF || VEC, "the bf16 variant rides on the vector gather");
  static_assert(!IO16 || (BF && !ROWS), "16-bit tensors: bf16 compute, generic row decode");
  constexpr int TM = BMc / (WM * 32), TN = BNn / (WN * 32);
  constexpr int NT = (WM * WN > 4 ? WM * WN : 4) * 64;   // threads: 4 waves (some idle for small tiles) or 8 waves
  constexpr int A_IT = 8 * BMc / NT, B_IT = 8 * BNn / NT;     // float4 per thread per 32-row tile
  constexpr int A_PR = BMc / 4, B_PR = BNn / 4;       // float4 per tile row
  // double-buffered tiles: tile t+1 is written to LDS in the middle of tile t's MFMAs (one barrier per tile)
  __shared__ __attribute__((aligned(16))) float As2[2][32 * BMc];
  __shared__ __attribute__((aligned(16))) float Bs2[2][32 * BNn];
  __shared__ int rowtab[2][3][32];

  const int tid = threadIdx.x;
  // (an XCD-contiguous work order was measured here and was slower for the 256-channel layers: 530 vs 465 us)
  const int tiles = p.co_tiles * p.nn_tiles;
  const int split = blockIdx.x / tiles, tile = blockIdx.x - split * tiles;
  const int ct = tile % p.co_tiles, nt = tile / p.co_tiles;
  const int co0 = ct * BMc, nn0 = nt * BNn;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(p.M, m_begin + p.rows_per_split);
  const int ntiles = (m_end - m_begin + 31) / 32;

  const int wave = tid >> 6, lane = tid & 63;
  const bool active = wave < WM * WN;   // small tiles use fewer than 4 MFMA waves
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const bool cd_vec = (p.Cd & 3) == 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 a_reg[A_IT], b_reg[B_IT];
  u32x2 a16[A_IT], b16[B_IT];          // IO16: the staged pieces are 4 bf16 values
  float a_msk[A_IT], b_msk[B_IT];      // 0 / 1 per staged float4, applied when the tile is written to LDS (keeps the loads in flight)
#pragma unroll
  for (int i = 0; i < A_IT; ++i) a_msk[i] = 1.f;
#pragma unroll
  for (int i = 0; i < B_IT; ++i) b_msk[i] = 1.f;

  auto fill_rowtab = [&](int t) {     // decode the 32 pixels of tile t (tid < 32 only)
    const int m = m_begin + t * 32 + tid;
    int n = 0, y0 = -(1 << 28), x0 = -(1 << 28);
    if (m < m_end) {
      n = m / (p.Hg * p.Wg);
      int rem = m - n * (p.Hg * p.Wg);
      int a = rem / p.Wg, b = rem - a * p.Wg;
      y0 = a * p.stride - p.pad; x0 = b * p.stride - p.pad;
    }
    rowtab[t & 1][0][tid] = n; rowtab[t & 1][1][tid] = y0; rowtab[t & 1][2][tid] = x0;
  };

  auto gather = [&](int buf, int r, int ty, int tx, bool& ok) -> size_t {
    int y = rowtab[buf][1][r] + ty, x = rowtab[buf][2][r] + tx;
    if (p.reflect) {
      if (y < 0) y = -y;
      if (y >= p.Hi) y = 2 * p.Hi - 2 - y;
      if (x < 0) x = -x;
      if (x >= p.Wi) x = 2 * p.Wi - 2 - x;
    }
    ok = (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
    return ((size_t)(rowtab[buf][0][r] * p.Hi + y) * p.Wi + x) * p.Cs;
  };

  // ---- ROWS fast path state (wave-uniform) ----
  int rw_n = 0, rw_a = 0, rw_b = 0;            // image, grid row, first grid column of the NEXT tile to load
  int tap_ty = 0, tap_tx = 0, tap_c0 = 0;
  if constexpr (ROWS) {
    const int hw = p.Hg * p.Wg;
    rw_n = m_begin / hw;
    const int rem = m_begin - rw_n * hw;
    rw_a = rem / p.Wg;
    rw_b = rem - rw_a * p.Wg;
    const int tp = nn0 / p.Cs;
    tap_c0 = nn0 - tp * p.Cs;
    tap_ty = tp / p.kw;
    tap_tx = tp - tap_ty * p.kw;
  }

  // generic-path column decode (K-invariant): columns nn0 + (tid % B_PR)*4 + e
  int gq_ty[4] = {0, 0, 0, 0}, gq_tx[4] = {0, 0, 0, 0}, gq_off[4] = {0, 0, 0, 0};
  bool gq_ok[4] = {false, false, false, false};
  if constexpr (!VEC) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int nn = nn0 + (tid % B_PR) * 4 + e;
      gq_ok[e] = nn < p.NN;
      const int tp = gq_ok[e] ? nn / p.Cs : 0, c = gq_ok[e] ? nn - tp * p.Cs : 0;
      gq_ty[e] = tp / p.kw;
      gq_tx[e] = tp - gq_ty[e] * p.kw;
      gq_off[e] = (gq_ty[e] * p.Wi + gq_tx[e]) * p.Cs + c;
    }
  }

  auto load_tile_rows = [&](int t) {
    const int mb = m_begin + t * 32;
    // A': 32 dense rows of dy starting at row mb (all valid: m_end - m_begin is a multiple of 32 on this path)
    const float* abase = p.dy + (size_t)mb * p.Cd + co0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int idx = tid + NT * i;
      const int r = idx / A_PR, c4 = idx - r * A_PR;
      const bool ok = co0 + c4 * 4 < p.Cd;      // unconditional load, masked at the LDS store (see load_tile)
      a_reg[i] = *reinterpret_cast<const f32x4*>(ok ? abase + (size_t)r * p.Cd + c4 * 4 : p.dy);
      a_msk[i] = ok ? 1.f : 0.f;
    }
    // B': source row y is uniform for the tile; x = (b0 + r)*stride - pad + tx varies with the thread's row
    int y = rw_a * p.stride - p.pad + tap_ty;
    if (p.reflect) {
      y = y < 0 ? -y : y;
      y = y >= p.Hi ? 2 * p.Hi - 2 - y : y;
    }
    const bool yok = (unsigned)y < (unsigned)p.Hi;
    const float* bbase = p.x + ((size_t)(rw_n * p.Hi + (yok ? y : 0)) * p.Wi) * p.Cs + tap_c0;
    const int x0 = rw_b * p.stride - p.pad + tap_tx;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int idx = tid + NT * i;
      const int r = idx / B_PR, c4 = idx - r * B_PR;
      int x = x0 + r * p.stride;
      if (p.reflect) {
        x = x < 0 ? -x : x;
        x = x >= p.Wi ? 2 * p.Wi - 2 - x : x;
      }
      const bool ok = yok && (unsigned)x < (unsigned)p.Wi;
      b_reg[i] = *reinterpret_cast<const f32x4*>(bbase + (size_t)(ok ? x : 0) * p.Cs + c4 * 4);
      b_msk[i] = ok ? 1.f : 0.f;        // applied when the tile is written to LDS (keeps the load in flight)
    }
    rw_b += 32;                                  // next tile: same image row, or wrap (Wg % 32 == 0)
    const int wb = rw_b == p.Wg;
    rw_b = wb ? 0 : rw_b;
    rw_a += wb;
    const int wa = rw_a == p.Hg;
    rw_a = wa ? 0 : rw_a;
    rw_n += wa;
  };

  auto load_tile = [&](int t) {
    if constexpr (ROWS) { load_tile_rows(t); return; }
    const int mb = m_begin + t * 32, buf = t & 1;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int idx = tid + NT * i;
      const int r = idx / A_PR, c4 = idx - r * A_PR;
      const int m = mb + r, co = co0 + c4 * 4;
      if (cd_vec) {
        // UNCONDITIONAL load from a clamped (always valid) address, masked at the LDS store: a load under a branch whose result is
        // merged with zeros makes the compiler wait for every load in turn (round 4: ~6 us per tile against 2 us of MFMAs)
        const bool ok = m < m_end && co < p.Cd;
        if constexpr (IO16) {
          const u32x2 t = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.dy) + (ok ? (size_t)m * p.Cd + co : (size_t)0));
          a16[i] = t;
        } else {
          a_reg[i] = *reinterpret_cast<const f32x4*>(p.dy + (ok ? (size_t)m * p.Cd + co : (size_t)0));
        }
        a_msk[i] = ok ? 1.f : 0.f;
      } else {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < m_end) {
          const float* src = p.dy + (size_t)m * p.Cd + co;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < p.Cd) v[e] = src[e];
        }
        a_reg[i] = v;
      }
    }
    if constexpr (VEC) {
      const int tp = nn0 / p.Cs, c0 = nn0 - tp * p.Cs;
      const int ty = tp / p.kw, tx = tp - ty * p.kw;
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = tid + NT * i;
        const int r = idx / B_PR, c4 = idx - r * B_PR;
        bool ok;
        const size_t off = gather(buf, r, ty, tx, ok);
        if constexpr (IO16) {
          const u32x2 t = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.x) + (ok ? off + c0 + c4 * 4 : (size_t)0));
          b16[i] = t;
        } else {
          b_reg[i] = *reinterpret_cast<const f32x4*>(p.x + (ok ? off + c0 + c4 * 4 : (size_t)0));      // (see the A operand above)
        }
        b_msk[i] = ok ? 1.f : 0.f;
      }
    } else {
      // generic channel counts: the thread's 4 columns nn = (tap, channel) are the same for every tile -> their
      // tap displacement / element offset were decoded once (gq_*); per tile only the row origin changes
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = tid + NT * i;
        const int r = idx / B_PR;
        const int y0 = rowtab[buf][1][r], x0 = rowtab[buf][2][r];
        const int lin = ((rowtab[buf][0][r] * p.Hi + y0) * p.Wi + x0) * p.Cs;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int y = y0 + gq_ty[e], x = x0 + gq_tx[e];
          int off = lin + gq_off[e];
          if (p.reflect) {
            const int yr = y < 0 ? -y : (y >= p.Hi ? 2 * p.Hi - 2 - y : y);
            const int xr = x < 0 ? -x : (x >= p.Wi ? 2 * p.Wi - 2 - x : x);
            off += ((yr - y) * p.Wi + (xr - x)) * p.Cs;
            y = yr; x = xr;
          }
          const bool ok = gq_ok[e] && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
          const float t = p.x[ok ? off : 0];
          v[e] = ok ? t : 0.f;
        }
        b_reg[i] = v;
      }
    }
  };

  constexpr int LDA = BMc + 32, LDB = BNn + 32;      // bf16 row strides (elements)
  auto store_tile = [&](int buf) {
    float* As = As2[buf];
    float* Bs = Bs2[buf];
    if constexpr (BF) {
      unsigned short* Ah = reinterpret_cast<unsigned short*>(As);
      unsigned short* Bh = reinterpret_cast<unsigned short*>(Bs);
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int idx = tid + NT * i;
        const int r = idx / A_PR, c4 = idx - r * A_PR;
        if constexpr (IO16) {
          const u32x2 z = {0u, 0u};
          *reinterpret_cast<u32x2*>(&Ah[r * LDA + c4 * 4]) = a_msk[i] != 0.f ? a16[i] : z;
        } else {
          *reinterpret_cast<bf16x4*>(&Ah[r * LDA + c4 * 4]) = __builtin_convertvector(a_reg[i] * a_msk[i], bf16x4);
        }
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = tid + NT * i;
        const int r = idx / B_PR, c4 = idx - r * B_PR;
        if constexpr (IO16) {
          const u32x2 z = {0u, 0u};
          *reinterpret_cast<u32x2*>(&Bh[r * LDB + c4 * 4]) = b_msk[i] != 0.f ? b16[i] : z;
        } else {
          *reinterpret_cast<bf16x4*>(&Bh[r * LDB + c4 * 4]) = __builtin_convertvector(b_reg[i] * b_msk[i], bf16x4);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) *reinterpret_cast<f32x4*>(&As[(tid + NT * i) * 4]) = a_reg[i] * a_msk[i];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) *reinterpret_cast<f32x4*>(&Bs[(tid + NT * i) * 4]) = b_reg[i] * b_msk[i];
  };
  auto mfma_range = [&](int buf, int kp0, int kp1) {
    const float* As = As2[buf];
    const float* Bs = Bs2[buf];
    float af[2][TM], bf[2][TN];          // fragments of step kp+1 are read before the MFMAs of step kp
    auto read = [&](int kp, int slot) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[slot][i] = As[(2 * kp + lh) * BMc + wm * TM * 32 + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[slot][j] = Bs[(2 * kp + lh) * BNn + wn * TN * 32 + j * 32 + lr];
    };
    read(kp0, 0);
#pragma unroll
    for (int kp = kp0; kp < kp1; ++kp) {
      const int slot = (kp - kp0) & 1;
      if (kp + 1 < kp1) read(kp + 1, slot ^ 1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[slot][i], bf[slot][j], acc[i][j], 0, 0, 0);
    }
  };

  // bf16: the 2 x 16-pixel K steps of a tile; lane = (channel lr, pixel half lh), group of 16 lanes = one transposed block
  auto mfma_bf16 = [&](int buf) {
    typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4;
    const unsigned short* Ah = reinterpret_cast<const unsigned short*>(As2[buf]);
    const unsigned short* Bh = reinterpret_cast<const unsigned short*>(Bs2[buf]);
    const int li = lane & 15, q = li >> 2, pq = li & 3, g1 = (lane >> 4) & 1;
    bf16x8 af[2][TM], bfr[2][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int px = 16 * s + 8 * lh + q;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const unsigned short* a0 = &Ah[px * LDA + wm * TM * 32 + i * 32 + 16 * g1 + 4 * pq];
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)a0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(a0 + 4 * LDA));
        af[s][i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned short* b0 = &Bh[px * LDB + wn * TN * 32 + j * 32 + 16 * g1 + 4 * pq];
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)b0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(b0 + 4 * LDB));
        bfr[s][j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i], bfr[s][j], acc[i][j], 0, 0, 0);
  };

  if (ntiles > 0) {
    if (!ROWS && tid < 32) fill_rowtab(0);
    __syncthreads();
    load_tile(0);
    if (!ROWS && ntiles > 1 && tid < 32) fill_rowtab(1);
    store_tile(0);
    __syncthreads();                    // tile 0 in LDS, rowtab[1] visible
    if (ntiles > 1) load_tile(1);
  }
  if constexpr (BF) {
    // the transposing read needs EXEC all ones: every wave executes it (idle waves of small tiles read valid addresses
    // and discard), no lane-dependent branch around it
    for (int t = 0; t < ntiles; ++t) {
      const int cur = t & 1;
      if (t + 1 < ntiles) store_tile(cur ^ 1);
      if (!ROWS && t + 2 < ntiles && tid < 32) fill_rowtab(t + 2);
      if (active) mfma_bf16(cur);
      __syncthreads();
      if (t + 2 < ntiles) load_tile(t + 2);
    }
  } else
  for (int t = 0; t < ntiles; ++t) {
    const int cur = t & 1;
    if (active) mfma_range(cur, 0, 8);
    if (t + 1 < ntiles) store_tile(cur ^ 1);               // tile t+1: its loads were issued half a tile ago
    if (!ROWS && t + 2 < ntiles && tid < 32) fill_rowtab(t + 2);   // into rowtab[cur]: tile t's rows are no longer needed
    if (active) mfma_range(cur, 8, 16);
    __syncthreads();                    // tile t+1 and rowtab[cur] visible; everyone is done reading tile t
    if (t + 2 < ntiles) load_tile(t + 2);                  // in flight during the first half of tile t+1
  }

  if (!active) return;
  float* slab = p.slab + (size_t)split * p.Cdpad * p.NNpad;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nn = nn0 + wn * TN * 32 + j * 32 + lr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wm * TM * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        slab[(size_t)co * p.NNpad + nn] = acc[i][j][e];
      }
  }
}

struct WgradReduceParams {
  const float* slab;
  float* dw;
  long long sO, sI, sH, sW;
  int O, I, kh, kw, splits, Cdpad, NNpad;
  int accumulate;      // 1: dw += the slab sum (a second use of the same weight in one backward pass: srgan_set_wgrad_accumulate)
};

// Workgroup = one output channel x 64 input channels x all taps: the slab row is read in its own order ([tap][i]: runs of 64
// floats), summed over the splits in split order, transposed in LDS and written in dW's order ([i][tap]: one contiguous run for
// the standard weight layout).  Round 3: the first version walked a flat 64-bit index (three 64-bit divisions per output, every
// store 36 bytes from its neighbour's) and ran at a sixth of the memory rate -- 145 launches, 1.7 ms of the step.
constexpr int kReduceIC = 64, kReduceMaxTaps = 64;      // (validate: at most 64 taps)

__device__ __forceinline__ int reduce_row_blocks(const WgradReduceParams& p) { return p.O * ((p.I + kReduceIC - 1) / kReduceIC); }

__device__ __forceinline__ void reduce_row_block(const WgradReduceParams& p, int blk, float* sm) {
  const int ichunks = (p.I + kReduceIC - 1) / kReduceIC;
  const int o = blk / ichunks, i0 = (blk - o * ichunks) * kReduceIC;
  const int ic = min(kReduceIC, p.I - i0), taps = p.kh * p.kw, n = ic * taps;
  const size_t slab_stride = (size_t)p.Cdpad * p.NNpad;
  const float* row = p.slab + (size_t)o * p.NNpad + i0;
  if (((p.I | p.NNpad | ic) & 3) == 0 && (slab_stride & 3) == 0) {
    // 16 bytes per lane and four splits in flight: with one dword load per wave between dependent adds the sums ran at the
    // round-trip latency (1.7 TB/s over the step's 2.4 GB of slabs)
    const int icq = ic >> 2, nq = icq * taps;
    for (int e = threadIdx.x; e < nq; e += blockDim.x) {
      const int tap = e / icq, il = (e - tap * icq) * 4;
      const float* src = row + tap * p.I + il;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      int s = 0;
      for (; s + 4 <= p.splits; s += 4) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (size_t)s * slab_stride);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 1) * slab_stride);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 2) * slab_stride);
        const f32x4 a3 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 3) * slab_stride);
        v += a0; v += a1; v += a2; v += a3;             // split order, one rounding per addition, as the scalar loop
      }
      for (; s < p.splits; ++s) v += *reinterpret_cast<const f32x4*>(src + (size_t)s * slab_stride);
#pragma unroll
      for (int c = 0; c < 4; ++c) sm[(il + c) * taps + tap] = v[c];
    }
  } else {
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
      const int tap = e / ic, il = e - tap * ic;
      const float* src = row + tap * p.I + il;
      float v = 0.f;
      for (int s = 0; s < p.splits; ++s) v += src[s * slab_stride];
      sm[il * taps + tap] = v;
    }
  }
  __syncthreads();
  float* dst0 = p.dw + o * p.sO;
  for (int w = threadIdx.x; w < n; w += blockDim.x) {
    const int il = w / taps, tap = w - il * taps;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    float* dst = dst0 + ((i0 + il) * p.sI + ky * p.sH + kx * p.sW);
    *dst = p.accumulate ? *dst + sm[w] : sm[w];
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradReduceParams p) {
  __shared__ float sm[kReduceIC * kReduceMaxTaps];
  reduce_row_block(p, blockIdx.x, sm);
}

// many slabs, few outputs (first-layer weights: 64x3x4x4 summed over 1024 pixel ranges; the discriminator heads): workgroup =
// 64 consecutive elements of one slab row x 4 waves; wave w sums splits w, w + 4, ... (eight 256-byte loads in flight), the
// four partial sums meet in LDS and are added in wave order: fixed order, deterministic.  (Round 3; before, one wave per output
// with the lanes striding over the splits: every load touched 64 different lines for 4 bytes each.)
__device__ __forceinline__ int reduce_col_blocks(const WgradReduceParams& p) { return p.O * ((p.kh * p.kw * p.I + 63) / 64); }

__device__ __forceinline__ void reduce_col_block(const WgradReduceParams& p, int blk, float* sm) {
  const int nn = p.kh * p.kw * p.I, cols = (nn + 63) / 64;
  const int o = blk / cols, e = (blk - o * cols) * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  const size_t slab_stride = (size_t)p.Cdpad * p.NNpad;
  float v = 0.f;
  if (e < nn) {
    const float* src = p.slab + (size_t)o * p.NNpad + e;
    int s = wave;
    for (; s + 28 < p.splits; s += 32) {
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = src[(size_t)(s + 4 * j) * slab_stride];
#pragma unroll
      for (int j = 0; j < 8; ++j) v += a[j];
    }
    for (; s < p.splits; s += 4) v += src[(size_t)s * slab_stride];
  }
  sm[threadIdx.x] = v;
  __syncthreads();
  if (wave == 0 && e < nn) {
    v = ((sm[threadIdx.x] + sm[threadIdx.x + 64]) + sm[threadIdx.x + 128]) + sm[threadIdx.x + 192];
    const int tap = e / p.I, i = e - tap * p.I;
    const int ky = tap / p.kw, kx = tap - ky * p.kw;
    float* dst = p.dw + (o * p.sO + i * p.sI + ky * p.sH + kx * p.sW);
    *dst = p.accumulate ? *dst + v : v;
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_wave_kernel(WgradReduceParams p) {
  __shared__ float sm[256];
  reduce_col_block(p, blockIdx.x, sm);
}

// Many slab sums in one launch (srgan_wgrad_defer_begin: the weight gradients of a whole backward pass leave their slabs in an
// arena and are summed together when the pass ends -- 145 launches of ~12 us per train step, each too small to fill the chip,
// become a handful that run at the memory rate).  The records travel in the kernel arguments (a captured train step replays
// them as they were); a workgroup = one row block (reduce_row_block) or one column block (reduce_col_block) of one record: the
// same code, order and single store as the one-record kernels above: identical bits.
constexpr int kMultiReduceMax = 40;
struct MultiReduceEntry {
  WgradReduceParams r;
  int first_block;     // first workgroup of this record
  int wave;            // 1: column blocks (wgrad_reduce_wave_kernel's rule: many splits, few outputs)
};
struct MultiReduceArgs {
  int n;
  MultiReduceEntry e[kMultiReduceMax];
};

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(MultiReduceArgs a) {
  __shared__ float sm[kReduceIC * kReduceMaxTaps];
  int k = 0;
  while (k + 1 < a.n && (int)blockIdx.x >= a.e[k + 1].first_block) ++k;
  const WgradReduceParams& p = a.e[k].r;
  const int blk = blockIdx.x - a.e[k].first_block;
  if (a.e[k].wave) reduce_col_block(p, blk, sm);
  else reduce_row_block(p, blk, sm);
}

// column sums of a dense [M][C] matrix -> out[C] (bias gradients); two-stage, deterministic.
__global__ void colsum_partial_kernel(const float* a, float* part, int M, int C, int rows_per_block) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float s = 0.f;
  for (int m = m0; m < m1; ++m) s += a[(size_t)m * C + c];
  part[(size_t)blockIdx.y * C + c] = s;
}
// one WAVE per column: lane l sums partials l, l + 64, ... (fixed assignment), then a shuffle tree -- deterministic, and the
// up-to-1024 partials no longer sit in one thread's serial chain (18 us per launch, 25 launches per step)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* part, float* out, int C, int nparts, int accumulate) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (c >= C) return;
  float s = 0.f;
  for (int i = lane; i < nparts; i += 64) s += part[(size_t)i * C + c];
  s = wave_sum(s);
  if (lane == 0) out[c] = accumulate ? out[c] + s : s;
}

// at most four columns (the discriminator's PatchGAN and class heads: the only biased convolutions of a train step): one
// workgroup, thread t sums rows t, t + 256, ... of every column, the 256 partial sums meet in LDS and are added in thread order
// -- one launch instead of the partial + final pair (25 pairs per step)
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float* __restrict__ a, float* __restrict__ out, int M, int C,
                                                            int accumulate) {
  __shared__ float sh[4][256];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int m = threadIdx.x; m < M; m += 256)
    for (int c = 0; c < C; ++c) s[c] += a[(size_t)m * C + c];
  for (int c = 0; c < C; ++c) sh[c][threadIdx.x] = s[c];
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += sh[threadIdx.x][i];
    out[threadIdx.x] = accumulate ? out[threadIdx.x] + t : t;
  }
}

// ---- host side ----------------------------------------------------------------------------
namespace {

struct TileChoice { int BM, BN; };

TileChoice choose_tile(long long M, int N) {
  static const bool big = !SRGAN_AB_SET("SRGAN_NO_BIG_TILE");
  if (N <= 32) return {128, 32};
  if (N <= 64) return (big && ceil_div(M, 256) >= 256) ? TileChoice{256, 64} : TileChoice{128, 64};
  long long tiles = ceil_div(M, 128) * ceil_div(N, 128);
  if (tiles < 256) return {64, 64};
  // 8-wave 256x128 workgroups (one per CU, both waves of a SIMD barrier-coupled) when the grid fills the chip in whole
  // rounds: avoids the tail in which one of two independent co-resident workgroups runs alone on its SIMDs
  long long big_tiles = ceil_div(M, 256) * ceil_div(N, 128);
  if (big && big_tiles >= 256) return {256, 128};
  return {128, 128};
}

template <int BM, int BN, int WM, int WN>
int launch_igemm(const IgemmParams& p, int phases, bool vec, hipStream_t st, double flops) {
  constexpr int tile_id = (BM == 256 && BN == 64) ? 6 : (BM == 256) ? 5 : (BM == 128 && BN == 128) ? 0 : (BM == 128 && BN == 64) ? 1 : (BM == 128 && BN == 32) ? 2 : 3;
  ProfScope scope(tile_id * 2 + (vec ? 1 : 0), flops, st);
  dim3 grid((unsigned)(p.m_tiles * p.n_tiles), (unsigned)phases, (unsigned)std::max(p.ksplit, 1));
  if (vec && compute_bf16())
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, true>), grid, dim3(WM * WN * 64), 0, st, p);
  else if (vec)
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true>), grid, dim3(WM * WN * 64), 0, st, p);
  else
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, false>), grid, dim3(WM * WN * 64), 0, st, p);
  return check_launch("igemm_kernel");
}

int run_igemm_tiles(IgemmParams p, int phases, hipStream_t st, double flops);
int run_igemm(IgemmParams p, int phases, hipStream_t st, double flops, float* slab = nullptr);

// bf16 mode: layers whose K tiles are 64 channels of one tap run on igemm16_kernel (128 x 128 tiles; 128 x 64 for 64 outputs)
static bool igemm16_ok(const IgemmParams& p) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_IGEMM16");
  if (off || !compute_bf16() || p.Cs % BK16 != 0 || p.Kpad != p.K || p.Cd < 64 || p.Ty > 8 || p.Tx > 8 || p.M >= (1 << 23)) return false;
  // byte offsets of the gather are 32-bit (as in igemm_kernel's vector path)
  return (long long)p.NB * p.Hs * p.Ws * p.Cs < (1LL << 29) && (long long)p.Npad * p.Kpad < (1LL << 29);
}
static int igemm16_bn(const IgemmParams& p) { return p.Cd <= 64 ? 64 : 128; }

// Round 6: of igemm16's layers, the 3x3 / stride-1 ones on maps large enough to fill the device run on halo16e_kernel
// (conv_halo16e.hip) -- the style encoder's convolutions on its 62- and 31-pixel maps and their input gradients.  Geometry only
// (the packed operand is laid out for the kernel that will read it, whatever the tensors' types turn out to be).
static bool halo16e_ok(const IgemmParams& p) {
  if (!igemm16_ok(p) || p.stride != 1 || p.Ty != 3 || p.Tx != 3 || p.xpad_off != 0) return false;
  if (!halo16e_shape_ok(p.Cs, p.Cd) || p.Npad != p.Cd || p.K != 9 * p.Cs) return false;
  if (p.mode == 0 && p.reflect && (p.pad != 1 || p.Hs < 2 || p.Ws < 2)) return false;
  if (p.mode == 1 && p.reflect) return false;
  // a destination a few pixels wider than a multiple of the 32-column patch wastes most of its last column of patches (the
  // 33 x 33 padded gradient of a 31 x 31 layer: 73 us against the implicit GEMM's 37): at least 3/4 of the patch columns filled
  if ((long long)p.Wd * 4 < 3LL * 32 * ceil_div(p.Wd, 32) && wino_threshold_scale() > 0) return false;
  // enough workgroups for about half a device round (tests: SRGAN_WINOGRAD_THRESHOLD_SCALE=0 drives small shapes through it)
  const long long wgs = (long long)p.NB * ceil_div(p.Hd, halo16e_patch_rows(p.Cs, p.Cd)) * ceil_div(p.Wd, 32);
  return wgs >= 120 * wino_threshold_scale() && (long long)p.NB * p.Hd * p.Wd * p.Cd < (1LL << 30);
}

static int launch_halo16e(const IgemmParams& p, hipStream_t st, double flops) {
  Halo16eParams q{};
  q.src = p.src; q.wp = reinterpret_cast<const unsigned short*>(p.wp); q.bias = p.bias; q.dst = p.dst;
  q.NB = p.NB; q.Hs = p.Hs; q.Ws = p.Ws; q.Cs = p.Cs; q.Hd = p.Hd; q.Wd = p.Wd; q.N = p.Cd;
  if (p.mode == 0) {            // destination row a reads source rows a - pad + ty
    q.oy0 = -p.pad; q.ox0 = -p.pad; q.flip = 0;
  } else {                      // destination row a reads source rows a + pad - ty = (a + pad - 2) + (2 - ty)
    q.oy0 = p.pad - 2; q.ox0 = p.pad - 2; q.flip = 1;
  }
  q.reflect = p.reflect; q.act = p.act; q.slope = p.slope; q.src16 = p.src16; q.dst16 = p.dst16;
  return halo16e_run(q, flops, st);
}

static int launch_igemm16(IgemmParams p, int phases, hipStream_t st, double flops) {
  const int bn = igemm16_bn(p);
#ifdef SRGAN_EXPERIMENTS
  p.exp = (int)SRGAN_AB_INT("SRGAN_IG16_EXP", 0);
#endif
  p.m_tiles = (int)ceil_div(p.M, 128);
  p.n_tiles = (int)ceil_div(p.Cd, bn);
  ProfScope scope(36, flops, st);
  // two resident workgroups per CU over all (phase, split) planes; a workgroup walks its plane's tiles with that stride
  const long long planes = (long long)phases * std::max(p.ksplit, 1), tiles = (long long)p.m_tiles * p.n_tiles;
  const long long gx = std::min<long long>(tiles, std::max<long long>(8, (512 / planes) & ~7LL));
  dim3 grid((unsigned)gx, (unsigned)phases, (unsigned)std::max(p.ksplit, 1));
  const bool o16 = p.dst16 != 0 && p.ksplit <= 1;
  static const bool no_dma = SRGAN_AB_SET("SRGAN_NO_IG16_DMA");       // (A/B, experiment builds only)
  const bool dma = !no_dma;
#define SRGAN_IG16(BN_)                                                                                      \
  do {                                                                                                        \
    if (p.src16 && o16 && dma) hipLaunchKernelGGL((igemm16_kernel<BN_, true, true, true>), grid, dim3(256), 0, st, p);      \
    else if (p.src16 && dma) hipLaunchKernelGGL((igemm16_kernel<BN_, true, false, true>), grid, dim3(256), 0, st, p);       \
    else if (p.src16 && o16) hipLaunchKernelGGL((igemm16_kernel<BN_, true, true>), grid, dim3(256), 0, st, p);      \
    else if (p.src16) hipLaunchKernelGGL((igemm16_kernel<BN_, true, false>), grid, dim3(256), 0, st, p);       \
    else if (o16) hipLaunchKernelGGL((igemm16_kernel<BN_, false, true>), grid, dim3(256), 0, st, p);           \
    else hipLaunchKernelGGL((igemm16_kernel<BN_, false, false>), grid, dim3(256), 0, st, p);                  \
  } while (0)
  if (bn == 64) SRGAN_IG16(64);
  else SRGAN_IG16(128);
#undef SRGAN_IG16
  return check_launch("igemm16_kernel");
}

// ---- split-K for layers with few pixel tiles and long K loops (the encoder's 7x7 / 3x3 maps, the discriminators' 8x8 / 4x4
// maps): 64-200 four-wave workgroups walking 64-144 K tiles leave most SIMDs with one wave and nothing to hide its loads
// behind (22-48 TFLOP/s).  The K range is cut into `ksplit` pieces (grid z), every piece writes raw partial sums into its own
// destination-shaped slab, and splitk_reduce_kernel adds the slabs in split order (deterministic), the bias and the activation.
template <bool OUT16 = false>
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int ksplit, long long n4, long long stride,
                                     const float* __restrict__ bias, int Cd, int act, float slope, float* __restrict__ dst) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = *reinterpret_cast<const f32x4*>(slab + i * 4);
    for (int k = 1; k < ksplit; ++k) v += *reinterpret_cast<const f32x4*>(slab + (size_t)k * stride + i * 4);
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + (i * 4) % Cd);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act, slope);
    if constexpr (OUT16) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(dst) + i * 4) = __builtin_convertvector(v, bf16x4);
    else *reinterpret_cast<f32x4*>(dst + i * 4) = v;
  }
}

struct SplitKPlan { int ksplit, kt_per_split; long long dst_elems; };

static TileChoice final_tile(const IgemmParams& p) {
  TileChoice tc = choose_tile(p.M, p.Cd);
  const bool vec = (p.Cs % BK) == 0;
  if (tc.BM == 256 && !vec) tc = {128, tc.BN};
  return tc;
}

static SplitKPlan plan_splitk(const IgemmParams& p, int phases) {
  SplitKPlan s{1, 0, (long long)p.NB * p.Hd * p.Wd * p.Cd};
  static const bool off = SRGAN_AB_SET("SRGAN_NO_SPLITK");
  static const int target = SRGAN_AB_INT("SRGAN_SPLITK_TARGET", 768);
  const bool k16 = igemm16_ok(p);
  if (phases == 1 && halo16e_ok(p)) return s;      // served by halo16e_kernel (its packed operand is the register image)
  const TileChoice tc = k16 ? TileChoice{128, igemm16_bn(p)} : final_tile(p);
  if (off || tc.BM == 256 || (p.Cd & 3) != 0 || s.dst_elems >= (1LL << 28)) return s;
  const long long wgs = ceil_div(p.M, tc.BM) * ceil_div(p.Cd, tc.BN) * phases;
  const int nk = p.Kpad / (k16 ? BK16 : BK);      // kt_per_split counts the serving kernel's K tiles
  long long ks = std::min<long long>(8, std::min<long long>(target / std::max<long long>(wgs, 1), nk / 8));
  if (ks < 2) return s;
  s.kt_per_split = (int)ceil_div(nk, ks);
  s.ksplit = (int)ceil_div(nk, s.kt_per_split);
  if (s.ksplit < 2) s.ksplit = 1;
  return s;
}

static size_t splitk_bytes(const IgemmParams& p, int phases) {
  const SplitKPlan s = plan_splitk(p, phases);
  return s.ksplit > 1 ? (size_t)s.ksplit * s.dst_elems * sizeof(float) : 0;
}

// `slab`: splitk_bytes(p, phases) bytes of scratch, or null (then the layer runs unsplit)
int run_igemm(IgemmParams p, int phases, hipStream_t st, double flops, float* slab) {
  const SplitKPlan sk = slab ? plan_splitk(p, phases) : SplitKPlan{1, 0, 0};
  float* const real_dst = p.dst;
  const float* const real_bias = p.bias;
  const int real_act = p.act;
  if (sk.ksplit > 1) {
    p.ksplit = sk.ksplit; p.kt_per_split = sk.kt_per_split; p.split_stride = sk.dst_elems;
    p.dst = slab; p.bias = nullptr; p.act = SRGAN_ACT_NONE;
  } else {
    p.ksplit = 1; p.kt_per_split = 0; p.split_stride = 0;
  }
  if (int e = run_igemm_tiles(p, phases, st, flops)) return e;
  if (sk.ksplit > 1) {
    const long long n4 = sk.dst_elems / 4;
    const dim3 rg((unsigned)std::max<long long>(1, std::min<long long>(ceil_div(n4, 256), 4096)));
    if (p.dst16)
      hipLaunchKernelGGL(splitk_reduce_kernel<true>, rg, dim3(256), 0, st, (const float*)slab, sk.ksplit, n4, sk.dst_elems, real_bias,
                         p.Cd, real_act, p.slope, real_dst);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<false>, rg, dim3(256), 0, st, (const float*)slab, sk.ksplit, n4, sk.dst_elems, real_bias,
                         p.Cd, real_act, p.slope, real_dst);
    return check_launch("splitk_reduce_kernel");
  }
  return 0;
}

int run_igemm_tiles(IgemmParams p, int phases, hipStream_t st, double flops) {
  if (phases == 1 && p.ksplit <= 1 && halo16e_ok(p)) return launch_halo16e(p, st, flops);
  if (igemm16_ok(p)) return launch_igemm16(p, phases, st, flops);
  TileChoice tc = choose_tile(p.M, p.Cd);
  p.m_tiles = (int)ceil_div(p.M, tc.BM);
  p.n_tiles = (int)ceil_div(p.Cd, tc.BN);
  const bool vec = (p.Cs % BK) == 0;
  if (tc.BM == 256 && tc.BN == 128 && vec) return launch_igemm<256, 128, 4, 2>(p, phases, true, st, flops);
  if (tc.BM == 256 && tc.BN == 64 && vec) return launch_igemm<256, 64, 4, 2>(p, phases, true, st, flops);
  if (tc.BM == 256) { tc = {128, tc.BN}; p.m_tiles = (int)ceil_div(p.M, 128); }
  if (tc.BM == 128 && tc.BN == 128) return launch_igemm<128, 128, 2, 2>(p, phases, vec, st, flops);
  if (tc.BM == 128 && tc.BN == 64) return launch_igemm<128, 64, 2, 2>(p, phases, vec, st, flops);
  if (tc.BM == 128 && tc.BN == 32) return launch_igemm<128, 32, 4, 1>(p, phases, vec, st, flops);
  return launch_igemm<64, 64, 2, 2>(p, phases, vec, st, flops);
}

int npad_for(long long M, int N) {
  TileChoice tc = choose_tile(M, N);
  return (int)round_up(N, tc.BN);
}

// algorithmic FLOPs of one conv pass: 2 * (N*Ho*Wo) * O * (kh*kw*I)
double conv_flops(const srgan_conv_desc* d) {
  return 2.0 * d->N * d->Ho * d->Wo * (double)d->O * d->kh * d->kw * d->I;
}

int validate(const srgan_conv_desc* d) {
  SRGAN_REQUIRE(d != nullptr, "conv desc is null");
  SRGAN_REQUIRE(d->N > 0 && d->Hi > 0 && d->Wi > 0 && d->I > 0 && d->O > 0 && d->kh > 0 && d->kw > 0,
                "conv desc: non-positive dimension");
  SRGAN_REQUIRE(d->stride >= 1 && d->pad >= 0, "conv desc: bad stride/pad");
  const int ho = (d->Hi + 2 * d->pad - d->kh) / d->stride + 1;
  const int wo = (d->Wi + 2 * d->pad - d->kw) / d->stride + 1;
  SRGAN_REQUIRE(ho == d->Ho && wo == d->Wo, "conv desc: Ho/Wo (%d,%d) do not match geometry (%d,%d)", d->Ho, d->Wo, ho, wo);
  SRGAN_REQUIRE(d->pad_mode == SRGAN_PAD_ZERO || d->pad_mode == SRGAN_PAD_REFLECT, "conv desc: bad pad_mode");
  if (d->pad_mode == SRGAN_PAD_REFLECT)
    SRGAN_REQUIRE(d->stride == 1 && d->pad < d->Hi && d->pad < d->Wi, "reflect padding needs stride 1 and pad < size");
  SRGAN_REQUIRE((long long)d->N * d->Hi * d->Wi * (long long)d->I < (1LL << 30) - (1LL << 24) &&
                (long long)d->N * d->Ho * d->Wo * (long long)d->O < (1LL << 30) - (1LL << 24),
                "tensor too large: the gather uses 32-bit byte offsets (< 4 GiB per activation tensor)");
  SRGAN_REQUIRE(d->kh * d->kw <= 64, "kernel window larger than 64 taps is not supported");
  return 0;
}

struct WgradPlan { int BMc, BNn, splits, rows_per_split, Cdpad, NNpad, co_tiles, nn_tiles; bool vec, rows; };

// the instantiation that serves a (tile, vec, rows) choice
struct WgradVariant { void (*fn)(WgradParams); int threads; };

template <int BMc, int BNn, int WM, int WN>
static WgradVariant wgrad_variant(bool vec, bool rows, bool io16 = false) {
  constexpr int NT = (WM * WN > 4 ? WM * WN : 4) * 64;
  if constexpr (BMc >= 64 && BNn >= 64) {
    if (io16) return {wgrad_kernel<BMc, BNn, WM, WN, true, false, true, true>, NT};      // (wgrad_io16_ok: vec, not rows, bf16 mode)
  }
  if constexpr (BMc >= 64 && BNn >= 64) {
    if (vec && rows && compute_bf16()) return {wgrad_kernel<BMc, BNn, WM, WN, true, true, true>, NT};
    if (vec && rows) return {wgrad_kernel<BMc, BNn, WM, WN, true, true>, NT};
  }
  if (vec && compute_bf16()) return {wgrad_kernel<BMc, BNn, WM, WN, true, false, true>, NT};
  if (vec) return {wgrad_kernel<BMc, BNn, WM, WN, true>, NT};
  return {wgrad_kernel<BMc, BNn, WM, WN, false>, NT};
}

static bool wgrad_lookup(int BMc, int BNn, bool vec, bool rows, WgradVariant* k, bool io16 = false) {
  if (BMc == 128 && BNn == 128) *k = wgrad_variant<128, 128, 2, 2>(vec, rows, io16);
  else if (BMc == 128 && BNn == 64) *k = wgrad_variant<128, 64, 2, 2>(vec, rows, io16);
  else if (BMc == 128 && BNn == 32) *k = wgrad_variant<128, 32, 4, 1>(vec, rows);
  else if (BMc == 64 && BNn == 128) *k = wgrad_variant<64, 128, 2, 2>(vec, rows, io16);
  else if (BMc == 64 && BNn == 64) *k = wgrad_variant<64, 64, 2, 2>(vec, rows, io16);
  else if (BMc == 64 && BNn == 32) *k = wgrad_variant<64, 32, 2, 1>(vec, rows);
  else if (BMc == 32 && BNn == 128) *k = wgrad_variant<32, 128, 1, 4>(vec, rows);
  else if (BMc == 32 && BNn == 64) *k = wgrad_variant<32, 64, 1, 2>(vec, rows);
  else if (BMc == 32 && BNn == 32) *k = wgrad_variant<32, 32, 1, 1>(vec, rows);
  else return false;
  return true;
}

// Workgroups of a weight-gradient variant the whole device holds at once (one "round" of the grid).  The split count is
// chosen so that the grid is a whole number of rounds: a grid of 2.04 rounds costs three (measured: 1044 blocks on 512
// slots ran 409 us where 1008 blocks need 373 us).  Falls back to the LDS bound when no device is present (the
// workspace query is a host-only call).
static long long wgrad_round_slots(const WgradPlan& w) {
  static std::mutex mu;
  static std::map<int, long long> cache;
  const int key = (w.BMc << 16) | (w.BNn << 4) | (compute_bf16() ? 4 : 0) | (w.vec ? 2 : 0) | (w.rows ? 1 : 0);
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int per_cu = 0, cus = 0, dev = 0;
  WgradVariant k{};
  if (wgrad_lookup(w.BMc, w.BNn, w.vec, w.rows, &k) && hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k.fn, k.threads, 0) == hipSuccess && per_cu > 0 && cus > 0) {
    // measured
  } else {
    (void)hipGetLastError();
    cus = 256;
    const int lds = 2 * 32 * (std::max(w.BMc, 64) + std::max(w.BNn, 64)) * 4 + 1024;
    per_cu = std::max(1, std::min(8, 163840 / lds));
  }
  return cache[key] = (long long)per_cu * cus;
}

WgradPlan plan_wgrad(const srgan_conv_desc* d) {
  WgradPlan w;
  const long long M = (long long)d->N * d->Ho * d->Wo;
  const int NN = d->kh * d->kw * d->I;
  // (8-wave 256x128 weight-gradient tiles were measured SLOWER in the train step: 88 vs 98 TFLOP/s, A/B on one device)
  w.BMc = d->O <= 32 ? 32 : (d->O <= 64 ? 64 : 128);
  w.vec = (d->I % 32) == 0;
  if (w.vec) w.BNn = (d->I % 128 == 0) ? 128 : ((d->I % 64 == 0) ? 64 : 32);
  else w.BNn = 64;
  // supported shapes: (128,128) (128,64) (128,32) (64,128) (64,64) (64,32) (32,128) (32,64) (32,32)
  // row-aligned fast path: every 32-pixel tile inside one image row, every split a whole number of tiles
  w.rows = w.vec && (d->Wo % 32) == 0 && (M % 32) == 0 && (d->O % 4) == 0;
  w.co_tiles = (int)ceil_div(d->O, w.BMc);
  w.nn_tiles = (int)ceil_div(NN, w.BNn);
  w.Cdpad = w.co_tiles * w.BMc;
  w.NNpad = w.nn_tiles * w.BNn;
  const long long tiles = (long long)w.co_tiles * w.nn_tiles;
  const long long max_splits = std::max<long long>(1, ceil_div(M, 256));   // at least 8 K-tiles per split
  static const long long wg_target = SRGAN_AB_INT("SRGAN_WGRAD_BLOCKS", 512);
  long long splits;
  {
    // about wg_target blocks, rounded to whole device rounds; among the candidate round counts take the best-filled
    const long long slots = wgrad_round_slots(w);
    const long long r0 = std::max<long long>(1, (wg_target + slots / 2) / slots);
    double best_fill = -1.0;
    splits = 1;
    for (long long r = r0; r <= r0 + 2; ++r) {
      long long s = std::min(std::max<long long>(1, (slots * r) / tiles), max_splits);
      const long long blocks = tiles * s;
      const long long rounds = ceil_div(blocks, slots);
      const double fill = (double)blocks / (double)(rounds * slots);
      if (fill > best_fill + 0.04) { best_fill = fill; splits = s; }
    }
  }
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  w.rows_per_split = (int)round_up(ceil_div(M, splits), 32);
  w.splits = (int)ceil_div(M, w.rows_per_split);
  return w;
}

size_t pack_bytes(const srgan_conv_desc* d) {
  // fwd pack and dgrad pack upper bounds
  const long long M_f = (long long)d->N * d->Ho * d->Wo;
  const long long Kf = round_up((long long)d->kh * d->kw * d->I, BK);
  const long long f = (long long)npad_for(M_f, d->O) * Kf;
  const int s = d->stride;
  const int Ty = (int)ceil_div(d->kh, s), Tx = (int)ceil_div(d->kw, s);
  const long long Kd = round_up((long long)Ty * Tx * d->O, BK);
  const long long g = (long long)s * s * round_up(d->I, 128) * Kd;
  size_t bytes = (size_t)((f > g ? f : g) * sizeof(float));
  if (rowconv_applicable(d)) bytes = std::max(bytes, rowconv_packed_elems(d) * sizeof(float));
  {
    srgan_conv_desc f;
    long long w_off;
    if (narrow_dgrad_desc(d, &f, &w_off) && rowconv_applicable(&f)) bytes = std::max(bytes, rowconv_packed_elems(&f) * sizeof(float));
  }
  if (wino_applicable(d, 0)) bytes = std::max(bytes, wino_packed_bytes(d, 0));
  if (wino_applicable(d, 1)) bytes = std::max(bytes, wino_packed_bytes(d, 1));
  if (d->I == 3 || d->O == 3) bytes = std::max(bytes, (size_t)d->kh * ((d->kw * 3 + 1) & ~1) * std::max(d->I, d->O) * sizeof(float));   // conv_rgbin.hip
  return bytes;
}

}  // namespace
}  // namespace srgan

using namespace srgan;

namespace srgan { static size_t conv_splitk_bytes(const srgan_conv_desc* d, int kind); }

extern "C" size_t srgan_conv2d_workspace(const srgan_conv_desc* d) {
  if (validate(d) != 0) return 0;
  size_t bytes = pack_bytes(d);
  // reflect dgrad: padded-gradient temp
  if (d->pad_mode == SRGAN_PAD_REFLECT)
    bytes += (size_t)d->N * (d->Hi + 2 * d->pad) * (d->Wi + 2 * d->pad) * d->I * sizeof(float);
  // split-K slabs of the implicit-GEMM forward / input gradient, behind the packed operand (and the reflect temp)
  bytes += 256 + std::max(conv_splitk_bytes(d, 0), conv_splitk_bytes(d, 1));
  // F(4x4,3x3) layers: the transformed-input image, behind the packed operand (64-float aligned)
  bytes = std::max(bytes, (size_t)round_up((long long)pack_bytes(d), 256) + std::max(wino_scratch_bytes(d, 0), wino_scratch_bytes(d, 1)));
  WgradPlan w = plan_wgrad(d);
  if (wino_wgrad_applicable(d)) wino_wgrad_slab(d, &w.splits, &w.Cdpad, &w.NNpad);
  if (rgb_wgrad_kind(d) >= 0) rgb_wgrad_slab(d, &w.splits, &w.Cdpad, &w.NNpad);
  size_t wg = (size_t)w.splits * w.Cdpad * w.NNpad * sizeof(float);
  size_t cs = (size_t)1024 * d->O * sizeof(float);
  if (wg + cs > bytes) bytes = wg + cs;
  if (narrow_applicable(d)) bytes = std::max(bytes, narrow_workspace(d) + cs);
  {
    Wino43WgradGeom wg43{};
    if (wino43_wgrad_geometry(d, &wg43)) bytes = std::max(bytes, (size_t)round_up((long long)(wg43.slab_bytes + cs), 256) + wg43.z_bytes);
  }
  return bytes + 4096;
}

namespace srgan {
// ---- forward: which kernel family serves this layer, and the packed-weight layout it wants ----
enum FwdPath { PATH_IGEMM = 0, PATH_NARROW = 1, PATH_WAVE = 2, PATH_DENSE = 3, PATH_WINO = 4, PATH_ROWCONV = 5, PATH_RGBIN = 6 };

// ---- 3-channel 7x7 heads (the generator's RGB output layer, model.py:232, and the input gradient of its 7x7 RGB input
// layer, model.py:212) on the matrix pipe.  With Cout = 3 a 32-wide MFMA tile would be 90 % padding, so the layer is split:
//   P[b][y][x'][(co, kx)] = sum_{ky, c} x[b][y + ky - pad][x'][c] * w[co][c][ky][kx]     -- a 7x1 conv with 21 (-> 32) outputs
//   y[b][y][x][co]        = bias[co] + sum_kx P[b][y][x + kx - pad][(co, kx)]             -- a shift-add over 7 columns
// The first is the ordinary implicit GEMM (N tile 32, 66 % useful), the second a memory-bound pass over P (21 floats per pixel,
// kept behind the packed weights).  1.3x faster than the VALU kernel of conv_narrow.hip on the 128x128 maps (230 vs 305 us at
// batch 32; the N = 32 tile runs at 74 TFLOP/s executed -- a 256x32 tile with 64x32 wave tiles was tried and was slower).
static bool rowconv_applicable(const srgan_conv_desc* d) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_ROWCONV");
  return !off && d->O == 3 && d->kh == 7 && d->kw == 7 && d->stride == 1 && d->pad_mode == SRGAN_PAD_ZERO && (d->I % BK) == 0 &&
         d->Wo >= 64 && d->Ho >= 8 && (long long)d->N * d->Ho * d->Wi * 21 < (1LL << 30);
}
constexpr int RC_N = 21, RC_NPAD = 32;
static size_t rowconv_weight_elems(const srgan_conv_desc* d) { return (size_t)RC_NPAD * d->kh * d->I; }
// (round 3: where conv_rgbout.hip's direct 4x4x1-MFMA kernel applies it takes the layer over -- same dispatch slot, own packed
// filter, no intermediate)
static size_t rowconv_packed_elems(const srgan_conv_desc* d) {
  if (rgbout_applicable(d)) return rgbout_packed_elems(d);
  return round_up((long long)rowconv_weight_elems(d), 64) + (size_t)d->N * d->Ho * d->Wi * RC_N;
}

// wp[n = co*7 + kx][k = ky*I + c] = w[co][c][ky][kx], rows 21..31 zero
__global__ void rowconv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, long long sO, long long sI, long long sH,
                                    long long sW, int I, int kh, int kw) {
  const int K = kh * I, total = RC_NPAD * K;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int k = idx % K, n = idx / K;
    const int ky = k / I, c = k - ky * I;
    const int co = n / kw, kx = n - co * kw;
    wp[idx] = n < RC_N ? w[co * sO + c * sI + ky * sH + kx * sW] : 0.f;
  }
}

// y[pix][co] = bias[co] + sum_kx P[row, x + kx - pad][co*7 + kx]
__global__ void rowconv_shift_add_kernel(const float* __restrict__ P, const float* __restrict__ bias, float* __restrict__ y,
                                         long long rows, int Wi, int Wo, int pad) {
  const long long total = rows * Wo;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % Wo);
    const long long row = idx / Wo;
    const float* pr = P + row * Wi * RC_N;
    float a0 = bias ? bias[0] : 0.f, a1 = bias ? bias[1] : 0.f, a2 = bias ? bias[2] : 0.f;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const int xs = x + kx - pad;
      if ((unsigned)xs < (unsigned)Wi) {
        const float* q = pr + (size_t)xs * RC_N + kx;
        a0 += q[0];
        a1 += q[7];
        a2 += q[14];
      }
    }
    float* o = y + idx * 3;
    o[0] = a0; o[1] = a1; o[2] = a2;
  }
}

static int rowconv_pack(const srgan_conv_desc* d, const float* w, float* dst, hipStream_t st) {
  if (rgbout_applicable(d)) return rgbout_pack(d, w, dst, st);
  const int total = RC_NPAD * d->kh * d->I;
  hipLaunchKernelGGL(rowconv_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, dst, d->sO, d->sI, d->sH, d->sW, d->I,
                     d->kh, d->kw);
  return check_launch("rowconv_pack_kernel");
}


static FwdPath fwd_path(const srgan_conv_desc* d, int act) {
  if (act == SRGAN_ACT_NONE) {
    if (dense_head_applicable(d)) return PATH_DENSE;
    if (narrow_wave_applicable(d)) return PATH_WAVE;
    if (rowconv_applicable(d)) return PATH_ROWCONV;
    if (narrow_applicable(d)) return PATH_NARROW;
  }
  if (rgbin_applicable(d)) return PATH_RGBIN;       // 3-channel 7x7 input layer: LDS-staged halo on the MFMA (conv_rgbin.hip)
  if (wino_applicable(d, 0)) return PATH_WINO;      // bias / activation fused in its epilogue too
  return PATH_IGEMM;
}

static void fwd_geometry(const srgan_conv_desc* d, FwdPath path, IgemmParams& p) {
  p.NB = d->N; p.Hs = d->Hi; p.Ws = d->Wi; p.Cs = d->I;
  p.Hg = d->Ho; p.Wg = d->Wo; p.Hd = d->Ho; p.Wd = d->Wo; p.Cd = d->O;
  p.mode = 0; p.stride = d->stride; p.pad = d->pad; p.Ty = d->kh; p.Tx = d->kw;
  p.K = d->kh * d->kw * d->I; p.Kpad = (int)round_up(p.K, BK);
  p.M = d->N * d->Ho * d->Wo;
  p.Npad = (path == PATH_WAVE || path == PATH_DENSE) ? d->O : npad_for(p.M, d->O);
  p.reflect = d->pad_mode == SRGAN_PAD_REFLECT;
}

static size_t fwd_packed_bytes(const srgan_conv_desc* d, int act) {
  const FwdPath path = fwd_path(d, act);
  if (path == PATH_NARROW) return (size_t)d->I * d->kh * d->kw * 4 * sizeof(float);
  if (path == PATH_ROWCONV) return rowconv_packed_elems(d) * sizeof(float);
  if (path == PATH_RGBIN) return rgbin_packed_elems(d) * sizeof(float);
  if (path == PATH_WINO) return wino_packed_bytes(d, 0);
  IgemmParams p{};
  fwd_geometry(d, path, p);
  return (size_t)p.Npad * p.Kpad * sizeof(float);
}

static PackParams fwd_pack_params(const srgan_conv_desc* d, FwdPath path, const float* w, float* dst) {
  IgemmParams p{};
  fwd_geometry(d, path, p);
  PackParams q{};
  q.w = w; q.dst = dst; q.sO = d->sO; q.sI = d->sI; q.sH = d->sH; q.sW = d->sW;
  q.O = d->O; q.I = d->I; q.kh = d->kh; q.kw = d->kw; q.mode = 0; q.stride = d->stride; q.pad = d->pad;
  q.Ty = d->kh; q.Tx = d->kw; q.Cs = d->I; q.N = d->O; q.K = p.K; q.Kpad = p.Kpad; q.Npad = p.Npad; q.phases = 1;
  q.out16 = (path == PATH_IGEMM && igemm16_ok(p)) ? 1 : 0;
  q.regimg = (q.out16 && halo16e_ok(p)) ? 1 : 0;
  return q;
}

static int fwd_pack(const srgan_conv_desc* d, int act, const float* w, float* dst, hipStream_t st) {
  const FwdPath path = fwd_path(d, act);
  if (path == PATH_NARROW) return narrow_pack(d, w, dst, st);
  if (path == PATH_ROWCONV) return rowconv_pack(d, w, dst, st);
  if (path == PATH_RGBIN) return rgbin_pack(d, w, dst, st);
  if (path == PATH_WINO) return wino_pack(d, 0, w, dst, st);
  const PackParams q = fwd_pack_params(d, path, w, dst);
  long long total = (long long)q.Npad * q.Kpad;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0, st, q);
  return check_launch("pack_weights_kernel");
}

// `packed` = [32 x 7*I weights | P scratch]
static int rowconv_run(const srgan_conv_desc* d, const float* x, const float* packed, const float* bias, float* y, hipStream_t st) {
  if (rgbout_applicable(d)) return rgbout_run(d, x, packed, bias, y, st);
  float* P = const_cast<float*>(packed) + round_up((long long)rowconv_weight_elems(d), 64);
  IgemmParams p{};
  p.src = x; p.wp = packed; p.bias = nullptr; p.dst = P;
  p.NB = d->N; p.Hs = d->Hi; p.Ws = d->Wi; p.Cs = d->I;
  p.Hg = d->Ho; p.Wg = d->Wi; p.Hd = d->Ho; p.Wd = d->Wi; p.Cd = RC_N;
  p.mode = 0; p.stride = 1; p.pad = d->pad; p.xpad_off = d->pad;      // rows padded, columns not
  p.Ty = d->kh; p.Tx = 1;
  p.K = d->kh * d->I; p.Kpad = p.K; p.Npad = RC_NPAD;
  p.M = d->N * d->Ho * d->Wi;
  p.reflect = 0; p.act = SRGAN_ACT_NONE; p.slope = 0.f;
  if (int e = run_igemm(p, 1, st, conv_flops(d))) return e;
  const long long rows = (long long)d->N * d->Ho, total = rows * d->Wo;
  hipLaunchKernelGGL(rowconv_shift_add_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 16384)), dim3(256), 0, st,
                     (const float*)P, bias, y, rows, d->Wi, d->Wo, d->pad);
  return check_launch("rowconv_shift_add_kernel");
}

static int fwd_run(const srgan_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act,
                   float slope, float* scratch, hipStream_t st) {
  const FwdPath path = fwd_path(d, act);
  if (path == PATH_ROWCONV) return rowconv_run(d, x, wp, bias, y, st);
  if (path == PATH_NARROW) return narrow_fwd_packed(d, x, wp, bias, y, st);
  if (path == PATH_RGBIN) return rgbin_run(d, x, wp, bias, y, act, slope, st);
  if (path == PATH_WINO) return wino_run(d, 0, x, wp, bias, y, act, slope, scratch, st);
  IgemmParams p{};
  fwd_geometry(d, path, p);
  p.src = x; p.bias = bias; p.dst = y; p.act = act; p.slope = slope; p.wp = wp;
  if (path == PATH_DENSE) return dense_head_fwd(d, x, wp, p.Kpad, bias, y, st);
  if (path == PATH_WAVE) return narrow_wave_fwd(d, x, wp, p.Kpad, bias, y, st);
  return run_igemm(p, 1, st, conv_flops(d), scratch);      // scratch (may be null): room for split-K slabs
}

// ---- input gradient / transposed-conv forward ----
// Input gradient of a stride-1 zero-padded conv with <= 4 INPUT channels (the generator's 7x7 RGB layer inside the
// cycle / identity passes) = a narrow-OUTPUT convolution of dy with the flipped, transposed filter.  The implicit
// GEMM would pad those 3 channels to a 32-wide MFMA tile (9 % useful); conv_narrow.hip serves them directly.
static bool narrow_dgrad_desc(const srgan_conv_desc* d, srgan_conv_desc* f, long long* w_off) {
  if (d->stride != 1 || d->pad_mode != SRGAN_PAD_ZERO || d->I > 4 || d->kh != d->kw || d->kh - 1 - d->pad < 0) return false;
  *f = *d;
  f->Hi = d->Ho; f->Wi = d->Wo; f->Ho = d->Hi; f->Wo = d->Wi; f->I = d->O; f->O = d->I;
  f->pad = d->kh - 1 - d->pad;
  f->sO = d->sI; f->sI = d->sO; f->sH = -d->sH; f->sW = -d->sW;     // w'[i][o][ky][kx] = w[o][i][kh-1-ky][kw-1-kx]
  *w_off = (long long)(d->kh - 1) * d->sH + (long long)(d->kw - 1) * d->sW;
  return narrow_applicable(f);
}

// Input gradient of a stride-1 zero-padded conv with 3 OUTPUT channels (the generator's 7x7 RGB head) = a 3-channel-INPUT
// convolution of dy with the flipped, transposed filter: conv_rgbin.hip.
static bool rgbin_dgrad_desc(const srgan_conv_desc* d, srgan_conv_desc* f, long long* w_off) {
  if (d->stride != 1 || d->pad_mode != SRGAN_PAD_ZERO || d->O != 3 || d->kh != d->kw || d->kh - 1 - d->pad < 0) return false;
  *f = *d;
  f->Hi = d->Ho; f->Wi = d->Wo; f->Ho = d->Hi; f->Wo = d->Wi; f->I = d->O; f->O = d->I;
  f->pad = d->kh - 1 - d->pad;
  f->sO = d->sI; f->sI = d->sO; f->sH = -d->sH; f->sW = -d->sW;     // w'[i][o][ky][kx] = w[o][i][kh-1-ky][kw-1-kx]
  *w_off = (long long)(d->kh - 1) * d->sH + (long long)(d->kw - 1) * d->sW;
  return rgbin_applicable(f);
}

struct DgradGeom { IgemmParams p; int phases; bool reflect, wino, narrow, rgbin, narrow_s2; int Hd, Wd; size_t packed_elems; };

// Input gradient of a STRIDE-2 zero-padded conv with <= 4 input channels (the encoder's 7x7 / stride-2 RGB layer inside phase 2,
// D's 4x4 / stride-2 RGB layers inside phase 1): the implicit GEMM pads the 3 channels to a 32-wide MFMA tile (6.7 TFLOP/s,
// 0.69 ms for one launch of the step).  Row parity pp of the input pixel selects the taps ky = ky0 + 2a (ky0 = (pp + pad) & 1),
// so phase image (pp, qq) = dx[2u + pp][2v + qq] is a STRIDE-1 narrow-output convolution of dy with the flipped sub-filter:
//   dx_pq[u][v][c] = sum_{o, a', b'} dy[u + a' - pad_y][v + b' - pad_x][o] * w[o][c][ky0 + 2 (na - 1 - a')][kx0 + 2 (nb - 1 - b')]
// with pad_y = (na - 1) - (pp + pad - ky0) / 2: four launches of conv_narrow.hip's generic kernel that scatter their pixels with
// stride 2 into dx.  Returns false when the layer does not qualify.
static bool narrow_s2_phase(const srgan_conv_desc* d, int pp, int qq, srgan_conv_desc* f, long long* w_off, int* pad_x,
                            size_t* packed_off) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_NARROW_S2");
  if (off || d->stride != 2 || d->pad_mode != SRGAN_PAD_ZERO || d->I > 4 || d->O % 16 != 0 || d->kh != d->kw || d->kh < 2 ||
      d->kh > 8 || d->Hi < 2 || d->Wi < 2)
    return false;
  size_t off_elems = 0;
  for (int ph = 0; ph < 4; ++ph) {
    const int p1 = ph >> 1, q1 = ph & 1;
    const int ky0 = (p1 + d->pad) & 1, kx0 = (q1 + d->pad) & 1;
    const int na = (d->kh - ky0 + 1) / 2, nb = (d->kw - kx0 + 1) / 2;
    if (p1 == pp && q1 == qq) {
      *f = *d;
      f->Hi = d->Ho; f->Wi = d->Wo; f->I = d->O; f->O = d->I;
      f->Ho = (d->Hi - pp + 1) / 2; f->Wo = (d->Wi - qq + 1) / 2;
      f->kh = na; f->kw = nb; f->stride = 1;
      f->pad = (na - 1) - (pp + d->pad - ky0) / 2;
      *pad_x = (nb - 1) - (qq + d->pad - kx0) / 2;
      f->sO = d->sI; f->sI = d->sO; f->sH = -2 * d->sH; f->sW = -2 * d->sW;
      *w_off = (long long)(ky0 + 2 * (na - 1)) * d->sH + (long long)(kx0 + 2 * (nb - 1)) * d->sW;
      *packed_off = off_elems;
      return f->Ho > 0 && f->Wo > 0 && na > 0 && nb > 0 && f->pad >= 0 && *pad_x >= 0;
    }
    off_elems += (size_t)d->O * na * nb * 4;
  }
  return false;
}
static size_t narrow_s2_packed_elems(const srgan_conv_desc* d) {
  size_t n = 0;
  for (int ph = 0; ph < 4; ++ph) {
    const int ky0 = ((ph >> 1) + d->pad) & 1, kx0 = ((ph & 1) + d->pad) & 1;
    n += (size_t)d->O * ((d->kh - ky0 + 1) / 2) * ((d->kw - kx0 + 1) / 2) * 4;
  }
  return n;
}

static DgradGeom dgrad_geometry(const srgan_conv_desc* d) {
  DgradGeom g{};
  g.reflect = d->pad_mode == SRGAN_PAD_REFLECT;
  const int s = d->stride;
  // reflect: gradient w.r.t. the padded image (a pad-0 conv over Hi+2P), then fold.
  const int P = g.reflect ? d->pad : 0;
  g.Hd = d->Hi + 2 * P; g.Wd = d->Wi + 2 * P;
  IgemmParams& p = g.p;
  p.bias = nullptr;
  p.NB = d->N; p.Hs = d->Ho; p.Ws = d->Wo; p.Cs = d->O;
  p.Hg = (int)ceil_div(g.Hd, s); p.Wg = (int)ceil_div(g.Wd, s);
  p.Hd = g.Hd; p.Wd = g.Wd; p.Cd = d->I;
  p.mode = 1; p.stride = s; p.pad = g.reflect ? 0 : d->pad;
  p.Ty = (int)ceil_div(d->kh, s); p.Tx = (int)ceil_div(d->kw, s);
  p.K = p.Ty * p.Tx * d->O; p.Kpad = (int)round_up(p.K, BK);
  p.M = d->N * p.Hg * p.Wg;
  p.Npad = npad_for(p.M, d->I);
  p.reflect = 0; p.act = SRGAN_ACT_NONE; p.slope = 0.f;
  g.phases = s * s;
  g.packed_elems = (size_t)g.phases * p.Npad * p.Kpad;
  g.wino = wino_applicable(d, 1);
  if (g.wino) g.packed_elems = wino_packed_bytes(d, 1) / sizeof(float);
  srgan_conv_desc f;
  long long w_off;
  g.narrow = !g.wino && narrow_dgrad_desc(d, &f, &w_off);
  if (g.narrow) g.packed_elems = rowconv_applicable(&f) ? rowconv_packed_elems(&f) : (size_t)f.I * f.kh * f.kw * 4;
  g.rgbin = !g.wino && !g.narrow && rgbin_dgrad_desc(d, &f, &w_off);
  if (g.rgbin) g.packed_elems = rgbin_packed_elems(&f);
  int px;
  size_t po;
  g.narrow_s2 = !g.wino && !g.narrow && !g.rgbin && narrow_s2_phase(d, 0, 0, &f, &w_off, &px, &po);
  if (g.narrow_s2) g.packed_elems = narrow_s2_packed_elems(d);
  return g;
}

static PackParams dgrad_pack_params(const srgan_conv_desc* d, const DgradGeom& g, const float* w, float* dst) {
  PackParams q{};
  q.w = w; q.dst = dst; q.sO = d->sO; q.sI = d->sI; q.sH = d->sH; q.sW = d->sW;
  q.O = d->O; q.I = d->I; q.kh = d->kh; q.kw = d->kw; q.mode = 1; q.stride = d->stride; q.pad = g.p.pad;
  q.Ty = g.p.Ty; q.Tx = g.p.Tx; q.Cs = d->O; q.N = d->I; q.K = g.p.K; q.Kpad = g.p.Kpad; q.Npad = g.p.Npad; q.phases = g.phases;
  q.out16 = (!g.wino && !g.narrow && !g.rgbin && !g.narrow_s2 && igemm16_ok(g.p)) ? 1 : 0;
  q.regimg = (q.out16 && g.phases == 1 && halo16e_ok(g.p)) ? 1 : 0;
  return q;
}

static int dgrad_pack(const srgan_conv_desc* d, const float* w, float* dst, hipStream_t st) {
  DgradGeom g = dgrad_geometry(d);
  if (g.wino) return wino_pack(d, 1, w, dst, st);
  if (g.narrow) {
    srgan_conv_desc f;
    long long w_off;
    narrow_dgrad_desc(d, &f, &w_off);
    if (rowconv_applicable(&f)) return rowconv_pack(&f, w + w_off, dst, st);
    return narrow_pack(&f, w + w_off, dst, st);
  }
  if (g.rgbin) {
    srgan_conv_desc f;
    long long w_off;
    rgbin_dgrad_desc(d, &f, &w_off);
    return rgbin_pack(&f, w + w_off, dst, st);
  }
  if (g.narrow_s2) {
    for (int ph = 0; ph < 4; ++ph) {
      srgan_conv_desc f;
      long long w_off;
      int px;
      size_t po;
      if (!narrow_s2_phase(d, ph >> 1, ph & 1, &f, &w_off, &px, &po)) { set_error("narrow stride-2 dgrad: phase geometry"); return -1; }
      if (int e = narrow_pack(&f, w + w_off, dst + po, st)) return e;
    }
    return 0;
  }
  const PackParams q = dgrad_pack_params(d, g, w, dst);
  long long total = (long long)g.packed_elems;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 4096)), dim3(256), 0, st, q);
  return check_launch("pack_weights_kernel");
}

// `scratch`: room for the padded gradient when the conv is reflect-padded
__global__ void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ r, long long n4, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
    reinterpret_cast<f32x4*>(y)[i] += reinterpret_cast<const f32x4*>(r)[i];
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] += r[n4 * 4 + threadIdx.x];
}

// bytes of split-K slabs the implicit-GEMM forward (kind 0) / input gradient (kind 1) of d wants (0: another kernel family
// serves it, or the layer is not split)
static size_t conv_splitk_bytes(const srgan_conv_desc* d, int kind) {
  if (kind == 0) {
    const FwdPath path = fwd_path(d, SRGAN_ACT_NONE);
    if (path != PATH_IGEMM) return 0;
    IgemmParams p{};
    fwd_geometry(d, path, p);
    return splitk_bytes(p, 1);
  }
  const DgradGeom g = dgrad_geometry(d);
  if (g.wino || g.narrow || g.rgbin || g.narrow_s2) return 0;
  return splitk_bytes(g.p, g.phases);
}

static int dgrad_run(const srgan_conv_desc* d, const float* dy, const float* wp, float* dx, float* scratch, hipStream_t st,
                     const float* res = nullptr, const float* mask = nullptr, float mask_slope = 0.f);
static int dgrad_run_core(const srgan_conv_desc* d, const float* dy, const float* wp, float* dx, float* scratch, hipStream_t st,
                          const float* res, bool* res_done, const float* mask, float mask_slope, bool* mask_done);

// dx = dgrad(dy) (+ res): the F(4x4,3x3) kernel adds `res` in its epilogue; every other dispatch gets one in-place add pass
// ... (* mask): the LeakyReLU backward of the layer that produced this layer's input (mask = that activated tensor): in the
// epilogue of the transposed F(3x3,2x2) kernel, one in-place elementwise pass behind every other dispatch
static int dgrad_run(const srgan_conv_desc* d, const float* dy, const float* wp, float* dx, float* scratch, hipStream_t st,
                     const float* res, const float* mask, float mask_slope) {
  bool done = false, mdone = false;
  if (int e = dgrad_run_core(d, dy, wp, dx, scratch, st, res, &done, mask, mask_slope, &mdone)) return e;
  if (res && !done) {
    const long long n = (long long)d->N * d->Hi * d->Wi * d->I, n4 = n / 4;
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(ceil_div(n4, 256), 8192))),
                       dim3(256), 0, st, dx, res, n4, n);
    if (int e = check_launch("add_inplace_kernel")) return e;
  }
  if (mask && !mdone)
    return srgan_act_bwd(mask, dx, dx, (long long)d->N * d->Hi * d->Wi * d->I, SRGAN_ACT_LRELU, mask_slope, st);
  return 0;
}

static int dgrad_run_core(const srgan_conv_desc* d, const float* dy, const float* wp, float* dx, float* scratch, hipStream_t st,
                          const float* res, bool* res_done, const float* mask, float mask_slope, bool* mask_done) {
  DgradGeom g = dgrad_geometry(d);
  g.p.src = dy; g.p.wp = wp;
  g.p.dst = g.reflect ? scratch : dx;
  if (g.narrow) {
    srgan_conv_desc f;
    long long w_off;
    narrow_dgrad_desc(d, &f, &w_off);
    if (rowconv_applicable(&f)) return rowconv_run(&f, dy, wp, nullptr, dx, st);
    return narrow_fwd_packed(&f, dy, wp, nullptr, dx, st);
  }
  if (g.rgbin) {
    srgan_conv_desc f;
    long long w_off;
    rgbin_dgrad_desc(d, &f, &w_off);
    return rgbin_run(&f, dy, wp, nullptr, dx, SRGAN_ACT_NONE, 0.f, st);
  }
  if (g.narrow_s2) {
    for (int ph = 0; ph < 4; ++ph) {
      srgan_conv_desc f;
      long long w_off;
      int px;
      size_t po;
      if (!narrow_s2_phase(d, ph >> 1, ph & 1, &f, &w_off, &px, &po)) { set_error("narrow stride-2 dgrad: phase geometry"); return -1; }
      if (int e = narrow_fwd_strided(&f, px, dy, wp + po, dx, d->Hi, d->Wi, ph >> 1, ph & 1, st)) return e;
    }
    return 0;
  }
  if (g.wino) {
    // (F(4,3) needs zero padding, the fold scratch needs reflect padding: the two uses of `scratch` never meet)
    // (the mask rides in the epilogue only when nothing is added to dx afterwards)
    if (int e = wino_run(d, 1, dy, wp, nullptr, g.p.dst, SRGAN_ACT_NONE, 0.f, g.reflect ? nullptr : scratch, st,
                         g.reflect ? nullptr : res, res_done, false, (g.reflect || res) ? nullptr : mask, mask_slope, mask_done)) return e;
  } else {
    // split-K slabs sit behind the padded-gradient temp of a reflect layer
    float* slab = scratch ? (g.reflect ? scratch + round_up((long long)d->N * g.Hd * g.Wd * d->I, 64) : scratch) : nullptr;
    if (int e = run_igemm(g.p, g.phases, st, conv_flops(d), slab)) return e;
  }
  if (g.reflect) {
    long long n = (long long)d->N * d->Hi * d->Wi * d->I;
    hipLaunchKernelGGL(reflect_fold_kernel<false>, dim3((unsigned)std::min<long long>(ceil_div(n, 256), 8192)), dim3(256), 0, st,
                       (const float*)scratch, dx, d->N, d->Hi, d->Wi, d->I, d->pad);
    return check_launch("reflect_fold_kernel");
  }
  return 0;
}
}  // namespace srgan

extern "C" int srgan_conv2d_fwd(const srgan_conv_desc* d, const float* x, const float* w, const float* bias,
                                float* y, int act, float slope, void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(x && w && y && ws, "conv2d_fwd: null pointer");
  SRGAN_REQUIRE(ws_bytes >= srgan_conv2d_workspace(d), "conv2d_fwd: workspace too small");
  hipStream_t st = as_stream(stream);
  if (int e = fwd_pack(d, act, w, (float*)ws, st)) return e;
  return fwd_run(d, x, (const float*)ws, bias, y, act, slope, (float*)ws + round_up((long long)(pack_bytes(d) / sizeof(float)), 64), st);
}

extern "C" int srgan_conv2d_dgrad(const srgan_conv_desc* d, const float* dy, const float* w, float* dx,
                                  void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(dy && w && dx && ws, "conv2d_dgrad: null pointer");
  SRGAN_REQUIRE(ws_bytes >= srgan_conv2d_workspace(d), "conv2d_dgrad: workspace too small");
  hipStream_t st = as_stream(stream);
  DgradGeom g = dgrad_geometry(d);
  if (int e = dgrad_pack(d, w, (float*)ws, st)) return e;
  return dgrad_run(d, dy, (const float*)ws, dx, (float*)ws + round_up((long long)g.packed_elems, 64), st);
}

// ---- packed-weight variants: pack once per optimiser step, reuse across the step's forwards / backwards ----
extern "C" size_t srgan_conv2d_packed_bytes(const srgan_conv_desc* d, int kind, int act) {
  if (validate(d) != 0) return 0;
  return kind == 0 ? fwd_packed_bytes(d, act) : dgrad_geometry(d).packed_elems * sizeof(float);
}

extern "C" int srgan_conv2d_pack(const srgan_conv_desc* d, int kind, int act, const float* w, void* packed, size_t bytes,
                                 void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(w && packed, "conv2d_pack: null pointer");
  SRGAN_REQUIRE(kind == 0 || kind == 1, "conv2d_pack: kind must be 0 (forward) or 1 (input gradient)");
  SRGAN_REQUIRE(bytes >= srgan_conv2d_packed_bytes(d, kind, act), "conv2d_pack: destination too small");
  return kind == 0 ? fwd_pack(d, act, w, (float*)packed, as_stream(stream)) : dgrad_pack(d, w, (float*)packed, as_stream(stream));
}

// ---- one launch for many packs: the host collects one record per cached operand (srgan_conv2d_pack_entry), copies the
// array to the device and calls srgan_conv2d_pack_multi after the optimiser step that changed those weights ----
extern "C" size_t srgan_pack_entry_bytes(void) { return sizeof(PackEntry); }

// fills `entry` (host memory, srgan_pack_entry_bytes() bytes).  Returns 0, or 1 when this layer's operand is produced by a
// kernel outside the multi-pack launch (narrow-output layers): pack it with srgan_conv2d_pack.
extern "C" int srgan_conv2d_pack_entry(const srgan_conv_desc* d, int kind, int act, const float* w, void* packed, void* entry) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(w && packed && entry, "conv2d_pack_entry: null pointer");
  SRGAN_REQUIRE(kind == 0 || kind == 1, "conv2d_pack_entry: kind must be 0 (forward) or 1 (input gradient)");
  PackEntry pe{};
  if (kind == 0) {
    const FwdPath path = fwd_path(d, act);
    if (path == PATH_NARROW || path == PATH_ROWCONV || path == PATH_RGBIN) return 1;
    if (path == PATH_WINO) { pe.type = 1; wino_pack_params(d, 0, w, (float*)packed, &pe.wn); }
    else { pe.type = 0; pe.ig = fwd_pack_params(d, path, w, (float*)packed); }
  } else {
    const DgradGeom g = dgrad_geometry(d);
    if (g.narrow || g.rgbin || g.narrow_s2) return 1;
    if (g.wino) { pe.type = 1; wino_pack_params(d, 1, w, (float*)packed, &pe.wn); }
    else { pe.type = 0; pe.ig = dgrad_pack_params(d, g, w, (float*)packed); }
  }
  std::memcpy(entry, &pe, sizeof(pe));
  return 0;
}

extern "C" int srgan_conv2d_pack_multi(const void* entries_dev, int n_entries, void* stream) {
  SRGAN_REQUIRE(entries_dev && n_entries > 0, "conv2d_pack_multi: bad argument");
  hipLaunchKernelGGL(pack_multi_kernel, dim3(256, (unsigned)n_entries), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const PackEntry*>(entries_dev));
  return check_launch("pack_multi_kernel");
}

// Non-zero when the packed operand of (d, kind, act) does not depend on the batch size or the map size (the Winograd filter
// images and the RGB-input image depend on the weights and the channel counts only): a caller may then share one packed buffer
// between descriptors that differ in N / H / W only and return the same signature.  0: the layout follows the geometry.
// Round 6: the implicit-GEMM operands too.  Their bytes are a function of the weights and of PackParams' layout fields; two
// descriptors of one weight that differ in N / H / W only often agree in all of them (the discriminators see 64 images in their
// own update and 32 in the generator's, the encoder 32 and 64: every one of their operands was packed -- and re-packed after
// every optimiser step -- twice).  The signature is a hash of those fields (bit 62 set: never one of the small values above).
static unsigned long long pack_layout_signature(const PackParams& q, int kind) {
  static const bool off = SRGAN_AB_SET("SRGAN_NO_PACK_SHARE");
  if (off) return 0;
  unsigned long long h = 1469598103934665603ull;
  const long long f[] = {q.O, q.I, q.kh, q.kw, q.mode, q.stride, q.pad, q.Ty, q.Tx, q.Cs, q.N, q.K, q.Kpad, q.Npad, q.phases, q.out16,
                         q.regimg, kind};
  for (long long v : f)
    for (int b = 0; b < 8; ++b) {
      h ^= (unsigned long long)((v >> (8 * b)) & 0xff);
      h *= 1099511628211ull;
    }
  return (h >> 2) | (1ull << 62);
}

extern "C" unsigned long long srgan_conv2d_pack_signature(const srgan_conv_desc* d, int kind, int act) {
  if (validate(d) != 0) return 0;
  if (kind == 0) {
    const FwdPath path = fwd_path(d, act);
    if (path == PATH_RGBIN) return 100;
    if (path == PATH_WINO) return 10 + (unsigned long long)(wino_packed_bytes(d, 0) % 1000003) * 16 + 1;
    if (path == PATH_NARROW || path == PATH_ROWCONV) return 0;      // (their buffers carry per-geometry scratch behind the filter)
    return pack_layout_signature(fwd_pack_params(d, path, nullptr, nullptr), 0);
  }
  const DgradGeom g = dgrad_geometry(d);
  if (g.rgbin) return 101;
  if (g.wino) return 10 + (unsigned long long)(wino_packed_bytes(d, 1) % 1000003) * 16 + 2;
  if (g.narrow || g.narrow_s2) return 0;
  return pack_layout_signature(dgrad_pack_params(d, g, nullptr, nullptr), 1);
}

extern "C" size_t srgan_conv2d_packed_scratch(const srgan_conv_desc* d, int kind) {
  if (validate(d) != 0) return 0;
  if (kind == 1 && d->pad_mode == SRGAN_PAD_REFLECT) return srgan_conv2d_workspace(d);      // padded-gradient temp (+ split-K slabs)
  if (kind == 0 ? fwd_path(d, SRGAN_ACT_NONE) == PATH_WINO : dgrad_geometry(d).wino) return wino_scratch_bytes(d, kind);
  return conv_splitk_bytes(d, kind);      // implicit-GEMM layers with few pixel tiles: split-K slabs (0 for everything else)
}

extern "C" int srgan_conv2d_fwd_packed(const srgan_conv_desc* d, const float* x, const void* packed, const float* bias,
                                       float* y, int act, float slope, void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(x && packed && y, "conv2d_fwd_packed: null pointer");
  const size_t need = srgan_conv2d_packed_scratch(d, 0);
  SRGAN_REQUIRE(need == 0 || (ws && ws_bytes >= need), "conv2d_fwd_packed: workspace too small (srgan_conv2d_packed_scratch)");
  return fwd_run(d, x, (const float*)packed, bias, y, act, slope, (float*)ws, as_stream(stream));
}

extern "C" int srgan_conv2d_dgrad_packed(const srgan_conv_desc* d, const float* dy, const void* packed, float* dx,
                                         void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(dy && packed && dx, "conv2d_dgrad_packed: null pointer");
  const size_t need = srgan_conv2d_packed_scratch(d, 1);
  SRGAN_REQUIRE(need == 0 || (ws && ws_bytes >= need), "conv2d_dgrad_packed: workspace too small (srgan_conv2d_packed_scratch)");
  return dgrad_run(d, dy, (const float*)packed, dx, (float*)ws, as_stream(stream));
}

// ---- instance norm + activation feeding an F(4x4,3x3) layer without materialising the normalised tensor ----
extern "C" int srgan_instnorm_conv_v_applicable(const srgan_conv_desc* d) {
  if (validate(d) != 0) return 0;
  return d->Hi == 32 && d->Wi == 32 && d->I % 32 == 0 && d->kh == 3 && d->stride == 1 && d->pad == 1 &&
         d->pad_mode == SRGAN_PAD_ZERO && fwd_path(d, SRGAN_ACT_NONE) == PATH_WINO && wino43_fwd_applicable(d) ? 1 : 0;
}

extern "C" int srgan_instnorm_fwd_v(const srgan_conv_desc* d, const float* x, const float* scale, const float* shift, float* mean,
                                    float* rstd, void* v_image, size_t v_bytes, float eps, int act, float slope, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(x && mean && rstd && v_image, "instnorm_fwd_v: null pointer");
  SRGAN_REQUIRE((scale == nullptr) == (shift == nullptr), "instnorm_fwd_v: scale and shift go together");
  SRGAN_REQUIRE(srgan_instnorm_conv_v_applicable(d), "instnorm_fwd_v: layer not applicable (32x32 map into an F(4x4,3x3) layer)");
  SRGAN_REQUIRE(v_bytes >= srgan_conv2d_packed_scratch(d, 0), "instnorm_fwd_v: V image too small (srgan_conv2d_packed_scratch)");
  return in_fwd_slab_v_launch(x, scale, shift, mean, rstd, static_cast<float*>(v_image), d->N, d->I, eps, act, slope, as_stream(stream));
}

extern "C" int srgan_conv2d_fwd_from_v(const srgan_conv_desc* d, const void* v_image, const void* packed, const float* bias,
                                       float* y, int act, float slope, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(v_image && packed && y, "conv2d_fwd_from_v: null pointer");
  SRGAN_REQUIRE(fwd_path(d, act) == PATH_WINO && wino43_fwd_applicable(d), "conv2d_fwd_from_v: layer does not run on F(4x4,3x3)");
  return wino_run(d, 0, nullptr, (const float*)packed, bias, y, act, slope, const_cast<float*>(static_cast<const float*>(v_image)),
                  as_stream(stream), nullptr, nullptr, true);
}

// ---- instance-norm backward feeding the F(4x4,3x3) input-gradient and weight-gradient kernels of the preceding convolution ----
// d describes that convolution (its OUTPUT is what the norm normalises).
extern "C" int srgan_instnorm_bwd_vz_applicable(const srgan_conv_desc* d) {
  if (validate(d) != 0) return 0;
  Wino43WgradGeom g{};
  const DgradGeom dg = dgrad_geometry(d);
  return d->Ho == 32 && d->Wo == 32 && d->Hi == 32 && d->Wi == 32 && d->O % 32 == 0 && d->kh == 3 && d->stride == 1 && d->pad == 1 &&
         d->pad_mode == SRGAN_PAD_ZERO && dg.wino && !dg.reflect && wino43_dgrad_applicable(d) && wino43_wgrad_geometry(d, &g) ? 1 : 0;
}

extern "C" size_t srgan_instnorm_bwd_vz_z_bytes(const srgan_conv_desc* d) {
  if (validate(d) != 0) return 0;
  Wino43WgradGeom g{};
  return wino43_wgrad_geometry(d, &g) ? g.z_bytes : 0;
}

extern "C" int srgan_instnorm_bwd_vz(const srgan_conv_desc* d, const float* x, const float* dy, const float* scale, const float* shift,
                                     const float* mean, const float* rstd, float* dscale, float* dshift, void* v_image, size_t v_bytes,
                                     void* z_image, size_t z_bytes, int act, float slope, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(x && dy && mean && rstd && dscale && dshift && v_image && z_image, "instnorm_bwd_vz: null pointer");
  SRGAN_REQUIRE((scale == nullptr) == (shift == nullptr), "instnorm_bwd_vz: scale and shift go together");
  SRGAN_REQUIRE(srgan_instnorm_bwd_vz_applicable(d), "instnorm_bwd_vz: layer not applicable");
  SRGAN_REQUIRE(v_bytes >= srgan_conv2d_packed_scratch(d, 1) && z_bytes >= srgan_instnorm_bwd_vz_z_bytes(d),
                "instnorm_bwd_vz: image buffers too small (srgan_conv2d_packed_scratch(d, 1), srgan_instnorm_bwd_vz_z_bytes)");
  return in_bwd_slab_vz_launch(x, dy, scale, shift, mean, rstd, dscale, dshift, static_cast<float*>(v_image),
                               static_cast<float*>(z_image), d->N, d->O, act, slope, as_stream(stream));
}

extern "C" int srgan_conv2d_dgrad_from_v(const srgan_conv_desc* d, const void* v_image, const void* packed, const float* res,
                                         float* dx, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(v_image && packed && dx, "conv2d_dgrad_from_v: null pointer");
  SRGAN_REQUIRE(res != dx, "conv2d_dgrad_from_v: res must not alias dx");
  const DgradGeom dg = dgrad_geometry(d);
  SRGAN_REQUIRE(dg.wino && !dg.reflect && wino43_dgrad_applicable(d), "conv2d_dgrad_from_v: the input gradient does not run on F(4x4,3x3)");
  return wino_run(d, 1, nullptr, (const float*)packed, nullptr, dx, SRGAN_ACT_NONE, 0.f,
                  const_cast<float*>(static_cast<const float*>(v_image)), as_stream(stream), res, nullptr, true);
}

extern "C" int srgan_conv2d_dgrad_packed_mask(const srgan_conv_desc* d, const float* dy, const void* packed, const float* mask,
                                              float slope, float* dx, void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(dy && packed && dx && mask, "conv2d_dgrad_packed_mask: null pointer");
  SRGAN_REQUIRE(mask != dx, "conv2d_dgrad_packed_mask: mask must not alias dx");
  const size_t need = srgan_conv2d_packed_scratch(d, 1);
  SRGAN_REQUIRE(need == 0 || (ws && ws_bytes >= need), "conv2d_dgrad_packed_mask: workspace too small (srgan_conv2d_packed_scratch)");
  return dgrad_run(d, dy, (const float*)packed, dx, (float*)ws, as_stream(stream), nullptr, mask, slope);
}

extern "C" int srgan_conv2d_dgrad_packed_add(const srgan_conv_desc* d, const float* dy, const void* packed, const float* res,
                                             float* dx, void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(dy && packed && dx && res, "conv2d_dgrad_packed_add: null pointer");
  SRGAN_REQUIRE(res != dx, "conv2d_dgrad_packed_add: res must not alias dx");
  const size_t need = srgan_conv2d_packed_scratch(d, 1);
  SRGAN_REQUIRE(need == 0 || (ws && ws_bytes >= need), "conv2d_dgrad_packed_add: workspace too small (srgan_conv2d_packed_scratch)");
  return dgrad_run(d, dy, (const float*)packed, dx, (float*)ws, as_stream(stream), res);
}

namespace srgan {
static int finish_wgrad(const srgan_conv_desc* d, const WgradPlan& w, const float* dy, float* dw, float* dbias, void* ws,
                        hipStream_t st);
static int defer_take(const srgan_conv_desc* d, void** ws, size_t* ws_bytes, hipStream_t st);
}

// Weight gradient of a layer whose forward kept its F(4x4,3x3) V image (srgan_conv2d_wgrad_v_bytes(d) != 0 bytes, handed to
// srgan_conv2d_fwd_packed as `ws` and kept alive by the caller instead of the shared scratch).
extern "C" size_t srgan_conv2d_wgrad_v_bytes(const srgan_conv_desc* d) {
  if (validate(d) != 0) return 0;
  Wino43WgradGeom g{};
  return wino43_wgrad_geometry(d, &g) ? g.v_bytes : 0;
}

extern "C" int srgan_conv2d_wgrad_v(const srgan_conv_desc* d, const float* v_image, const float* dy, float* dw, float* dbias,
                                    void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(v_image && dy && dw && ws, "conv2d_wgrad_v: null pointer");
  SRGAN_REQUIRE(ws_bytes >= srgan_conv2d_workspace(d), "conv2d_wgrad_v: workspace too small");
  Wino43WgradGeom g{};
  SRGAN_REQUIRE(wino43_wgrad_geometry(d, &g), "conv2d_wgrad_v: layer not applicable");
  hipStream_t st = as_stream(stream);
  if (int e = defer_take(d, &ws, &ws_bytes, st)) return e;
  const size_t cs = (size_t)1024 * d->O * sizeof(float);
  float* zimg = reinterpret_cast<float*>(static_cast<char*>(ws) + round_up((long long)(g.slab_bytes + cs), 256));
  if (int e = wino43_wgrad_launch(g, v_image, dy, zimg, (float*)ws, conv_flops(d), st)) return e;
  if (int e = check_launch("wino43_wgrad_kernel")) return e;
  WgradPlan w = plan_wgrad(d);
  w.splits = g.splits; w.Cdpad = d->O; w.NNpad = 9 * d->I;
  return finish_wgrad(d, w, dy, dw, dbias, ws, st);
}

// The same with the Z image (A dy A^T) already written by srgan_instnorm_bwd_vz.
extern "C" int srgan_conv2d_wgrad_vz(const srgan_conv_desc* d, const float* v_image, const float* z_image, float* dw, void* ws,
                                     size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(v_image && z_image && dw && ws, "conv2d_wgrad_vz: null pointer");
  SRGAN_REQUIRE(ws_bytes >= srgan_conv2d_workspace(d), "conv2d_wgrad_vz: workspace too small");
  Wino43WgradGeom g{};
  SRGAN_REQUIRE(wino43_wgrad_geometry(d, &g), "conv2d_wgrad_vz: layer not applicable");
  hipStream_t st = as_stream(stream);
  if (int e = defer_take(d, &ws, &ws_bytes, st)) return e;
  if (int e = wino43_wgrad_launch(g, v_image, nullptr, const_cast<float*>(z_image), (float*)ws, conv_flops(d), st, true)) return e;
  if (int e = check_launch("wino43_wgrad_kernel")) return e;
  WgradPlan w = plan_wgrad(d);
  w.splits = g.splits; w.Cdpad = d->O; w.NNpad = 9 * d->I;
  return finish_wgrad(d, w, nullptr, dw, nullptr, ws, st);
}

// ---- bf16 mode, residual-trunk shapes (conv_halo16.hip) with bf16 TENSORS on either side: the convolution kernels of the fused
// residual block (srgan_amd.ops._ResBlockBf16Fn), whose intermediates (conv outputs, normalised activation, the gradients between
// the norm and conv backward kernels) live in HBM as bf16.  `packed`: the ordinary packed operand of (d, kind) in bf16 mode
// (srgan_conv2d_pack / the pack cache).
extern "C" int srgan_halo16_applicable(const srgan_conv_desc* d) {
  if (validate(d) != 0) return 0;
  return compute_bf16() && halo16_applicable(d, 0) && halo16_wgrad_applicable(d) && wino_applicable(d, 0) && wino_applicable(d, 1) ? 1 : 0;
}

// 4x4 / stride-2 / pad-1 layers whose forward (halo16s_kernel), input gradient (halo16t_kernel) and weight gradient
// (halo16s2_wgrad_kernel) all run on the LDS-resident-patch kernels in the bf16 mode: these take and write bf16 tensors
// (srgan_halo16_conv / srgan_halo16_wgrad with the *_bf16 flags) -- the generator's down / up convolutions.
extern "C" int srgan_halo16s2_applicable(const srgan_conv_desc* d) {
  if (validate(d) != 0) return 0;
  return compute_bf16() && halo16s_applicable(d) && halo16t_applicable(d) && halo16_wgrad_applicable(d) && wino_applicable(d, 0) &&
         wino_applicable(d, 1) ? 1 : 0;
}

extern "C" int srgan_halo16_conv(const srgan_conv_desc* d, int kind, const void* src, int src_bf16, const void* packed,
                                 const float* res, void* dst, int dst_bf16, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(src && packed && dst, "halo16_conv: null pointer");
  SRGAN_REQUIRE(kind == 0 || kind == 1, "halo16_conv: kind must be 0 (forward) or 1 (input gradient)");
  if (srgan_halo16s2_applicable(d)) {
    SRGAN_REQUIRE(!res, "halo16_conv: no skip tensor on the stride-2 layers");
    if (kind == 0) return halo16s_run(d, src, packed, nullptr, dst, SRGAN_ACT_NONE, 0.f, conv_flops(d), as_stream(stream), src_bf16 != 0, dst_bf16 != 0);
    return halo16t_run(d, src, packed, dst, conv_flops(d), as_stream(stream), src_bf16 != 0, dst_bf16 != 0);
  }
  SRGAN_REQUIRE(srgan_halo16_applicable(d), "halo16_conv: layer / compute mode not applicable (srgan_halo16_applicable / srgan_halo16s2_applicable)");
  return halo16_run(d, kind, src, packed, nullptr, res, dst, SRGAN_ACT_NONE, 0.f, conv_flops(d), as_stream(stream), src_bf16 != 0,
                    dst_bf16 != 0);
}

extern "C" int srgan_halo16_wgrad(const srgan_conv_desc* d, const void* x, int x_bf16, const void* dy, int dy_bf16, float* dw,
                                  void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(x && dy && dw && ws, "halo16_wgrad: null pointer");
  // (round 6: also the stride-2 layers no patch kernel serves forward -- the discriminators' 16- and 8-pixel maps,
  //  srgan_conv2d_io_applicable -- whose weight gradient halo16s2_wgrad_kernel takes all the same)
  if (compute_bf16() && rgb_wgrad16_served(d)) {      // round 6: the 7x7 RGB layers, 64-channel side fp32 or bf16
    const int kind = rgb_wgrad_kind(d);
    SRGAN_REQUIRE(kind == 0 ? !x_bf16 : !dy_bf16, "halo16_wgrad: the 3-channel tensor of an RGB layer is fp32");
    SRGAN_REQUIRE(ws_bytes >= srgan_conv2d_workspace(d), "halo16_wgrad: workspace too small (srgan_conv2d_workspace)");
    hipStream_t st = as_stream(stream);
    if (int e = defer_take(d, &ws, &ws_bytes, st)) return e;
    WgradPlan w = plan_wgrad(d);
    rgb_wgrad_slab(d, &w.splits, &w.Cdpad, &w.NNpad);
    if (int e = rgb_wgrad_run(d, x, dy, (float*)ws, st, (kind == 0 ? dy_bf16 : x_bf16) != 0)) return e;
    return finish_wgrad(d, w, nullptr, dw, nullptr, ws, st);
  }
  SRGAN_REQUIRE(compute_bf16() && halo16_wgrad_applicable(d),
                "halo16_wgrad: layer / compute mode not applicable (srgan_halo16_applicable / srgan_halo16s2_applicable / srgan_conv2d_io_applicable)");
  SRGAN_REQUIRE(ws_bytes >= srgan_conv2d_workspace(d), "halo16_wgrad: workspace too small (srgan_conv2d_workspace)");
  hipStream_t st = as_stream(stream);
  if (int e = defer_take(d, &ws, &ws_bytes, st)) return e;
  WgradPlan w = plan_wgrad(d);
  halo16_wgrad_slab(d, &w.splits, &w.Cdpad, &w.NNpad);
  if (int e = halo16_wgrad_run(d, x, dy, (float*)ws, conv_flops(d), st, x_bf16 != 0, dy_bf16 != 0)) return e;
  return finish_wgrad(d, w, nullptr, dw, nullptr, ws, st);
}

namespace srgan {
static int launch_wgrad(const WgradParams& p, const WgradPlan& w, hipStream_t st, bool io16 = false) {
  WgradVariant k{};
  if (!wgrad_lookup(w.BMc, w.BNn, w.vec, w.rows, &k, io16)) { set_error("wgrad: unsupported tile %dx%d", w.BMc, w.BNn); return -1; }
  ProfScope scope(8 + (w.vec ? 1 : 0), 2.0 * p.M * (double)p.Cd * p.NN, st);
  dim3 grid((unsigned)(w.co_tiles * w.nn_tiles * w.splits), 1, 1);
  hipLaunchKernelGGL(k.fn, grid, dim3(k.threads), 0, st, p);
  return check_launch("wgrad_kernel");
}
}  // namespace srgan

namespace srgan {
// srgan_set_wgrad_accumulate: while on, the weight-gradient entry points of this host thread ADD to dw / dbias
static thread_local int g_wgrad_accumulate = 0;
// bit 1 of the same switch: nobody reads dw before srgan_wgrad_defer_end (the caller's gradient sink) -- the call may be deferred
static thread_local int g_wgrad_deferrable = 0;

// srgan_wgrad_defer_begin .. _end (process-wide: autograd runs the backward functions on its own device thread, not on the thread
// that opened the scope; one backward pass at a time, g_defer_mutex only keeps a misuse from corrupting the queue):
// deferrable weight-gradient calls take their workspace (slab first) from the arena
// instead of the caller's shared scratch and queue their slab sum; the queue is launched as wgrad_reduce_multi_kernel when it
// is full, when the arena is, when a second record for the same dw arrives (stream order then keeps the two sums in call
// order), and at _end.
struct WgradDefer {
  bool active = false;
  char* arena = nullptr;
  size_t bytes = 0, used = 0;
  size_t asked = 0;         // workspace bytes the deferrable calls of the scope asked for (srgan_wgrad_defer_need)
  hipStream_t st = nullptr;
  std::vector<MultiReduceEntry> pending;
};
static WgradDefer g_defer;
static std::mutex g_defer_mutex;
static long long g_defer_sums = 0, g_defer_launches = 0;      // srgan_wgrad_defer_stats
static thread_local bool g_call_deferred = false;

static bool reduce_by_wave(const WgradReduceParams& r) {
  return r.splits >= 64 && (long long)r.O * r.kh * r.kw * r.I <= 131072;
}
static long long reduce_row_blocks_host(const WgradReduceParams& r) { return (long long)r.O * ceil_div(r.I, kReduceIC); }
static long long reduce_col_blocks_host(const WgradReduceParams& r) { return (long long)r.O * ceil_div((long long)r.kh * r.kw * r.I, 64); }

static int defer_launch_pending(bool reset_arena) {
  WgradDefer& q = g_defer;
  if (!q.pending.empty()) {
    MultiReduceArgs a{};
    a.n = (int)q.pending.size();
    long long blocks = 0;
    for (int k = 0; k < a.n; ++k) {
      a.e[k] = q.pending[k];
      a.e[k].first_block = (int)blocks;
      blocks += a.e[k].wave ? reduce_col_blocks_host(a.e[k].r) : reduce_row_blocks_host(a.e[k].r);
    }
    q.pending.clear();
    static const bool dbg = std::getenv("SRGAN_DEBUG_REDUCE") != nullptr;
    if (dbg) {
      double bytes = 0;
      for (int k = 0; k < a.n; ++k) {
        const WgradReduceParams& r = a.e[k].r;
        bytes += 4.0 * r.O * r.kh * r.kw * r.I * (r.splits + 1 + r.accumulate);
        std::fprintf(stderr, "  sum O %d I %d k %dx%d splits %d Cdpad %d NNpad %d wave %d acc %d\n", r.O, r.I, r.kh, r.kw, r.splits, r.Cdpad,
                     r.NNpad, a.e[k].wave, r.accumulate);
      }
      std::fprintf(stderr, "multi launch: %d sums, %lld blocks, %.1f MB\n", a.n, blocks, bytes / 1e6);
    }
    g_defer_sums += a.n;
    ++g_defer_launches;
    SRGAN_REQUIRE(blocks < (1LL << 31), "wgrad defer: grid too large");
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, q.st, a);
    if (int e = check_launch("wgrad_reduce_multi_kernel")) return e;
  }
  if (reset_arena) q.used = 0;
  return 0;
}

// Called first by every weight-gradient entry point: when the call can be deferred, *ws / *ws_bytes become a fresh piece of the arena
static int defer_take(const srgan_conv_desc* d, void** ws, size_t* ws_bytes, hipStream_t st) {
  g_call_deferred = false;
  if (!g_wgrad_deferrable) return 0;
  std::lock_guard<std::mutex> lock(g_defer_mutex);
  WgradDefer& q = g_defer;
  if (!q.active) return 0;
  if (st != q.st) return 0;          // a call on another stream (the discriminator's second scale): immediate sum, own scratch
  const size_t need = (size_t)round_up((long long)srgan_conv2d_workspace(d), 256);
  q.asked += need;
  if (need == 0 || need > q.bytes) return 0;
  if (q.used + need > q.bytes)
    if (int e = defer_launch_pending(true)) return e;
  *ws = q.arena + q.used;
  *ws_bytes = need;
  q.used += need;
  g_call_deferred = true;
  return 0;
}

// slab sum -> dW (through the weight strides) and optional bias column sums
static int finish_wgrad(const srgan_conv_desc* d, const WgradPlan& w, const float* dy, float* dw, float* dbias, void* ws,
                        hipStream_t st) {
  WgradReduceParams r{};
  r.slab = (const float*)ws; r.dw = dw; r.sO = d->sO; r.sI = d->sI; r.sH = d->sH; r.sW = d->sW;
  r.O = d->O; r.I = d->I; r.kh = d->kh; r.kw = d->kw; r.splits = w.splits; r.Cdpad = w.Cdpad; r.NNpad = w.NNpad;
  r.accumulate = g_wgrad_accumulate;
  long long total = (long long)d->O * d->kh * d->kw * d->I;
  if (g_call_deferred) {
    g_call_deferred = false;
    std::lock_guard<std::mutex> lock(g_defer_mutex);
    WgradDefer& q = g_defer;
    bool again = false;
    for (const MultiReduceEntry& e : q.pending) again = again || e.r.dw == dw;
    if (again || (int)q.pending.size() == kMultiReduceMax)
      if (int e = defer_launch_pending(false)) return e;
    MultiReduceEntry me{};
    me.r = r;
    me.wave = reduce_by_wave(r) ? 1 : 0;
    q.pending.push_back(me);
  } else if (reduce_by_wave(r))
    hipLaunchKernelGGL(wgrad_reduce_wave_kernel, dim3((unsigned)reduce_col_blocks_host(r)), dim3(256), 0, st, r);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)reduce_row_blocks_host(r)), dim3(256), 0, st, r);
  if (int e2 = check_launch("wgrad_reduce_kernel")) return e2;
  (void)total;
  if (dbias) {
    const int M = d->N * d->Ho * d->Wo;
    if (d->O <= 4 && M <= (1 << 16)) {
      hipLaunchKernelGGL(colsum_narrow_kernel, dim3(1), dim3(256), 0, st, dy, dbias, M, d->O, g_wgrad_accumulate);
      return check_launch("colsum_narrow_kernel");
    }
    float* part = (float*)ws + (size_t)w.splits * w.Cdpad * w.NNpad;
    int nparts = (int)std::min<long long>(1024, ceil_div(M, 64));
    int rpb = (int)ceil_div(M, nparts);
    nparts = (int)ceil_div(M, rpb);
    dim3 g1((unsigned)ceil_div(d->O, 64), (unsigned)nparts);
    hipLaunchKernelGGL(colsum_partial_kernel, g1, dim3(64), 0, st, dy, part, M, d->O, rpb);
    if (int e3 = check_launch("colsum_partial_kernel")) return e3;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)ceil_div(d->O, 4)), dim3(256), 0, st, (const float*)part, dbias, d->O, nparts,
                       g_wgrad_accumulate);
    return check_launch("colsum_final_kernel");
  }
  return 0;
}
}  // namespace srgan

namespace srgan {
// the implicit-GEMM weight gradient (wgrad_kernel + slab sum); io16: x and dy are bf16 tensors (igemm16_io_ok layers)
static int generic_wgrad(const srgan_conv_desc* d, const WgradPlan& w, const float* x, const float* dy, float* dw, float* dbias,
                         void* ws, hipStream_t st, bool io16) {
  WgradParams p{};
  p.x = x; p.dy = dy; p.slab = (float*)ws;
  p.NB = d->N; p.Hi = d->Hi; p.Wi = d->Wi; p.Cs = d->I; p.Hg = d->Ho; p.Wg = d->Wo; p.Cd = d->O;
  p.stride = d->stride; p.pad = d->pad; p.kw = d->kw; p.reflect = d->pad_mode == SRGAN_PAD_REFLECT;
  p.NN = d->kh * d->kw * d->I; p.NNpad = w.NNpad; p.Cdpad = w.Cdpad;
  p.M = d->N * d->Ho * d->Wo; p.rows_per_split = w.rows_per_split;
  p.co_tiles = w.co_tiles; p.nn_tiles = w.nn_tiles;
  if (int e = launch_wgrad(p, w, st, io16)) return e;
  return finish_wgrad(d, w, io16 ? nullptr : dy, dw, dbias, ws, st);
}

// every direction of the layer on the generic bf16 kernels, with the vector paths their 16-bit loads need
static bool igemm16_io_ok(const srgan_conv_desc* d, int act) {
  if (!compute_bf16() || (d->I & 3) || (d->O & 3)) return false;
  if (fwd_path(d, act) != PATH_IGEMM) return false;
  IgemmParams p{};
  fwd_geometry(d, PATH_IGEMM, p);
  if (!igemm16_ok(p)) return false;
  const DgradGeom g = dgrad_geometry(d);
  if (g.wino || g.narrow || g.rgbin || g.narrow_s2 || !igemm16_ok(g.p)) return false;
  if (wino_wgrad_applicable(d) || rgb_wgrad_kind(d) >= 0 || narrow_applicable(d)) return false;
  const WgradPlan w = plan_wgrad(d);
  return w.vec && !w.rows && w.BMc >= 64 && w.BNn >= 64;
}
}  // namespace srgan

extern "C" int srgan_conv2d_wgrad(const srgan_conv_desc* d, const float* x, const float* dy, float* dw,
                                  float* dbias, void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(x && dy && dw && ws, "conv2d_wgrad: null pointer");
  SRGAN_REQUIRE(ws_bytes >= srgan_conv2d_workspace(d), "conv2d_wgrad: workspace too small");
  hipStream_t st = as_stream(stream);
  if (int e = defer_take(d, &ws, &ws_bytes, st)) return e;
  WgradPlan w = plan_wgrad(d);
  if (wino_wgrad_applicable(d)) {
    wino_wgrad_slab(d, &w.splits, &w.Cdpad, &w.NNpad);
    if (int e = wino_wgrad_run(d, x, dy, (float*)ws, st)) return e;
    return finish_wgrad(d, w, dy, dw, dbias, ws, st);
  }
  if (rgb_wgrad_kind(d) >= 0) {
    rgb_wgrad_slab(d, &w.splits, &w.Cdpad, &w.NNpad);
    if (int e = rgb_wgrad_run(d, x, dy, (float*)ws, st)) return e;
    return finish_wgrad(d, w, dy, dw, dbias, ws, st);
  }
  if (narrow_applicable(d)) {
    int n_slabs = 0;
    if (int e = narrow_wgrad(d, x, dy, ws, &n_slabs, st)) return e;
    w.splits = n_slabs; w.Cdpad = 4; w.NNpad = d->kh * d->kw * d->I;
    return finish_wgrad(d, w, dy, dw, dbias, ws, st);
  }
  return generic_wgrad(d, w, x, dy, dw, dbias, ws, st, false);
}

// ---- bf16 mode, the generic layers (igemm16_kernel, wgrad_kernel<BF>) with bf16 TENSORS on either side: the 3x3 reflect-padded
// convolutions of the style encoder's blocks (pyfiles/model.py:413-437), whose normalised inputs, outputs and the gradients of
// both live in HBM as bf16 (srgan_amd.ops._ConvIoFn).  `packed`: the ordinary packed operand of (d, kind, act) in bf16 mode.
extern "C" int srgan_igemm16_io_applicable(const srgan_conv_desc* d, int act) {
  if (validate(d) != 0) return 0;
  return igemm16_io_ok(d, act) ? 1 : 0;
}

extern "C" int srgan_igemm16_conv(const srgan_conv_desc* d, int kind, const void* src, int src_bf16, const void* packed,
                                  const float* bias, void* dst, int dst_bf16, int act, float slope, void* ws, size_t ws_bytes,
                                  void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(src && packed && dst, "igemm16_conv: null pointer");
  SRGAN_REQUIRE(kind == 0 || kind == 1, "igemm16_conv: kind must be 0 (forward) or 1 (input gradient)");
  SRGAN_REQUIRE(kind == 0 || (!bias && act == SRGAN_ACT_NONE), "igemm16_conv: bias / activation belong to the forward");
  SRGAN_REQUIRE(igemm16_io_ok(d, kind == 0 ? act : SRGAN_ACT_NONE), "igemm16_conv: layer / compute mode not applicable (srgan_igemm16_io_applicable)");
  const size_t need = srgan_conv2d_packed_scratch(d, kind);
  SRGAN_REQUIRE(need == 0 || (ws && ws_bytes >= need), "igemm16_conv: workspace too small (srgan_conv2d_packed_scratch)");
  hipStream_t st = as_stream(stream);
  if (kind == 0) {
    IgemmParams p{};
    fwd_geometry(d, PATH_IGEMM, p);
    p.src = static_cast<const float*>(src); p.bias = bias; p.dst = static_cast<float*>(dst); p.act = act; p.slope = slope;
    p.wp = static_cast<const float*>(packed); p.src16 = src_bf16 != 0; p.dst16 = dst_bf16 != 0;
    return run_igemm(p, 1, st, conv_flops(d), need ? static_cast<float*>(ws) : nullptr);
  }
  DgradGeom g = dgrad_geometry(d);
  float* scratch = need ? static_cast<float*>(ws) : nullptr;
  g.p.src = static_cast<const float*>(src); g.p.wp = static_cast<const float*>(packed); g.p.src16 = src_bf16 != 0;
  // reflect: the gradient with respect to the padded image stays an fp32 temp; the fold writes the tensor's type
  g.p.dst = g.reflect ? scratch : static_cast<float*>(dst);
  g.p.dst16 = g.reflect ? 0 : (dst_bf16 != 0);
  float* slab = scratch ? (g.reflect ? scratch + round_up((long long)d->N * g.Hd * g.Wd * d->I, 64) : scratch) : nullptr;
  if (int e = run_igemm(g.p, g.phases, st, conv_flops(d), slab)) return e;
  if (g.reflect) {
    const long long n = (long long)d->N * d->Hi * d->Wi * d->I;
    const dim3 fg((unsigned)std::min<long long>(ceil_div(n, 256), 8192));
    if (dst_bf16)
      hipLaunchKernelGGL(reflect_fold_kernel<true>, fg, dim3(256), 0, st, (const float*)scratch, static_cast<float*>(dst), d->N, d->Hi,
                         d->Wi, d->I, d->pad);
    else
      hipLaunchKernelGGL(reflect_fold_kernel<false>, fg, dim3(256), 0, st, (const float*)scratch, static_cast<float*>(dst), d->N, d->Hi,
                         d->Wi, d->I, d->pad);
    return check_launch("reflect_fold_kernel");
  }
  return 0;
}

// x, dy: bf16 tensors.  Same workspace, gradient sink and deferred slab sum as srgan_conv2d_wgrad.
extern "C" int srgan_igemm16_wgrad(const srgan_conv_desc* d, const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(x && dy && dw && ws, "igemm16_wgrad: null pointer");
  SRGAN_REQUIRE(igemm16_io_ok(d, SRGAN_ACT_NONE), "igemm16_wgrad: layer / compute mode not applicable (srgan_igemm16_io_applicable)");
  SRGAN_REQUIRE(ws_bytes >= srgan_conv2d_workspace(d), "igemm16_wgrad: workspace too small (srgan_conv2d_workspace)");
  hipStream_t st = as_stream(stream);
  if (int e = defer_take(d, &ws, &ws_bytes, st)) return e;
  const WgradPlan w = plan_wgrad(d);
  return generic_wgrad(d, w, static_cast<const float*>(x), static_cast<const float*>(dy), dw, nullptr, ws, st, true);
}

// ---- bf16 mode, round 6: 16-bit activations in the discriminator trunks (VERDICT r5 item 1 iii) ----
// SingleDiscriminator_solo (pyfiles/model.py:294-316) is [conv 4x4 / stride 2 / pad 1, no bias -> LeakyReLU(0.01)] x num_cls with
// no norm between: the tensors between the layers are written by one convolution's epilogue and read by the next one's gather.
// These entry points run such a layer with bf16 tensors on either side on whatever kernel serves the direction in the bf16 mode
// -- forward: halo16s_kernel or igemm16_kernel (bias / activation in the epilogue, LDS-DMA tiles from a bf16 source); input
// gradient: halo16t_kernel or igemm16_kernel; weight gradient: srgan_halo16_wgrad (halo16s2_wgrad_kernel takes bf16 x / dy).
// `packed`: the ordinary packed operand of (d, kind, act) in the bf16 mode (srgan_conv2d_pack / the pack cache).
namespace srgan {
// The 7x7 / stride-1 / pad-3 layers between a 3-channel and a 64-channel tensor (the generator's RGB input and output layers,
// pyfiles/model.py:212, 232) with the 64-channel side in bf16: 3 = input layer (forward rgbin16 writes bf16, input gradient =
// rgbout16 form reads bf16 dy, weight gradient takes bf16 dy), 4 = output layer (forward rgbout16 reads bf16, input gradient =
// rgbin16 form writes bf16, weight gradient takes bf16 x).  The 3-channel side is always fp32.
static int conv_io_rgb(const srgan_conv_desc* d, int act) {
  if (!compute_bf16() || act != SRGAN_ACT_NONE || !rgb_wgrad16_served(d)) return 0;
  const DgradGeom g = dgrad_geometry(d);
  srgan_conv_desc f;
  long long w_off;
  if (d->I == 3) {
    if (fwd_path(d, act) != PATH_RGBIN || !rgbin16_served(d)) return 0;
    if (!g.narrow || !narrow_dgrad_desc(d, &f, &w_off) || !rowconv_applicable(&f) || !rgbout16_served(&f)) return 0;
    return 3;
  }
  if (fwd_path(d, act) != PATH_ROWCONV || !rgbout16_served(d)) return 0;
  if (!g.rgbin || !rgbin_dgrad_desc(d, &f, &w_off) || !rgbin16_served(&f)) return 0;
  return 4;
}

static bool conv_io_dirs(const srgan_conv_desc* d, int act, int* fwd, int* bwd) {      // 1: patch kernel, 2: igemm16_kernel
  if (const int r = conv_io_rgb(d, act)) { *fwd = *bwd = r; return true; }
  if (!compute_bf16() || d->kh != 4 || d->kw != 4 || d->stride != 2 || d->pad != 1 || d->pad_mode != SRGAN_PAD_ZERO) return false;
  if ((d->I & 7) || (d->O & 7)) return false;
  const FwdPath path = fwd_path(d, act);
  if (path == PATH_WINO && halo16s_applicable(d)) *fwd = 1;
  else if (path == PATH_IGEMM) {
    IgemmParams p{};
    fwd_geometry(d, PATH_IGEMM, p);
    if (!igemm16_ok(p)) return false;
    *fwd = 2;
  } else return false;
  const DgradGeom g = dgrad_geometry(d);
  if (g.wino && halo16t_applicable(d)) *bwd = 1;
  else if (!g.wino && !g.narrow && !g.rgbin && !g.narrow_s2 && !g.reflect && igemm16_ok(g.p)) *bwd = 2;
  else return false;
  return halo16_wgrad_applicable(d);
}
}  // namespace srgan

extern "C" int srgan_conv2d_io_applicable(const srgan_conv_desc* d, int act) {
  if (validate(d) != 0) return 0;
  int f = 0, b = 0;
  return conv_io_dirs(d, act, &f, &b) ? 1 : 0;
}

extern "C" int srgan_conv2d_io_fwd(const srgan_conv_desc* d, const void* x, int x_bf16, const void* packed, const float* bias, void* y,
                                   int y_bf16, int act, float slope, void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(x && packed && y, "conv2d_io_fwd: null pointer");
  int f = 0, b = 0;
  SRGAN_REQUIRE(conv_io_dirs(d, act, &f, &b), "conv2d_io_fwd: layer / compute mode not applicable (srgan_conv2d_io_applicable)");
  hipStream_t st = as_stream(stream);
  if (f == 3) {
    SRGAN_REQUIRE(!x_bf16, "conv2d_io_fwd: the 3-channel input of the RGB input layer is fp32");
    return rgbin_run(d, static_cast<const float*>(x), static_cast<const float*>(packed), bias, static_cast<float*>(y), act, slope, st, y_bf16 != 0);
  }
  if (f == 4) {
    SRGAN_REQUIRE(!y_bf16, "conv2d_io_fwd: the 3-channel result of the RGB output layer is fp32");
    return rgbout_run(d, x, static_cast<const float*>(packed), bias, static_cast<float*>(y), st, x_bf16 != 0);
  }
  if (f == 1) return halo16s_run(d, x, packed, bias, y, act, slope, conv_flops(d), st, x_bf16 != 0, y_bf16 != 0);
  const size_t need = srgan_conv2d_packed_scratch(d, 0);
  SRGAN_REQUIRE(need == 0 || (ws && ws_bytes >= need), "conv2d_io_fwd: workspace too small (srgan_conv2d_packed_scratch)");
  IgemmParams p{};
  fwd_geometry(d, PATH_IGEMM, p);
  p.src = static_cast<const float*>(x); p.bias = bias; p.dst = static_cast<float*>(y); p.act = act; p.slope = slope;
  p.wp = static_cast<const float*>(packed); p.src16 = x_bf16 != 0; p.dst16 = y_bf16 != 0;
  return run_igemm(p, 1, st, conv_flops(d), need ? static_cast<float*>(ws) : nullptr);
}

extern "C" int srgan_conv2d_io_dgrad(const srgan_conv_desc* d, const void* dy, int dy_bf16, const void* packed, void* dx, int dx_bf16,
                                     void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SRGAN_REQUIRE(dy && packed && dx, "conv2d_io_dgrad: null pointer");
  int f = 0, b = 0;
  SRGAN_REQUIRE(conv_io_dirs(d, SRGAN_ACT_NONE, &f, &b) || conv_io_dirs(d, SRGAN_ACT_LRELU, &f, &b),
                "conv2d_io_dgrad: layer / compute mode not applicable (srgan_conv2d_io_applicable)");
  hipStream_t st = as_stream(stream);
  if (b == 3 || b == 4) {
    srgan_conv_desc f;
    long long w_off;
    if (b == 3) {          // the RGB input layer's input gradient: a 64 -> 3 channel convolution of dy (rgbout16 form)
      SRGAN_REQUIRE(!dx_bf16, "conv2d_io_dgrad: the 3-channel input gradient of the RGB input layer is fp32");
      narrow_dgrad_desc(d, &f, &w_off);
      return rgbout_run(&f, dy, static_cast<const float*>(packed), nullptr, static_cast<float*>(dx), st, dy_bf16 != 0);
    }
    SRGAN_REQUIRE(!dy_bf16, "conv2d_io_dgrad: the 3-channel gradient of the RGB output layer's result is fp32");
    rgbin_dgrad_desc(d, &f, &w_off);
    return rgbin_run(&f, static_cast<const float*>(dy), static_cast<const float*>(packed), nullptr, static_cast<float*>(dx), SRGAN_ACT_NONE, 0.f, st,
                     dx_bf16 != 0);
  }
  if (b == 1) return halo16t_run(d, dy, packed, dx, conv_flops(d), st, dy_bf16 != 0, dx_bf16 != 0);
  const size_t need = srgan_conv2d_packed_scratch(d, 1);
  SRGAN_REQUIRE(need == 0 || (ws && ws_bytes >= need), "conv2d_io_dgrad: workspace too small (srgan_conv2d_packed_scratch)");
  DgradGeom g = dgrad_geometry(d);
  g.p.src = static_cast<const float*>(dy); g.p.wp = static_cast<const float*>(packed); g.p.src16 = dy_bf16 != 0;
  g.p.dst = static_cast<float*>(dx); g.p.dst16 = dx_bf16 != 0;
  return run_igemm(g.p, g.phases, st, conv_flops(d), need ? static_cast<float*>(ws) : nullptr);
}

// A weight used more than once in one backward pass (the generator runs twice inside util_notebook.py:664 and :689) gets its
// later contributions added by the split-K slab reduce instead of by a separate elementwise pass of the caller.
extern "C" int srgan_set_wgrad_accumulate(int on) {
  srgan::g_wgrad_accumulate = (on & 1) != 0;
  srgan::g_wgrad_deferrable = (on & 2) != 0;
  return 0;
}

extern "C" int srgan_wgrad_defer_begin(void* arena, size_t arena_bytes, void* stream) {
  SRGAN_REQUIRE(arena && arena_bytes >= 4096, "wgrad_defer_begin: null or tiny arena");
  SRGAN_REQUIRE((reinterpret_cast<uintptr_t>(arena) & 255) == 0, "wgrad_defer_begin: the arena must be 256-byte aligned");
  std::lock_guard<std::mutex> lock(srgan::g_defer_mutex);
  SRGAN_REQUIRE(!srgan::g_defer.active, "wgrad_defer_begin: already active");
  srgan::g_defer.active = true;
  srgan::g_defer.arena = static_cast<char*>(arena);
  srgan::g_defer.bytes = arena_bytes & ~(size_t)255;
  srgan::g_defer.used = 0;
  srgan::g_defer.asked = 0;
  srgan::g_defer.st = as_stream(stream);
  srgan::g_defer.pending.clear();
  return 0;
}

extern "C" int srgan_wgrad_defer_stats(long long* sums, long long* launches) {
  SRGAN_REQUIRE(sums && launches, "wgrad_defer_stats: null pointer");
  std::lock_guard<std::mutex> lock(srgan::g_defer_mutex);
  *sums = srgan::g_defer_sums;
  *launches = srgan::g_defer_launches;
  return 0;
}

extern "C" int srgan_wgrad_defer_need(long long* bytes) {
  SRGAN_REQUIRE(bytes, "wgrad_defer_need: null pointer");
  std::lock_guard<std::mutex> lock(srgan::g_defer_mutex);
  *bytes = (long long)srgan::g_defer.asked;
  return 0;
}

extern "C" int srgan_wgrad_defer_end(void) {
  std::lock_guard<std::mutex> lock(srgan::g_defer_mutex);
  if (!srgan::g_defer.active) return 0;
  const int e = srgan::defer_launch_pending(true);
  srgan::g_defer.active = false;
  return e;
}

// ---- compute mode (BASELINE configs [2]-[4] are bf16): process-wide, set between steps ----
extern "C" int srgan_set_compute_mode(int mode) {
  SRGAN_REQUIRE(mode == 0 || mode == 1, "set_compute_mode: 0 (fp32) or 1 (bf16 MFMA, fp32 accumulate)");
  g_compute_mode = mode;
  return 0;
}
extern "C" int srgan_get_compute_mode(void) { return g_compute_mode; }

// ---- launch-timer API (used only by bench.py) ------------------------------------------------
extern "C" int srgan_prof_enable(int on) {
  if (on) {
    g_prof_slots.clear();
    g_prof_next = 0;
  }
  g_prof_on = on != 0;
  return 0;
}
extern "C" int srgan_prof_num_kernels(void) { return kProfKernels; }
extern "C" const char* srgan_prof_kernel_name(int kid) {
  return (kid >= 0 && kid < kProfKernels) ? kProfNames[kid] : "";
}
// One timed launch of the session (index < srgan_prof_num_slots()): kernel id, duration, algorithmic FLOPs.
extern "C" int srgan_prof_num_slots(void) { return (int)g_prof_slots.size(); }
extern "C" int srgan_prof_slot(int index, int* kid, double* ms, double* flops) {
  SRGAN_REQUIRE(kid && ms && flops, "prof_slot: null pointer");
  SRGAN_REQUIRE(index >= 0 && (size_t)index < g_prof_slots.size(), "prof_slot: index out of range");
  const ProfSlot& s = g_prof_slots[index];
  float t = 0.f;
  hipError_t e = hipEventElapsedTime(&t, s.a, s.b);
  if (e != hipSuccess) { set_error("prof_slot: %s", hipGetErrorString(e)); return (int)e; }
  *kid = s.kid; *ms = t; *flops = s.flops;
  return 0;
}
// Totals of the session for one kernel id; call after the streams have been synchronised.
extern "C" int srgan_prof_collect(int kid, double* total_ms, long long* launches, double* total_flops) {
  SRGAN_REQUIRE(total_ms && launches && total_flops, "prof_collect: null pointer");
  double ms = 0.0, fl = 0.0;
  long long n = 0;
  for (const ProfSlot& s : g_prof_slots) {
    if (s.kid != kid) continue;
    float t = 0.f;
    hipError_t e = hipEventElapsedTime(&t, s.a, s.b);
    if (e != hipSuccess) { set_error("prof_collect: %s", hipGetErrorString(e)); return (int)e; }
    ms += t; fl += s.flops; ++n;
  }
  *total_ms = ms; *launches = n; *total_flops = fl;
  return 0;
}

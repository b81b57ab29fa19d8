// Shared helpers for the gfx950 kernels of libsrgan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "srgan_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace srgan {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

#define SRGAN_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      srgan::set_error(__VA_ARGS__);        \
      return -1;                            \
    }                                       \
  } while (0)

// Timing ablations (wrong results) and phase stamps are compile-time variants of the kernels: never in the product library.
#if !defined(SRGAN_EXPERIMENTS) && (defined(WINO_EXP) || defined(WINO42_EXP) || defined(RGBOUT_EXP) || defined(W43_DIAG) || defined(H16R_EXP))
#error "WINO_EXP / WINO42_EXP / RGBOUT_EXP / W43_DIAG / H16R_EXP are experiment builds: make exp EXPFLAGS=-DWINO_EXP=n (writes scratch/libsrgan_exp.so)"
#endif

// A/B switches of the measurement scripts under scratch/: only `make exp` (-DSRGAN_EXPERIMENTS, which writes
// scratch/libsrgan_exp.so, loaded through SRGAN_HIP_LIB) reads them from the environment.  In the product library every switch is
// a compile-time constant and its name is not in the binary (tests/test_abi_cpu.py checks the strings).
#ifdef SRGAN_EXPERIMENTS
#include <cstdlib>
#define SRGAN_AB_SET(name) (std::getenv(name) != nullptr)
#define SRGAN_AB_INT(name, dflt) (std::getenv(name) ? std::atoll(std::getenv(name)) : (long long)(dflt))
#else
#define SRGAN_AB_SET(name) false
#define SRGAN_AB_INT(name, dflt) ((long long)(dflt))
#endif

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline long long ceil_div(long long a, long long b) { return (a + b - 1) / b; }
inline long long round_up(long long a, long long b) { return ceil_div(a, b) * b; }

// 64-lane wavefront reductions (gfx950 wave = 64).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `red` must hold 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// The activation as one select: `act` is uniform, and written as `if (act == ...)` the compiler keeps a scalar branch per
// ELEMENT of an unrolled epilogue (~30 instructions and two taken branches around each 16-byte store: the row pass + stores of
// wino42_kernel's epilogue took 7000 cycles, round 5).  Negative-side slope: 1 (none), 0 (ReLU), slope (LeakyReLU); the
// `+ 0.f` keeps ReLU's zero positive.  Same values as the branch form for every FINITE input and for NaN (a NaN stays a NaN, as
// in torch.relu); the one difference: ReLU of -inf is fma(-inf, 0, 0) = NaN here where the branch form and
// torch.relu give 0 -- an overflowed activation now poisons what follows instead of being clamped.  Accepted: an infinite
// pre-activation means the step has already diverged, and bench.py / the trainer tests check the losses for finiteness.
__device__ __forceinline__ float act_neg_slope(int act, float slope) {
  return act == SRGAN_ACT_LRELU ? slope : (act == SRGAN_ACT_RELU ? 0.f : 1.f);
}
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  return v > 0.f ? v : __builtin_fmaf(v, act_neg_slope(act, slope), 0.f);
}
// derivative from the (pre- or post-activation) sign; slope > 0 keeps signs equal
__device__ __forceinline__ float act_grad(float v, int act, float slope) {
  return v > 0.f ? 1.f : act_neg_slope(act, slope);
}


// process-wide compute mode (conv_igemm.hip): true = bf16 MFMA operands, fp32 accumulate
bool compute_bf16();

// launch-timer bracket (conv_igemm.hip) usable from the other translation units
struct ProfToken { size_t idx; bool on; };
ProfToken prof_begin(int kid, double flops, hipStream_t st);
void prof_end(ProfToken t, hipStream_t st);

// parameters of the fused Winograd forward / input-gradient kernels (conv_wino.hip, conv_wino43.hip)
struct WinoParams {
  const float* src;   // [NB][H][W][C]
  const float* u;     // [n_tiles][nchunk][16 pos][2 halves][64 couts][4 ch] transformed filters
  const float* bias;  // [Cd] or null
  float* dst;         // [NB][Ho][Wo][Cd]
  int NB, H, W, C, Ho, Wo, Cd;   // source tensor, destination tensor (Cd = its channels)
  int pad, reflect;
  int TH, TW, T;      // tile grid per image (of the phase image in mode 2), tiles in total
  int nchunk, n_tiles, m_tiles;
  int cpp;            // mode 1: chunks per input phase (C / 8; F(4x4,2x2): C / 16)
  int ipw;            // F(4x4,2x2): items per workgroup (set by wino42_launch)
  int act;            // fused activation of the epilogue (SRGAN_ACT_*)
  float slope;
  const float* res;   // F(4x4,3x3) only: tensor of the destination's shape added in the epilogue (residual gradient), or null
  const float* mask;  // transposed F(3x3,2x2) only: tensor of the destination's shape whose sign selects 1 / mask_slope per element
  float mask_slope;   //   (the LeakyReLU backward of the layer that produced the destination's forward tensor), or null
};

// conv_wino42.hip: F(4x4,2x2) kernels of the 4x4 / stride-2 layers behind wino_run (variant 7): kind 0 = strided form, 1 = transposed
int wino42_launch(const WinoParams& p, int kind, long long grid, bool mask, double flops, hipStream_t st);
// conv_wino43.hip: F(4x4,3x3) kernel behind wino_run
size_t wino43_scratch_floats(long long T, int C);
int wino43_launch(const WinoParams& p, float* vimg, long long grid, double flops, hipStream_t st, bool v_ready = false);
// instance norm + activation of a 32x32 map written straight as the V image of the following F(4x4,3x3) layer (conv_wino43.hip)
int in_fwd_slab_v_launch(const float* x, const float* scale, const float* shift, float* mean, float* rstd, float* vimg, int N,
                         int C, float eps, int act, float slope, hipStream_t st);
// F(4x4,3x3) weight gradient from the forward's V image (conv_wino43.hip; geometry and dispatch in conv_wino.hip)
struct Wino43WgradGeom { int NB, H, W, C, O, ntc, splits, chunks_per_split; size_t v_bytes, z_bytes, slab_bytes; };
bool wino43_wgrad_geometry(const srgan_conv_desc* d, Wino43WgradGeom* g);      // false: not applicable
int wino43_wgrad_launch(const Wino43WgradGeom& g, const float* vimg, const float* dy, float* zimg, float* slab, double flops, hipStream_t st, bool z_ready = false);
// instance-norm backward of a 32x32 map writing the two F(4x4,3x3) transforms of its result instead of the result (conv_wino43.hip)
int in_bwd_slab_vz_launch(const float* x, const float* gup, const float* scale, const float* shift, const float* mean,
                          const float* rstd, float* dscale, float* dshift, float* vimg, float* zimg, int N, int C, int act,
                          float slope, hipStream_t st);
bool wino43_dgrad_applicable(const srgan_conv_desc* d);    // the input gradient of d runs on F(4x4,3x3)
// conv_wino.hip: Winograd F(2x2,3x3) for 3x3 stride-1 pad-1 layers; kind 0 = forward, 1 = input gradient
bool wino_applicable(const srgan_conv_desc* d, int kind);
double wino_threshold_scale();     // test hook SRGAN_WINOGRAD_THRESHOLD_SCALE: scales the minimum-workgroup thresholds of the dispatch
size_t wino_packed_bytes(const srgan_conv_desc* d, int kind);
int wino_pack(const srgan_conv_desc* d, int kind, const float* w, float* dst, hipStream_t st);
size_t wino_scratch_bytes(const srgan_conv_desc* d, int kind);
int wino_run(const srgan_conv_desc* d, int kind, const float* src, const float* packed, const float* bias, float* dst, int act, float slope, float* scratch, hipStream_t st, const float* res = nullptr, bool* res_done = nullptr, bool v_ready = false, const float* mask = nullptr, float mask_slope = 0.f, bool* mask_done = nullptr);
bool wino43_fwd_applicable(const srgan_conv_desc* d);      // the forward of d runs on F(4x4,3x3)
bool wino_wgrad_applicable(const srgan_conv_desc* d);
void wino_wgrad_slab(const srgan_conv_desc* d, int* splits, int* Cdpad, int* NNpad);
int wino_wgrad_run(const srgan_conv_desc* d, const float* x, const float* dy, float* slab, hipStream_t st);

// conv_halo16.hip: bf16-mode 3x3 stride-1 trunk convolution with the activation patch resident in LDS (dispatch: variant 4 of
// conv_wino.hip's slot -- not a Winograd transform, it shares the packed-filter plumbing)
bool halo16_applicable(const srgan_conv_desc* d, int kind);
size_t halo16_packed_bytes(const srgan_conv_desc* d);
int halo16_run(const srgan_conv_desc* d, int kind, const void* src, const void* packed, const float* bias, const float* res,
               void* dst, int act, float slope, double flops, hipStream_t st, bool in16 = false, bool out16 = false);
bool halo16t_applicable(const srgan_conv_desc* d);        // kind 1 of a 4x4 / stride-2 layer (transposed form)
size_t halo16t_packed_bytes(const srgan_conv_desc* d);
int halo16t_run(const srgan_conv_desc* d, const void* dy, const void* packed, void* dx, double flops, hipStream_t st,
                bool in16 = false, bool out16 = false);
bool halo16s_applicable(const srgan_conv_desc* d);        // kind 0 of a 4x4 / stride-2 layer (strided form, variant 6)
size_t halo16s_packed_bytes(const srgan_conv_desc* d);
int halo16s_run(const srgan_conv_desc* d, const void* x, const void* packed, const float* bias, void* y, int act, float slope,
                double flops, hipStream_t st, bool in16 = false, bool out16 = false);
bool halo16_wgrad_applicable(const srgan_conv_desc* d);
void halo16_wgrad_slab(const srgan_conv_desc* d, int* splits, int* Cdpad, int* NNpad);
int halo16_wgrad_run(const srgan_conv_desc* d, const void* x, const void* dy, float* slab, double flops, hipStream_t st,
                     bool x16 = false, bool d16 = false);

// conv_halo16e.hip (round 6): the style encoder's 3x3 / stride-1 layers on 62- and 31-pixel maps in the bf16 mode -- LDS-resident
// halo of a ragged destination patch, filter operand from the packed register image (PackParams::regimg).  Geometry as the
// implicit GEMM states it (IgemmParams): source map Hs x Ws x Cs, destination map Hd x Wd x N, halo origin = patch origin +
// (oy0, ox0), `flip`: the filter image holds tap 8 - t at halo tap t (input gradient), `reflect`: mirrored halo coordinates.
struct Halo16eParams {
  const void* src;             // [NB][Hs][Ws][Cs] fp32, or bf16 (src16)
  const unsigned short* wp;    // register image, bf16
  const float* bias;           // [N] or null
  void* dst;                   // [NB][Hd][Wd][N] fp32, or bf16 (dst16)
  int NB, Hs, Ws, Cs, Hd, Wd, N;
  int oy0, ox0, flip, reflect, act;
  float slope;
  int src16, dst16;
  int tiles_y, tiles_x, n_tiles;       // set by halo16e_run
};
bool halo16e_shape_ok(int Cs, int N);
int halo16e_patch_rows(int Cs, int N);
int halo16e_run(Halo16eParams p, double flops, hipStream_t st);

// conv_rgbin.hip: 3-channel-input 7x7 stride-1 layers on the MFMA (LDS-staged halo)
// conv_rgbout.hip: 7x7 / stride-1 / pad-3 layers with <= 4 OUTPUT channels on the 4x4x1 MFMA (direct, LDS-resident halo and filter)
bool rgbout_applicable(const srgan_conv_desc* d);
size_t rgbout_packed_elems(const srgan_conv_desc* d);
int rgbout_pack(const srgan_conv_desc* d, const float* w, float* dst, hipStream_t st);
// src16 (bf16 mode, rgbout16_served layers): x is a bf16 tensor
int rgbout_run(const srgan_conv_desc* d, const void* x, const float* packed, const float* bias, float* y, hipStream_t st, bool src16 = false);
bool rgbout16_served(const srgan_conv_desc* d);
bool rgbin_applicable(const srgan_conv_desc* d);
size_t rgbin_packed_elems(const srgan_conv_desc* d);
int rgbin_pack(const srgan_conv_desc* d, const float* w, float* dst, hipStream_t st);
// dst16 (bf16 mode, rgbin16_served layers): y is written as bf16
int rgbin_run(const srgan_conv_desc* d, const float* x, const float* packed, const float* bias, float* y, int act, float slope, hipStream_t st,
              bool dst16 = false);
bool rgbin16_served(const srgan_conv_desc* d);
int rgb_wgrad_kind(const srgan_conv_desc* d);      // -1: not applicable
void rgb_wgrad_slab(const srgan_conv_desc* d, int* splits, int* Cdpad, int* NNpad);
int rgb_wgrad_run(const srgan_conv_desc* d, const void* x, const void* dy, float* slab, hipStream_t st, bool c64_bf16 = false);
bool rgb_wgrad16_served(const srgan_conv_desc* d);      // bf16 mode: the bf16-MFMA kernel runs (and takes a bf16 64-channel tensor)

// conv_narrow.hip: direct kernels for Cout <= 4, stride-1, zero-pad layers
bool narrow_applicable(const srgan_conv_desc* d);
size_t narrow_workspace(const srgan_conv_desc* d);
int narrow_pack(const srgan_conv_desc* d, const float* w, float* wp, hipStream_t st);
int narrow_fwd_packed(const srgan_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, hipStream_t st);
int narrow_fwd_strided(const srgan_conv_desc* f, int pad_x, const float* x, const float* wp, float* y, int Hd, int Wd, int oy_off,
                       int ox_off, hipStream_t st);
bool narrow_wave_applicable(const srgan_conv_desc* d);
int narrow_wave_fwd(const srgan_conv_desc* d, const float* x, const float* wp, int Kpad, const float* bias, float* y, hipStream_t st);
bool dense_head_applicable(const srgan_conv_desc* d);
int dense_head_fwd(const srgan_conv_desc* d, const float* x, const float* wp, int Kpad, const float* bias, float* y, hipStream_t st);
int narrow_wgrad(const srgan_conv_desc* d, const float* x, const float* dy, void* ws, int* n_slabs, hipStream_t st);

}  // namespace srgan

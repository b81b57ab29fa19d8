/*
 * srgan_hip.h -- C ABI of libsrgan_hip.so: the MI355X (gfx950) kernels behind the
 * Style-Restricted GAN G+D+E train step.
 *
 * The reference (shinshoji01/Style-Restricted_GAN) has no FFI layer: its "operator API" is
 * the set of torch.nn / torch.nn.functional calls made by pyfiles/model.py, pyfiles/util.py
 * and pyfiles/util_notebook.py.  Each entry point below names the reference call site(s) it
 * replaces.  The Python host (style-restricted_gan_amd/srgan_amd) binds these with ctypes and
 * mirrors the reference's nn.Module / loss / trainer signatures on top of them.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to dense fp32 data unless stated otherwise;
 *   - activations are NHWC ([N][H][W][C], C fastest); 2-D tensors are row-major;
 *   - conv weights are read through explicit element strides (sO,sI,sH,sW) of the logical
 *     [O][I][kh][kw] tensor, so PyTorch's OIHW parameters are consumed in place;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing syncs;
 *   - `ws` / `ws_bytes`: caller-owned scratch, size from the matching *_workspace() query;
 *   - return value: 0 on success, <0 invalid argument, >0 a hipError_t.  The text of the last
 *     error of the calling thread is available from srgan_last_error().
 */
#ifndef SRGAN_HIP_H
#define SRGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRGAN_ABI_VERSION 1

enum { SRGAN_ACT_NONE = 0, SRGAN_ACT_RELU = 1, SRGAN_ACT_LRELU = 2 };
enum { SRGAN_PAD_ZERO = 0, SRGAN_PAD_REFLECT = 1 };

int srgan_abi_version(void);
const char* srgan_last_error(void);

/* Geometry of one convolution y[N,Ho,Wo,O] = conv(x[N,Hi,Wi,I], w[O,I,kh,kw]) (+bias). */
typedef struct srgan_conv_desc {
  int N, Hi, Wi, I;      /* input  */
  int Ho, Wo, O;         /* output */
  int kh, kw, stride, pad;
  int pad_mode;          /* SRGAN_PAD_* (reflect: pyfiles/model.py:358,364,419,425) */
  long long sO, sI, sH, sW; /* element strides of the logical [O][I][kh][kw] weight */
} srgan_conv_desc;

/* nn.Conv2d forward (pyfiles/model.py:191,193,212,215,232,262,269,302,309,328-331,358,364,
 * 369,385,445) with optional bias and fused activation epilogue (nn.LeakyReLU model.py:263).
 * Also the input-gradient of nn.ConvTranspose2d (model.py:227,230) when called with the
 * transposed-conv weight viewed as [O=Cin_t][I=Cout_t]. */
size_t srgan_conv2d_workspace(const srgan_conv_desc* d);
int srgan_conv2d_fwd(const srgan_conv_desc* d, const float* x, const float* w, const float* bias,
                     float* y, int act, float slope, void* ws, size_t ws_bytes, void* stream);

/* Gradient w.r.t. the conv input (autograd of the call sites above): dx[N,Hi,Wi,I].
 * Also the FORWARD of nn.ConvTranspose2d (model.py:227,230).  Weights are read at call time. */
int srgan_conv2d_dgrad(const srgan_conv_desc* d, const float* dy, const float* w, float* dx,
                       void* ws, size_t ws_bytes, void* stream);

/* Packed-weight variants: the repacked weight operand ([phase][Npad][Kpad], or the narrow-output layout) depends only
 * on (desc, kind, act); pack it once per optimiser step and reuse it for every forward (kind 0) / input-gradient
 * (kind 1) of that step.  srgan_conv2d_fwd / _dgrad are exactly pack-into-workspace + the packed call. */
size_t srgan_conv2d_packed_bytes(const srgan_conv_desc* d, int kind, int act);
int srgan_conv2d_pack(const srgan_conv_desc* d, int kind, int act, const float* w, void* packed, size_t bytes, void* stream);
/* Scratch the packed calls need beside the operand (0 for most layers): the transformed-input image of the F(4x4,3x3)
 * Winograd layers (kind 0 and 1), the padded-gradient temp of a reflect-padded layer (kind 1).  Caller-owned like every
 * workspace of this library; only live between the launches of one call, so one buffer serves every layer of a stream. */
size_t srgan_conv2d_packed_scratch(const srgan_conv_desc* d, int kind);
/* Non-zero when the packed operand depends on the weights and channel counts only (Winograd filter images, RGB-input image):
 * descriptors of the same weight that differ in N / H / W only and return the same signature may share one packed buffer. */
unsigned long long srgan_conv2d_pack_signature(const srgan_conv_desc* d, int kind, int act);
int srgan_conv2d_fwd_packed(const srgan_conv_desc* d, const float* x, const void* packed, const float* bias, float* y,
                            int act, float slope, void* ws, size_t ws_bytes, void* stream);
int srgan_conv2d_dgrad_packed(const srgan_conv_desc* d, const float* dy, const void* packed, float* dx,
                              void* ws, size_t ws_bytes, void* stream);
/* conv3x3(act(instance_norm(x) * scale + shift)) without the intermediate tensor: inside SingleResidualBlock (model.py:196-201:
 * c1 -> cn1 -> ReLU -> c2) the normalised activation is read by c2 only.  srgan_instnorm_fwd_v = srgan_instnorm_fwd (same
 * statistics, same expression, mean / rstd saved for the backward) whose output is written directly as the transformed-input
 * ("V") image of the F(4x4,3x3) layer `d` (v_bytes >= srgan_conv2d_packed_scratch(d, 0)); srgan_conv2d_fwd_from_v then runs
 * that layer's multiply + output transform on it, and srgan_conv2d_wgrad_v its weight gradient.  Applicable
 * (srgan_instnorm_conv_v_applicable != 0) to 32x32 maps with Cin % 32 == 0 whose forward dispatches to F(4x4,3x3). */
int srgan_instnorm_conv_v_applicable(const srgan_conv_desc* d);
int srgan_instnorm_fwd_v(const srgan_conv_desc* d, const float* x, const float* scale, const float* shift, float* mean,
                         float* rstd, void* v_image, size_t v_bytes, float eps, int act, float slope, void* stream);
int srgan_conv2d_fwd_from_v(const srgan_conv_desc* d, const void* v_image, const void* packed, const float* bias, float* y,
                            int act, float slope, void* stream);
/* The backward counterpart.  `d` describes a convolution whose OUTPUT y is normalised (y -> (CB)IN -> activation): the gradient
 * dy w.r.t. y, produced by the norm's backward, is read by exactly two kernels -- the convolution's input gradient and its
 * F(4x4,3x3) weight gradient -- each through a transform.  srgan_instnorm_bwd_vz = srgan_instnorm_bwd (same dscale / dshift,
 * same dy values) that writes those two transforms instead of dy: v_image (srgan_conv2d_packed_scratch(d, 1) bytes) for
 * srgan_conv2d_dgrad_from_v (multiply + output transform, optional skip-path gradient `res` as in ..._dgrad_packed_add), z_image
 * (srgan_instnorm_bwd_vz_z_bytes(d) bytes) for srgan_conv2d_wgrad_vz (with the V image the forward kept).  x = y (the conv
 * output the forward normalised), dy = gradient w.r.t. the norm's output. */
int srgan_instnorm_bwd_vz_applicable(const srgan_conv_desc* d);
size_t srgan_instnorm_bwd_vz_z_bytes(const srgan_conv_desc* d);
int srgan_instnorm_bwd_vz(const srgan_conv_desc* d, const float* x, const float* dy, const float* scale, const float* shift,
                          const float* mean, const float* rstd, float* dscale, float* dshift, void* v_image, size_t v_bytes,
                          void* z_image, size_t z_bytes, int act, float slope, void* stream);
int srgan_conv2d_dgrad_from_v(const srgan_conv_desc* d, const void* v_image, const void* packed, const float* res, float* dx,
                              void* stream);
int srgan_conv2d_wgrad_vz(const srgan_conv_desc* d, const float* v_image, const float* z_image, float* dw, void* ws,
                          size_t ws_bytes, void* stream);
/* dx = (input gradient of the convolution) + res: SingleResidualBlock (model.py:196-201) feeds its input to its first
 * convolution AND to the skip connection, so the gradient of the block input is the sum of the two paths; handing the skip
 * path's gradient to the convolution's input-gradient kernel (added in the F(4x4,3x3) epilogue; one in-place pass on the other
 * dispatches) replaces autograd's separate accumulation pass over the tensor.  res: dense, shape of dx, must not alias dx. */
/* dx = (input gradient of the convolution) * LeakyReLU'(mask): a discriminator trunk is conv -> LeakyReLU(0.2) -> conv -> ...
 * (model.py:303-309), so the gradient this layer hands to the previous one must still be multiplied by that layer's LeakyReLU
 * derivative -- 1 or `slope` after the sign of its activated output, which IS this layer's input tensor (`mask`).  Folded into
 * the epilogue of the transposed F(3x3,2x2) kernel (one in-place elementwise pass on the other dispatches) it replaces the
 * read-read-write pass in front of the previous layer's backward.  mask: dense, shape of dx, must not alias dx. */
int srgan_conv2d_dgrad_packed_mask(const srgan_conv_desc* d, const float* dy, const void* packed, const float* mask, float slope,
                                   float* dx, void* ws, size_t ws_bytes, void* stream);
int srgan_conv2d_dgrad_packed_add(const srgan_conv_desc* d, const float* dy, const void* packed, const float* res, float* dx,
                                  void* ws, size_t ws_bytes, void* stream);

/* Many packs in one launch (after an optimiser step every cached operand of the network is stale at once):
 * srgan_conv2d_pack_entry writes one host record (srgan_pack_entry_bytes() bytes) per operand -- it returns 1 for the few
 * layers whose operand is made by another kernel (narrow-output layers; use srgan_conv2d_pack) -- the caller copies the array
 * of records to the device and srgan_conv2d_pack_multi repacks them all. */
size_t srgan_pack_entry_bytes(void);
int srgan_conv2d_pack_entry(const srgan_conv_desc* d, int kind, int act, const float* w, void* packed, void* entry);
int srgan_conv2d_pack_multi(const void* entries_dev, int n_entries, void* stream);

/* Gradient w.r.t. the conv weight, written through (sO,sI,sH,sW); dbias[O] (may be NULL) = column sums of dy.
 * Overwrites dw / dbias, unless srgan_set_wgrad_accumulate(1) is in effect on the calling host thread: then every weight-gradient
 * entry point (srgan_conv2d_wgrad, _wgrad_v, _wgrad_vz, srgan_halo16_wgrad) ADDS its result in the split-K slab reduce
 * (dw += ...): the reference's autograd sums the two uses of the generator's weights inside one backward call
 * (util_notebook.py:664, :689; torch's AccumulateGrad input buffer) -- here without a separate elementwise pass. */
int srgan_set_wgrad_accumulate(int on);

/* Deferred slab sums (round 3).  Every weight-gradient entry point ends in a split-K slab sum of ~12 us that cannot fill the
 * chip; the reference's step has 145 of them.  Between srgan_wgrad_defer_begin(arena, bytes) and srgan_wgrad_defer_end()
 * (process-wide -- autograd calls the entry points from its own thread -- so one backward pass at a time) a call made with bit 1 of the switch above set -- srgan_set_wgrad_accumulate(2 | accumulate): "nobody reads dw
 * before _end", true for the gradient buffers autograd's AccumulateGrad (util_notebook.py:664, :689 `.backward()`) would fill --
 * takes its workspace from the arena (256-byte aligned device memory, the caller keeps it alive and untouched until _end; `ws`
 * is then unused) and queues its sum; queued sums run as ONE launch when 40 are waiting, when the arena is full, when a second
 * sum for the same dw arrives, and at _end -- all on `stream`; calls made on any other stream are not deferred.  Same per-output loop and rounding as the
 * immediate sums: identical bits.  dbias column sums are not deferred. */
int srgan_wgrad_defer_begin(void* arena, size_t arena_bytes, void* stream);
int srgan_wgrad_defer_end(void);
/* Process totals since load: slab sums that went through the queue, and the launches that ran them. */
int srgan_wgrad_defer_stats(long long* sums, long long* launches);
/* Workspace bytes the deferrable calls since the last srgan_wgrad_defer_begin asked for (whether or not they fitted): an arena
 * of that size holds a whole pass without an early flush -- the caller sizes it from its first pass instead of guessing. */
int srgan_wgrad_defer_need(long long* bytes);
int srgan_conv2d_wgrad(const srgan_conv_desc* d, const float* x, const float* dy, float* dw,
                       float* dbias, void* ws, size_t ws_bytes, void* stream);

/* The same gradient from the transformed input the FORWARD of an F(4x4,3x3) layer already wrote (pyfiles/model.py:191,193: the
 * residual-trunk convs): when srgan_conv2d_wgrad_v_bytes(d) != 0, hand srgan_conv2d_fwd_packed a buffer of that many bytes as
 * `ws` and keep it until the backward pass; srgan_conv2d_wgrad_v then needs neither x nor a second input transform. */
size_t srgan_conv2d_wgrad_v_bytes(const srgan_conv_desc* d);
int srgan_conv2d_wgrad_v(const srgan_conv_desc* d, const float* v_image, const float* dy, float* dw,
                         float* dbias, void* ws, size_t ws_bytes, void* stream);

/* Instance norm + per-(n,c) affine + activation (+ residual):
 *   xh = (x - mean_nc) * rstd_nc ; y = act(xh * scale[n,c] + shift[n,c]) (+ res)
 * F.instance_norm(eps=1e-5, biased var) at model.py:58-60 (CBIN), nn.InstanceNorm2d at
 * model.py:178 (scale/shift NULL), nn.ReLU model.py:199,240,246, LeakyReLU(0.2) model.py:418,
 * residual add model.py:201.  mean/rstd [N*C] are written for the backward. */
size_t srgan_instnorm_workspace(int N, int HW, int C);
int srgan_instnorm_fwd(const float* x, const float* scale, const float* shift, const float* res,
                       float* y, float* mean, float* rstd, int N, int HW, int C, float eps,
                       int act, float slope, void* ws, size_t ws_bytes, void* stream);
/* Backward: dx, and (when scale!=NULL or wanted) dscale[n,c]=sum dy_act*xh, dshift[n,c]=sum dy_act.
 * dres is dy itself (caller aliases it).  y_act_src: the forward's x (pre-norm) is re-normalised
 * to rebuild the activation mask. */
int srgan_instnorm_bwd(const float* x, const float* dy, const float* scale, const float* shift,
                       const float* mean, const float* rstd, float* dx, float* dscale,
                       float* dshift, int N, int HW, int C, int act, float slope,
                       void* ws, size_t ws_bytes, void* stream);

/* Central-biasing affine of _CBINorm.forward (model.py:54-67): t = tanh(c W^T + b);
 * scale[n,ch] = gamma[ch]; shift[n,ch] = t*gamma + beta.  c:[N,num_con] W:[C,num_con]. */
int srgan_cbin_affine_fwd(const float* c, const float* W, const float* b, const float* gamma,
                          const float* beta, float* t, float* scale, float* shift,
                          int N, int C, int num_con, void* stream);
/* Backward of the above from dscale/dshift[N,C]: dgamma, dbeta, dW, db (overwritten), dc [N,num_con]. */
int srgan_cbin_affine_bwd(const float* c, const float* W, const float* gamma, const float* t,
                          const float* dscale, const float* dshift, float* dgamma, float* dbeta,
                          float* dW, float* db, float* dc, int N, int C, int num_con,
                          void* ws, size_t ws_bytes /* >= N*C floats */, void* stream);

/* The same affine for ALL central-biasing layers of a network in one launch (forward) / two launches (backward): the 15
 * `_CBINorm` layers of SingleGenerator (model.py:213-224) depend on the style code and their own parameters only.
 * The caller fills one host record per layer (srgan_cbin_rec_bytes() bytes each; pointers a pass does not use may be NULL),
 * copies the array to the device and passes it as `table_dev`; outputs of a layer are dense [N][C] blocks.
 * Backward: dscale / dshift must be non-NULL (zeros for absent gradients), `da` is [N][C] scratch per layer, `gamma` is the
 * forward-time copy (row 0 of the saved scale); dc[N][num_con] sums the layers in table order (dc may be NULL: the style
 * code's gradient is only wanted when the code came from the encoder, not for the noise codes of most generator passes). */
size_t srgan_cbin_rec_bytes(void);
int srgan_cbin_rec_fill(void* rec, const float* W, const float* b, const float* gamma, const float* beta, float* t,
                        float* scale, float* shift, const float* dscale, const float* dshift, float* dgamma,
                        float* dbeta, float* dW, float* db, float* da, int C);
/* on != 0: the backward of this record ADDS to dgamma / dbeta / dW / db (a layer reached twice in one backward pass: the
 * generator is back-propagated through two graphs inside util_notebook.py:664 and again inside :689). */
int srgan_cbin_rec_set_accumulate(void* rec, int on);
int srgan_cbin_affine_multi_fwd(const float* c, const void* table_dev, int n_layers, int N, int max_C, int num_con,
                                void* stream);
int srgan_cbin_affine_multi_bwd(const float* c, const void* table_dev, int n_layers, int N, int max_C, int num_con,
                                float* dc, void* stream);

/* Pointwise: y = act(x) and dx = dy * act'(y) (mask from the OUTPUT sign, slope>0). tanh: model.py:248 */
int srgan_act_fwd(const float* x, float* y, long long n, int act, float slope, void* stream);
int srgan_act_bwd(const float* y, const float* dy, float* dx, long long n, int act, float slope, void* stream);
int srgan_tanh_fwd(const float* x, float* y, long long n, void* stream);
int srgan_tanh_bwd(const float* y, const float* dy, float* dx, long long n, void* stream);
/* y = a + b (E block: cmp(...) + shortcut(x), model.py:436) */
int srgan_add(const float* a, const float* b, float* y, long long n, void* stream);

/* nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False)  model.py:286,324 */
int srgan_avgpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int srgan_avgpool3s2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* nn.AvgPool2d(2,2) (floor)  model.py:365,368,426,429 */
int srgan_avgpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int srgan_avgpool2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* The same with an fp32 or bf16 tensor on either side (C % 4 == 0): around the 16-bit convolutions of the style encoder in the
 * bf16 mode (pyfiles/model.py:409-411). */
int srgan_avgpool2_fwd_io(const void* x, int x_bf16, void* y, int y_bf16, int N, int H, int W, int C, void* stream);
int srgan_avgpool2_bwd_io(const void* dy, int dy_bf16, void* dx, int dx_bf16, int N, int H, int W, int C, void* stream);
/* LeakyReLU(slope) -> AdaptiveAvgPool2d(1)  (Encoder.last_layer, model.py:454,475-480) */
int srgan_lrelu_gap_fwd(const float* x, float* y, int N, int HW, int C, float slope, void* stream);
int srgan_lrelu_gap_bwd(const float* x, const float* dy, float* dx, int N, int HW, int C, float slope, void* stream);

/* Encoder.reparametrize (model.py:459-463): out = eps * exp(logvar / 2) + mu, stdv = exp(logvar / 2) (kept for the backward);
 * backward: dmu = g, dlogvar = g * eps * stdv / 2.  n = B * ndim; every operation rounded on its own, as the reference's chain. */
int srgan_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* out, float* stdv, long long n, void* stream);
int srgan_reparam_bwd(const float* g, const float* eps, const float* stdv, float* dlogvar, long long n, void* stream);

/* nn.Linear: y[M,N] = x[M,K] W[N,K]^T + b  (model.py:455-457; small M) */
int srgan_linear_fwd(const float* x, const float* W, const float* b, float* y, int M, int N, int K, void* stream);
int srgan_linear_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db,
                     int M, int N, int K, void* stream);

/* bf16 compute mode, residual-trunk layers (3x3 / stride 1 / zero pad 1, Cin = Cout in {64, 128, 256}, maps of whole 4 x 32
 * patches; reference pyfiles/model.py:188-201) with bf16 TENSORS in HBM on either side: the kernels of the fused residual block
 * whose intermediates -- conv outputs, the normalised activation, the gradients between the norm and conv backward kernels --
 * are stored as bf16 (fp32 accumulation, fp32 statistics, fp32 master weights and residual stream).  `*_bf16` flags say which
 * side is bf16 ([N][H][W][C] dense NHWC either way).  `packed`: the ordinary packed operand of (d, kind) made in bf16 mode.
 * srgan_halo16_conv: kind 0 forward, kind 1 input gradient (+ `res`, fp32, added to an fp32 result).  srgan_halo16_wgrad: dw
 * (fp32, through the descriptor's weight strides); bf16 x requires bf16 dy.  ws: srgan_conv2d_workspace(d) bytes. */
int srgan_halo16_applicable(const srgan_conv_desc* d);
/* Round 4: the same two entry points also serve the 4x4 / stride-2 / pad-1 layers whose three directions run on the
 * LDS-resident-patch kernels in the bf16 mode -- (Cin, Cout) = (64, 128) / (128, 256), output maps of whole 4 x 32 patches: the
 * generator's down convolutions (pyfiles/model.py:212-215) and, through their transposed form, its up convolutions (:227-230).
 * kind 0 = strided form (x -> y, or a ConvTranspose2d's input gradient), kind 1 = transposed form (dy -> dx, or a ConvTranspose2d
 * forward); no `res`.  srgan_halo16_wgrad on these layers: bf16 x with fp32 or bf16 dy, or both fp32. */
int srgan_halo16s2_applicable(const srgan_conv_desc* d);
int srgan_halo16_conv(const srgan_conv_desc* d, int kind, const void* src, int src_bf16, const void* packed, const float* res,
                      void* dst, int dst_bf16, void* stream);
int srgan_halo16_wgrad(const srgan_conv_desc* d, const void* x, int x_bf16, const void* dy, int dy_bf16, float* dw, void* ws,
                       size_t ws_bytes, void* stream);
/* Round 5: the generic layers of the bf16 mode (no LDS-resident-patch kernel: the style encoder's 3x3 reflect-padded
 * convolutions on 62 / 31 / 15 / 7-pixel maps, pyfiles/model.py:413-437) with bf16 TENSORS on either side.  srgan_igemm16_io_applicable:
 * forward (with `act`), input gradient and weight gradient of d all run on the 64-deep-K-tile implicit GEMM and the vector
 * weight-gradient kernel (bf16 mode, Cin % 64 == 0, Cout % 64 == 0 ...).  srgan_igemm16_conv: kind 0 forward (bias / activation as
 * srgan_conv2d_fwd_packed), kind 1 input gradient (no bias / activation; reflect padding folds into dst's type); `packed`: the
 * ordinary packed operand of (d, kind, act); ws: srgan_conv2d_packed_scratch(d, kind) bytes.  srgan_igemm16_wgrad: x and dy both
 * bf16, dw fp32 through the descriptor's weight strides; ws: srgan_conv2d_workspace(d) bytes; honours
 * srgan_set_wgrad_accumulate / the deferred slab sums like srgan_conv2d_wgrad. */
int srgan_igemm16_io_applicable(const srgan_conv_desc* d, int act);
int srgan_igemm16_conv(const srgan_conv_desc* d, int kind, const void* src, int src_bf16, const void* packed, const float* bias,
                       void* dst, int dst_bf16, int act, float slope, void* ws, size_t ws_bytes, void* stream);
int srgan_igemm16_wgrad(const srgan_conv_desc* d, const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, void* stream);
/* bf16 mode, round 6: the 4x4 / stride-2 / pad-1 layers of the discriminator trunks (pyfiles/model.py:302-309: conv without bias
 * -> LeakyReLU(0.01), no norm between the layers) with bf16 tensors on either side: forward with the bias / activation epilogue
 * (halo16s_kernel, or igemm16_kernel with LDS-DMA tiles), input gradient (halo16t_kernel or igemm16_kernel); the weight gradient is
 * srgan_halo16_wgrad.  `packed`: the ordinary packed operand of (d, kind, act).  ws: srgan_conv2d_packed_scratch(d, kind) bytes. */
/* Also (act = none): the generator's 7x7 / stride-1 / pad-3 layers between a 3-channel and a 64-channel tensor (pyfiles/model.py:212,
 * 232) with the 64-CHANNEL side in bf16 -- the RGB input layer writes a bf16 y and takes a bf16 dy, the RGB output layer reads a
 * bf16 x and writes a bf16 dx; the 3-channel side is fp32 (a bf16 flag there is refused).  Their weight gradient is
 * srgan_halo16_wgrad too (bf16 allowed on the 64-channel tensor only). */
int srgan_conv2d_io_applicable(const srgan_conv_desc* d, int act);
int srgan_conv2d_io_fwd(const srgan_conv_desc* d, const void* x, int x_bf16, const void* packed, const float* bias, void* y, int y_bf16,
                        int act, float slope, void* ws, size_t ws_bytes, void* stream);
int srgan_conv2d_io_dgrad(const srgan_conv_desc* d, const void* dy, int dy_bf16, const void* packed, void* dx, int dx_bf16, void* ws,
                          size_t ws_bytes, void* stream);
/* LeakyReLU / ReLU backward dx = dy * f'(y) over n elements (n % 8 == 0), every tensor fp32 or bf16 (nn.LeakyReLU, model.py:303,310). */
int srgan_act_bwd_io(const void* y, int y_bf16, const void* dy, int dy_bf16, void* dx, int dx_bf16, long long n, int act, float slope,
                     void* stream);
/* Single-pass instance norm (+ per-sample scale / shift, activation, optional fp32 skip tensor) of maps with <= 1024 pixels with
 * bf16 tensors on either side; statistics, sums, scale / shift gradients in fp32.  Backward: x = the normalised tensor's INPUT
 * (fp32 or bf16), dy fp32 or bf16, dx of x's type.  srgan_instnorm_slab_applicable: the shape is served (C % 32 == 0, HW <= 1024, enough slabs). */
int srgan_instnorm_slab_applicable(int N, int HW, int C);
int srgan_instnorm_slab_fwd_io(const void* x, int x_bf16, const float* scale, const float* shift, const float* res, void* y,
                               int y_bf16, float* mean, float* rstd, int N, int HW, int C, float eps, int act, float slope,
                               void* stream);
int srgan_instnorm_slab_bwd_io(const void* x, int x_bf16, const void* dy, int dy_bf16, const float* scale, const float* shift,
                               const float* mean, const float* rstd, void* dx, int dx_bf16, float* dscale, float* dshift, int N,
                               int HW, int C, int act, float slope, void* stream);
/* Instance norm (+ affine, activation) with fp32 or bf16 tensors on either side, for the shapes the slab kernels or the fast
 * two-pass kernels serve (srgan_instnorm_io_applicable: HW <= 1024 with C % 32 == 0, or C | 1024 with C % 4 == 0): the norms
 * between the generator's down / up convolutions when the bf16 mode keeps those activations in 16 bits (pyfiles/model.py:54-67,
 * 228, 231, 245-246) and, on maps too large for the slab kernels, the norms inside a residual block with bf16 intermediates
 * (`res`: fp32 skip tensor added to an fp32 result; pyfiles/model.py:86-94).  Statistics, scale / shift and their gradients are
 * fp32.  Backward: dx has x's type.  ws:
 * srgan_instnorm_workspace(N, HW, C) bytes. */
int srgan_instnorm_io_applicable(int N, int HW, int C);
int srgan_instnorm_fwd_io(const void* x, int x_bf16, const float* scale, const float* shift, const float* res, void* y, int y_bf16,
                          float* mean, float* rstd, int N, int HW, int C, float eps, int act, float slope, void* ws, size_t ws_bytes,
                          void* stream);
int srgan_instnorm_bwd_io(const void* x, int x_bf16, const void* dy, int dy_bf16, const float* scale, const float* shift,
                          const float* mean, const float* rstd, void* dx, int dx_bf16, float* dscale, float* dshift, int N, int HW,
                          int C, int act, float slope, void* ws, size_t ws_bytes, void* stream);

/* NCHW <-> NHWC repack at the module boundary. */
int srgan_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream);
int srgan_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream);

/* ---- fused loss reductions (value + gradient in one launch) -------------------------------
 * Each writes loss[0] (overwrite) and the gradient of `weight * loss` w.r.t. its input. */
/* get_loss_D, util.py:457-462 with nn.MSELoss: one scale; loss = mean((o-target)^2)*weight */
int srgan_mse_const(const float* o, long long n, float target, float weight, float* loss, float* d_o, void* stream);
/* nn.Softmax(dim=1) + get_domainloss_D (util.py:464-468, model.py:333-346): z:[B,n_class] logits,
 * label:[B] int64; q=softmax(z) is written to `q`; loss = mean((q-onehot)^2)*weight; dz via softmax Jacobian. */
int srgan_softmax_mse(const float* z, const long long* label, int B, int n_class, float weight,
                      float* q, float* loss, float* dz, void* stream);
/* nn.CrossEntropyLoss() (mean reduction) of the encoder pre-training job, 04_Facial_Recognition-Encoder.ipynb cell 18/22:
 * loss = weight * mean_b(logsumexp(z_b) - z_b[label_b]); dz = weight * (softmax(z) - onehot) / B. */
int srgan_softmax_xent(const float* z, const long long* label, int B, int n_class, float weight, float* loss,
                       float* dz, void* stream);
/* torch.mean(torch.abs(a-b)) util_notebook.py:625,639,676,686 ; da = weight*sign(a-b)/n, db = -da */
size_t srgan_l1_workspace(long long n);
int srgan_l1_mean(const float* a, const float* b, long long n, float weight, float* loss,
                  float* da, float* db, void* ws, size_t ws_bytes, void* stream);
/* batch-KL + correlation + histogram-imitation on mu[B,d] (util_notebook.py:644-662, util.py:470-553).
 * vals[4] = (bKL, corr, hist, w_bkl*bKL + w_corr*corr + w_hist*hist); dmu = gradient of vals[3].
 * hist_target: [bins] device floats.  d <= 16, bins <= 64, B <= 4096. */
int srgan_latent_losses(const float* mu, int B, int d, float n_batch, const float* hist_target, int bins,
                        float range_max, float sigma, float w_bkl, float w_corr, float w_hist,
                        float* vals, float* dmu, float* corr_out /* [d*d] Pearson matrix or NULL */, void* stream);
/* Every loss of ONE discriminator evaluation in one launch (util.py:457-468 as used by util_notebook.py:582-590 and :622-624):
 * per scale s an LSGAN map o[s] = [rows][per_row[s]] and class logits z[s] = [rows][n_class] (z / dz may be NULL: no class
 * head).  Rows [0, rows_first) are compared with the constant t_first and carry the softmax + class-MSE loss against label[];
 * rows [rows_first, rows) (the translated half of a real | fake batch; may be empty) with t_rest.
 * vals[4] = {lsgan_first, class, lsgan_rest, total}, each the mean over scales of the per-scale nn.MSELoss,
 * total = lsgan_first + w_class * class + lsgan_rest; d_o[s] / dz[s] = d total / d o[s], z[s] (zero rows where unused).
 * o, per_row, z, d_o, dz are HOST arrays of n_scales (<= 4) entries. */
int srgan_d_losses(const float* const* o, const long long* per_row, const float* const* z, int n_scales, int rows,
                   int rows_first, int n_class, const long long* label, float t_first, float t_rest, float w_class,
                   float* vals, float* const* d_o, float* const* dz, void* stream);
/* out[0] = sum_i w[i] * x[i][0] over n <= 16 device scalars (x, w: host arrays) -- the weighted sum of a phase's loss terms
 * (util_notebook.py:585, :626-662, :677-687) in one launch; srgan_lincomb_bwd: dx[i] = w[i] * g[0]. */
int srgan_lincomb(const float* const* x, const float* w, int n, float* out, void* stream);
int srgan_lincomb_bwd(const float* w, int n, const float* g, float* dx, void* stream);
/* conventional KL term of the encoder (util_notebook.py:630-634; SingleGAN: :302): weight * -0.5 * sum(1 + logvar - mu^2 -
 * exp(logvar)) over all n = B*ndim values, with dmu / dlogvar (either may be NULL) in the same launch */
int srgan_kl_normal(const float* mu, const float* logvar, long long n, float weight, float* loss, float* dmu,
                    float* dlogvar, void* stream);
/* generic nn.MSELoss(a, b) * weight with both gradients (get_domainloss_D on probabilities, util.py:464-468) */
int srgan_mse_pair(const float* a, const float* b, long long n, float weight, float* loss, float* da, float* db,
                   void* stream);
/* GaussianHistogram.forward (util.py:521-537) of a 1-D sample x[n] -> h[bins], and its vector-Jacobian product */
size_t srgan_soft_histogram_workspace(long long n, int bins);
int srgan_soft_histogram_fwd(const float* x, long long n, int bins, float lo, float hi, float sigma, float* h,
                             void* ws, size_t ws_bytes, void* stream);
int srgan_soft_histogram_bwd(const float* x, const float* g, long long n, int bins, float lo, float hi, float sigma,
                             float* dx, void* stream);

/* optim.Adam.step, torch 1.4 arithmetic (util_notebook.py:500-506) on flat buffers.
 * step_count is the 1-based step number t.  Writes through raw pointers (no autograd version bump). */
int srgan_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                    float beta2, float eps, int step_count, void* stream);

/* The same update for a whole parameter list in ONE launch.  `table` (device memory): n_tensors records of five
 * 64-bit words {p, g, m, v, numel}; all tensors must share the step count (they do: one optimiser = one list). */
int srgan_adam_multi(const void* table, int n_tensors, long long max_numel, float lr, float beta1, float beta2,
                     float eps, int step_count, void* stream);

/* The optimiser as a hipGraph-capturable op (BASELINE configs[4] "hipGraph step"; the step captured is
 * util_notebook.py:696-734): the step counter t and the hyper-parameters live in a 32-byte DEVICE record, so replaying a
 * captured train step advances t like calling optimizer.step() would.  srgan_adam_multi_dev = t += 1, bias corrections of
 * torch 1.4's Adam from the device-side t (in double), then the update of srgan_adam_multi -- two launches, no host value
 * baked into the launch.  `steps_done` seeds t (0 for a fresh optimiser; the checkpoint's count when resuming);
 * srgan_adam_state_set_lr follows lr_scheduler.ExponentialLR (util_notebook.py:501-507) between steps. */
size_t srgan_adam_state_bytes(void);
int srgan_adam_state_init(void* state, float lr, float beta1, float beta2, float eps, int steps_done, void* stream);
int srgan_adam_state_set_lr(void* state, float lr, void* stream);
int srgan_adam_multi_dev(const void* table, int n_tensors, long long max_numel, void* state, void* stream);

/* Small host -> device upload (pointer tables: <= 1 MiB, multiple of 4 bytes) carried in kernel arguments: nothing to keep
 * alive on the host after the call returns, and a captured hipGraph stores the bytes in its node instead of re-reading a host
 * address at replay (no reference counterpart; plumbing of the multi-tensor ops above). */
int srgan_upload_small(void* dst_dev, const void* src_host, size_t nbytes, void* stream);

/* ---- compute mode.  0 (default): exact fp32 products on v_mfma_f32_32x32x2_f32 (BASELINE configs[0], [1]).
 * 1: bf16 MFMA compute for configs [2]-[4]: conv operands are rounded to bf16 (nearest even) on their way into LDS and
 * multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- torch.autocast(bfloat16) semantics for the convolutions;
 * tensors in HBM, norms, losses and Adam stay fp32; the Winograd kernels are not used.  Process-wide; packed weights made
 * under the other mode must be re-packed (srgan_amd.ops.set_compute_dtype does). */
int srgan_set_compute_mode(int mode);
int srgan_get_compute_mode(void);

/* ---- input pipeline: transforms.CenterCrop((178,178)) -> Resize((128,128)) -> RandomHorizontalFlip ->
 * ToTensor -> MinMax(True) of the training notebooks (05-train cell 9; MinMax: pyfiles/util.py:108-155) on a batch of decoded
 * uint8 RGB images src[B][Hs][Ws][3] (device).  Resize is Pillow's antialiased BILINEAR resample: the caller passes the
 * window / 22-bit fixed-point coefficient tables of its two passes (bounds[2*i] = first source index, bounds[2*i+1] = taps;
 * coeffs[i*ksize + t]), built as Pillow's precompute_coeffs / normalize_coeffs_8bpc do -- the resized bytes are bit-exact.
 * flip[b] != 0 mirrors image b; minmax / mean0 select the per-image scaling to [0,1] / [-1,1].  dst is NHWC fp32. */
size_t srgan_preprocess_workspace(int B, int crop_h, int out_h, int out_w);
int srgan_preprocess_u8(const unsigned char* src, int B, int Hs, int Ws, int top, int left, int crop_h, int crop_w,
                        int out_h, int out_w, const int* h_bounds, const int* h_coeffs, int h_ksize,
                        const int* v_bounds, const int* v_coeffs, int v_ksize, const unsigned char* flip,
                        int minmax, int mean0, float* dst, void* ws, size_t ws_bytes, void* stream);

/* ---- evaluation path (pyfiles/evaluation.py:13-110) ------------------------------------------------
 * The VGG19-bn feature extractor (evaluation.py:13-36: torchvision's vgg19_bn features + avgpool + classifier[:6]) runs its
 * 3x3 convolutions (BatchNorm folded in eval mode, bias + ReLU in the epilogue) and Linear layers on srgan_conv2d_fwd; the
 * one op it needs beyond the train step is nn.MaxPool2d(2, 2): */
int srgan_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* PRDC (evaluation.py:98-110 -> prdc.compute_prdc, prdc==0.2, Docker/requirements.txt:13), in the steps of that package:
 * compute_pairwise_distance: dist[i][j] = ||x_i - y_j||_2 for x[N][D], y[M][D] (row-major fp32);
 * get_kth_value: the k-th smallest entry of every row (1-based, k <= 16) -- with k = nearest_k + 1 on the self-distance
 *   matrix this is compute_nearest_neighbour_distances (the radii);
 * compute_prdc: out4 = {precision, recall, density, coverage} from dist[real][fake] and the two radius vectors. */
int srgan_pairwise_dist(const float* x, int N, const float* y, int M, int D, float* dist, void* stream);
int srgan_kth_smallest_rows(const float* dist, int N, int M, int k, float* out, void* stream);
size_t srgan_prdc_workspace(int N, int M);
int srgan_prdc_from_dist(const float* dist, int N, int M, const float* r_real, const float* r_fake, int nearest_k,
                         float* out4, void* ws, size_t ws_bytes, void* stream);

/* ---- collectives of the data-parallel step (SURVEY.md 8 e) -----------------------------------------------------------
 * What replaces torch.nn.DataParallel(net, devices=[0,1,2,3]) of notebook/05-train_Style-Restricted_GAN.ipynb:404-407,446
 * (scatter the batch, replicate the module, gather the outputs, reduce the gradients on device 0): one process per GPU with
 * persistent replicas, a bucketed in-place gradient all-reduce and an all-gather of the encoder's mu rows, over RCCL (xGMI).
 * librccl.so is bound lazily at the first call (srgan_comm_available() asks without a communicator); `comm` is an opaque
 * ncclComm_t.  The calls only enqueue on `stream` (capturable into a hipGraph on a communication stream); buffers are device
 * memory owned by the caller; rendezvous -- moving the 128-byte id from rank 0 to the others -- is the caller's.
 * Return values: 0, <0 invalid argument / library unavailable, 1000 + ncclResult_t for an RCCL error. */
int srgan_comm_available(void);
int srgan_comm_unique_id(void* id128);                          /* rank 0: ncclGetUniqueId */
int srgan_comm_init(const void* id128, int nranks, int rank, void** comm);
int srgan_comm_size(void* comm, int* nranks);
int srgan_comm_destroy(void* comm);
/* gradient bucket: in-place all-reduce of `count` elements (fp32, or bf16 when `bf16` = 1), sum or average over the ranks --
 * DataParallel's reduce_add of the replicas' gradients (torch/nn/parallel/_functions.py as used by the notebook's wrapper) */
int srgan_allreduce_bucket(void* comm, void* buf, long long count, int bf16, int average, void* stream);
/* mu rows of every rank, rank-major: the gather of the replicas' encoder outputs DataParallel performs before the batch-statistics
 * losses of pyfiles/util.py:455-553 see the GLOBAL batch */
int srgan_allgather_rows(void* comm, const float* rows, float* all_rows, long long count_per_rank, void* stream);

/* ---- launch timer for bench.py's roofline leg (no reference counterpart) ---------------------
 * While enabled every implicit-GEMM / weight-gradient launch is bracketed by HIP events on its own
 * stream and tagged with its algorithmic FLOPs (2*N*Ho*Wo*O*kh*kw*I).  Collect after a device sync. */
int srgan_prof_enable(int on);
int srgan_prof_num_kernels(void);
const char* srgan_prof_kernel_name(int kid);
int srgan_prof_collect(int kid, double* total_ms, long long* launches, double* total_flops);
int srgan_prof_num_slots(void);
int srgan_prof_slot(int index, int* kid, double* ms, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* SRGAN_HIP_H */

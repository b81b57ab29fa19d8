"""nn.Module surface of the reference's ``pyfiles/model.py`` over the gfx950 HIP kernels.

Same class names, constructor signatures, forward signatures / return structures and
``state_dict()`` keys, shapes and dtypes as the reference, so that
``05-train_Style-Restricted_GAN.ipynb``'s construction code and ``.pth`` checkpoints carry over.
torch.nn containers (nn.Conv2d, nn.Linear, ...) are used only as PARAMETER HOLDERS -- this keeps the
key names and PyTorch's default initialisation (the reference's ``weights_init`` is a no-op,
util.py:193-203) -- every ``forward`` below dispatches to hand-written kernels via ``srgan_amd.ops``.
Modules are built in the reference's construction order so that a given ``torch.manual_seed``
yields the same initial weights.

4-D activations returned by these modules are NHWC-dense tensors with logical NCHW shape.
"""
import functools

import numpy as np
import os
import torch
import torch.nn as nn

from . import _lib, ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, PAD_REFLECT, PAD_ZERO

__all__ = ["CBINorm2d", "get_norm_layer", "SingleResidualBlock", "SingleGenerator",
           "SingleDiscriminator_original", "SingleDiscriminator_original_multi", "SingleDiscriminator_solo",
           "SingleDiscriminator_solo_multi", "BasicBlock", "Encoder_original", "BasicBlock_classification",
           "Encoder", "Encoder_classifier", "MinMax"]

D_SLOPE = 0.01   # nn.LeakyReLU() default in the discriminators   (model.py:263,303)
E_SLOPE = 0.2    # nn.LeakyReLU(0.2) in the encoders               (model.py:357,418,454)


# ------------------------------------------------------------------------------------------------
# parameter holders with HIP forwards
# ------------------------------------------------------------------------------------------------
class _Conv2d(nn.Conv2d):
    """nn.Conv2d parameters; forward = implicit-GEMM MFMA kernel (optionally fused LeakyReLU)."""

    def forward(self, x, act=ACT_NONE, slope=0.0, in_slope=None, act_bwd_by_consumer=False, io16=None):
        """io16 (bf16 mode, ops.s2_io_applicable layers only): None = ordinary fp32 tensors; True / False = the input may be a
        bf16 tensor and the output is written as bf16 / fp32 (the generator's down path, ops.py "bf16 activation storage")."""
        if io16 is not None:
            return ops.conv2d_s2_io(x, self.weight, io16)
        mode = PAD_REFLECT if self.padding_mode == "reflect" else PAD_ZERO
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], mode, act, slope, in_slope, act_bwd_by_consumer)

    def s2_io_applicable(self, x):
        n, ci, hi, wi = x.shape
        return (self.bias is None and self.kernel_size == (4, 4) and self.stride == (2, 2) and self.padding == (1, 1)
                and self.padding_mode == "zeros" and ops.s2_io_applicable(n, ci, hi, wi, self.out_channels, self.weight, False))

    def act_io_applicable_shape(self, n, ci, hi, wi, act):
        return (self.bias is None and self.kernel_size == (4, 4) and self.stride == (2, 2) and self.padding == (1, 1)
                and self.padding_mode == "zeros" and ci == self.in_channels and ops.conv_act_io_applicable(n, ci, hi, wi, self.weight, act))

    def act_io_applicable(self, x, act):
        """bf16 mode: this 4x4 / stride-2 layer + activation runs with 16-bit tensors (ops.conv2d_act_io; the discriminator trunks)."""
        n, ci, hi, wi = x.shape
        return self.act_io_applicable_shape(n, ci, hi, wi, act)

    def forward_io(self, x, out_bf16=True):
        """bf16 mode, io_applicable layers: the input may be a bf16 tensor, the output is written as bf16 / fp32 (the style
        encoder's blocks, ops.py "16-bit activations around the generic convolutions")."""
        mode = PAD_REFLECT if self.padding_mode == "reflect" else PAD_ZERO
        return ops.conv2d_io(x, self.weight, self.padding[0], mode, out_bf16)

    def io_applicable(self, x):
        n, ci, hi, wi = x.shape
        mode = PAD_REFLECT if self.padding_mode == "reflect" else PAD_ZERO
        return (self.bias is None and self.stride == (1, 1) and self.padding[0] == self.padding[1]
                and ops.conv_io_applicable(n, ci, hi, wi, self.weight, self.padding[0], mode))


class _ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x, io16=None):
        if io16 is not None:
            return ops.conv_transpose2d_io(x, self.weight, io16)
        return ops.conv_transpose2d(x, self.weight, self.stride[0], self.padding[0])

    def s2_io_applicable(self, x):
        n, ci, hi, wi = x.shape
        return (self.bias is None and self.kernel_size == (4, 4) and self.stride == (2, 2) and self.padding == (1, 1)
                and self.output_padding == (0, 0) and ops.s2_io_applicable(n, ci, hi, wi, self.out_channels, self.weight, True))


class _Linear(nn.Linear):
    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class _LeakyReLU(nn.Module):
    """Parameter-free slot that keeps the reference's Sequential indices (convs at 0,2,4,...)."""

    def __init__(self, slope=0.01):
        super().__init__()
        self.negative_slope = slope

    def forward(self, x):
        return ops.activation(x, ACT_LRELU, self.negative_slope)


class _Tanh(nn.Module):
    def forward(self, x):
        return ops.tanh(x)


class _Softmax(nn.Module):
    """nn.Softmax() placeholder of classification_layer{1,2}: the softmax itself is fused into the
    head/loss kernels; called stand-alone it normalises over dim 1 like the reference's implicit dim."""

    def forward(self, x):
        z = x.reshape(x.shape[0], -1)
        _, q = ops.softmax_mse(z, torch.zeros(z.shape[0], dtype=torch.int64, device=z.device), 0.0)
        return q.view_as(x)


class _InstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False): no parameters, no buffers."""

    def __init__(self, num_features, affine=False):
        super().__init__()
        if affine:
            raise NotImplementedError("InstanceNorm2d(affine=True) is not used by the reference")
        self.num_features = num_features

    def forward(self, x, act=ACT_NONE, slope=0.0, out_bf16=False):
        if out_bf16 or x.dtype == torch.bfloat16:
            return ops.instance_norm_act_io(x, None, None, act, slope, 1e-5, out_bf16)
        return ops.instance_norm_act(x, None, None, None, act, slope)


class _AvgPool3s2(nn.Module):
    def forward(self, x):
        return ops.avgpool3s2(x)


# The two scales of a multi-scale discriminator (model.py:303-337, :339-346: discriminator1 on the image, discriminator2 at half
# the width on the 3x3 / stride-2 average of it) share nothing but the input.  The second scale is 1/16 of the first one's FLOPs
# in launches of 4-128 workgroups that cannot fill 256 CUs (1.07 GFLOP layers at 23-28 us each: 40-47 TFLOP/s), so it runs on a
# SIDE STREAM forked from the caller's: forward here, backward wherever autograd replays it (the engine runs a node's backward on
# its forward's stream and orders the streams with events) -- inside a captured train step the two scales become parallel
# branches of the hipGraph.  Same kernels, same operands, same accumulation order of the two input gradients (the engine's,
# by topological order): identical results.  ``SRGAN_NO_PARALLEL_SCALES=1`` keeps everything on one stream.
_PARALLEL_SCALES = not _lib.ab("SRGAN_NO_PARALLEL_SCALES")
_side_streams = {}


class _second_scale:
    """``with _second_scale(x) as fork: ...`` runs the block on the device's side stream after everything queued on the current
    one; ``fork.join(*tensors)`` (after the block) makes the current stream wait for it and tells the allocator where the
    block's results are used."""

    def __init__(self, x):
        self._on = _PARALLEL_SCALES and x.is_cuda
        if self._on:
            self._main = torch.cuda.current_stream(x.device)
            side = _side_streams.get(x.device.index)
            if side is None:
                side = _side_streams[x.device.index] = torch.cuda.Stream(x.device)
            self._side = side
            self._x = x
            ops.register_compute_streams(x.device, self._main, side)

    def __enter__(self):
        if self._on:
            self._side.wait_stream(self._main)
            self._ctx = torch.cuda.stream(self._side)
            self._ctx.__enter__()
            self._x.record_stream(self._side)
        return self

    def __exit__(self, *exc):
        if self._on:
            self._ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        if self._on:
            self._main.wait_stream(self._side)
            for t in tensors:
                t.record_stream(self._main)


# The shortcut of an encoder block (model.py:409-411, :427-431: AvgPool2d -> 1x1 conv, 1 / 18 of the block's FLOPs in launches
# of 8-20 us) shares only the block's input with the main path: the same fork, forward here and backward where autograd
# replays it.  ``SRGAN_NO_PARALLEL_SHORTCUT=1`` (experiment build) keeps it on the caller's stream.
_PARALLEL_SHORTCUT = not _lib.ab("SRGAN_NO_PARALLEL_SHORTCUT")


class _side_branch(_second_scale):
    def __init__(self, x):
        super().__init__(x)
        self._on = self._on and _PARALLEL_SHORTCUT and x.is_cuda


def _fork_input(x):
    """A leaf input that wants a gradient gets it from both scales; behind a view node of the caller's stream the two meet in an
    ordinary input buffer and the leaf's AccumulateGrad node sees one producer on its own stream (no stream-mismatch warning)."""
    return x.view_as(x) if (_PARALLEL_SCALES and x.is_cuda and x.is_leaf and x.requires_grad) else x


class _AvgPool2(nn.Module):
    def forward(self, x):
        if x.dtype == torch.bfloat16:        # a 16-bit conv output of the bf16 mode: the mean goes back to the fp32 stream
            return ops.avgpool2_io(x, False)
        return ops.avgpool2(x)


def _block_io16(x, conv1, conv2):
    """Can an encoder block (norm -> conv1 -> norm -> conv2 -> pool) keep its four intermediates in bf16?  Both convolutions and
    both norms must be served for THIS shape (each by its own geometry)."""
    if not x.is_cuda or ops.get_compute_dtype() != "bf16":
        return False
    # (asked on every forward: the answer depends on the geometry and the switches only)
    key = (tuple(x.shape), tuple(conv1.weight.shape), conv1.weight.stride(), tuple(conv2.weight.shape), conv2.weight.stride(),
           conv1.padding_mode, conv1.padding, conv1.stride, conv2.padding, conv2.stride, conv1.bias is None, conv2.bias is None,
           ops.STORAGE_BF16, ops.pack_cache_active())
    hit = _io16_memo.get(key)
    if hit is None:
        n, c, h, w = x.shape
        hit = _io16_memo[key] = bool((c & 3) == 0 and (conv2.out_channels & 3) == 0 and ops.norm_io_applicable(n, c, h, w)
                                     and conv1.io_applicable(x) and conv2.io_applicable(x))
    return hit


_io16_memo = {}


# ------------------------------------------------------------------------------------------------
# central-biasing instance norm        (reference: _CBINorm / CBINorm2d, model.py:12-73)
# ------------------------------------------------------------------------------------------------
class CBINorm2d(nn.Module):
    def __init__(self, num_features, num_con=8, eps=1e-5, momentum=0.1, affine=False, track_running_stats=False):
        super().__init__()
        if track_running_stats:
            raise NotImplementedError("CBINorm2d(track_running_stats=True) is not used by the reference")
        self.num_features, self.num_con, self.eps, self.momentum = num_features, num_con, eps, momentum
        self.affine, self.track_running_stats = affine, track_running_stats
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.ConBias = nn.Sequential(_Linear(num_con, num_features), _Tanh())

    def _check_input_dim(self, input):
        if input.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(input.dim()))

    def forward(self, input, ConInfor, act=ACT_NONE, slope=0.0, res=None, out_bf16=False):
        """(IN(x) + tanh(Linear(c))) * weight + bias, with optional fused activation / residual.  out_bf16 / a bf16 input: the
        bf16 mode's 16-bit activation storage (no residual there)."""
        self._check_input_dim(input)
        scale, shift = self.scale_shift(ConInfor, input.device)
        if out_bf16 or input.dtype == torch.bfloat16:
            if res is not None:
                raise ValueError("CBINorm2d: no residual on the 16-bit storage path")
            return ops.instance_norm_act_io(input, scale, shift, act, slope, self.eps, out_bf16)
        return ops.instance_norm_act(input, scale, shift, res, act, slope, self.eps)

    def scale_shift(self, ConInfor, device):
        """Per-sample (scale, shift) [N, C] of the layer: y = IN(x) * scale + shift."""
        if isinstance(ConInfor, PrecomputedCon):       # the network computed every layer's affine in one launch
            return ConInfor.affine[id(self)]
        return ops.cbin_affine(ConInfor, *self.affine_params(device))

    def affine_params(self, device):
        lin = self.ConBias[0]
        if self.affine:
            return lin.weight, lin.bias, self.weight, self.bias
        return lin.weight, lin.bias, torch.ones(self.num_features, device=device), torch.zeros(self.num_features, device=device)


class PrecomputedCon:
    """Style code plus the (scale, shift) pair of every central-biasing layer of a network, made by ONE launch
    (ops.cbin_affine_multi): the affine depends on the code and the layer's parameters only."""

    def __init__(self, c, layers):
        self.c = c
        affs = ops.cbin_affine_multi(c, [l.affine_params(c.device) for l in layers])
        self.affine = {id(l): a for l, a in zip(layers, affs)}


def get_norm_layer(layer_type='instance', num_con=2):
    """(norm_layer, c_norm_layer) factories, as model.py:173-182."""
    if layer_type == 'instance':
        norm_layer = functools.partial(_InstanceNorm2d, affine=False)
        c_norm_layer = functools.partial(CBINorm2d, affine=True, num_con=num_con)
    elif layer_type == 'batch':
        # dead code in the reference (CBBNorm2d._load_from_state_dict references an undefined name,
        # model.py:163; no notebook passes norm_type="batch") -- out of scope, SURVEY.md section 2 row 1
        raise NotImplementedError('normalization layer [batch] is out of scope of the MI355X path')
    else:
        raise NotImplementedError('normalization layer [%s] is not found' % layer_type)
    return norm_layer, c_norm_layer


# ------------------------------------------------------------------------------------------------
# Generator                              (reference: model.py:188-249)
# ------------------------------------------------------------------------------------------------
class SingleResidualBlock(nn.Module):
    def __init__(self, nch, c_norm_layer):
        super().__init__()
        self.c1 = _Conv2d(nch, nch, kernel_size=3, stride=1, padding=1, bias=False)
        self.cn1 = c_norm_layer(nch)
        self.c2 = _Conv2d(nch, nch, kernel_size=3, stride=1, padding=1, bias=False)
        self.cn2 = c_norm_layer(nch)

    def forward(self, x):
        data, con = x[0], x[1]
        if data.is_cuda and isinstance(self.cn1, CBINorm2d) and self.cn1.affine:
            s1, h1 = self.cn1.scale_shift(con, data.device)
            s2, h2 = self.cn2.scale_shift(con, data.device)
            if ops.res_block_fusable(data, self.c1.weight, self.c2.weight, s1, s2):
                # the whole block as one autograd node: neither the normalised activation nor the gradients w.r.t. the two
                # conv outputs are ever written (ops._ResBlockFn)
                self.cn1._check_input_dim(data)
                return ops.residual_block(data, s1, h1, s2, h2, self.c1.weight, self.c2.weight, self.cn1.eps), con
            if ops.res_block_bf16_fusable(data, self.c1.weight, self.c2.weight, s1, s2):
                # bf16 mode: one node, its intermediates stored as bf16 (ops._ResBlockBf16Fn)
                self.cn1._check_input_dim(data)
                return ops.residual_block_bf16(data, s1, h1, s2, h2, self.c1.weight, self.c2.weight, self.cn1.eps), con
        # the block input feeds c1 and the skip connection: conv2d_skip routes the skip path's gradient into c1's
        # input-gradient kernel (added in its epilogue) instead of a separate accumulation pass
        y1, skip = ops.conv2d_skip(data, self.c1.weight, None, self.c1.stride[0], self.c1.padding[0], PAD_ZERO)
        if ops.norm_act_conv_fusable(y1, self.c2.weight):
            # cn1 + ReLU + c2 without the normalised tensor: the norm kernel writes c2's transformed-input image directly
            self.cn1._check_input_dim(y1)
            scale, shift = self.cn1.scale_shift(con, y1.device)
            y2 = ops.instance_norm_act_conv(y1, scale, shift, self.c2.weight, ACT_RELU, 0.0, self.cn1.eps)
        else:
            y2 = self.c2(self.cn1(y1, con, ACT_RELU))
        return self.cn2(y2, con, ACT_NONE, 0.0, skip), con


class SingleGenerator(nn.Module):
    def __init__(self, nch_in, nch, reduce=2, num_cls=3, res_num=6, norm_type="instance", num_con=2, nch_out=None):
        super().__init__()
        if nch_out is None:
            nch_out = nch_in
        norm_layer, c_norm_layer = get_norm_layer(layer_type=norm_type, num_con=num_con)
        self.num_cls = num_cls
        k, s, p = 2 * reduce, reduce, int(reduce / 2)

        convs = [_Conv2d(nch_in, nch, kernel_size=7, stride=1, padding=3, bias=False)]
        cnorms = [c_norm_layer(nch)]
        for i in range(num_cls):
            convs.append(_Conv2d(nch * 2 ** i, nch * 2 ** (i + 1), kernel_size=k, stride=s, padding=p, bias=False))
            cnorms.append(c_norm_layer(nch * 2 ** (i + 1)))
        self.down_convs = nn.ModuleList(convs)
        self.down_cnorms = nn.ModuleList(cnorms)

        self.resBlocks = nn.Sequential(*[SingleResidualBlock(nch * 2 ** num_cls, c_norm_layer) for _ in range(res_num)])

        ups = [_ConvTranspose2d(nch * 2 ** num_cls, nch * 2 ** (num_cls - 1), kernel_size=k, stride=s, padding=p, bias=False)]
        norms = [norm_layer(nch * 2 ** (num_cls - 1))]
        for i in range(1, num_cls)[::-1]:
            ups.append(_ConvTranspose2d(nch * 2 ** i, nch * 2 ** (i - 1), kernel_size=k, stride=s, padding=p, bias=False))
            norms.append(norm_layer(nch * 2 ** (i - 1)))
        ups.append(_Conv2d(nch, nch_out, kernel_size=7, stride=1, padding=3, bias=False))
        self.up_convs = nn.ModuleList(ups)
        self.up_norms = nn.ModuleList(norms)

    def forward(self, x, c):
        if c.is_cuda and not _lib.ab("SRGAN_NO_CBIN_MULTI"):
            c = PrecomputedCon(c, list(self.down_cnorms) + [n for blk in self.resBlocks for n in (blk.cn1, blk.cn2)])
        # bf16 mode: the tensors between the stride-2 convolutions and their norms are kept in bf16 where the patch kernels and the
        # 16-bit norm kernels serve the shapes (ops.py, "bf16 activation storage"); everything else -- and every other mode --
        # takes the ordinary fp32 path.  io[i]: down conv i (i >= 1) takes and writes bf16.
        n, _, h, w = x.shape
        io = [False] * (self.num_cls + 2)
        if x.is_cuda and ops.get_compute_dtype() == "bf16":
            hh, ww = h, w
            for i in range(1, self.num_cls + 1):
                cv = self.down_convs[i]
                probe = torch.empty((n, cv.in_channels, hh, ww), device="meta")
                io[i] = (cv.s2_io_applicable(probe) and ops.norm_io_applicable(n, cv.in_channels, hh, ww)
                         and ops.norm_io_applicable(n, cv.out_channels, hh // 2, ww // 2))
                hh, ww = hh // 2, ww // 2
        # round 6: the 64-channel side of the two 7x7 RGB layers too -- the first conv writes bf16 for norm 0, the last norm
        # writes bf16 for the RGB head (ops.conv2d_act_io with (k, stride, pad) = (7, 1, 3); the 3-channel side stays fp32)
        cv0, cvl = self.down_convs[0], self.up_convs[-1]
        rgb16 = [x.is_cuda and cv.kernel_size == (7, 7) and cv.bias is None and
                 ops.conv_act_io_applicable(n, cv.in_channels, h, w, cv.weight, ACT_NONE, 7, 1, 3) and
                 ops.norm_io_applicable(n, ch, h, w) for cv, ch in ((cv0, cv0.out_channels), (cvl, cvl.in_channels))]
        for i in range(self.num_cls + 1):
            if i == 0 and rgb16[0]:
                y = ops.conv2d_act_io(x, cv0.weight, ACT_NONE, 0.0, True, 7, 1, 3)
            else:
                y = self.down_convs[i](x, io16=True) if io[i] else self.down_convs[i](x)
            x = self.down_cnorms[i](y, c, ACT_RELU, out_bf16=io[i + 1])
        x = self.resBlocks([x, c])[0]
        n, _, h, w = x.shape
        iot = [False] * (self.num_cls + 1)
        nio = [False] * (self.num_cls + 1)       # norm i can run on the 16-bit I/O kernels (its OWN shape: ADVICE r4)
        if x.is_cuda and ops.get_compute_dtype() == "bf16":
            hh, ww = h, w
            for i in range(self.num_cls):
                cv = self.up_convs[i]
                probe = torch.empty((n, cv.in_channels, hh, ww), device="meta")
                nio[i] = ops.norm_io_applicable(n, cv.out_channels, 2 * hh, 2 * ww)
                iot[i] = cv.s2_io_applicable(probe) and nio[i]
                hh, ww = 2 * hh, 2 * ww
        for i in range(self.num_cls):
            y = self.up_convs[i](x, io16=True) if iot[i] else self.up_convs[i](x)
            # norm i writes bf16 for a 16-bit up conv i + 1 only if norm i itself is served (with iot[i] false and iot[i + 1] true
            # -- a 16 x 64 input: the trunk map is 4 x 16 -- the fp32-in / bf16-out norm used to run unchecked)
            x = self.up_norms[i](y, ACT_RELU, out_bf16=(iot[i + 1] and nio[i]) or (i == self.num_cls - 1 and rgb16[1] and nio[i]))
        if x.dtype == torch.bfloat16:
            return ops.tanh(ops.conv2d_act_io(x, cvl.weight, ACT_NONE, 0.0, False, 7, 1, 3))
        return ops.tanh(self.up_convs[-1](x))


# ------------------------------------------------------------------------------------------------
# Discriminators                        (reference: model.py:255-346)
# ------------------------------------------------------------------------------------------------
def _d_trunk_layers(nch_in, nch, reduce, num_cls):
    layers = [_Conv2d(nch_in, nch, kernel_size=4, stride=2, padding=1, bias=False), _LeakyReLU(D_SLOPE)]
    dim_in = nch
    for _ in range(1, num_cls):
        dim_out = min(dim_in * 2, nch * 8)
        layers.append(_Conv2d(dim_in, dim_out, kernel_size=2 * reduce, stride=reduce, padding=int(reduce / 2), bias=False))
        layers.append(_LeakyReLU(D_SLOPE))
        dim_in = dim_out
    return layers, dim_in


_CHAIN_ACT_BWD = not _lib.ab("SRGAN_NO_CHAIN_ACT_BWD")


def _conv_out_hw(m, x):
    kh, kw = m.kernel_size
    return ((x.shape[2] + 2 * m.padding[0] - kh) // m.stride[0] + 1, (x.shape[3] + 2 * m.padding[1] - kw) // m.stride[1] + 1)


def _run_trunk(seq, x):
    """conv + LeakyReLU pairs run as one fused kernel each; a trailing bias conv runs plain.  Inside the chain a pair's
    LeakyReLU backward is done by the NEXT conv's input-gradient kernel (ops._Conv2dFn: the intermediate tensors have no other
    consumer); the last pair -- its output leaves the trunk -- keeps its own."""
    mods = list(seq)
    steps = []                                   # (conv, slope or None)
    i = 0
    while i < len(mods):
        if i + 1 < len(mods) and isinstance(mods[i + 1], _LeakyReLU):
            steps.append((mods[i], mods[i + 1].negative_slope))
            i += 2
        else:
            steps.append((mods[i], None))
            i += 1
    prev_slope = None                            # slope of the previous pair when its backward was handed to this conv
    for j, (m, slope) in enumerate(steps):
        # bf16 mode, round 6: a conv 4x4 / stride 2 + LeakyReLU pair whose three directions take bf16 tensors runs on them
        # (ops.conv2d_act_io); its output is bf16 when the next pair reads it the same way (the heads take fp32)
        if slope is not None and prev_slope is None and isinstance(m, _Conv2d) and m.act_io_applicable(x, ACT_LRELU):
            nxt = steps[j + 1] if j + 1 < len(steps) else None
            out16 = bool(nxt is not None and nxt[1] is not None and isinstance(nxt[0], _Conv2d)
                         and nxt[0].act_io_applicable_shape(x.shape[0], m.out_channels, x.shape[2] // 2, x.shape[3] // 2, ACT_LRELU))
            x = ops.conv2d_act_io(x, m.weight, ACT_LRELU, slope, out16)
            prev_slope = None
            continue
        hand_on = (_CHAIN_ACT_BWD and slope is not None and j + 1 < len(steps) and isinstance(m, _Conv2d)
                   and isinstance(steps[j + 1][0], _Conv2d) and torch.is_grad_enabled()
                   and not (steps[j + 1][1] is not None
                            and steps[j + 1][0].act_io_applicable_shape(x.shape[0], m.out_channels, *_conv_out_hw(m, x), ACT_LRELU)))
        if isinstance(m, _Conv2d):
            x = m(x, ACT_LRELU if slope is not None else ACT_NONE, slope or 0.0, prev_slope, hand_on)
        elif slope is not None:
            x = m(x, ACT_LRELU, slope)
        else:
            x = m(x)
        prev_slope = slope if hand_on else None
    return x


class SingleDiscriminator_original(nn.Module):
    def __init__(self, nch_in, nch, reduce=2, num_cls=3, norm_type="instance", num_con=2):
        super().__init__()
        self.num_cls = num_cls
        layers, dim_in = _d_trunk_layers(nch_in, nch, reduce, num_cls)
        layers.append(_Conv2d(dim_in, 1, kernel_size=4, stride=1, padding=1, bias=True))
        self.down_convs = nn.Sequential(*layers)

    def forward(self, x):
        return _run_trunk(self.down_convs, x)


class SingleDiscriminator_original_multi(nn.Module):
    def __init__(self, nch_in, nch, reduce=2, num_cls=3, norm_type="instance", num_con=2):
        super().__init__()
        self.discriminator1 = SingleDiscriminator_original(nch_in, nch, reduce, num_cls, norm_type, num_con)
        self.down = _AvgPool3s2()
        self.discriminator2 = SingleDiscriminator_original(nch_in, nch // 2, reduce, num_cls, norm_type, num_con)

    def forward(self, x):
        x = _fork_input(x)
        with _second_scale(x) as fork:
            d2 = self.discriminator2(self.down(x))
        d1 = self.discriminator1(x)
        fork.join(d2)
        return [d1, d2]


class SingleDiscriminator_solo(nn.Module):
    def __init__(self, nch_in, nch, reduce=2, num_cls=3, norm_type="instance", num_con=2):
        super().__init__()
        self.num_cls = num_cls
        layers, _ = _d_trunk_layers(nch_in, nch, reduce, num_cls)
        self.down_convs = nn.Sequential(*layers)

    def forward(self, x):
        return _run_trunk(self.down_convs, x)


class SingleDiscriminator_solo_multi(nn.Module):
    def __init__(self, nch_in, nch, reduce=2, num_cls=3, norm_type="instance", n_class=4):
        super().__init__()
        self.n_class = n_class
        self.discriminator1 = SingleDiscriminator_solo(nch_in, nch, reduce, num_cls, norm_type, None)
        self.down = _AvgPool3s2()
        self.discriminator2 = SingleDiscriminator_solo(nch_in, nch // 2, reduce, num_cls, norm_type, None)
        dim_in = min(nch * 2 ** num_cls, nch * 8)
        self.last_layer1 = _Conv2d(dim_in, 1, kernel_size=4, stride=1, padding=1, bias=True)
        self.last_layer2 = _Conv2d(dim_in // 2, 1, kernel_size=4, stride=1, padding=1, bias=True)
        self.classification_layer1 = nn.Sequential(_Conv2d(dim_in, n_class, kernel_size=8, stride=1, padding=0, bias=True), _Softmax())
        self.classification_layer2 = nn.Sequential(_Conv2d(dim_in // 2, n_class, kernel_size=4, stride=1, padding=0, bias=True), _Softmax())

    def forward_logits(self, x):
        """-> ([out1, out2], [logits1, logits2]); logits are [B, n_class] pre-softmax (used by the
        fused softmax+MSE loss kernel in the trainer)."""
        x = _fork_input(x)
        with _second_scale(x) as fork:
            d2 = self.discriminator2(self.down(x))
            o2 = self.last_layer2(d2)
            z2 = self.classification_layer2[0](d2).reshape(-1, self.n_class)
        d1 = self.discriminator1(x)
        o1 = self.last_layer1(d1)
        z1 = self.classification_layer1[0](d1).reshape(-1, self.n_class)
        fork.join(o2, z2)
        return [o1, o2], [z1, z2]

    def forward(self, x):
        outs, logits = self.forward_logits(x)
        return outs, [_softmax_rows(z) for z in logits]


class _SoftmaxRowsFn(torch.autograd.Function):
    """Row softmax with its Jacobian-vector product, both through the fused softmax kernel."""

    @staticmethod
    def forward(ctx, z):
        _, q = ops.softmax_mse(z, torch.zeros(z.shape[0], dtype=torch.int64, device=z.device), 0.0)
        ctx.save_for_backward(q)
        return q

    @staticmethod
    def backward(ctx, gq):
        (q,) = ctx.saved_tensors
        return q * (gq - (q * gq).sum(dim=1, keepdim=True))


def _softmax_rows(z):
    return _SoftmaxRowsFn.apply(z)


# ------------------------------------------------------------------------------------------------
# Encoders                               (reference: model.py:352-507)
# ------------------------------------------------------------------------------------------------
class BasicBlock_classification(nn.Module):
    def __init__(self, nch_in, nch_out, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(nch_in)
        self.nl1 = _LeakyReLU(E_SLOPE)
        self.conv1 = _Conv2d(nch_in, nch_in, kernel_size=3, stride=1, padding=1, bias=False, padding_mode="reflect")
        self.norm2 = norm_layer(nch_in)
        self.nl2 = _LeakyReLU(E_SLOPE)
        self.cmp = nn.Sequential(
            _Conv2d(nch_in, nch_out, kernel_size=3, stride=1, padding=1, bias=False, padding_mode="reflect"),
            _AvgPool2())
        self.shortcut = nn.Sequential(
            _AvgPool2(),
            _Conv2d(nch_in, nch_out, kernel_size=1, stride=1, padding=0, bias=True))

    def forward(self, input):
        x = input
        with _side_branch(x) as fork:        # pool + 1x1 conv: a few small launches beside the main path's large ones
            sc = self.shortcut(x)
        if _block_io16(x, self.conv1, self.cmp[0]):
            h = self.conv1.forward_io(self.norm1(x, ACT_LRELU, self.nl1.negative_slope, out_bf16=True))
            h = self.cmp[0].forward_io(self.norm2(h, ACT_LRELU, self.nl2.negative_slope, out_bf16=True))
            h = self.cmp[1](h)
        else:
            h = self.conv1(self.norm1(x, ACT_LRELU, self.nl1.negative_slope))
            h = self.cmp(self.norm2(h, ACT_LRELU, self.nl2.negative_slope))
        fork.join(sc)
        return ops.add(h, sc)


class BasicBlock(nn.Module):
    def __init__(self, nch_in, nch_out, c_norm_layer=None):
        super().__init__()
        self.cnorm1 = c_norm_layer(nch_in)
        self.nl1 = _LeakyReLU(E_SLOPE)
        self.conv1 = _Conv2d(nch_in, nch_in, kernel_size=3, stride=1, padding=1, bias=False, padding_mode="reflect")
        self.cnorm2 = c_norm_layer(nch_in)
        self.nl2 = _LeakyReLU(E_SLOPE)
        self.cmp = nn.Sequential(
            _Conv2d(nch_in, nch_out, kernel_size=3, stride=1, padding=1, bias=False, padding_mode="reflect"),
            _AvgPool2())
        self.shortcut = nn.Sequential(
            _AvgPool2(),
            _Conv2d(nch_in, nch_out, kernel_size=1, stride=1, padding=0, bias=True))

    def forward(self, input):
        x, d = input
        with _side_branch(x) as fork:
            sc = self.shortcut(x)
        if _block_io16(x, self.conv1, self.cmp[0]):
            h = self.conv1.forward_io(self.cnorm1(x, d, ACT_LRELU, self.nl1.negative_slope, out_bf16=True))
            h = self.cmp[0].forward_io(self.cnorm2(h, d, ACT_LRELU, self.nl2.negative_slope, out_bf16=True))
            h = self.cmp[1](h)
        else:
            h = self.conv1(self.cnorm1(x, d, ACT_LRELU, self.nl1.negative_slope))
            h = self.cmp(self.cnorm2(h, d, ACT_LRELU, self.nl2.negative_slope))
        fork.join(sc)
        return [ops.add(h, sc), d]


def host_to_device(t, device):
    """Small host tensor -> device without stalling the host: a pageable ``.to(device)`` makes the host wait until the stream
    has drained (the GPU then idles while the next launches are being queued); a copy out of PyTorch's cached pinned
    allocator is asynchronous and the allocator keeps the staging block alive until the copy has run."""
    device = torch.device(device)
    if device.type != "cuda" or t.device.type != "cpu" or _lib.ab("SRGAN_SYNC_H2D"):
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def _cpu_normal_like(t):
    """Noise from the CPU default generator, then moved -- the reference's reparametrize does
    ``torch.FloatTensor(size).normal_().to(device)`` (model.py:461), so seeds reproduce."""
    return host_to_device(torch.FloatTensor(t.size()).normal_(), t.device)


class _ReparamFn(torch.autograd.Function):
    """eps * exp(logvar/2) + mu on [B, ndim]: one launch forward, one backward (``srgan_reparam_fwd / _bwd``; the reference's
    chain of elementwise passes rounds every product and sum on its own, and so do the kernels)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        ops._require_gpu(mu, "reparametrize")
        mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
        out, std = torch.empty_like(mu), torch.empty_like(mu)
        _lib.check(_lib.load().srgan_reparam_fwd(ops._ptr(mu), ops._ptr(logvar), ops._ptr(eps), ops._ptr(out), ops._ptr(std),
                                                 mu.numel(), ops._stream()), "reparam_fwd")
        ctx.save_for_backward(eps, std)
        return out

    @staticmethod
    def backward(ctx, g):
        eps, std = ctx.saved_tensors
        g = g.contiguous()
        dlogvar = torch.empty_like(std)
        _lib.check(_lib.load().srgan_reparam_bwd(ops._ptr(g), ops._ptr(eps), ops._ptr(std), ops._ptr(dlogvar), g.numel(),
                                                 ops._stream()), "reparam_bwd")
        return g, dlogvar, None


class _EncoderBase(nn.Module):
    def reparametrize(self, mu, logvar):
        return _ReparamFn.apply(mu, logvar, _cpu_normal_like(mu))

    def reparam_with(self, mu, logvar, eps):
        """reparametrize with caller-supplied N(0, I) noise (already on the device)."""
        return _ReparamFn.apply(mu, logvar, eps)


class Encoder_original(_EncoderBase):
    def __init__(self, nch_in, nch_out, nch=64, num_cls=3, norm_type="instance", num_con=2, device="cpu"):
        super().__init__()
        _, c_norm_layer = get_norm_layer(layer_type=norm_type, num_con=num_con)
        self.num_cls, self.device = num_cls, device
        self.first_layer = _Conv2d(nch_in, nch, kernel_size=7, stride=2, padding=1, bias=True)
        blocks, in_nch = [], nch
        for _ in range(num_cls):
            blocks.append(BasicBlock(in_nch, in_nch * 2, c_norm_layer))
            in_nch *= 2
        self.layers = nn.Sequential(*blocks)
        self.last_layer = nn.Sequential(_LeakyReLU(E_SLOPE), nn.AdaptiveAvgPool2d(1))
        self.fcmean = _Linear(in_nch, nch_out)
        self.fcvar = _Linear(in_nch, nch_out)

    def forward(self, x, c):
        h = self.layers([self.first_layer(x), c])[0]
        feat = ops.lrelu_global_avgpool(h, self.last_layer[0].negative_slope)
        mu, logvar = self.fcmean(feat), self.fcvar(feat)
        return self.reparametrize(mu, logvar), mu, logvar


class Encoder(_EncoderBase):
    def __init__(self, nch_in, nch_out, nch=64, num_cls=3, norm_type="instance", num_con=2, device="cpu"):
        super().__init__()
        norm_layer, _ = get_norm_layer(layer_type=norm_type, num_con=num_con)
        self.num_cls, self.device = num_cls, device
        self.first_layer = _Conv2d(nch_in, nch, kernel_size=7, stride=2, padding=1, bias=True)
        blocks, in_nch = [], nch
        for _ in range(num_cls):
            blocks.append(BasicBlock_classification(in_nch, in_nch * 2, norm_layer))
            in_nch *= 2
        self.layers = nn.Sequential(*blocks)
        self.last_layer = nn.Sequential(_LeakyReLU(E_SLOPE), nn.AdaptiveAvgPool2d(1))
        self.fcmean = _Linear(in_nch, nch_out)
        self.fcvar = _Linear(in_nch, nch_out)
        self.fcclass = _Linear(in_nch, num_con)

    def freeze_melt(self, classifier_layers, mode="freeze"):
        """Toggle requires_grad of the parameters whose state_dict key is listed (model.py:465-472)."""
        keys = list(self.state_dict().keys())
        for i, param in enumerate(self.parameters()):
            if keys[i] in classifier_layers:
                if mode == "freeze":
                    param.requires_grad = False
                elif mode == "melt":
                    param.requires_grad = True

    def features(self, x):
        h = self.layers(self.first_layer(x))
        return ops.lrelu_global_avgpool(h, self.last_layer[0].negative_slope)

    def forward(self, x):
        feat = self.features(x)
        mu, logvar = self.fcmean(feat), self.fcvar(feat)
        c_code = self.reparametrize(mu, logvar)
        class_output = self.fcclass(feat)
        return c_code, mu, logvar, class_output, None


class Encoder_classifier(nn.Module):
    """Pre-training twin of Encoder (keys = Encoder's minus fcmean/fcvar); softmax class output."""

    def __init__(self, nch_in, nch_out, nch=64, num_cls=3, norm_type="instance", num_con=2):
        super().__init__()
        norm_layer, _ = get_norm_layer(layer_type=norm_type, num_con=num_con)
        self.num_cls = num_cls
        self.first_layer = _Conv2d(nch_in, nch, kernel_size=7, stride=2, padding=1, bias=True)
        blocks, in_nch = [], nch
        for _ in range(num_cls):
            blocks.append(BasicBlock_classification(in_nch, in_nch * 2, norm_layer))
            in_nch *= 2
        self.layers = nn.Sequential(*blocks)
        self.last_layer = nn.Sequential(_LeakyReLU(E_SLOPE), nn.AdaptiveAvgPool2d(1))
        self.fcclass = _Linear(in_nch, num_con)

    def forward(self, x):
        h = self.layers(self.first_layer(x))
        feat = ops.lrelu_global_avgpool(h, self.last_layer[0].negative_slope)
        return _softmax_rows(self.fcclass(feat))


class MinMax(object):
    """Per-image min-max scaling to [0,1] or [-1,1] (reference util.py:107-155; the notebooks import it
    from ``model``).  Host-side data-pipeline transform, not part of the GPU hot path."""

    def __init__(self, mean0=True):
        self.mean0 = mean0

    def __call__(self, img):
        x = img.detach().cpu().numpy() if torch.is_tensor(img) else np.asarray(img)
        lo, hi = x.min(), x.max()
        out = (x - lo) / (hi - lo + 1e-8)
        if self.mean0:
            out = out * 2 - 1
        return torch.Tensor(out)

    def __repr__(self):
        return self.__class__.__name__

"""torch.autograd.Function wrappers over the C ABI of libsrgan_hip.so.

PyTorch is used for device memory, streams and the autograd graph only; every forward and
backward computation below is a hand-written gfx950 kernel reached through ctypes.

Layout: 4-D activations are *NHWC-dense* tensors with logical NCHW shape (what PyTorch calls
channels_last), so they interoperate with callers written against the reference's NCHW API.

Stale-graph semantics: convolution / transposed-convolution weights
are kept on the autograd context by reference and read when backward RUNS (torch 1.4 behaviour
the reference's train step relies on, util_notebook.py:664-690); activations, norm statistics
and the CBIN gamma copy are the forward-time values.
"""
import ctypes
import weakref

import torch
from torch.autograd import Function

from . import _lib
from ._lib import ConvDesc

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
PAD_ZERO, PAD_REFLECT = 0, 1

_workspaces = {}
import os as _os
_DEBUG_FRESH_WS = _lib.ab('SRGAN_DEBUG_FRESH_WS')
_DEBUG_CLONE = _lib.ab('SRGAN_DEBUG_CLONE')


def _require_gpu(t, what):
    if not t.is_cuda:
        raise _lib.SrganHipError(f"{what}: tensor is on {t.device}; srgan_amd runs on the MI355X HIP path only "
                                 "(no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.SrganHipError(f"{what}: expected float32, got {t.dtype}")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def workspace(device, nbytes):
    """Cached scratch buffer (per device); grown on demand, reused in stream order."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    if _DEBUG_FRESH_WS:
        return torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes * 1.25), 1 << 22), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# Streams that carry parts of one backward pass (model._second_scale: the discriminator's second scale runs on a side stream).
# Code that reads gradients from INSIDE the pass -- the eager data-parallel reducer sends a bucket from the hook of its last
# parameter, on whatever stream that parameter's node ran -- first makes its stream wait for the others.
_compute_streams = {}


def register_compute_streams(device, *streams):
    known = _compute_streams.setdefault(device.index, [])
    for s in streams:
        if all(s != k for k in known):
            known.append(s)
        if len(known) > 8:          # callers' streams come and go (tests); keep the recent ones
            del known[0]


def wait_compute_streams(device):
    cur = torch.cuda.current_stream(device)
    for s in _compute_streams.get(device.index, ()):
        if s != cur:
            cur.wait_stream(s)


def upload_small(blob, device, out=None):
    """bytes / bytearray (a pointer table of a few KB, multiple of 4 bytes) -> device uint8 tensor.  The bytes travel in the
    arguments of a tiny kernel (C ABI ``srgan_upload_small``): no pinned staging buffer to keep alive, and inside a captured
    hipGraph the node stores the bytes, so a replay rewrites the same table instead of re-reading a host address."""
    n = len(blob)
    if out is None:
        out = torch.empty(n, dtype=torch.uint8, device=device)
    elif out.numel() < n:
        raise _lib.SrganHipError("upload_small: destination too small")
    buf = (ctypes.c_char * n).from_buffer_copy(bytes(blob))
    _lib.check(_lib.load().srgan_upload_small(_ptr(out), ctypes.byref(buf), n, _stream()), "upload_small")
    return out


def nhwc_empty(n, c, h, w, device, dtype=torch.float32):
    return torch.empty((n, h, w, c), dtype=dtype, device=device).permute(0, 3, 1, 2)


def is_nhwc_dense(t):
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def to_nhwc(t):
    """Return an NHWC-dense tensor with the same logical NCHW shape/values (HIP repack kernel)."""
    if t.dim() != 4:
        raise ValueError('expected 4D input (got {}D input)'.format(t.dim()))
    if t.dtype == torch.bfloat16 and t.is_cuda and is_nhwc_dense(t):      # a 16-bit activation of the bf16 mode (already dense)
        return t
    _require_gpu(t, "to_nhwc")
    if is_nhwc_dense(t):
        return t
    if t.dtype != torch.float32:
        raise _lib.SrganHipError(f"to_nhwc: a {t.dtype} tensor must already be NHWC-dense (the repack kernel is fp32)")
    t = t.contiguous()
    n, c, h, w = t.shape
    out = nhwc_empty(n, c, h, w, t.device)
    _lib.check(_lib.load().srgan_nchw_to_nhwc(_ptr(t), _ptr(out), n, c, h, w, _stream()), "nchw_to_nhwc")
    return out


def to_nchw(t):
    """NCHW-contiguous copy of an NHWC-dense tensor (HIP repack kernel)."""
    _require_gpu(t, "to_nchw")
    if t.is_contiguous():
        return t
    t = to_nhwc(t)
    n, c, h, w = t.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=t.device)
    _lib.check(_lib.load().srgan_nhwc_to_nchw(_ptr(t), _ptr(out), n, c, h, w, _stream()), "nhwc_to_nchw")
    return out


def cat_batch(tensors):
    """Concatenate NHWC-dense activations along the batch dimension (stays NHWC-dense, differentiable).
    Every layer of G / D / E is per-sample (instance norm, no batch norm), so running two batches as one is exact
    and gives the GEMM kernels twice the rows per launch."""
    tensors = [to_nhwc(t) for t in tensors]
    return torch.cat([t.permute(0, 2, 3, 1) for t in tensors], 0).permute(0, 3, 1, 2)


def _dense2d(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------
def _conv_desc(n, hi, wi, i, ho, wo, o, kh, kw, stride, pad, pad_mode, weight):
    so, si, sh, sw = weight.stride()
    return ConvDesc(n, hi, wi, i, ho, wo, o, kh, kw, stride, pad, pad_mode, so, si, sh, sw)


def _conv_ws(desc, device):
    lib = _lib.load()
    nbytes = lib.srgan_conv2d_workspace(ctypes.byref(desc))
    if nbytes == 0:
        raise _lib.SrganHipError("conv2d: " + lib.srgan_last_error().decode())
    return workspace(device, nbytes), nbytes


# ---- packed-weight cache ---------------------------------------------------------------------------
# The repacked weight operand of a conv depends only on (weight values, geometry, kind).  Inside a trainer step the
# weights change only at optimiser steps, so the trainer turns the cache on for the duration of ``train()`` and, after
# each optimiser step, re-packs every cached operand of that network in ONE launch (``refresh_packed``).  Entries persist
# across ``train()`` calls; an entry is only trusted while the parameter's autograd version counter is unchanged (any
# ordinary in-place update -- ``load_state_dict``, ``copy_`` -- bumps it; the raw-pointer Adam does not, which is why the
# trainer refreshes explicitly), and everything is re-packed once when a scope is entered.  Outside such a scope (plain
# module calls) every call repacks.
_pack_cache_on = False
_pack_cache = {}          # (id(weight), kind, act_flag, geometry) -> _Packed
_structure_epoch = 0      # bumped whenever a buffer a captured hipGraph may point at is dropped or re-allocated


def structure_epoch():
    return _structure_epoch


def bump_structure_epoch():
    global _structure_epoch
    _structure_epoch += 1


def graph_keepalive():
    """Every persistent device buffer the conv / norm ops hand to kernels by pointer (packed operands, repack tables,
    workspaces).  A captured train step holds these references for its lifetime: the caches may move on to new buffers
    (a new geometry rebuilds a repack table), but memory a recorded launch points at is never handed to anyone else."""
    keep = [h.buf for h in _pack_cache.values()]
    keep += [t[1] for t in _tables.values() if t[1] is not None]
    keep += list(_workspaces.values())
    return keep


class _Packed:
    """A cached packed operand.  The parameter is held WEAKLY: when a network is dropped (a notebook re-run, a sweep, the test
    suite building many trainers) its entries die with it and are purged at the next scope entry, instead of pinning the weight,
    its packed device buffer and a slot of every later refresh for the life of the process."""
    __slots__ = ("buf", "_wref", "desc", "kind", "act", "version", "fresh", "ptr", "scratch")

    def __init__(self, buf, weight, desc, kind, act, scratch):
        self.buf, self._wref, self.desc, self.kind, self.act = buf, weakref.ref(weight), desc, kind, act
        self.scratch = scratch                 # bytes of per-call scratch the packed run wants (shared workspace)
        self.version, self.fresh, self.ptr = weight._version, True, weight.data_ptr()

    @property
    def weight(self):
        return self._wref()


def _purge_dead_packed():
    dead = [k for k, h in _pack_cache.items() if h.weight is None]
    if not dead:
        return
    for k in dead:
        del _pack_cache[k]
    for k in [k for k, g in _geo_cache.items() if g[0].weight is None]:
        del _geo_cache[k]
    _tables.clear()


class pack_cache:
    """Context manager: cache packed conv weights (see above).  ``refresh_on_entry=False``: the caller vouches that no
    parameter was written behind the cache's back since the last scope (the captured train step does: it checks the
    parameters' version counters itself before every replay)."""

    def __init__(self, refresh_on_entry=True):
        self._refresh = refresh_on_entry

    def __enter__(self):
        global _pack_cache_on
        self._prev = _pack_cache_on
        _pack_cache_on = True
        if not self._prev:
            _purge_dead_packed()
        if self._refresh and not self._prev and _pack_cache:
            # entries persist between scopes; a ``.data`` update made outside (no version bump) would go unseen, so the
            # whole cache is re-packed on entry -- one multi-pack launch
            refresh_packed([h.weight for h in _pack_cache.values()])
        return self

    def __exit__(self, *exc):
        global _pack_cache_on
        _pack_cache_on = self._prev
        return False


def set_compute_dtype(name):
    """"fp32" (default: exact fp32 MFMA, Winograd where it applies) or "bf16" (BASELINE configs [2]-[4]: conv operands rounded
    to bf16 on their way into LDS, v_mfma_f32_32x32x16_bf16 with fp32 accumulation; everything else stays fp32).
    Process-wide; drops the packed-weight cache because the two modes use different kernels / layouts."""
    mode = {"fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}[str(name).replace("torch.", "")]
    _lib.check(_lib.load().srgan_set_compute_mode(mode), "set_compute_mode")
    bump_structure_epoch()
    _pack_cache.clear()
    _geo_cache.clear()
    _tables.clear()


def pack_cache_active():
    """True inside a ``pack_cache()`` scope (the 16-bit activation paths of the bf16 mode live on cached operands)."""
    return _pack_cache_on


def get_compute_dtype():
    return "bf16" if _lib.load().srgan_get_compute_mode() == 1 else "fp32"


def invalidate_packed(params=None):
    """Forget cached operands (all, or those of ``params``): they are re-packed one by one at their next use."""
    bump_structure_epoch()
    if params is None:
        _pack_cache.clear()
        _geo_cache.clear()
        _tables.clear()
        return
    ids = {id(p) for p in params}
    for key in [k for k in _pack_cache if k[0] in ids]:
        del _pack_cache[key]
    for key in [k for k in _geo_cache if k[0] in ids]:
        del _geo_cache[key]
    _tables.clear()


def mark_stale(params):
    """The raw-pointer optimiser calls this after writing ``params``: their cached operands must be re-packed before the
    next use (in one launch by ``refresh_packed``, else one by one)."""
    if not _pack_cache:
        return
    ids = {id(p) for p in params}
    for k, h in _pack_cache.items():
        if k[0] in ids:
            h.fresh = False


def _pack_one(hit):
    lib = _lib.load()
    _lib.check(lib.srgan_conv2d_pack(ctypes.byref(hit.desc), hit.kind, hit.act, _ptr(hit.weight), _ptr(hit.buf), hit.buf.numel(),
                                     _stream()), "conv2d_pack")
    hit.version, hit.fresh = hit.weight._version, True
    if hit.ptr != hit.weight.data_ptr():       # storage swapped under the parameter (``p.data = ...``): tables hold the old pointer
        hit.ptr = hit.weight.data_ptr()
        _tables.clear()


_geo_cache = {}           # (id(weight), kind, act_flag, full geometry) -> (_Packed, scratch bytes of that geometry)


def _packed(desc, weight, kind, act):
    """-> (_Packed, scratch bytes).  Operands whose layout does not depend on batch / map size (srgan_conv2d_pack_signature:
    the Winograd filter images) are shared between the geometries a weight is used with (the trainer runs the generator at
    batch 32, 64 and 128: one repack per optimiser step instead of three)."""
    gkey = (id(weight), kind, int(act != ACT_NONE), desc.N, desc.Hi, desc.Wi, desc.I, desc.O, desc.kh, desc.stride,
            desc.pad, desc.pad_mode)
    g = _geo_cache.get(gkey)
    if g is not None and g[0].weight is weight and _pack_cache.get(g[2]) is g[0]:
        hit, scratch = g[0], g[1]
    else:
        lib = _lib.load()
        sig = lib.srgan_conv2d_pack_signature(ctypes.byref(desc), kind, act)
        key = gkey if sig == 0 else (id(weight), kind, int(act != ACT_NONE), "sig", sig, desc.I, desc.O, desc.kh, desc.stride,
                                     desc.pad, desc.pad_mode)
        scratch = lib.srgan_conv2d_packed_scratch(ctypes.byref(desc), kind)
        hit = _pack_cache.get(key)
        if hit is None or hit.weight is not weight:
            nbytes = lib.srgan_conv2d_packed_bytes(ctypes.byref(desc), kind, act)
            if nbytes == 0:
                raise _lib.SrganHipError("conv2d pack: " + lib.srgan_last_error().decode())
            own = _lib.ConvDesc()
            ctypes.memmove(ctypes.byref(own), ctypes.byref(desc), ctypes.sizeof(own))
            hit = _Packed(torch.empty(nbytes, dtype=torch.uint8, device=weight.device), weight, own, kind, act, scratch)
            _pack_cache[key] = hit
            _tables.clear()
            _pack_one(hit)
            _geo_cache[gkey] = (hit, scratch, key)
            return hit, scratch
        _geo_cache[gkey] = (hit, scratch, key)
    if not hit.fresh or hit.version != weight._version or hit.ptr != weight.data_ptr():
        _pack_one(hit)
    return hit, scratch


_tables = {}              # frozenset of parameter ids -> (entries, device table, singles)


def refresh_packed(params, force=False):
    """Re-pack every cached operand of ``params`` after their optimiser step: one multi-pack launch for the implicit-GEMM /
    Winograd operands (device table built once per parameter set), single launches for the few narrow-output layers."""
    if not (_pack_cache_on or force):
        return
    params = list(params)
    if not params:
        return
    lib = _lib.load()
    tkey = frozenset(id(p) for p in params)
    tab = _tables.get(tkey)
    if tab is None:
        ids = set(tkey)
        hits = [h for k, h in _pack_cache.items() if k[0] in ids and h.weight is not None]
        for h in hits:
            # the layout of an operand follows the kernel dispatch (compute mode, SRGAN_* switches): an entry made under another
            # dispatch gets a buffer of the size the current one wants before anything is packed into it
            want = lib.srgan_conv2d_packed_bytes(ctypes.byref(h.desc), h.kind, h.act)
            if want != h.buf.numel():
                bump_structure_epoch()
                h.buf = torch.empty(want, dtype=torch.uint8, device=h.buf.device)
                h.scratch = lib.srgan_conv2d_packed_scratch(ctypes.byref(h.desc), h.kind)
        nb = lib.srgan_pack_entry_bytes()
        multi, singles, blob = [], [], bytearray()
        for h in hits:
            rec = (ctypes.c_char * nb)()
            rc = lib.srgan_conv2d_pack_entry(ctypes.byref(h.desc), h.kind, h.act, _ptr(h.weight), _ptr(h.buf), ctypes.byref(rec))
            if rc == 0:
                multi.append(h)
                blob += bytes(rec)
            elif rc == 1:
                singles.append(h)
            else:
                _lib.check(rc, "conv2d_pack_entry")
        dev = None
        if multi:
            dev = upload_small(blob, params[0].device)
        tab = (multi, dev, singles)
        _tables[tkey] = tab
    multi, dev, singles = tab
    if any(h.ptr != h.weight.data_ptr() for h in multi):       # a parameter's storage moved: rebuild the table next time
        _tables.pop(tkey, None)
        for h in multi + singles:
            _pack_one(h)
        return
    if multi:
        _lib.check(lib.srgan_conv2d_pack_multi(_ptr(dev), len(multi), _stream()), "conv2d_pack_multi")
        for h in multi:
            h.version, h.fresh = h.weight._version, True
    # The operands outside the multi-pack launch (narrow-output layers, the RGB layers: one small kernel each, four for a
    # stride-2 input gradient) are re-packed at their next USE instead: the discriminator's image-gradient operands are used
    # once per train step (generator update) but went stale -- and were re-packed -- after each of its k optimiser steps.
    for h in singles:
        if _EAGER_SINGLE_PACKS:
            _pack_one(h)
        else:
            h.fresh = False


_EAGER_SINGLE_PACKS = _lib.ab("SRGAN_EAGER_SINGLE_PACKS")


def mark_singles_stale():
    """Before a train step is RECORDED: every operand that is re-packed at use (see refresh_packed) is marked stale, so that the
    recording holds a pack launch at the first use of each one -- a replay starts from whatever the previous step's optimisers
    left, not from the state the cache happened to be in when the step was captured."""
    lib = _lib.load()
    nb = lib.srgan_pack_entry_bytes()
    rec = (ctypes.c_char * nb)()
    for h in _pack_cache.values():
        if h.weight is not None and lib.srgan_conv2d_pack_entry(ctypes.byref(h.desc), h.kind, h.act, _ptr(h.weight), _ptr(h.buf),
                                                                 ctypes.byref(rec)) == 1:
            h.fresh = False


def _run_conv_fwd(desc, x, weight, bias, y, act, slope, keep_v=None):
    lib = _lib.load()
    if _pack_cache_on:
        hit, scratch = _packed(desc, weight, 0, act)
        # keep_v: the caller's own buffer for the F(4x4,3x3) transformed input (kept for the weight gradient) instead of scratch
        ws = keep_v if keep_v is not None else (workspace(x.device, scratch) if scratch else None)
        _lib.check(lib.srgan_conv2d_fwd_packed(ctypes.byref(desc), _ptr(x), _ptr(hit.buf), _ptr(bias), _ptr(y), act,
                                               float(slope), _ptr(ws), scratch, _stream()), "conv2d_fwd_packed")
        return
    ws, nb = _conv_ws(desc, x.device)
    _lib.check(lib.srgan_conv2d_fwd(ctypes.byref(desc), _ptr(x), _ptr(weight), _ptr(bias), _ptr(y), act,
                                    float(slope), _ptr(ws), nb, _stream()), "conv2d_fwd")


def _run_conv_dgrad(desc, dy, weight, dx, res=None, mask=None, mask_slope=0.0):
    """dx = input gradient (+ res: the gradient of a skip connection that shares the conv's input, added in the kernel's
    epilogue where the dispatch supports it) (* LeakyReLU'(mask): the activation backward of the layer that produced the
    conv's input, in the epilogue of the transposed stride-2 Winograd kernel, else one in-place pass)."""
    lib = _lib.load()
    if mask is not None and (res is not None or not _pack_cache_on):
        _run_conv_dgrad(desc, dy, weight, dx, res)
        _lib.check(lib.srgan_act_bwd(_ptr(mask), _ptr(dx), _ptr(dx), dx.numel(), ACT_LRELU, float(mask_slope), _stream()), "act_bwd")
        return
    if _pack_cache_on:
        hit, scratch = _packed(desc, weight, 1, ACT_NONE)
        ws = workspace(dy.device, scratch) if scratch else None
        if mask is not None:
            _lib.check(lib.srgan_conv2d_dgrad_packed_mask(ctypes.byref(desc), _ptr(dy), _ptr(hit.buf), _ptr(mask), float(mask_slope),
                                                          _ptr(dx), _ptr(ws), scratch, _stream()), "conv2d_dgrad_packed_mask")
            return
        if res is not None:
            _lib.check(lib.srgan_conv2d_dgrad_packed_add(ctypes.byref(desc), _ptr(dy), _ptr(hit.buf), _ptr(res), _ptr(dx), _ptr(ws),
                                                         scratch, _stream()), "conv2d_dgrad_packed_add")
            return
        _lib.check(lib.srgan_conv2d_dgrad_packed(ctypes.byref(desc), _ptr(dy), _ptr(hit.buf), _ptr(dx), _ptr(ws), scratch,
                                                 _stream()), "conv2d_dgrad_packed")
        return
    ws, nb = _conv_ws(desc, dy.device)
    _lib.check(lib.srgan_conv2d_dgrad(ctypes.byref(desc), _ptr(dy), _ptr(weight), _ptr(dx), _ptr(ws), nb,
                                      _stream()), "conv2d_dgrad")
    if res is not None:
        _lib.check(lib.srgan_add(_ptr(dx), _ptr(res), _ptr(dx), dx.numel(), _stream()), "add")


def _run_conv_wgrad(desc, x, dy, dw, dbias, v_image=None):
    ws, nb = _conv_ws(desc, dy.device)
    if v_image is not None:
        _lib.check(_lib.load().srgan_conv2d_wgrad_v(ctypes.byref(desc), _ptr(v_image), _ptr(dy), _ptr(dw), _ptr(dbias), _ptr(ws),
                                                    nb, _stream()), "conv2d_wgrad_v")
        return
    _lib.check(_lib.load().srgan_conv2d_wgrad(ctypes.byref(desc), _ptr(x), _ptr(dy), _ptr(dw), _ptr(dbias), _ptr(ws),
                                              nb, _stream()), "conv2d_wgrad")


# ---- parameter-gradient sink ------------------------------------------------------------------------------
# A parameter reached through several graphs in ONE backward call (the generator inside util_notebook.py:664 and :689: the
# reconstruction / identity graph and the kept ``target_image`` graph) gets one gradient per use; torch's engine sums them in
# the AccumulateGrad input buffer -- 156 small elementwise launches per train step.  Inside a ``fused_param_grads`` scope the
# weight-gradient kernels of this module write a parameter's FIRST contribution of the pass into a fresh buffer and ADD the
# later ones in their own epilogue (split-K slab reduce with beta = 1, central-biasing records with the accumulate flag); the
# Functions return no gradient for such a parameter, and when the scope closes ``p.grad`` is bound to the buffer (added to an
# existing ``p.grad``, as AccumulateGrad would).  Same operands, same single rounding per addition as the engine's ``a += b``:
# bit-identical results.  A parameter's post-accumulate-grad hooks still fire once per backward call -- the engine runs the
# AccumulateGrad node with an undefined gradient -- but DURING the pass, before ``p.grad`` is bound: fine for hooks that count
# (``dp.GradReducer`` while a step is recorded), wrong for hooks that read the gradient, which is why the eager data-parallel
# step (buckets sent from the hooks) keeps the ordinary path.  Opt-in (the trainer wraps its backward calls): plain
# ``.backward()`` / ``autograd.grad`` on the modules are untouched.
#
# The sink's buffers are read by nobody until the scope closes, which also lets the library DEFER the split-K slab sums of the
# weight-gradient calls (``srgan_wgrad_defer_begin``): they wait in an arena and run as a few large launches instead of one
# ~12 us launch per layer (145 per train step).  ``SRGAN_NO_WGRAD_DEFER=1`` keeps the immediate sums.
_grad_sink = None
_sink_alloc = None          # optional callable(parameter) -> buffer for its first contribution of a pass (dp.grad_slot), or None
_sink_seed = False
_defer_depth = 0
_NO_WGRAD_DEFER = _lib.ab("SRGAN_NO_WGRAD_DEFER")
_WGRAD_ARENA_BYTES = 32 << 20      # first size; grown to what a pass asks for (below)
_WGRAD_ARENA_MAX = 2048 << 20
_arena_want = {}                    # device index -> bytes the largest pass so far asked for


def _wgrad_arena(device=None):
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    # one arena per DEVICE: its uses are serialised by stream order (srgan_wgrad_defer_begin records the stream, a
    # recorded step replays on the caller's stream), so a re-recorded step -- a new capture stream each time -- takes no new GiB.
    # Sized from use (it was a fixed 1 GiB): the library reports what the deferrable calls of a pass asked
    # for (srgan_wgrad_defer_need); a pass that did not fit -- it still runs right, with an early flush of the queued sums --
    # makes the NEXT scope take a larger arena.  A trainer's step 0 is eager, so the arena has its final size before a step is
    # recorded; growing later drops the recordings (structure epoch), never while a capture is in flight.
    key = ("wgrad_arena", dev.index)
    a = _workspaces.get(key)
    want = min(max(_WGRAD_ARENA_BYTES, _arena_want.get(dev.index, 0)), _WGRAD_ARENA_MAX)
    if a is None or (a.numel() < want and not torch.cuda.is_current_stream_capturing()):
        if a is not None:
            bump_structure_epoch()
        want = (want + (16 << 20) - 1) // (16 << 20) * (16 << 20)
        a = _workspaces[key] = torch.empty(want, dtype=torch.uint8, device=dev)
    return a


class fused_param_grads:
    """``device``: where the pass runs (default: the current device) -- the arena of the deferred slab sums lives there and the
    sums run on that device's current stream."""

    def __init__(self, enabled=True, device=None, seed=False):
        """``seed``: a parameter that already HAS a gradient (an earlier scope of the same backward pass bound it) takes this
        scope's contributions on top of it -- its ``p.grad`` is the sink slot and the kernels add into it -- instead of into a
        fresh buffer that the scope's exit would add with one elementwise launch per parameter."""
        self._enabled = enabled
        self._device = device
        self._seed = seed

    def __enter__(self):
        global _grad_sink, _defer_depth
        global _sink_seed
        self._prev = _grad_sink
        self._prev_seed = _sink_seed
        _grad_sink = {} if self._enabled else None
        _sink_seed = bool(self._seed and self._enabled)
        self._defer = False
        if self._enabled and not _NO_WGRAD_DEFER and _defer_depth == 0:
            a = _wgrad_arena(self._device)
            self._arena_index = a.device.index
            st = ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
            _lib.check(_lib.load().srgan_wgrad_defer_begin(_ptr(a), a.numel(), st), "wgrad_defer_begin")
            self._defer = True
            _defer_depth = 1
        return self

    def __exit__(self, et, ev, tb):
        global _grad_sink, _defer_depth
        global _sink_seed
        sink, _grad_sink = _grad_sink, self._prev
        _sink_seed = self._prev_seed
        if self._defer:
            _defer_depth = 0
            err = _lib.load().srgan_wgrad_defer_end()          # the queued sums, before anyone binds or reads the buffers
            if et is None:
                _lib.check(err, "wgrad_defer_end")
            need = ctypes.c_longlong(0)
            _lib.load().srgan_wgrad_defer_need(ctypes.byref(need))
            idx = self._arena_index
            if need.value > _arena_want.get(idx, 0):
                _arena_want[idx] = int(need.value)
        if et is None and sink:
            with torch.no_grad():
                for p, buf in sink.values():
                    if p.grad is None:
                        p.grad = buf
                    elif p.grad is not buf:      # (a seeded slot IS p.grad: the kernels have added into it)
                        p.grad.add_(buf)
        return False


def _sink_slots(*params):
    """-> ([buffer per parameter], accumulate) when every parameter can take its gradient through the sink of the current
    scope (leaf parameters, all visited before or none), else None: the caller then returns ordinary gradients."""
    if _grad_sink is None:
        return None
    if not all(p.is_leaf and p.requires_grad for p in params):
        return None
    seen = [id(p) in _grad_sink for p in params]
    if any(seen) and not all(seen):
        return None
    if not seen[0]:
        if _sink_seed and all(p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in params):
            for p in params:                     # seeded scope: add into the gradient an earlier scope left
                _grad_sink[id(p)] = (p, p.grad)
            return [p.grad for p in params], True
        bufs = [_sink_alloc(p) if _sink_alloc is not None else None for p in params]   # data parallel: the bucket slices
        # a bucket slice is the SAME storage an earlier pass bound to p.grad (GradReducer.finish / a previous
        # scope's exit).  Without a zero_grad in between (gradient accumulation, two backward calls) that gradient is still
        # live: writing this pass's first contribution over it would lose it, and the exit's `p.grad is not buf` test would
        # skip the add.  Such a slot is a seeded one -- the kernels add into it; all or none, since one launch serves the group.
        live = [b is not None and p.grad is not None and p.grad.data_ptr() == b.data_ptr() for p, b in zip(params, bufs)]
        if any(live):
            if all(live) and all(p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in params):
                for p in params:
                    _grad_sink[id(p)] = (p, p.grad)
                return [p.grad for p in params], True
            bufs = [None] * len(params)          # mixed: fresh buffers, added to p.grad at the exit like any unseeded slot
        for p, buf in zip(params, bufs):
            if buf is None:
                buf = torch.empty(p.shape, dtype=torch.float32, device=p.device)
            _grad_sink[id(p)] = (p, buf)
    return [_grad_sink[id(p)][1] for p in params], seen[0]


class _wgrad_accumulate:
    """``srgan_set_wgrad_accumulate`` around the weight-gradient calls of one Function.backward (thread-local in the library):
    bit 0 = add to the buffer (a later use of the parameter in this pass), bit 1 = the buffer is a sink slot, so the slab sum
    may wait for the end of the scope."""

    def __init__(self, on, sink=False):
        self._mode = (1 if on else 0) | (2 if sink and _defer_depth else 0)

    def __enter__(self):
        if self._mode:
            _lib.load().srgan_set_wgrad_accumulate(self._mode)

    def __exit__(self, *exc):
        if self._mode:
            _lib.load().srgan_set_wgrad_accumulate(0)
        return False


def _act_bwd(y, gy, act, slope):
    g = torch.empty_like(gy)
    _lib.check(_lib.load().srgan_act_bwd(_ptr(y), _ptr(gy), _ptr(g), gy.numel(), act, float(slope), _stream()), "act_bwd")
    return g


class _Conv2dFn(Function):
    """``skip=True``: also returns the input itself as a second output -- the tensor a residual connection should use.  Its
    gradient then arrives in THIS backward next to the conv output's, and the input-gradient kernel adds it in its epilogue
    (``srgan_conv2d_dgrad_packed_add``) instead of autograd accumulating the two paths with a separate pass.

    A chain conv + LeakyReLU -> conv (+ LeakyReLU) -> ... whose intermediate tensors have no other consumer (the discriminator
    trunk, ``model._run_trunk``) moves each activation's backward into the NEXT layer's input-gradient kernel: the producer is
    called with ``act_bwd_by_consumer=True`` (its backward takes the incoming gradient as already multiplied by its own
    activation's derivative) and the consumer with ``in_slope`` = the producer's slope (its input gradient is multiplied by 1 or
    ``in_slope`` after the sign of its input: ``srgan_conv2d_dgrad_packed_mask``)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, pad_mode, act, slope, skip=False, in_slope=None, act_bwd_by_consumer=False):
        x = to_nhwc(x)
        _require_gpu(weight, "conv2d weight")
        n, i, hi, wi = x.shape
        o, i2, kh, kw = weight.shape
        if i != i2:
            raise _lib.SrganHipError(f"conv2d: input has {i} channels, weight expects {i2}")
        ho = (hi + 2 * pad - kh) // stride + 1
        wo = (wi + 2 * pad - kw) // stride + 1
        desc = _conv_desc(n, hi, wi, i, ho, wo, o, kh, kw, stride, pad, pad_mode, weight)
        y = nhwc_empty(n, o, ho, wo, x.device)
        # F(4x4,3x3) layers: the forward's transformed input serves the weight gradient too -- keep it instead of scratch
        keep_v = None
        if _pack_cache_on and ctx.needs_input_grad[1]:
            nv = _lib.load().srgan_conv2d_wgrad_v_bytes(ctypes.byref(desc))
            if nv:
                keep_v = torch.empty(nv, dtype=torch.uint8, device=x.device)
        _run_conv_fwd(desc, x, weight, bias, y, act, slope, keep_v)
        ctx.v_image = keep_v
        ctx.desc, ctx.act, ctx.slope = desc, act, slope
        ctx.weight = weight            # by reference: read at backward time (torch 1.4 semantics)
        ctx.has_bias = bias is not None
        ctx.bias = bias
        ctx.in_slope, ctx.act_bwd_by_consumer = in_slope, bool(act_bwd_by_consumer) and act == ACT_LRELU
        if in_slope is not None and skip:
            raise _lib.SrganHipError("conv2d: in_slope and skip do not combine")
        ctx.save_for_backward(x, y if (act != ACT_NONE and not ctx.act_bwd_by_consumer) else None)
        if skip:
            return y, x
        return y

    @staticmethod
    def backward(ctx, gy, gskip=None):
        x, y = ctx.saved_tensors
        weight = ctx.weight
        dx = dw = db = None
        if gy is None:                     # only the skip path was used downstream
            return (to_nhwc(gskip) if gskip is not None else None), None, None, None, None, None, None, None, None, None, None
        gy = to_nhwc(gy)
        if ctx.act != ACT_NONE and not ctx.act_bwd_by_consumer:
            gy = _act_bwd(y, gy, ctx.act, ctx.slope)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            res = to_nhwc(gskip) if gskip is not None else None
            if res is not None and res.data_ptr() == dx.data_ptr():
                res = res.clone()
            if ctx.in_slope is not None:
                _run_conv_dgrad(ctx.desc, gy, weight, dx, res, x, ctx.in_slope)
            else:
                _run_conv_dgrad(ctx.desc, gy, weight, dx, res)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            slots = _sink_slots(*((weight, ctx.bias) if ctx.has_bias else (weight,)))
            acc = False
            if slots is not None:
                (dw, *rest), acc = slots
                db = rest[0] if rest else None
            else:
                dw = torch.empty(weight.shape, dtype=torch.float32, device=weight.device)
                if ctx.has_bias:
                    db = torch.empty(weight.shape[0], dtype=torch.float32, device=weight.device)
            desc = ConvDesc.from_buffer_copy(ctx.desc)
            desc.sO, desc.sI, desc.sH, desc.sW = dw.stride()
            with _wgrad_accumulate(acc, slots is not None):
                _run_conv_wgrad(desc, x, gy, dw, db, ctx.v_image)
            if slots is not None:
                dw = db = None
        return dx, dw, db, None, None, None, None, None, None, None, None


def conv2d(x, weight, bias=None, stride=1, padding=0, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.0, in_slope=None,
           act_bwd_by_consumer=False):
    return _Conv2dFn.apply(x, weight, bias, stride, padding, pad_mode, act, slope, False, in_slope, act_bwd_by_consumer)


def conv2d_skip(x, weight, bias=None, stride=1, padding=0, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.0):
    """-> (conv2d(x, ...), x): use the second result for a skip connection around the convolution (see _Conv2dFn)."""
    return _Conv2dFn.apply(x, weight, bias, stride, padding, pad_mode, act, slope, True)


class _ConvTranspose2dFn(Function):
    """y = conv_transpose2d(x, w[Cin,Cout,kh,kw]): the input-gradient kernel of the conv C whose
    weight is w viewed as [O=Cin][I=Cout]; backward = C forward (dx) and C's weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        x = to_nhwc(x)
        _require_gpu(weight, "conv_transpose2d weight")
        n, ci, hi, wi = x.shape
        ci2, co, kh, kw = weight.shape
        if ci != ci2:
            raise _lib.SrganHipError(f"conv_transpose2d: input has {ci} channels, weight expects {ci2}")
        ho = (hi - 1) * stride - 2 * pad + kh
        wo = (wi - 1) * stride - 2 * pad + kw
        # conv C: input = y-space [n,co,ho,wo], output = x-space [n,ci,hi,wi]
        desc = _conv_desc(n, ho, wo, co, hi, wi, ci, kh, kw, stride, pad, PAD_ZERO, weight)
        y = nhwc_empty(n, co, ho, wo, x.device)
        _run_conv_dgrad(desc, x, weight, y)
        ctx.desc, ctx.weight = desc, weight
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = to_nhwc(gy)
        weight = ctx.weight
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _run_conv_fwd(ctx.desc, gy, weight, None, dx, ACT_NONE, 0.0)
        if ctx.needs_input_grad[1]:
            slots = _sink_slots(weight)
            (dw,), acc = slots if slots is not None else ([torch.empty(weight.shape, dtype=torch.float32, device=weight.device)], False)
            desc = ConvDesc.from_buffer_copy(ctx.desc)
            desc.sO, desc.sI, desc.sH, desc.sW = dw.stride()
            with _wgrad_accumulate(acc, slots is not None):
                _run_conv_wgrad(desc, gy, x, dw, None)
            if slots is not None:
                dw = None
        return dx, dw, None, None


def conv_transpose2d(x, weight, stride=2, padding=1):
    return _ConvTranspose2dFn.apply(x, weight, stride, padding)


# ------------------------------------------------------------------------------------------
# instance norm (+affine, activation, residual) and the CBIN affine
# ------------------------------------------------------------------------------------------
class _InstNormFn(Function):
    @staticmethod
    def forward(ctx, x, scale, shift, res, act, slope, eps):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        lib = _lib.load()
        if res is not None:
            res = to_nhwc(res)
        y = torch.empty_like(x)
        mean = torch.empty(n * c, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        nb = lib.srgan_instnorm_workspace(n, h * w, c)
        ws = workspace(x.device, nb)
        _lib.check(lib.srgan_instnorm_fwd(_ptr(x), _ptr(scale), _ptr(shift), _ptr(res), _ptr(y), _ptr(mean), _ptr(rstd),
                                          n, h * w, c, float(eps), act, float(slope), _ptr(ws), nb, _stream()),
                   "instnorm_fwd")
        ctx.act, ctx.slope, ctx.has_res = act, slope, res is not None
        ctx.save_for_backward(x, scale, shift, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, scale, shift, mean, rstd = ctx.saved_tensors
        gy = to_nhwc(gy)
        n, c, h, w = x.shape
        lib = _lib.load()
        dx = torch.empty_like(x)
        dscale = torch.empty(n, c, dtype=torch.float32, device=x.device)
        dshift = torch.empty_like(dscale)
        nb = lib.srgan_instnorm_workspace(n, h * w, c)
        ws = workspace(x.device, nb)
        _lib.check(lib.srgan_instnorm_bwd(_ptr(x), _ptr(gy), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _ptr(dx),
                                          _ptr(dscale), _ptr(dshift), n, h * w, c, ctx.act, float(ctx.slope), _ptr(ws), nb,
                                          _stream()), "instnorm_bwd")
        has_aff = scale is not None
        return (dx, dscale if has_aff else None, dshift if has_aff else None, (gy.clone() if _DEBUG_CLONE else gy) if ctx.has_res else None,
                None, None, None)


def instance_norm_act(x, scale=None, shift=None, res=None, act=ACT_NONE, slope=0.0, eps=1e-5):
    """act((x - mean)/sqrt(var+eps) * scale[n,c] + shift[n,c]) (+ res)."""
    return _InstNormFn.apply(x, scale, shift, res, act, slope, eps)


class _NormActConvFn(Function):
    """conv3x3(act(instance_norm(x) * scale + shift), weight) for a 32x32 map feeding an F(4x4,3x3) layer, without the
    normalised tensor: the norm kernel writes the convolution's transformed-input (V) image directly
    (``srgan_instnorm_fwd_v``), the multiply runs on it (``srgan_conv2d_fwd_from_v``) and the weight gradient re-uses it
    (``srgan_conv2d_wgrad_v``).  Backward = the three kernels of the unfused chain: input gradient of the convolution, its
    weight gradient, instance-norm backward.  Same kept semantics as _Conv2dFn (weight by reference, read at backward time)."""

    @staticmethod
    def forward(ctx, x, scale, shift, weight, act, slope, eps):
        lib = _lib.load()
        x = to_nhwc(x)
        n, c, h, w = x.shape
        o = weight.shape[0]
        desc = _conv_desc(n, h, w, c, h, w, o, 3, 3, 1, 1, PAD_ZERO, weight)
        hit, scratch = _packed(desc, weight, 0, ACT_NONE)
        keep = None
        if ctx.needs_input_grad[3]:
            nv = lib.srgan_conv2d_wgrad_v_bytes(ctypes.byref(desc))
            if nv:
                keep = torch.empty(max(nv, scratch), dtype=torch.uint8, device=x.device)
        v = keep if keep is not None else workspace(x.device, scratch)
        mean = torch.empty(n * c, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _lib.check(lib.srgan_instnorm_fwd_v(ctypes.byref(desc), _ptr(x), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _ptr(v),
                                            v.numel(), float(eps), act, float(slope), _stream()), "instnorm_fwd_v")
        y = nhwc_empty(n, o, h, w, x.device)
        _lib.check(lib.srgan_conv2d_fwd_from_v(ctypes.byref(desc), _ptr(v), _ptr(hit.buf), None, _ptr(y), ACT_NONE, 0.0, _stream()),
                   "conv2d_fwd_from_v")
        ctx.desc, ctx.weight, ctx.v_image = desc, weight, keep
        ctx.act, ctx.slope, ctx.eps = act, slope, eps
        ctx.save_for_backward(x, scale, shift, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, scale, shift, mean, rstd = ctx.saved_tensors
        lib = _lib.load()
        gy = to_nhwc(gy)
        weight = ctx.weight
        n, c, h, w = x.shape
        dw = None
        if ctx.needs_input_grad[3]:
            slots = _sink_slots(weight)
            (dw,), acc = slots if slots is not None else ([torch.empty(weight.shape, dtype=torch.float32, device=weight.device)], False)
            desc = ConvDesc.from_buffer_copy(ctx.desc)
            desc.sO, desc.sI, desc.sH, desc.sW = dw.stride()
            with _wgrad_accumulate(acc, slots is not None):
                if ctx.v_image is not None:
                    _run_conv_wgrad(desc, None, gy, dw, None, ctx.v_image)
                else:   # no V kept (layer outside the F(4x4,3x3) weight-gradient geometry): recompute the normalised input
                    hx = _InstNormFn.apply(x.detach(), scale, shift, None, ctx.act, ctx.slope, ctx.eps)
                    _run_conv_wgrad(desc, hx, gy, dw, None)
            if slots is not None:
                dw = None
        dx = dscale = dshift = None
        if ctx.needs_input_grad[0] or (scale is not None and ctx.needs_input_grad[1]):
            dh = torch.empty_like(x)
            _run_conv_dgrad(ctx.desc, gy, weight, dh)
            dx = torch.empty_like(x)
            dscale = torch.empty(n, c, dtype=torch.float32, device=x.device)
            dshift = torch.empty_like(dscale)
            nb = lib.srgan_instnorm_workspace(n, h * w, c)
            ws = workspace(x.device, nb)
            _lib.check(lib.srgan_instnorm_bwd(_ptr(x), _ptr(dh), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _ptr(dx),
                                              _ptr(dscale), _ptr(dshift), n, h * w, c, ctx.act, float(ctx.slope), _ptr(ws), nb,
                                              _stream()), "instnorm_bwd")
            if scale is None:
                dscale = dshift = None
        return dx, dscale, dshift, dw, None, None, None


class _ResBlockFn(Function):
    """SingleResidualBlock (pyfiles/model.py:196-201) as ONE autograd node on 32x32 maps whose convolutions run on F(4x4,3x3):

        y1 = c1(x);  y2 = c2(relu(cbin1(y1)));  out = cbin2(y2) + x

    Forward: c1 keeps its transformed input V0; cbin1 + ReLU is written straight as c2's transformed input V1
    (``srgan_instnorm_fwd_v``); cbin2 + skip in the slab norm kernel.  Backward: each norm backward writes the two transforms of
    its result -- the input-gradient kernel's image and the weight-gradient kernel's Z image -- instead of the result
    (``srgan_instnorm_bwd_vz``), the multiplies run on them (``srgan_conv2d_dgrad_from_v`` / ``srgan_conv2d_wgrad_vz`` with V1 /
    V0), and the skip path's gradient is added in c1's input-gradient epilogue.  Six launches fewer than the chain and the
    gradients w.r.t. y1 / y2 are never written or re-read.  Weights are held by reference and read at backward time (stale-graph
    semantics of SURVEY.md Appendix C-1), the packed operands come from the packed-weight scope."""

    @staticmethod
    def forward(ctx, x, s1, h1, s2, h2, w1, w2, eps):
        lib = _lib.load()
        x = to_nhwc(x)
        n, c, h, w = x.shape
        dev = x.device
        d1 = _conv_desc(n, h, w, c, h, w, w1.shape[0], 3, 3, 1, 1, PAD_ZERO, w1)
        d2 = _conv_desc(n, h, w, w1.shape[0], h, w, w2.shape[0], 3, 3, 1, 1, PAD_ZERO, w2)
        need_w = ctx.needs_input_grad[5] or ctx.needs_input_grad[6]
        hit1, sc1 = _packed(d1, w1, 0, ACT_NONE)
        hit2, sc2 = _packed(d2, w2, 0, ACT_NONE)
        v0 = torch.empty(max(sc1, 16), dtype=torch.uint8, device=dev) if need_w else workspace(dev, sc1)
        y1 = nhwc_empty(n, w1.shape[0], h, w, dev)
        _lib.check(lib.srgan_conv2d_fwd_packed(ctypes.byref(d1), _ptr(x), _ptr(hit1.buf), None, _ptr(y1), ACT_NONE, 0.0, _ptr(v0), sc1,
                                               _stream()), "conv2d_fwd_packed")
        v1 = torch.empty(max(sc2, 16), dtype=torch.uint8, device=dev) if need_w else workspace(dev, sc2)
        c1 = w1.shape[0]
        mean1 = torch.empty(n * c1, dtype=torch.float32, device=dev)
        rstd1 = torch.empty_like(mean1)
        _lib.check(lib.srgan_instnorm_fwd_v(ctypes.byref(d2), _ptr(y1), _ptr(s1), _ptr(h1), _ptr(mean1), _ptr(rstd1), _ptr(v1), sc2,
                                            float(eps), ACT_RELU, 0.0, _stream()), "instnorm_fwd_v")
        y2 = nhwc_empty(n, w2.shape[0], h, w, dev)
        _lib.check(lib.srgan_conv2d_fwd_from_v(ctypes.byref(d2), _ptr(v1), _ptr(hit2.buf), None, _ptr(y2), ACT_NONE, 0.0, _stream()),
                   "conv2d_fwd_from_v")
        c2 = w2.shape[0]
        out = torch.empty_like(y2)
        mean2 = torch.empty(n * c2, dtype=torch.float32, device=dev)
        rstd2 = torch.empty_like(mean2)
        nb = lib.srgan_instnorm_workspace(n, h * w, c2)
        ws = workspace(dev, nb)
        _lib.check(lib.srgan_instnorm_fwd(_ptr(y2), _ptr(s2), _ptr(h2), _ptr(x), _ptr(out), _ptr(mean2), _ptr(rstd2), n, h * w, c2,
                                          float(eps), ACT_NONE, 0.0, _ptr(ws), nb, _stream()), "instnorm_fwd")
        ctx.d1, ctx.d2, ctx.w1, ctx.w2 = d1, d2, w1, w2
        ctx.v0, ctx.v1 = (v0, v1) if need_w else (None, None)
        ctx.save_for_backward(y1, y2, s1, h1, s2, h2, mean1, rstd1, mean2, rstd2)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        y1, y2, s1, h1, s2, h2, mean1, rstd1, mean2, rstd2 = ctx.saved_tensors
        g = to_nhwc(g)
        n, c2, h, w = y2.shape
        c1 = y1.shape[1]
        dev = g.device
        d1, d2, w1, w2 = ctx.d1, ctx.d2, ctx.w1, ctx.w2
        st = _stream()

        def norm_bwd_vz(desc, y, gup, sc, sh, mean, rstd, act, ch):
            vb = lib.srgan_conv2d_packed_scratch(ctypes.byref(desc), 1)
            zb = lib.srgan_instnorm_bwd_vz_z_bytes(ctypes.byref(desc))
            vimg = torch.empty(vb, dtype=torch.uint8, device=dev)
            zimg = torch.empty(zb, dtype=torch.uint8, device=dev)
            dsc = torch.empty(n, ch, dtype=torch.float32, device=dev)
            dsh = torch.empty_like(dsc)
            _lib.check(lib.srgan_instnorm_bwd_vz(ctypes.byref(desc), _ptr(y), _ptr(gup), _ptr(sc), _ptr(sh), _ptr(mean), _ptr(rstd),
                                                 _ptr(dsc), _ptr(dsh), _ptr(vimg), vb, _ptr(zimg), zb, act, 0.0, st), "instnorm_bwd_vz")
            return vimg, zimg, dsc, dsh

        def wgrad_vz(desc, weight, v_fwd, zimg):
            slots = _sink_slots(weight)
            (dw,), acc = slots if slots is not None else ([torch.empty(weight.shape, dtype=torch.float32, device=dev)], False)
            dd = ConvDesc.from_buffer_copy(desc)
            dd.sO, dd.sI, dd.sH, dd.sW = dw.stride()
            ws, nb = _conv_ws(dd, dev)
            with _wgrad_accumulate(acc, slots is not None):
                _lib.check(lib.srgan_conv2d_wgrad_vz(ctypes.byref(dd), _ptr(v_fwd), _ptr(zimg), _ptr(dw), _ptr(ws), nb, st),
                           "conv2d_wgrad_vz")
            return None if slots is not None else dw

        def dgrad_from_v(desc, weight, vimg, res, ch):
            hit, _ = _packed(desc, weight, 1, ACT_NONE)
            dx = nhwc_empty(n, ch, h, w, dev)
            _lib.check(lib.srgan_conv2d_dgrad_from_v(ctypes.byref(desc), _ptr(vimg), _ptr(hit.buf), _ptr(res), _ptr(dx), st),
                       "conv2d_dgrad_from_v")
            return dx

        # cbin2 backward -> c2's two images; c2 weight gradient (with V1) and input gradient
        vimg, zimg, ds2, dh2 = norm_bwd_vz(d2, y2, g, s2, h2, mean2, rstd2, ACT_NONE, c2)
        dw2 = wgrad_vz(d2, w2, ctx.v1, zimg) if (ctx.needs_input_grad[6] and ctx.v1 is not None) else None
        dh = dgrad_from_v(d2, w2, vimg, None, c1)
        # cbin1 + ReLU backward -> c1's two images; c1 weight gradient (with V0); input gradient + the skip path's gradient
        vimg, zimg, ds1, dh1 = norm_bwd_vz(d1, y1, dh, s1, h1, mean1, rstd1, ACT_RELU, c1)
        dw1 = wgrad_vz(d1, w1, ctx.v0, zimg) if (ctx.needs_input_grad[5] and ctx.v0 is not None) else None
        dx = dgrad_from_v(d1, w1, vimg, g, d1.I) if ctx.needs_input_grad[0] else None
        return dx, ds1, dh1, ds2, dh2, dw1, dw2, None


class _ResBlockBf16Fn(Function):
    """SingleResidualBlock (pyfiles/model.py:196-201) in the bf16 compute mode as ONE autograd node whose INTERMEDIATES live in
    HBM as bf16:

        y1 = c1(x) [bf16];  h = relu(cbin1(y1)) [bf16];  y2 = c2(h) [bf16];  out = cbin2(y2) + x [fp32]

    and in the backward  dy2 = cbin2'(y2, g) [bf16];  dh = c2^T(dy2) [bf16];  dy1 = (relu . cbin1)'(y1, dh) [bf16];
    dx = c1^T(dy1) + g [fp32].  The residual stream (x, out, g, dx), the statistics, the scale / shift gradients, the weight
    gradients and the master weights stay fp32; every product is bf16 x bf16 with fp32 accumulation on the LDS-resident-patch
    kernels of csrc/conv_halo16.hip (which read a bf16 tensor without the fp32 -> bf16 conversion and at half the bytes), the
    norms are the single-pass slab kernels with bf16 I/O.  Against the unfused bf16-mode chain the conv outputs are rounded to
    bf16 before their statistics are taken (as under autocast), the normalised activation and the conv-output gradients are
    rounded once instead of on every read.  Weights are held by reference and read at backward time (stale-graph semantics,
    SURVEY.md Appendix C-1)."""

    @staticmethod
    def forward(ctx, x, s1, h1, s2, h2, w1, w2, eps):
        lib = _lib.load()
        x = to_nhwc(x)
        n, c, h, w = x.shape
        dev = x.device
        st = _stream()
        d = _conv_desc(n, h, w, c, h, w, c, 3, 3, 1, 1, PAD_ZERO, w1)
        d2 = _conv_desc(n, h, w, c, h, w, c, 3, 3, 1, 1, PAD_ZERO, w2)
        hit1, _ = _packed(d, w1, 0, ACT_NONE)
        hit2, _ = _packed(d2, w2, 0, ACT_NONE)

        def b16():
            return torch.empty((n, h, w, c), dtype=torch.bfloat16, device=dev)

        y1, hh, y2 = b16(), b16(), b16()
        _lib.check(lib.srgan_halo16_conv(ctypes.byref(d), 0, _ptr(x), 0, _ptr(hit1.buf), None, _ptr(y1), 1, st), "halo16_conv")
        mean1 = torch.empty(n * c, dtype=torch.float32, device=dev)
        rstd1, mean2, rstd2 = torch.empty_like(mean1), torch.empty_like(mean1), torch.empty_like(mean1)
        # (the slab kernels on maps of <= 1024 pixels, the two-pass kernels with 16-bit I/O on the 64 x 64 trunk maps of 256 x 256)
        nb = lib.srgan_instnorm_workspace(n, h * w, c)
        ws = workspace(dev, nb)
        _lib.check(lib.srgan_instnorm_fwd_io(_ptr(y1), 1, _ptr(s1), _ptr(h1), None, _ptr(hh), 1, _ptr(mean1), _ptr(rstd1),
                                             n, h * w, c, float(eps), ACT_RELU, 0.0, _ptr(ws), nb, st), "instnorm_fwd_io")
        _lib.check(lib.srgan_halo16_conv(ctypes.byref(d2), 0, _ptr(hh), 1, _ptr(hit2.buf), None, _ptr(y2), 1, st), "halo16_conv")
        out = torch.empty_like(x)
        _lib.check(lib.srgan_instnorm_fwd_io(_ptr(y2), 1, _ptr(s2), _ptr(h2), _ptr(x), _ptr(out), 0, _ptr(mean2), _ptr(rstd2),
                                             n, h * w, c, float(eps), ACT_NONE, 0.0, _ptr(ws), nb, st), "instnorm_fwd_io")
        ctx.d1, ctx.d2, ctx.w1, ctx.w2 = d, d2, w1, w2
        ctx.save_for_backward(x, y1, hh, y2, s1, h1, s2, h2, mean1, rstd1, mean2, rstd2)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, y1, hh, y2, s1, h1, s2, h2, mean1, rstd1, mean2, rstd2 = ctx.saved_tensors
        g = to_nhwc(g)
        n, c, h, w = x.shape
        dev = g.device
        st = _stream()
        d1, d2, w1, w2 = ctx.d1, ctx.d2, ctx.w1, ctx.w2

        def b16():
            return torch.empty((n, h, w, c), dtype=torch.bfloat16, device=dev)

        def norm_bwd(y, gup, gup16, sc, sh, mean, rstd, act):
            dy = b16()
            dsc = torch.empty(n, c, dtype=torch.float32, device=dev)
            dsh = torch.empty_like(dsc)
            nb = lib.srgan_instnorm_workspace(n, h * w, c)
            ws = workspace(dev, nb)
            _lib.check(lib.srgan_instnorm_bwd_io(_ptr(y), 1, _ptr(gup), gup16, _ptr(sc), _ptr(sh), _ptr(mean), _ptr(rstd),
                                                 _ptr(dy), 1, _ptr(dsc), _ptr(dsh), n, h * w, c, act, 0.0, _ptr(ws), nb, st),
                       "instnorm_bwd_io")
            return dy, dsc, dsh

        def wgrad(desc, weight, xin, xin16, dy):
            slots = _sink_slots(weight)
            (dw,), acc = slots if slots is not None else ([torch.empty(weight.shape, dtype=torch.float32, device=dev)], False)
            dd = ConvDesc.from_buffer_copy(desc)
            dd.sO, dd.sI, dd.sH, dd.sW = dw.stride()
            ws, nb = _conv_ws(dd, dev)
            with _wgrad_accumulate(acc, slots is not None):
                _lib.check(lib.srgan_halo16_wgrad(ctypes.byref(dd), _ptr(xin), xin16, _ptr(dy), 1, _ptr(dw), _ptr(ws), nb, st),
                           "halo16_wgrad")
            return None if slots is not None else dw

        dy2, ds2, dh2 = norm_bwd(y2, g, 0, s2, h2, mean2, rstd2, ACT_NONE)
        dw2 = wgrad(d2, w2, hh, 1, dy2) if ctx.needs_input_grad[6] else None
        hit2, _ = _packed(d2, w2, 1, ACT_NONE)
        dh = b16()
        _lib.check(lib.srgan_halo16_conv(ctypes.byref(d2), 1, _ptr(dy2), 1, _ptr(hit2.buf), None, _ptr(dh), 1, st), "halo16_conv")
        dy1, ds1, dh1 = norm_bwd(y1, dh, 1, s1, h1, mean1, rstd1, ACT_RELU)
        dw1 = wgrad(d1, w1, x, 0, dy1) if ctx.needs_input_grad[5] else None
        dx = None
        if ctx.needs_input_grad[0]:
            hit1, _ = _packed(d1, w1, 1, ACT_NONE)
            dx = torch.empty_like(g)
            _lib.check(lib.srgan_halo16_conv(ctypes.byref(d1), 1, _ptr(dy1), 1, _ptr(hit1.buf), _ptr(g), _ptr(dx), 0, st), "halo16_conv")
        return dx, ds1, dh1, ds2, dh2, dw1, dw2, None


# ---- bf16 activation storage outside the residual trunk (round 4) -------------------------------------------------------
# In the bf16 mode the generator's down / up path keeps the tensors BETWEEN its stride-2 convolutions and their norms in bf16
# (pyfiles/model.py:212-215, 227-231, 245-246):
#     norm0 out (fp32 -> bf16) -> down1 (bf16 -> bf16) -> norm1 (bf16 -> bf16) -> down2 (bf16 -> bf16) -> norm2 (bf16 -> fp32) -> trunk
#     trunk out (fp32) -> up0 (fp32 -> bf16) -> norm (bf16 -> bf16) -> up1 (bf16 -> bf16) -> norm (bf16 -> fp32) -> RGB head
# The convolutions round their operands to bf16 anyway, so a bf16 norm OUTPUT changes no product; a bf16 conv OUTPUT is what
# torch.autocast stores (its statistics are then taken from the rounded values, as in the residual-block node above).  The
# gradients take the types of the tensors they belong to.  Served by the LDS-resident-patch kernels (csrc/conv_halo16.hip:
# halo16s / halo16t / halo16s2_wgrad with IN16 / OUT16) and the instance-norm kernels with 16-bit I/O (srgan_instnorm_fwd_io /
# _bwd_io): half the bytes in the norm passes, region loads without conversion in the convolutions.
STORAGE_BF16 = True               # tests switch it off to compare against the fp32-tensor chain


def _is16(t):
    return 1 if t.dtype == torch.bfloat16 else 0


def s2_io_applicable(n, ci, hi, wi, co, weight, transposed):
    """True when a 4x4 / stride-2 / pad-1 Conv2d (ci -> co, hi x wi -> half) -- or, transposed=True, the ConvTranspose2d
    (ci -> co, hi x wi -> double) -- can take / write bf16 tensors: bf16 mode, packed-weight scope, all three directions on the
    patch kernels."""
    if not (STORAGE_BF16 and _pack_cache_on and get_compute_dtype() == "bf16" and weight.is_cuda):
        return False
    if tuple(weight.shape[2:]) != (4, 4):
        return False
    if transposed:          # conv C of _ConvTranspose2dFn: input = y-space [n, co, 2 hi, 2 wi], output = x-space
        desc = _conv_desc(n, 2 * hi, 2 * wi, co, hi, wi, ci, 4, 4, 2, 1, PAD_ZERO, weight)
    else:
        if hi % 2 or wi % 2:
            return False
        desc = _conv_desc(n, hi, wi, ci, hi // 2, wi // 2, co, 4, 4, 2, 1, PAD_ZERO, weight)
    return bool(_lib.load().srgan_halo16s2_applicable(ctypes.byref(desc)))


def norm_io_applicable(n, c, h, w):
    return bool(STORAGE_BF16 and get_compute_dtype() == "bf16" and _lib.load().srgan_instnorm_io_applicable(n, h * w, c))


def _s2_wgrad(desc, weight, xin, dy, needs):
    """dW of conv C (desc) from xin (C's input side) and dy (C's output side), either fp32 or bf16; through the gradient sink."""
    if not needs:
        return None
    slots = _sink_slots(weight)
    (dw,), acc = slots if slots is not None else ([torch.empty(weight.shape, dtype=torch.float32, device=weight.device)], False)
    dd = ConvDesc.from_buffer_copy(desc)
    dd.sO, dd.sI, dd.sH, dd.sW = dw.stride()
    ws, nb = _conv_ws(dd, xin.device)
    with _wgrad_accumulate(acc, slots is not None):
        _lib.check(_lib.load().srgan_halo16_wgrad(ctypes.byref(dd), _ptr(xin), _is16(xin), _ptr(dy), _is16(dy), _ptr(dw), _ptr(ws), nb,
                                                   _stream()), "halo16_wgrad")
    return None if slots is not None else dw


class _ConvS2IoFn(Function):
    """4x4 / stride-2 / pad-1 Conv2d without bias in the bf16 mode, input fp32 or bf16, output bf16 or fp32 (see above)."""

    @staticmethod
    def forward(ctx, x, weight, out_bf16):
        x = to_nhwc(x)
        n, i, hi, wi = x.shape
        o = weight.shape[0]
        desc = _conv_desc(n, hi, wi, i, hi // 2, wi // 2, o, 4, 4, 2, 1, PAD_ZERO, weight)
        hit, _ = _packed(desc, weight, 0, ACT_NONE)
        y = nhwc_empty(n, o, hi // 2, wi // 2, x.device, torch.bfloat16 if out_bf16 else torch.float32)
        _lib.check(_lib.load().srgan_halo16_conv(ctypes.byref(desc), 0, _ptr(x), _is16(x), _ptr(hit.buf), None, _ptr(y), _is16(y),
                                                 _stream()), "halo16_conv")
        ctx.desc, ctx.weight = desc, weight
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = to_nhwc(gy)
        weight = ctx.weight
        dx = None
        if ctx.needs_input_grad[0]:
            hit, _ = _packed(ctx.desc, weight, 1, ACT_NONE)
            dx = torch.empty_like(x)
            _lib.check(_lib.load().srgan_halo16_conv(ctypes.byref(ctx.desc), 1, _ptr(gy), _is16(gy), _ptr(hit.buf), None, _ptr(dx),
                                                     _is16(dx), _stream()), "halo16_conv")
        dw = _s2_wgrad(ctx.desc, weight, x, gy, ctx.needs_input_grad[1])
        return dx, dw, None


# ---- round 6: 16-bit activations in the discriminator trunks (conv 4x4 / stride 2 without bias -> LeakyReLU, model.py:302-309) ----
_NO_D_IO16 = _lib.ab("SRGAN_NO_D_IO16")          # A/B (experiment build only): the discriminators on fp32 tensors


_NO_RGB_IO16 = _lib.ab("SRGAN_NO_RGB_IO16")      # A/B (experiment build only): fp32 tensors around the generator's 7x7 RGB layers


def conv_act_io_applicable(n, ci, hi, wi, weight, act, k=4, stride=2, pad=1):
    """True when a Conv2d without bias followed by `act` can take and write bf16 tensors in all three directions (bf16 mode,
    packed-weight scope; ``srgan_conv2d_io_applicable``): the 4x4 / stride-2 / pad-1 layers of the discriminator trunks, and --
    (k, stride, pad) = (7, 1, 3), act none -- the generator's RGB input and output layers, whose 64-channel side may be bf16."""
    if not (STORAGE_BF16 and _pack_cache_on and get_compute_dtype() == "bf16" and weight.is_cuda):
        return False
    if (_NO_D_IO16 and k == 4) or (_NO_RGB_IO16 and k == 7):
        return False
    if tuple(weight.shape[2:]) != (k, k) or (stride == 2 and (hi % 2 or wi % 2)):
        return False
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    desc = _conv_desc(n, hi, wi, ci, ho, wo, weight.shape[0], k, k, stride, pad, PAD_ZERO, weight)
    return bool(_lib.load().srgan_conv2d_io_applicable(ctypes.byref(desc), act))


class _ConvActIoFn(Function):
    """Conv2d without bias + activation (fused epilogue) in the bf16 mode with fp32 or bf16 tensors on either side (the layers of
    ``conv_act_io_applicable``).  Backward: the activation's derivative is applied to the incoming gradient by one 16-bit
    elementwise pass whose result -- always bf16: both kernels that read it round it to bf16 anyway -- feeds the input-gradient
    and the weight-gradient kernel; dx has x's type."""

    @staticmethod
    def forward(ctx, x, weight, act, slope, out_bf16, k, stride, pad):
        x = to_nhwc(x)
        n, i, hi, wi = x.shape
        o, i2 = weight.shape[:2]
        if i != i2:
            raise _lib.SrganHipError(f"conv2d_act_io: input has {i} channels, weight expects {i2}")
        ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
        desc = _conv_desc(n, hi, wi, i, ho, wo, o, k, k, stride, pad, PAD_ZERO, weight)
        hit, scratch = _packed(desc, weight, 0, act)
        ws = workspace(x.device, scratch) if scratch else None
        y = nhwc_empty(n, o, ho, wo, x.device, torch.bfloat16 if out_bf16 else torch.float32)
        _lib.check(_lib.load().srgan_conv2d_io_fwd(ctypes.byref(desc), _ptr(x), _is16(x), _ptr(hit.buf), None, _ptr(y), _is16(y), act,
                                                   float(slope), _ptr(ws), scratch, _stream()), "conv2d_io_fwd")
        ctx.desc, ctx.weight, ctx.act, ctx.slope = desc, weight, act, slope
        ctx.save_for_backward(x, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y = ctx.saved_tensors
        gy = to_nhwc(gy)
        weight = ctx.weight
        lib = _lib.load()
        if ctx.act != ACT_NONE:
            g = torch.empty_like(gy, dtype=torch.bfloat16)          # (keeps gy's NHWC-dense strides: no copy)
            _lib.check(lib.srgan_act_bwd_io(_ptr(y), _is16(y), _ptr(gy), _is16(gy), _ptr(g), 1, gy.numel(), ctx.act, float(ctx.slope),
                                            _stream()), "act_bwd_io")
        else:
            g = gy
        dx = None
        if ctx.needs_input_grad[0]:
            hit, scratch = _packed(ctx.desc, weight, 1, ACT_NONE)
            ws = workspace(g.device, scratch) if scratch else None
            dx = torch.empty_like(x)
            _lib.check(lib.srgan_conv2d_io_dgrad(ctypes.byref(ctx.desc), _ptr(g), _is16(g), _ptr(hit.buf), _ptr(dx), _is16(dx), _ptr(ws),
                                                 scratch, _stream()), "conv2d_io_dgrad")
        dw = _s2_wgrad(ctx.desc, weight, x, g, ctx.needs_input_grad[1])
        return dx, dw, None, None, None, None, None, None


def conv2d_act_io(x, weight, act, slope, out_bf16, k=4, stride=2, pad=1):
    return _ConvActIoFn.apply(x, weight, act, slope, bool(out_bf16), k, stride, pad)


class _ConvT2IoFn(Function):
    """4x4 / stride-2 / pad-1 ConvTranspose2d in the bf16 mode with fp32 or bf16 tensors on either side: the transposed form of
    the conv C whose weight is w viewed as [O = Cin][I = Cout] (as _ConvTranspose2dFn)."""

    @staticmethod
    def forward(ctx, x, weight, out_bf16):
        x = to_nhwc(x)
        n, ci, hi, wi = x.shape
        co = weight.shape[1]
        desc = _conv_desc(n, 2 * hi, 2 * wi, co, hi, wi, ci, 4, 4, 2, 1, PAD_ZERO, weight)
        hit, _ = _packed(desc, weight, 1, ACT_NONE)
        y = nhwc_empty(n, co, 2 * hi, 2 * wi, x.device, torch.bfloat16 if out_bf16 else torch.float32)
        _lib.check(_lib.load().srgan_halo16_conv(ctypes.byref(desc), 1, _ptr(x), _is16(x), _ptr(hit.buf), None, _ptr(y), _is16(y),
                                                 _stream()), "halo16_conv")
        ctx.desc, ctx.weight = desc, weight
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = to_nhwc(gy)
        weight = ctx.weight
        dx = None
        if ctx.needs_input_grad[0]:
            hit, _ = _packed(ctx.desc, weight, 0, ACT_NONE)
            dx = torch.empty_like(x)
            _lib.check(_lib.load().srgan_halo16_conv(ctypes.byref(ctx.desc), 0, _ptr(gy), _is16(gy), _ptr(hit.buf), None, _ptr(dx),
                                                     _is16(dx), _stream()), "halo16_conv")
        dw = _s2_wgrad(ctx.desc, weight, gy, x, ctx.needs_input_grad[1])      # C's input side is y-space
        return dx, dw, None


class _InstNormIoFn(Function):
    """instance_norm_act with an fp32 or bf16 input and a bf16 or fp32 output (no residual); dx has x's type."""

    @staticmethod
    def forward(ctx, x, scale, shift, act, slope, eps, out_bf16):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        lib = _lib.load()
        y = nhwc_empty(n, c, h, w, x.device, torch.bfloat16 if out_bf16 else torch.float32)
        mean = torch.empty(n * c, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        nb = lib.srgan_instnorm_workspace(n, h * w, c)
        ws = workspace(x.device, nb)
        _lib.check(lib.srgan_instnorm_fwd_io(_ptr(x), _is16(x), _ptr(scale), _ptr(shift), None, _ptr(y), _is16(y), _ptr(mean), _ptr(rstd),
                                             n, h * w, c, float(eps), act, float(slope), _ptr(ws), nb, _stream()), "instnorm_fwd_io")
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(x, scale, shift, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, scale, shift, mean, rstd = ctx.saved_tensors
        gy = to_nhwc(gy)
        n, c, h, w = x.shape
        lib = _lib.load()
        dx = torch.empty_like(x)
        dscale = torch.empty(n, c, dtype=torch.float32, device=x.device)
        dshift = torch.empty_like(dscale)
        nb = lib.srgan_instnorm_workspace(n, h * w, c)
        ws = workspace(x.device, nb)
        _lib.check(lib.srgan_instnorm_bwd_io(_ptr(x), _is16(x), _ptr(gy), _is16(gy), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd),
                                             _ptr(dx), _is16(dx), _ptr(dscale), _ptr(dshift), n, h * w, c, ctx.act, float(ctx.slope),
                                             _ptr(ws), nb, _stream()), "instnorm_bwd_io")
        has_aff = scale is not None
        return dx, dscale if has_aff else None, dshift if has_aff else None, None, None, None, None


def conv2d_s2_io(x, weight, out_bf16):
    return _ConvS2IoFn.apply(x, weight, bool(out_bf16))


def conv_transpose2d_io(x, weight, out_bf16):
    return _ConvT2IoFn.apply(x, weight, bool(out_bf16))


def instance_norm_act_io(x, scale, shift, act=ACT_NONE, slope=0.0, eps=1e-5, out_bf16=False):
    return _InstNormIoFn.apply(x, scale, shift, act, slope, eps, bool(out_bf16))


# ---- bf16 compute mode: 16-bit activations around the GENERIC convolutions (round 5) --------------------------------------
# The style encoder's blocks (reference pyfiles/model.py:413-437: norm -> LeakyReLU -> conv3x3 -> norm -> LeakyReLU -> conv3x3 ->
# AvgPool2d, all reflect-padded, on 62 / 31 / 15 / 7-pixel maps) run on the implicit GEMM with 64-deep K tiles
# (csrc/conv_igemm.hip: igemm16_kernel) and the vector weight-gradient kernel.  With fp32 tensors those layers are bound by their
# own result and by the fp32 -> bf16 conversion on the way into LDS (profiles/LOG.md, round 4); here the tensors between the
# norms, the convolutions and the pool are bf16:
#     x (fp32, residual stream) -> norm1 (fp32 -> bf16) -> conv1 (bf16 -> bf16) -> norm2 (bf16 -> bf16) -> conv2 (bf16 -> bf16)
#       -> AvgPool2d (bf16 -> fp32) -> + shortcut (fp32)
# A bf16 norm OUTPUT changes no product (the convolution rounds its operand anyway), a bf16 conv OUTPUT is what torch.autocast
# stores; gradients take the types of the tensors they belong to, and both operands of the weight gradient are bf16 tensors.


_NO_CONV_IO16 = _lib.ab("SRGAN_NO_CONV_IO16")      # A/B (experiment build only): the encoder's blocks on fp32 tensors


def conv_io_applicable(n, ci, hi, wi, weight, pad, pad_mode):
    """True when a stride-1 Conv2d without bias (ci -> weight.shape[0], hi x wi) can take and write bf16 tensors in all three
    directions: bf16 mode, packed-weight scope, every direction on the generic bf16 kernels."""
    if not (STORAGE_BF16 and _pack_cache_on and get_compute_dtype() == "bf16" and weight.is_cuda) or _NO_CONV_IO16:
        return False
    co, _, kh, kw = weight.shape
    ho, wo = hi + 2 * pad - kh + 1, wi + 2 * pad - kw + 1
    if ho < 1 or wo < 1:
        return False
    desc = _conv_desc(n, hi, wi, ci, ho, wo, co, kh, kw, 1, pad, pad_mode, weight)
    return bool(_lib.load().srgan_igemm16_io_applicable(ctypes.byref(desc), ACT_NONE))


def _to16(t):
    return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)


class _ConvIoFn(Function):
    """Stride-1 Conv2d without bias in the bf16 mode, input fp32 or bf16, output bf16 or fp32 (see above)."""

    @staticmethod
    def forward(ctx, x, weight, pad, pad_mode, out_bf16):
        x = to_nhwc(x)
        n, i, hi, wi = x.shape
        o, i2, kh, kw = weight.shape
        if i != i2:
            raise _lib.SrganHipError(f"conv2d_io: input has {i} channels, weight expects {i2}")
        ho, wo = hi + 2 * pad - kh + 1, wi + 2 * pad - kw + 1
        desc = _conv_desc(n, hi, wi, i, ho, wo, o, kh, kw, 1, pad, pad_mode, weight)
        hit, scratch = _packed(desc, weight, 0, ACT_NONE)
        ws = workspace(x.device, scratch) if scratch else None
        y = nhwc_empty(n, o, ho, wo, x.device, torch.bfloat16 if out_bf16 else torch.float32)
        _lib.check(_lib.load().srgan_igemm16_conv(ctypes.byref(desc), 0, _ptr(x), _is16(x), _ptr(hit.buf), None, _ptr(y), _is16(y),
                                                  ACT_NONE, 0.0, _ptr(ws), scratch, _stream()), "igemm16_conv")
        ctx.desc, ctx.weight = desc, weight
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = to_nhwc(gy)
        weight = ctx.weight
        lib = _lib.load()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            hit, scratch = _packed(ctx.desc, weight, 1, ACT_NONE)
            ws = workspace(gy.device, scratch) if scratch else None
            dx = torch.empty_like(x)
            _lib.check(lib.srgan_igemm16_conv(ctypes.byref(ctx.desc), 1, _ptr(gy), _is16(gy), _ptr(hit.buf), None, _ptr(dx), _is16(dx),
                                              ACT_NONE, 0.0, _ptr(ws), scratch, _stream()), "igemm16_conv")
        if ctx.needs_input_grad[1]:
            slots = _sink_slots(weight)
            (dw,), acc = slots if slots is not None else ([torch.empty(weight.shape, dtype=torch.float32, device=weight.device)], False)
            dd = ConvDesc.from_buffer_copy(ctx.desc)
            dd.sO, dd.sI, dd.sH, dd.sW = dw.stride()
            ws, nb = _conv_ws(dd, gy.device)
            # both operands as bf16 tensors (the kernel rounds them to bf16 anyway; an fp32 side -- a block boundary -- is
            # rounded by one elementwise pass here)
            x16, g16 = _to16(x), _to16(gy)
            with _wgrad_accumulate(acc, slots is not None):
                _lib.check(lib.srgan_igemm16_wgrad(ctypes.byref(dd), _ptr(x16), _ptr(g16), _ptr(dw), _ptr(ws), nb, _stream()),
                           "igemm16_wgrad")
            if slots is not None:
                dw = None
        return dx, dw, None, None, None


def conv2d_io(x, weight, padding=0, pad_mode=PAD_ZERO, out_bf16=True):
    return _ConvIoFn.apply(x, weight, padding, pad_mode, bool(out_bf16))


class _AvgPool2IoFn(Function):
    """AvgPool2d(2, 2) with an fp32 or bf16 input and a bf16 or fp32 output; dx has x's type."""

    @staticmethod
    def forward(ctx, x, out_bf16):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = nhwc_empty(n, c, h // 2, w // 2, x.device, torch.bfloat16 if out_bf16 else torch.float32)
        _lib.check(_lib.load().srgan_avgpool2_fwd_io(_ptr(x), _is16(x), _ptr(y), _is16(y), n, h, w, c, _stream()), "avgpool2_fwd_io")
        ctx.shape, ctx.x_dtype = (n, c, h, w), x.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        n, c, h, w = ctx.shape
        gy = to_nhwc(gy)
        dx = nhwc_empty(n, c, h, w, gy.device, ctx.x_dtype)
        _lib.check(_lib.load().srgan_avgpool2_bwd_io(_ptr(gy), _is16(gy), _ptr(dx), _is16(dx), n, h, w, c, _stream()), "avgpool2_bwd_io")
        return dx, None


def avgpool2_io(x, out_bf16=False):
    return _AvgPool2IoFn.apply(x, bool(out_bf16))


RESBLOCK_BF16_STORAGE = True      # tests/test_ops_gpu.py switches the bf16-storage node off to compare it with the unfused chain


def res_block_bf16_fusable(x, w1, w2, s1, s2):
    """True when ``residual_block_bf16`` applies: bf16 compute mode, packed-weight scope, affine (scale, shift) pairs, both
    convolutions 3x3 square-channel layers on the LDS-resident-patch kernels, a map the slab norm kernels hold."""
    if not (_pack_cache_on and get_compute_dtype() == "bf16" and x.is_cuda and x.dim() == 4 and s1 is not None and s2 is not None):
        return False
    if not RESBLOCK_BF16_STORAGE:
        return False
    n, c, h, w = x.shape
    if tuple(w1.shape) != (c, c, 3, 3) or tuple(w2.shape) != (c, c, 3, 3):
        return False
    lib = _lib.load()
    if not lib.srgan_instnorm_io_applicable(n, h * w, c):
        return False
    desc = _conv_desc(n, h, w, c, h, w, c, 3, 3, 1, 1, PAD_ZERO, w1)
    return bool(lib.srgan_halo16_applicable(ctypes.byref(desc)))


def residual_block_bf16(x, s1, h1, s2, h2, w1, w2, eps=1e-5):
    """cbin2(c2(relu(cbin1(c1(x))))) + x in the bf16 mode with bf16 intermediates -- see _ResBlockBf16Fn."""
    return _ResBlockBf16Fn.apply(x, s1, h1, s2, h2, w1, w2, eps)


def res_block_fusable(x, w1, w2, s1, s2):
    """True when ``residual_block`` applies: packed-weight scope, 32x32 map, affine (scale, shift) pairs present, both
    convolutions 3x3 square-channel layers whose forward, input gradient and weight gradient all dispatch to F(4x4,3x3)."""
    if not (_pack_cache_on and x.is_cuda and x.dim() == 4 and s1 is not None and s2 is not None):
        return False
    if _lib.ab("SRGAN_NO_RESBLOCK_FUSION"):
        return False
    n, c, h, w = x.shape
    if (h, w) != (32, 32) or tuple(w1.shape) != (c, c, 3, 3) or tuple(w2.shape) != (c, c, 3, 3):
        return False
    lib = _lib.load()
    for wt in (w1, w2):
        desc = _conv_desc(n, h, w, c, h, w, c, 3, 3, 1, 1, PAD_ZERO, wt)
        if not (lib.srgan_instnorm_conv_v_applicable(ctypes.byref(desc)) and lib.srgan_instnorm_bwd_vz_applicable(ctypes.byref(desc))
                and lib.srgan_conv2d_wgrad_v_bytes(ctypes.byref(desc))):
            return False
    return True


def residual_block(x, s1, h1, s2, h2, w1, w2, eps=1e-5):
    """cbin2(c2(relu(cbin1(c1(x))))) + x with (scale, shift) = (s1, h1), (s2, h2) -- see _ResBlockFn."""
    return _ResBlockFn.apply(x, s1, h1, s2, h2, w1, w2, eps)


def norm_act_conv_fusable(x, weight):
    """True when ``instance_norm_act_conv`` applies: packed-weight scope, 32x32 map, the conv dispatches to F(4x4,3x3)."""
    if not (_pack_cache_on and x.is_cuda and x.dim() == 4 and weight.dim() == 4 and weight.shape[2] == 3 and weight.shape[3] == 3):
        return False
    if _lib.ab("SRGAN_NO_NORM_CONV_FUSION"):
        return False
    n, c, h, w = x.shape
    if (h, w) != (32, 32) or weight.shape[1] != c:
        return False
    desc = _conv_desc(n, h, w, c, h, w, weight.shape[0], 3, 3, 1, 1, PAD_ZERO, weight)
    return bool(_lib.load().srgan_instnorm_conv_v_applicable(ctypes.byref(desc)))


def instance_norm_act_conv(x, scale, shift, weight, act=ACT_NONE, slope=0.0, eps=1e-5):
    """conv2d(act(instance_norm(x) * scale[n,c] + shift[n,c]), weight, stride 1, pad 1) -- see _NormActConvFn."""
    return _NormActConvFn.apply(x, scale, shift, weight, act, slope, eps)


class _CbinAffineFn(Function):
    @staticmethod
    def forward(ctx, c, W, b, gamma, beta):
        _require_gpu(c, "cbin condition vector")
        c = _dense2d(c)
        n, nc = c.shape
        ch = W.shape[0]
        t = torch.empty(n, ch, dtype=torch.float32, device=c.device)
        scale = torch.empty_like(t)
        shift = torch.empty_like(t)
        _lib.check(_lib.load().srgan_cbin_affine_fwd(_ptr(c), _ptr(W), _ptr(b), _ptr(gamma), _ptr(beta), _ptr(t),
                                                     _ptr(scale), _ptr(shift), n, ch, nc, _stream()), "cbin_affine_fwd")
        ctx.W = W                      # Linear weight: by reference (read at backward time)
        ctx.save_for_backward(c, t, scale)
        return scale, shift

    @staticmethod
    def backward(ctx, dscale, dshift):
        c, t, scale = ctx.saved_tensors
        W = ctx.W
        n, nc = c.shape
        ch = W.shape[0]
        dev = c.device
        if dscale is None:
            dscale = torch.zeros(n, ch, dtype=torch.float32, device=dev)
        if dshift is None:
            dshift = torch.zeros(n, ch, dtype=torch.float32, device=dev)
        dgamma = torch.empty(ch, dtype=torch.float32, device=dev)
        dbeta = torch.empty_like(dgamma)
        dW = torch.empty(ch, nc, dtype=torch.float32, device=dev)
        db = torch.empty_like(dgamma)
        dc = torch.empty(n, nc, dtype=torch.float32, device=dev)
        ws = workspace(dev, n * ch * 4)
        # gamma as seen by the forward = row 0 of the saved scale (the reference multiplies by a
        # repeat()-copy of gamma made at forward time, model.py:64)
        _lib.check(_lib.load().srgan_cbin_affine_bwd(_ptr(c), _ptr(W), _ptr(scale), _ptr(t), _ptr(_dense2d(dscale)),
                                                     _ptr(_dense2d(dshift)), _ptr(dgamma), _ptr(dbeta), _ptr(dW), _ptr(db),
                                                     _ptr(dc), n, ch, nc, _ptr(ws), n * ch * 4, _stream()),
                   "cbin_affine_bwd")
        return dc, dW, db, dgamma, dbeta


def cbin_affine(c, W, b, gamma, beta):
    """-> (scale[N,C], shift[N,C]) with scale = gamma, shift = tanh(c W^T + b)*gamma + beta."""
    return _CbinAffineFn.apply(c, W, b, gamma, beta)


def _cbin_table(records, device):
    """Host records (ctypes buffers) -> one device array (kernel-argument upload: capturable, nothing to keep alive)."""
    return upload_small(b"".join(bytes(r) for r in records), device)


class _CbinAffineMultiFn(Function):
    """cbin_affine of L layers in one launch (backward: two).  params = (W_1, b_1, gamma_1, beta_1, ..., W_L, ...);
    returns (scale_1, shift_1, ..., scale_L, shift_L).  Same kept semantics as _CbinAffineFn: the Linear weights are held by
    reference and read at backward time, gamma in backward is the forward-time copy (row 0 of scale)."""

    @staticmethod
    def forward(ctx, c, *params):
        _require_gpu(c, "cbin condition vector")
        lib = _lib.load()
        c = _dense2d(c)
        n, nc = c.shape
        L = len(params) // 4
        chs = [params[4 * l].shape[0] for l in range(L)]
        dev = c.device
        outs, ts, recs = [], [], []
        nb = lib.srgan_cbin_rec_bytes()
        for l in range(L):
            W, b, g, be = params[4 * l: 4 * l + 4]
            t = torch.empty(n, chs[l], dtype=torch.float32, device=dev)
            scale = torch.empty_like(t)
            shift = torch.empty_like(t)
            rec = (ctypes.c_char * nb)()
            _lib.check(lib.srgan_cbin_rec_fill(ctypes.byref(rec), _ptr(W), _ptr(b), _ptr(g), _ptr(be), _ptr(t), _ptr(scale),
                                               _ptr(shift), None, None, None, None, None, None, None, chs[l]), "cbin_rec_fill")
            recs.append(rec)
            ts.append(t)
            outs += [scale, shift]
        table = _cbin_table(recs, dev)
        _lib.check(lib.srgan_cbin_affine_multi_fwd(_ptr(c), _ptr(table), L, n, max(chs), nc, _stream()), "cbin_affine_multi_fwd")
        ctx.Ws = [params[4 * l] for l in range(L)]          # by reference: read at backward time
        ctx.params = params                                 # the parameter objects (gradient sink: see fused_param_grads)
        ctx.chs = chs
        ctx.save_for_backward(c, *ts, *outs[0::2])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        saved = ctx.saved_tensors
        L = len(ctx.chs)
        c, ts, scales = saved[0], saved[1:1 + L], saved[1 + L:1 + 2 * L]
        n, nc = c.shape
        dev = c.device
        nb = lib.srgan_cbin_rec_bytes()
        zeros = None
        res, recs = [], []
        da = workspace(dev, n * sum(ctx.chs) * 4).view(torch.float32)
        off = 0
        for l in range(L):
            ch = ctx.chs[l]
            dscale, dshift = grads[2 * l], grads[2 * l + 1]
            if dscale is None or dshift is None:
                if zeros is None:
                    zeros = torch.zeros(n, max(ctx.chs), dtype=torch.float32, device=dev)
                dscale = zeros if dscale is None else dscale
                dshift = zeros if dshift is None else dshift
            dscale, dshift = _dense2d(dscale), _dense2d(dshift)
            slots = _sink_slots(*ctx.params[4 * l: 4 * l + 4]) if all(ctx.needs_input_grad[1 + 4 * l: 5 + 4 * l]) else None
            if slots is not None:
                (dW, db, dgamma, dbeta), acc = slots
            else:
                acc = False
                dgamma = torch.empty(ch, dtype=torch.float32, device=dev)
                dbeta = torch.empty_like(dgamma)
                dW = torch.empty(ch, nc, dtype=torch.float32, device=dev)
                db = torch.empty_like(dgamma)
            rec = (ctypes.c_char * nb)()
            _lib.check(lib.srgan_cbin_rec_fill(ctypes.byref(rec), _ptr(ctx.Ws[l]), None, _ptr(scales[l]), None, _ptr(ts[l]), None,
                                               None, _ptr(dscale), _ptr(dshift), _ptr(dgamma), _ptr(dbeta), _ptr(dW), _ptr(db),
                                               da.data_ptr() + off * 4, ch), "cbin_rec_fill")
            if acc:
                _lib.check(lib.srgan_cbin_rec_set_accumulate(ctypes.byref(rec), 1), "cbin_rec_set_accumulate")
            off += n * ch
            recs.append((rec, dscale, dshift))
            res += [None] * 4 if slots is not None else [dW, db, dgamma, dbeta]
        table = _cbin_table([r[0] for r in recs], dev)
        dc = torch.empty(n, nc, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        _lib.check(lib.srgan_cbin_affine_multi_bwd(_ptr(c), _ptr(table), L, n, max(ctx.chs), nc, _ptr(dc), _stream()),
                   "cbin_affine_multi_bwd")
        return (dc, *res)


def cbin_affine_multi(c, layer_params):
    """layer_params: [(W, b, gamma, beta), ...] -> [(scale, shift), ...], one launch for all layers."""
    flat = [p for lp in layer_params for p in lp]
    out = _CbinAffineMultiFn.apply(c, *flat)
    return [(out[2 * l], out[2 * l + 1]) for l in range(len(layer_params))]


# ------------------------------------------------------------------------------------------
# pointwise / pools / heads
# ------------------------------------------------------------------------------------------
class _TanhFn(Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, "tanh")
        x = to_nhwc(x) if x.dim() == 4 else x.contiguous()
        y = torch.empty_like(x)
        _lib.check(_lib.load().srgan_tanh_fwd(_ptr(x), _ptr(y), x.numel(), _stream()), "tanh_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        if y.dim() == 4:
            gy = to_nhwc(gy)
        dx = torch.empty_like(y)
        _lib.check(_lib.load().srgan_tanh_bwd(_ptr(y), _ptr(gy), _ptr(dx), y.numel(), _stream()), "tanh_bwd")
        return dx


def tanh(x):
    return _TanhFn.apply(x)


class _ActFn(Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        _require_gpu(x, "activation")
        x = to_nhwc(x) if x.dim() == 4 else x.contiguous()
        y = torch.empty_like(x)
        _lib.check(_lib.load().srgan_act_fwd(_ptr(x), _ptr(y), x.numel(), act, float(slope), _stream()), "act_fwd")
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        if y.dim() == 4:
            gy = to_nhwc(gy)
        return _act_bwd(y, gy, ctx.act, ctx.slope), None, None


def activation(x, act, slope=0.0):
    return _ActFn.apply(x, act, slope)


class _AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = to_nhwc(a), to_nhwc(b)
        y = torch.empty_like(a)
        _lib.check(_lib.load().srgan_add(_ptr(a), _ptr(b), _ptr(y), a.numel(), _stream()), "add")
        return y

    @staticmethod
    def backward(ctx, gy):
        return gy, gy


def add(a, b):
    return _AddFn.apply(a, b)


def _pool_fn(fwd_name, bwd_name, out_hw):
    class _Pool(Function):
        @staticmethod
        def forward(ctx, x):
            x = to_nhwc(x)
            n, c, h, w = x.shape
            ho, wo = out_hw(h, w)
            y = nhwc_empty(n, c, ho, wo, x.device)
            _lib.check(getattr(_lib.load(), fwd_name)(_ptr(x), _ptr(y), n, h, w, c, _stream()), fwd_name)
            ctx.shape = (n, c, h, w)
            return y

        @staticmethod
        def backward(ctx, gy):
            n, c, h, w = ctx.shape
            gy = to_nhwc(gy)
            dx = nhwc_empty(n, c, h, w, gy.device)
            _lib.check(getattr(_lib.load(), bwd_name)(_ptr(gy), _ptr(dx), n, h, w, c, _stream()), bwd_name)
            return dx
    return _Pool


_AvgPool3s2 = _pool_fn("srgan_avgpool3s2_fwd", "srgan_avgpool3s2_bwd", lambda h, w: ((h - 1) // 2 + 1, (w - 1) // 2 + 1))
_AvgPool2 = _pool_fn("srgan_avgpool2_fwd", "srgan_avgpool2_bwd", lambda h, w: (h // 2, w // 2))


def avgpool3s2(x):
    """AvgPool2d(3, stride=2, padding=1, count_include_pad=False)."""
    return _AvgPool3s2.apply(x)


def avgpool2(x):
    """AvgPool2d(2, 2)."""
    return _AvgPool2.apply(x)


class _LreluGapFn(Function):
    @staticmethod
    def forward(ctx, x, slope):
        x = to_nhwc(x)
        n, c, h, w = x.shape
        y = torch.empty(n, c, dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().srgan_lrelu_gap_fwd(_ptr(x), _ptr(y), n, h * w, c, float(slope), _stream()), "lrelu_gap_fwd")
        ctx.slope = slope
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        _lib.check(_lib.load().srgan_lrelu_gap_bwd(_ptr(x), _ptr(_dense2d(gy)), _ptr(dx), n, h * w, c, float(ctx.slope),
                                                   _stream()), "lrelu_gap_bwd")
        return dx, None


def lrelu_global_avgpool(x, slope):
    """LeakyReLU(slope) followed by AdaptiveAvgPool2d(1), flattened to [N, C]."""
    return _LreluGapFn.apply(x, slope)


class _LinearFn(Function):
    @staticmethod
    def forward(ctx, x, W, b):
        _require_gpu(x, "linear")
        x = _dense2d(x)
        m, k = x.shape
        n = W.shape[0]
        y = torch.empty(m, n, dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().srgan_linear_fwd(_ptr(x), _ptr(W), _ptr(b), _ptr(y), m, n, k, _stream()), "linear_fwd")
        ctx.W, ctx.has_bias = W, b is not None
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        W = ctx.W
        m, k = x.shape
        n = W.shape[0]
        gy = _dense2d(gy)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        dW = torch.empty(n, k, dtype=torch.float32, device=x.device) if need_w else None
        db = torch.empty(n, dtype=torch.float32, device=x.device) if (need_w and ctx.has_bias) else None
        _lib.check(_lib.load().srgan_linear_bwd(_ptr(x), _ptr(W), _ptr(gy), _ptr(dx), _ptr(dW), _ptr(db), m, n, k, _stream()),
                   "linear_bwd")
        return dx, dW, db


def linear(x, W, b=None):
    return _LinearFn.apply(x, W, b)


# ------------------------------------------------------------------------------------------
# fused losses (value + gradient in one launch)
# ------------------------------------------------------------------------------------------
class _MseConstFn(Function):
    @staticmethod
    def forward(ctx, o, target, weight):
        _require_gpu(o, "mse_const")
        o = o if (o.is_contiguous() or is_nhwc_dense(o)) else o.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=o.device)
        d_o = torch.empty_like(o)
        _lib.check(_lib.load().srgan_mse_const(_ptr(o), o.numel(), float(target), float(weight), _ptr(loss), _ptr(d_o),
                                               _stream()), "mse_const")
        ctx.save_for_backward(d_o)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_o,) = ctx.saved_tensors
        return d_o * g, None, None


def mse_const(o, target, weight=1.0):
    """weight * mean((o - target)^2)."""
    return _MseConstFn.apply(o, target, weight)


class _SoftmaxMseFn(Function):
    @staticmethod
    def forward(ctx, z, label, weight):
        _require_gpu(z, "softmax_mse")
        z = _dense2d(z)
        b, nc = z.shape
        label = label.to(device=z.device, dtype=torch.int64).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=z.device)
        q = torch.empty_like(z)
        dz = torch.empty_like(z)
        _lib.check(_lib.load().srgan_softmax_mse(_ptr(z), _ptr(label), b, nc, float(weight), _ptr(q), _ptr(loss), _ptr(dz),
                                                 _stream()), "softmax_mse")
        ctx.save_for_backward(dz)
        ctx.mark_non_differentiable(q)
        return loss, q

    @staticmethod
    def backward(ctx, g, _gq):
        (dz,) = ctx.saved_tensors
        return dz * g, None, None


def softmax_mse(z, label, weight=1.0):
    """-> (weight * mean((softmax(z) - onehot(label))^2), softmax(z))."""
    return _SoftmaxMseFn.apply(z, label, weight)


class _SoftmaxXentFn(Function):
    @staticmethod
    def forward(ctx, z, label, weight):
        _require_gpu(z, "softmax_xent")
        z = _dense2d(z)
        b, nc = z.shape
        label = label.to(device=z.device, dtype=torch.int64).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        _lib.check(_lib.load().srgan_softmax_xent(_ptr(z), _ptr(label), b, nc, float(weight), _ptr(loss), _ptr(dz),
                                                  _stream()), "softmax_xent")
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None, None


def softmax_xent(z, label, weight=1.0):
    """weight * mean cross-entropy of row-softmax(z) against integer labels (nn.CrossEntropyLoss semantics)."""
    return _SoftmaxXentFn.apply(z, label, weight)


class _L1MeanFn(Function):
    @staticmethod
    def forward(ctx, a, b, weight):
        _require_gpu(a, "l1_mean")
        if a.dim() == 4:
            a, b = to_nhwc(a), to_nhwc(b)
        else:
            a, b = a.contiguous(), b.contiguous()
        lib = _lib.load()
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        da = torch.empty_like(a) if need_a else None
        db = torch.empty_like(b) if need_b else None
        nb = lib.srgan_l1_workspace(a.numel())
        ws = workspace(a.device, nb)
        _lib.check(lib.srgan_l1_mean(_ptr(a), _ptr(b), a.numel(), float(weight), _ptr(loss), _ptr(da), _ptr(db), _ptr(ws),
                                     nb, _stream()), "l1_mean")
        ctx.save_for_backward(da, db)
        return loss

    @staticmethod
    def backward(ctx, g):
        da, db = ctx.saved_tensors
        return (da * g if da is not None else None, db * g if db is not None else None, None)


def l1_mean(a, b, weight=1.0):
    """weight * mean(|a - b|)."""
    return _L1MeanFn.apply(a, b, weight)


class _LatentLossFn(Function):
    @staticmethod
    def forward(ctx, mu, n_batch, target, bins, range_max, sigma, w_bkl, w_corr, w_hist):
        _require_gpu(mu, "latent_losses")
        mu = _dense2d(mu)
        b, d = mu.shape
        vals = torch.empty(4, dtype=torch.float32, device=mu.device)
        dmu = torch.empty_like(mu)
        corr = torch.empty(d, d, dtype=torch.float32, device=mu.device)
        _lib.check(_lib.load().srgan_latent_losses(_ptr(mu), b, d, float(n_batch), _ptr(target), bins, float(range_max),
                                                   float(sigma), float(w_bkl), float(w_corr), float(w_hist), _ptr(vals),
                                                   _ptr(dmu), _ptr(corr), _stream()), "latent_losses")
        ctx.save_for_backward(dmu)
        total, parts = vals[3], vals[:3]
        ctx.mark_non_differentiable(parts, corr)
        return total, parts, corr

    @staticmethod
    def backward(ctx, g, _gp, _gc):
        (dmu,) = ctx.saved_tensors
        return (dmu * g,) + (None,) * 8


def latent_losses(mu, n_batch, hist_target, w_bkl, w_corr, w_hist, bins=50, range_max=10.0, sigma=0.2):
    """-> (w_bkl*bKL + w_corr*corr + w_hist*hist, tensor([bKL, corr, hist]), Pearson matrix [d,d])."""
    return _LatentLossFn.apply(mu, n_batch, hist_target, bins, range_max, sigma, w_bkl, w_corr, w_hist)


class _MsePairFn(Function):
    @staticmethod
    def forward(ctx, a, b, weight):
        _require_gpu(a, "mse_pair")
        a, b = a.contiguous(), b.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.load().srgan_mse_pair(_ptr(a), _ptr(b), a.numel(), float(weight), _ptr(loss), _ptr(da), _ptr(db),
                                              _stream()), "mse_pair")
        ctx.save_for_backward(da, db)
        return loss

    @staticmethod
    def backward(ctx, g):
        da, db = ctx.saved_tensors
        return (da * g if da is not None else None, db * g if db is not None else None, None)


def mse_pair(a, b, weight=1.0):
    """weight * mean((a - b)^2) for two small tensors of equal shape."""
    return _MsePairFn.apply(a, b, weight)


class _DLossesFn(Function):
    @staticmethod
    def forward(ctx, label, rows_first, t_first, t_rest, w_class, n_scales, *tensors):
        outs, logits = tensors[:n_scales], tensors[n_scales:]
        lib = _lib.load()
        dev = outs[0].device
        rows = outs[0].shape[0]
        o = [t if (t.is_contiguous() or is_nhwc_dense(t)) else t.contiguous() for t in outs]
        for t in o:
            _require_gpu(t, "d_losses")
            if t.dim() == 4 and t.shape[1] != 1:
                raise _lib.SrganHipError("d_losses: LSGAN maps must have one channel")
        z = [_dense2d(t) for t in logits]
        nc = z[0].shape[1] if z else 0
        d_o = [torch.empty_like(t) for t in o]
        dz = [torch.empty_like(t) for t in z]
        vals = torch.empty(4, dtype=torch.float32, device=dev)
        S = n_scales
        arr = ctypes.c_void_p * S
        per = (ctypes.c_longlong * S)(*[t.numel() // rows for t in o])
        if label is not None:
            label = label.to(device=dev, dtype=torch.int64).contiguous()
        _lib.check(lib.srgan_d_losses(arr(*[t.data_ptr() for t in o]), per, arr(*[t.data_ptr() for t in z]) if z else None, S, rows,
                                      int(rows_first), int(nc), _ptr(label), float(t_first), float(t_rest), float(w_class),
                                      _ptr(vals), arr(*[t.data_ptr() for t in d_o]), arr(*[t.data_ptr() for t in dz]) if z else None,
                                      _stream()), "d_losses")
        ctx.save_for_backward(*d_o, *dz)
        total, parts = vals[3], vals[:3]
        ctx.mark_non_differentiable(parts)
        return total, parts

    @staticmethod
    def backward(ctx, g, _gp):
        grads = torch._foreach_mul(list(ctx.saved_tensors), g)      # one multi-tensor launch
        return (None,) * 6 + tuple(grads)


def d_losses(outs, logits, label, rows_first, t_first, t_rest, w_class):
    """Every loss of one discriminator evaluation in one launch -> (total, tensor([lsgan_first, class, lsgan_rest])):
    the first ``rows_first`` rows against ``t_first`` (+ softmax / class-MSE against ``label``), the rest against ``t_rest``;
    total = lsgan_first + w_class * class + lsgan_rest, each term the mean over the scales (util.py:457-468)."""
    logits = list(logits) if logits else []
    return _DLossesFn.apply(label, rows_first, t_first, t_rest, w_class, len(outs), *outs, *logits)


class _LincombFn(Function):
    @staticmethod
    def forward(ctx, weights, *terms):
        n = len(terms)
        dev = terms[0].device
        terms = [t.reshape(()) for t in terms]
        out = torch.empty((), dtype=torch.float32, device=dev)
        w = (ctypes.c_float * n)(*[float(v) for v in weights])
        _lib.check(_lib.load().srgan_lincomb((ctypes.c_void_p * n)(*[t.data_ptr() for t in terms]), w, n, _ptr(out), _stream()),
                   "lincomb")
        ctx.weights, ctx.dev = [float(v) for v in weights], dev
        ctx.keep = terms                  # the launch reads them asynchronously
        return out

    @staticmethod
    def backward(ctx, g):
        n = len(ctx.weights)
        dx = torch.empty(n, dtype=torch.float32, device=ctx.dev)
        w = (ctypes.c_float * n)(*ctx.weights)
        _lib.check(_lib.load().srgan_lincomb_bwd(w, n, _ptr(g.contiguous()), _ptr(dx), _stream()), "lincomb_bwd")
        return (None,) + tuple(dx[i] if ctx.needs_input_grad[1 + i] else None for i in range(n))


def lincomb(pairs):
    """sum of weight * scalar over [(0-dim device tensor, python float), ...] (<= 16 terms) in one launch; differentiable."""
    pairs = [(t, w) for t, w in pairs if t is not None]
    if len(pairs) > 16:
        return lincomb([(lincomb(pairs[:16]), 1.0)] + pairs[16:])
    for t, _ in pairs:
        _require_gpu(t, "lincomb")
    return _LincombFn.apply([w for _, w in pairs], *[t for t, _ in pairs])


class _KlNormalFn(Function):
    @staticmethod
    def forward(ctx, mu, logvar, weight):
        _require_gpu(mu, "kl_normal")
        mu, logvar = mu.contiguous(), logvar.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=mu.device)
        dmu = torch.empty_like(mu) if ctx.needs_input_grad[0] else None
        dlv = torch.empty_like(logvar) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.load().srgan_kl_normal(_ptr(mu), _ptr(logvar), mu.numel(), float(weight), _ptr(loss), _ptr(dmu), _ptr(dlv),
                                               _stream()), "kl_normal")
        ctx.save_for_backward(dmu, dlv)
        return loss

    @staticmethod
    def backward(ctx, g):
        dmu, dlv = ctx.saved_tensors
        return (dmu * g if dmu is not None else None, dlv * g if dlv is not None else None, None)


def kl_normal(mu, logvar, weight=1.0):
    """weight * -0.5 * sum(1 + logvar - mu^2 - exp(logvar)): the conventional KL term (util_notebook.py:630-634), value and
    both gradients in one launch."""
    return _KlNormalFn.apply(mu, logvar, weight)


class _SoftHistFn(Function):
    @staticmethod
    def forward(ctx, x, bins, lo, hi, sigma):
        _require_gpu(x, "soft_histogram")
        x = x.contiguous()
        lib = _lib.load()
        h = torch.empty(bins, dtype=torch.float32, device=x.device)
        nb = lib.srgan_soft_histogram_workspace(x.numel(), bins)
        ws = workspace(x.device, nb)
        _lib.check(lib.srgan_soft_histogram_fwd(_ptr(x), x.numel(), bins, float(lo), float(hi), float(sigma), _ptr(h),
                                                _ptr(ws), nb, _stream()), "soft_histogram_fwd")
        ctx.cfg = (bins, lo, hi, sigma)
        ctx.save_for_backward(x)
        return h

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        bins, lo, hi, sigma = ctx.cfg
        dx = torch.empty_like(x)
        _lib.check(_lib.load().srgan_soft_histogram_bwd(_ptr(x), _ptr(g.contiguous()), x.numel(), bins, float(lo), float(hi),
                                                        float(sigma), _ptr(dx), _stream()), "soft_histogram_bwd")
        return dx, None, None, None, None


def soft_histogram(x, bins, lo, hi, sigma):
    """Gaussian-kernel soft histogram of a 1-D sample (differentiable)."""
    return _SoftHistFn.apply(x, bins, lo, hi, sigma)


def adam_step_(p, g, m, v, lr, beta1, beta2, eps, step):
    """In-place torch-1.4 Adam on dense buffers, through raw pointers (no autograd version bump)."""
    _require_gpu(p, "adam")
    _lib.check(_lib.load().srgan_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(beta1),
                                           float(beta2), float(eps), int(step), _stream()), "adam_step")


def adam_state_new(device, lr, beta1, beta2, eps, steps_done):
    """Device-resident Adam record {t, lr, betas, eps, derived step size} seeded with ``steps_done`` completed steps."""
    lib = _lib.load()
    st = torch.empty(lib.srgan_adam_state_bytes(), dtype=torch.uint8, device=device)
    _lib.check(lib.srgan_adam_state_init(_ptr(st), float(lr), float(beta1), float(beta2), float(eps), int(steps_done), _stream()),
               "adam_state_init")
    return st


def adam_state_set_lr(state, lr):
    _lib.check(_lib.load().srgan_adam_state_set_lr(_ptr(state), float(lr), _stream()), "adam_state_set_lr")


def adam_multi_dev_(table_dev, n_tensors, max_numel, state):
    """t += 1 on the device, then one launch of torch-1.4 Adam over the tensors of the pointer table (hipGraph-capturable:
    no host value is baked into the launches)."""
    _lib.check(_lib.load().srgan_adam_multi_dev(_ptr(table_dev), int(n_tensors), int(max_numel), _ptr(state), _stream()),
               "adam_multi_dev")


def adam_multi_step_(table_dev, n_tensors, max_numel, lr, beta1, beta2, eps, step):
    """One launch of torch-1.4 Adam over a list of tensors described by a device pointer table (see optim.Adam)."""
    _lib.check(_lib.load().srgan_adam_multi(_ptr(table_dev), int(n_tensors), int(max_numel), float(lr), float(beta1),
                                            float(beta2), float(eps), int(step), _stream()), "adam_multi")

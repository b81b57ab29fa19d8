"""Data parallelism for the SRGAN train step: one process per GPU, persistent replicas, RCCL over xGMI.

Replaces the reference's ``torch.nn.DataParallel(net, devices)`` (05-train notebook cell 20), which
re-broadcasts every parameter on each of the 24 forwards of a step and gathers activations to GPU 0
(SURVEY.md 2.3).  Here every rank owns its batch shard and a full replica; the only exchanges are
  * ``all_gather`` of mu [B_local, ndim] before the batch-statistics losses (batch-KL / correlation /
    histogram are statistics of the GLOBAL batch), and
  * one bucketed ``all_reduce`` of gradients per optimiser step, launched on a side HIP stream so it can
    overlap independent compute (the next G forward during the D updates).
Backend: ``nccl`` (= RCCL on ROCm) for GPU tensors, ``gloo`` for the CPU tests.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn


def is_distributed():
    """True under a multi-rank process group.  ``SRGAN_DP_FORCE=1`` also takes the data-parallel code path with a
    one-rank group: the 1-GPU test box then drives the real RCCL calls (side-stream all-reduce, all-gather inside autograd),
    which two ranks cannot do on one device."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("SRGAN_DP_FORCE") == "1"


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, device)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("SRGAN_DP_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # override: tests share 1 GPU
    backend = backend or os.environ.get("SRGAN_DP_BACKEND")
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if (ws > 1 or os.environ.get("SRGAN_DP_FORCE") == "1") and not (dist.is_available() and dist.is_initialized()):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if use_gpu else "gloo")
        with _stdout_to_stderr():
            dist.init_process_group(backend, rank=rk, world_size=ws)
        if backend == "nccl":
            # create the RCCL communicator HERE, on the main thread and on this rank's device: otherwise the first collective
            # -- a bucket all-reduce issued from an autograd hook in the backward thread of step 0 -- would also be the one
            # that initialises it
            # (RCCL prints its version banner on stdout when the communicator comes up: keep it off the JSON line's channel)
            with _stdout_to_stderr():
                warm = torch.zeros(1, device=device)
                dist.all_reduce(warm)
                torch.cuda.synchronize(device)
        global _control
        _control = None
        control_group()
        if _transport == "abi" and use_gpu:
            _abi_comm()              # here, collectively and on the main thread -- not in the first bucket's autograd hook
        _gather_into_tensor.clear()
        _probe_gather_into_tensor(device)
    if os.environ.get("SRGAN_DP_BUCKET_DTYPE"):          # "fp32" / "bf16": see set_bucket_dtype (default: follow the compute mode)
        set_bucket_dtype(os.environ["SRGAN_DP_BUCKET_DTYPE"])
    return rk, ws, device


class DataParallel(nn.Module):
    """Drop-in for the notebooks' ``nn.DataParallel(net, devices)``: exposes ``.module`` (used when saving
    checkpoints, 05-train cell 24) and forwards calls to the local replica.  ``device_ids`` is accepted and
    ignored: placement is one process per GPU."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super().__init__()
        self.module = module
        self.device_ids = device_ids

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("module"), name)


def unwrap(net):
    return net.module if hasattr(net, "module") and isinstance(getattr(net, "module"), nn.Module) else net


# Segmented step recording (trainer._StepGraph under data parallelism): while a step is being RECORDED into hipGraphs the
# collectives are not issued -- each one ends the graph segment being recorded, is remembered as a host callable over static
# device buffers, and a new segment begins.  A replay launches segment, collective, segment, ...: the collectives stay the
# plain eager RCCL calls (nothing of RCCL is captured), everything between them is one graph launch.
_recorder = None


def hooks_need_live_grads():
    """True when the gradient hooks must fire WHILE the backward runs: the eager data-parallel step sends a bucket the moment
    its last gradient is final (``GradReducer._on_grad``), so parameter gradients have to pass through autograd's AccumulateGrad
    there; everywhere else the trainer lets the weight-gradient kernels accumulate (``ops.fused_param_grads``)."""
    return is_distributed() and _recorder is None


def recording():
    """True while a data-parallel step is being recorded into hipGraph segments (collectives become cuts of the recording)."""
    return is_distributed() and _recorder is not None


def set_recorder(rec):
    """rec: object with ``cut(comm)`` (end the current graph segment, run ``comm()`` after it on every replay) or None."""
    global _recorder
    _recorder = rec


# ---- transport of the GPU collectives --------------------------------------------------------------------------------
# "torch" (default): torch.distributed's nccl backend (= RCCL) issues them.  "abi" (SRGAN_DP_COMM=abi, or set_transport): the
# library's own entry points over RCCL (csrc/comm.cpp: srgan_allreduce_bucket / srgan_allgather_rows) -- the C-ABI form of the
# same two exchanges, enqueued on the current stream like any other kernel of the step.  The communicator is
# made once per process; its 128-byte id travels from rank 0 over the host-side control group.  Only GPU tensors take it (a
# gloo / CPU run has nothing to hand to RCCL); torch.distributed stays the rendezvous either way.
_transport = os.environ.get("SRGAN_DP_COMM", "torch")
_abi = None                   # (communicator handle, library) once made


def set_transport(name):
    global _transport
    if name not in ("torch", "abi"):
        raise ValueError(f"set_transport: {name!r}")
    _transport = name


def transport():
    return _transport


def _abi_comm():
    """The process's RCCL communicator behind the C ABI, created collectively at first use."""
    global _abi
    if _abi is None:
        import ctypes
        from . import _lib
        lib = _lib.load()
        if not lib.srgan_comm_available():
            raise _lib.SrganHipError("SRGAN_DP_COMM=abi: librccl.so is not loadable on this host")
        buf = ctypes.create_string_buffer(128)
        if dist.get_rank() == 0:
            _lib.check(lib.srgan_comm_unique_id(buf), "srgan_comm_unique_id")
        t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        dist.broadcast(t, 0, group=control_group())
        ident = ctypes.create_string_buffer(bytes(t.tolist()), 128)
        comm = ctypes.c_void_p()
        with _stdout_to_stderr():
            _lib.check(lib.srgan_comm_init(ident, dist.get_world_size(), dist.get_rank(), ctypes.byref(comm)), "srgan_comm_init")
        _abi = (comm, lib)
        import atexit
        atexit.register(abi_comm_destroy)      # the communicator leaves with the process (idempotent; explicit calls are fine)
    return _abi


def abi_comm_destroy():
    global _abi
    if _abi is not None:
        comm, lib = _abi
        _abi = None
        lib.srgan_comm_destroy(comm)


def _use_abi(t):
    return _transport == "abi" and t.is_cuda


_gather_into_tensor = {}      # backend name -> bool: does it implement all_gather_into_tensor (probed once, collectively)


def _probe_gather_into_tensor(device):
    """Feature-detect ``all_gather_into_tensor`` ONCE per backend, at a point every rank reaches together
    (``init_from_env`` right after the group is made, else the first gather of a step).  Whether the call exists is a property
    of the backend build, identical on every rank, so all ranks take the same branch; afterwards the chosen call's errors
    propagate instead of being answered with a DIFFERENT collective on one rank."""
    be = dist.get_backend()
    if be not in _gather_into_tensor:
        src = torch.zeros(1, device=device)
        out = torch.zeros(dist.get_world_size(), device=device)
        try:
            dist.all_gather_into_tensor(out, src)
            _gather_into_tensor[be] = True
        except (RuntimeError, NotImplementedError):
            dist.all_gather(list(out.chunk(dist.get_world_size(), 0)), src)      # keep the ranks' collective sequences equal
            _gather_into_tensor[be] = False
    return _gather_into_tensor[be]


def _all_gather_into(out, src):
    """Rank-ordered rows of every rank's ``src`` into ONE preallocated tensor (``all_gather_into_tensor``: RCCL writes the
    destination directly; a list of per-rank views may be staged through an internal flat buffer and copied out).  Backends
    without the call get the list form on views of ``out`` -- same result.  Errors of the chosen call propagate."""
    if _use_abi(src) and src.dtype == torch.float32:
        from . import _lib
        comm, lib = _abi_comm()
        _lib.check(lib.srgan_allgather_rows(comm, src.data_ptr(), out.data_ptr(), src.numel(), torch.cuda.current_stream(src.device).cuda_stream),
                   "srgan_allgather_rows")
        return
    if _probe_gather_into_tensor(src.device):
        dist.all_gather_into_tensor(out, src)
    else:
        dist.all_gather(list(out.chunk(dist.get_world_size(), 0)), src)


class _AllGatherRows(torch.autograd.Function):
    """Concatenate each rank's [B_local, d] rows in rank order; backward hands every rank the rows of the
    incoming gradient that belong to it (the gathered loss is evaluated redundantly on every rank)."""

    @staticmethod
    def forward(ctx, x):
        ws = dist.get_world_size()
        ctx.rows, ctx.rank = x.shape[0], dist.get_rank()
        src = x.contiguous()
        out = torch.empty((ws * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if _recorder is not None:
            launch_pending()
            _recorder.cut(_Comm("all_gather", lambda: _all_gather_into(out, src)))
            return out
        _all_gather_into(out, src)
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows].contiguous()


def all_gather_rows(x):
    return _AllGatherRows.apply(x) if is_distributed() else x


BUCKET_BYTES = 16 << 20   # per all-reduce message (fp32 bytes of its gradients): several buckets per network, so the first ones run under the backward

_wire_dtype = None        # None: follow the compute mode (bf16 convolutions -> bf16 gradient messages)


def set_bucket_dtype(name):
    """What travels in a gradient all-reduce: "fp32", "bf16" or None = follow ``ops.get_compute_dtype()`` (default).  In the
    bf16 mode (BASELINE configs[2]-[4]) the weight gradients are sums of bf16 products, so averaging them in bf16 on the wire
    costs nothing the mode has not already spent and halves the ~180 MB a step moves over xGMI; the averaged
    message is widened into the bucket's fp32 buffer on the communication stream and THAT is what ``p.grad`` is bound to --
    Adam and the master weights never see a bf16 tensor."""
    global _wire_dtype
    if name not in (None, "fp32", "bf16"):
        raise ValueError(f"set_bucket_dtype: {name!r}")
    _wire_dtype = name


def bucket_dtype(on_gpu=True):
    """The wire dtype in force: an explicit ``set_bucket_dtype`` choice, else the compute mode's (a CPU / gloo run -- ``on_gpu``
    false -- has no compute mode to follow and does not load the HIP library to ask)."""
    if _wire_dtype is not None:
        return _wire_dtype
    if not on_gpu:
        return "fp32"
    from . import ops
    return "bf16" if ops.get_compute_dtype() == "bf16" else "fp32"


class _Bucket:
    """One all-reduce message: a PERSISTENT flat fp32 buffer and, per parameter, the view of it that becomes ``p.grad``."""
    __slots__ = ("params", "flat", "views", "wire", "wire_views", "ready", "done", "launched", "count", "expected", "hit")

    def __init__(self, params):
        self.params = params
        self.flat = None               # allocated at first use (parameters may still move between devices before that)
        self.views = None
        self.wire = self.wire_views = None   # bf16 message (bucket_dtype() == "bf16"); None: the fp32 buffer itself travels
        self.ready = self.done = None  # HIP events: compute -> comm (bucket flattened), comm -> compute (all-reduce finished)
        self.launched = False
        self.count = 0
        self.expected = sum(1 for p in params if p.requires_grad)
        self.hit = [False] * len(params)

    def materialise(self, want16):
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=p0.device)
            self.views, off = [], 0
            for p in self.params:
                n = p.numel()
                self.views.append(self.flat[off:off + n].view(p.shape))      # dense, contiguous: what the fused Adam wants
                off += n
            self.wire = self.wire_views = None
            if p0.is_cuda:
                self.ready, self.done = torch.cuda.Event(), torch.cuda.Event()
        if want16 and self.wire is None:
            self.wire = torch.zeros(self.flat.numel(), dtype=torch.bfloat16, device=self.flat.device)
            self.wire_views = [self.wire[v.storage_offset():v.storage_offset() + v.numel()].view(v.shape) for v in self.views]
        elif not want16 and self.wire is not None:
            self.wire = self.wire_views = None

    def message(self):
        """The tensor the all-reduce runs on."""
        return self.wire if self.wire is not None else self.flat

    def reduce_(self, avg):
        """In-place average of the message across the ranks, then (bf16 message) widened into the fp32 buffer.  Runs on whatever
        stream is current: the communication stream."""
        msg = self.message()
        if _use_abi(msg):
            from . import _lib
            comm, lib = _abi_comm()
            _lib.check(lib.srgan_allreduce_bucket(comm, msg.data_ptr(), msg.numel(), 1 if msg.dtype == torch.bfloat16 else 0, 1,
                                                  torch.cuda.current_stream(msg.device).cuda_stream), "srgan_allreduce_bucket")
            if self.wire is not None:
                self.flat.copy_(msg)
            return
        dist.all_reduce(msg, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM)
        if self.wire is not None:
            self.flat.copy_(msg)
        if not avg:
            self.flat.mul_(1.0 / dist.get_world_size())


_slot_of = {}                 # id(parameter) -> (weakref to its GradReducer, bucket index, index in the bucket)


def _forget_slots(ref, pids):
    for pid in pids:
        hit = _slot_of.get(pid)
        if hit is not None and hit[0] is ref:
            del _slot_of[pid]
    if not _slot_of:
        import sys
        ops = sys.modules.get(__name__.rsplit(".", 1)[0] + ".ops")      # (a finaliser may run at interpreter shutdown)
        if ops is not None and getattr(ops, "_sink_alloc", None) is grad_slot:
            ops._sink_alloc = None


def grad_slot(p):
    """The slice of ``p``'s gradient bucket, for the weight-gradient kernels to write into (``ops.fused_param_grads`` asks when
    it needs a buffer for a parameter's first contribution of a pass): the bucket then IS the gradients and ``_launch`` has
    nothing to lay out -- 15 multi-tensor copies (0.37 ms) of a recorded data-parallel step (VERDICT r4 item 3 i).  None when
    the parameter belongs to no reducer, outside a process group, or off the GPU."""
    hit = _slot_of.get(id(p))
    if hit is None or not p.is_cuda or not is_distributed():
        return None
    red = hit[0]()
    if red is None:
        del _slot_of[id(p)]
        return None
    bk = red._buckets_cache[hit[1]]
    if bk.params[hit[2]] is not p:
        return None
    # the wire type of an exchange is the owning reducer's (resolved once in _reset()); asking the compute mode here
    # could create or drop the bf16 message buffer between arm() and the backward and split the exchange again
    bk.materialise(red._wire16 if red._wire16 is not None else bucket_dtype(True) == "bf16")
    return bk.views[hit[2]]


class GradReducer:
    """Averages ``.grad`` of a parameter list across ranks: bucketed, in-place all-reduce on a side HIP stream.

    Every bucket owns a persistent flat buffer.  When a bucket is sent, its gradients are laid into the buffer by ONE
    multi-tensor copy, the buffer is all-reduced IN PLACE on the communication stream (``ReduceOp.AVG`` on RCCL: no scaling
    pass), and ``finish()`` binds each ``p.grad`` to its slice of the buffer -- nothing is copied back.  The two stream
    dependencies are explicit events, so they hold for an asynchronous backend (RCCL enqueues and returns) as well as for
    gloo: ``ready`` (recorded on the compute stream after the flatten, awaited by the communication stream before the
    all-reduce) and ``done`` (recorded on the communication stream after the all-reduce, awaited by the compute stream in
    ``finish()`` -- and therefore also ordered before the next flatten into the same buffer).

    ``arm()`` before the backward: every parameter carries a post-accumulate-grad hook, and a bucket is sent the moment its
    last gradient is final -- the collectives of the layers near the loss run under the backward conv stack of the layers
    below them (buckets are filled in reverse parameter order, the order in which autograd finishes them).  ``start()``
    after the backward sends whatever is left and returns immediately; ``finish()`` waits and binds the averages.

    Parameters without a gradient take part with zeros so that every rank issues the same calls.  After an ARMED pass a
    parameter that received no gradient (no hook hit -- a structural property of the graph, identical on every rank) keeps
    ``p.grad = None``, so the optimiser skips it exactly as the single-process step does; ``reduce()`` without ``arm()``
    (manual use) hands every parameter its averaged gradient.
    """

    def __init__(self, params):
        self.params = [p for p in params]
        self._pending = None
        self._comm_stream = None
        self._armed = False
        self._waited = False           # recording: the wait for this reducer's buckets already rides in the cut that started them
        self._wire16 = None            # resolved by _reset() (arm() / start()); grad_slot() follows it
        self._buckets_cache = [_Bucket(b) for b in self._buckets()]
        self._where = {}
        for b, bucket in enumerate(self._buckets_cache):
            for j, p in enumerate(bucket.params):
                self._where[id(p)] = (b, j)
        self._work = []
        import weakref
        me = weakref.ref(self)
        for pid, (b, j) in self._where.items():
            _slot_of[pid] = (me, b, j)
        # the registry is keyed by id(parameter): entries leave with their reducer, and the sink allocator of
        # ops.fused_param_grads is dropped when the last reducer of the process has gone
        weakref.finalize(self, _forget_slots, me, list(self._where))
        if hasattr(torch.Tensor, "register_post_accumulate_grad_hook"):
            for p in self.params:
                if p.requires_grad:
                    p.register_post_accumulate_grad_hook(self._on_grad)

    def _buckets(self):
        buckets, cur, size = [], [], 0
        for p in reversed(self.params):               # gradients become final roughly in reverse parameter order
            nbytes = p.numel() * 4
            if cur and size + nbytes > BUCKET_BYTES:
                buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            buckets.append(cur)
        return buckets

    def _reset(self):
        # the wire dtype of this exchange is resolved ONCE, here (arm() / start()), not per bucket launch: bucket_dtype() asks the
        # compute mode, which loads the HIP library -- needless on a CPU / gloo run, and a mode switch between two buckets of one
        # exchange must not split it
        self._wire16 = bucket_dtype(any(p.is_cuda for p in self.params)) == "bf16"
        for bk in self._buckets_cache:
            bk.launched, bk.count = False, 0
            bk.expected = sum(1 for p in bk.params if p.requires_grad)
            bk.hit = [False] * len(bk.params)
        self._work = []

    def arm(self):
        if not is_distributed() or not self.params:
            return
        self._armed = True
        self._reset()

    def _on_grad(self, p):
        if not self._armed:
            return
        b, j = self._where[id(p)]
        bk = self._buckets_cache[b]
        if bk.launched:
            raise RuntimeError("GradReducer: a gradient arrived for a bucket that was already all-reduced "
                               "(a parameter's requires_grad changed between arm() and the backward?)")
        bk.count += 1
        bk.hit[j] = True
        if bk.count == bk.expected and _recorder is None:      # while recording, buckets are sent after the backward (start())
            self._launch(b)

    def _launch(self, b):
        bk = self._buckets_cache[b]
        bk.materialise(self._wire16)
        on_gpu = bk.flat.is_cuda
        if on_gpu and _recorder is None:
            # sent from a hook, inside the backward pass: the bucket's gradients may have been written on more than one stream
            # (the discriminator's second scale runs on a side stream) -- this one waits for the others before it reads them
            from . import ops
            ops.wait_compute_streams(bk.flat.device)
        srcs, dsts, holes = [], [], []
        for p, v in zip(bk.params, bk.wire_views if bk.wire is not None else bk.views):
            g = p.grad
            if g is None:
                holes.append(v)
            elif g.data_ptr() != v.data_ptr():        # already the bucket's own slice (no zero_grad since the last pass)
                srcs.append(g)
                dsts.append(v)                        # (a bf16 message: the multi-tensor copy rounds to nearest even)
        with torch.no_grad():
            if dsts:
                torch._foreach_copy_(dsts, srcs)       # one multi-tensor launch lays the bucket out
            if holes:
                torch._foreach_zero_(holes)
        avg = _backend_has_avg()
        if _recorder is not None:
            # recording: the flatten above is part of the graph segment; the all-reduce is issued between segments, on the
            # communication stream (launch_pending), and awaited at another cut (finish())
            _recorder.pending.append((bk, avg))
        elif on_gpu:
            self._comm_stream = comm_stream(bk.flat.device)
            bk.ready.record(torch.cuda.current_stream(bk.flat.device))
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(bk.ready)
                bk.reduce_(avg)
                bk.done.record(self._comm_stream)
        else:
            bk.reduce_(False)
        bk.launched = True
        self._work.append(bk)

    def start(self):
        if not is_distributed() or not self.params:
            return
        armed = self._armed
        if not armed:
            self._reset()
        self._armed = False
        for b in range(len(self._buckets_cache)):      # same order on every rank
            if not self._buckets_cache[b].launched:
                self._launch(b)
        self._pending = (self._work, armed)
        self._work = []

    def finish(self):
        if self._pending is None:
            return
        work, armed = self._pending
        recording = _recorder is not None
        if recording:
            launch_pending(then_wait=self)      # (a no-op when the caller already started the collectives to put work under them)
            if not self._waited:
                _recorder.cut(_Comm("wait", lambda: _wait_done(work)))
            self._waited = False
        self._pending = None
        for bk in work:
            if bk.done is not None and not recording:
                torch.cuda.current_stream(bk.flat.device).wait_event(bk.done)
            for j, (p, v) in enumerate(zip(bk.params, bk.views)):
                if p.grad is None and armed and not bk.hit[j]:
                    continue                            # received no gradient on any rank: the optimiser skips it
                p.grad = v

    def reduce(self):
        self.start()
        self.finish()


_comm_streams = {}


def comm_stream(device):
    """The ONE communication stream of a device: every reducer's all-reduces are enqueued on it, in program order -- the same
    order on every rank."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _comm_streams.get(key)
    if st is None:
        st = _comm_streams[key] = torch.cuda.Stream(device=device)
    return st


class _Comm:
    """A host callable between two graph segments of a recorded step (``trainer._Recording``), with a name for the trace."""
    __slots__ = ("kind", "fn")

    def __init__(self, kind, fn):
        self.kind, self.fn = kind, fn

    def __call__(self):
        self.fn()


def _all_reduce_async(items):
    """In-place average of flat buckets, issued on the communication stream behind everything the compute stream has been
    given so far; returns at once -- the graph segments launched next run UNDER the collectives (``_wait_done`` joins)."""
    bk0 = items[0][0]
    if not bk0.flat.is_cuda:
        for bk, avg in items:
            bk.reduce_(False)
        return
    dev = bk0.flat.device
    comm = comm_stream(dev)
    bk0.ready.record(torch.cuda.current_stream(dev))
    with torch.cuda.stream(comm):
        comm.wait_event(bk0.ready)
        for bk, avg in items:
            bk.reduce_(avg)
            bk.done.record(comm)


def _wait_done(buckets):
    for bk in buckets:
        if bk.done is not None:
            torch.cuda.current_stream(bk.flat.device).wait_event(bk.done)


def launch_pending(then_wait=None):
    """Recorded data-parallel step: end the graph segment here and, on every replay, start the all-reduce of every bucket
    flattened since the last cut on the communication stream.  What the caller records next runs under those collectives until
    the owning reducer's ``finish()`` (another cut: the compute stream waits for the ``done`` events).  ``then_wait``: a
    reducer whose result is needed at once -- its wait joins this cut instead of making one of its own (an empty graph
    segment between the two is avoided); the other reducers' buckets stay in flight.  Eager step: nothing to do (the buckets
    went out from the gradient hooks / ``start()``)."""
    rec = _recorder
    if rec is None or not rec.pending:
        return
    items, rec.pending = list(rec.pending), []
    joined = []
    if then_wait is not None and then_wait._pending is not None:
        joined = list(then_wait._pending[0])
        then_wait._waited = True

    def comm():
        _all_reduce_async(items)
        _wait_done(joined)
    rec.cut(_Comm("all_reduce", comm))


_control = None


import contextlib


@contextlib.contextmanager
def _stdout_to_stderr():
    """gloo announces its connections on the process's stdout ("[Gloo] Rank 0 is connected to ..."): a launcher that expects ONE
    JSON line there (bench.py's contract) must not see it -- route file descriptor 1 to stderr while a gloo group is made."""
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def control_group():
    """Host-side agreement between ranks (which execution mode a step takes, whether a recording succeeded) travels over a
    gloo group on CPU tensors: no RCCL kernel in the compute stream and no device synchronisation to read the answer.
    Created collectively: ``init_from_env`` makes it right after the process group; otherwise at the first agreement, which
    every rank reaches at the same point of its program."""
    global _control
    if _control is None:
        if dist.get_backend() == "gloo":
            _control = dist.group.WORLD
        else:
            # (a bounded wait: a rank that reaches a vote its peers never join -- see trainer._StepGraph.accepts -- raises
            #  after this instead of sitting in it for the default half hour)
            import datetime
            with _stdout_to_stderr():
                _control = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=5))
    return _control


def all_min(value):
    """min of an int over the ranks (the plain value without a process group).  Every rank must call it at the same point."""
    if not is_distributed():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=control_group())
    return int(t.item())


def all_agree(ok):
    """True iff ``ok`` is true on EVERY rank (one MIN all-reduce on the control group).  Every rank must call it at the same
    point of its program."""
    return all_min(1 if ok else 0) == 1


def _backend_has_avg():
    """RCCL reduces with ncclAvg in the collective itself; gloo has no AVG (sum, then one scaling launch per bucket)."""
    try:
        return dist.get_backend() == "nccl"
    except Exception:
        return False

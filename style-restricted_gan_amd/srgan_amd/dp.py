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
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


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
    if ws > 1 and not (dist.is_available() and dist.is_initialized()):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rk, world_size=ws)
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


class _AllGatherRows(torch.autograd.Function):
    """Concatenate each rank's [B_local, d] rows in rank order; backward hands every rank the rows of the
    incoming gradient that belong to it (the gathered loss is evaluated redundantly on every rank)."""

    @staticmethod
    def forward(ctx, x):
        ws = dist.get_world_size()
        ctx.rows, ctx.rank = x.shape[0], dist.get_rank()
        parts = [torch.empty_like(x) for _ in range(ws)]
        dist.all_gather(parts, x.contiguous())
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        return g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows].contiguous()


def all_gather_rows(x):
    return _AllGatherRows.apply(x) if is_distributed() else x


BUCKET_BYTES = 16 << 20   # per all-reduce message: several buckets per network, so the first ones run under the backward


class GradReducer:
    """Averages ``.grad`` of a parameter list across ranks with bucketed all-reduce on a side stream.

    ``arm()`` before the backward: every parameter carries a post-accumulate-grad hook, and a bucket's all-reduce is
    enqueued on the side stream the moment its last gradient is final -- so the collectives of the layers near the loss
    run under the backward conv stack of the layers below them (buckets are filled in reverse parameter order, the order
    in which autograd finishes them).  ``start()`` after the backward enqueues whatever is left (parameters without a
    gradient take part with zeros so every rank issues the same calls) and returns immediately; ``finish()`` makes the
    compute stream wait for the collectives and scatters the averaged values back.  Without ``arm()`` everything is
    enqueued by ``start()``.
    """

    def __init__(self, params):
        self.params = [p for p in params]
        self._pending = None
        self._comm_stream = None
        self._armed = False
        self._buckets_cache = self._buckets()
        self._where = {}
        for b, bucket in enumerate(self._buckets_cache):
            for p in bucket:
                self._where[id(p)] = b
        self._count = [0] * len(self._buckets_cache)
        self._launched = [False] * len(self._buckets_cache)
        self._work = []
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

    def arm(self):
        if not is_distributed() or not self.params:
            return
        self._armed = True
        self._count = [0] * len(self._buckets_cache)
        self._launched = [False] * len(self._buckets_cache)
        self._work = []

    def _on_grad(self, p):
        if not self._armed:
            return
        b = self._where[id(p)]
        self._count[b] += 1
        if self._count[b] == sum(1 for q in self._buckets_cache[b] if q.requires_grad) and not self._launched[b]:
            self._launch(b)

    def _launch(self, b):
        bucket = self._buckets_cache[b]
        on_gpu = self.params[0].is_cuda
        if on_gpu and self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
        flat = torch.cat([g.reshape(-1) for g in grads])
        if on_gpu:
            # the side stream must see the flattened bucket (and the backward kernels that produced it) complete
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            flat.record_stream(self._comm_stream)
            with torch.cuda.stream(self._comm_stream):
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        self._launched[b] = True
        self._work.append((bucket, flat))

    def start(self):
        if not is_distributed() or not self.params:
            return
        if not self._armed:
            self._launched = [False] * len(self._buckets_cache)
            self._work = []
        self._armed = False
        for b in range(len(self._buckets_cache)):      # same order on every rank
            if not self._launched[b]:
                self._launch(b)
        self._pending = (self._work, dist.get_world_size())
        self._work = []

    def finish(self):
        if self._pending is None:
            return
        work, ws = self._pending
        self._pending = None
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        inv = 1.0 / ws
        for bucket, flat in work:
            flat.mul_(inv)                      # one launch for the whole bucket
            views, dsts, off = [], [], 0
            for p in bucket:
                n = p.numel()
                v = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = v.clone()
                else:
                    views.append(v)
                    dsts.append(p.grad)
                off += n
            if dsts:
                torch._foreach_copy_(dsts, views)   # one multi-tensor launch instead of one copy per parameter

    def reduce(self):
        self.start()
        self.finish()

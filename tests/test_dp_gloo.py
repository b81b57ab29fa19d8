"""CPU, world_size=2 (gloo): the data-parallel plumbing of srgan_amd.dp.

* GradReducer averages gradients across ranks (several buckets, a parameter without grad on one rank);
* all_gather_rows is differentiable and hands every rank its own rows of the gradient;
* the convention used by SRGAN_training for the GLOBAL-batch latent losses (evaluate on the gathered mu on every
  rank, scale by world size, then average-all-reduce) reproduces the single-process gradient exactly, together with
  per-sample-mean losses -- checked with the oracle's loss functions on a toy encoder.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _toy_losses(theta, bias, x, n_batch, mu_all_fn, ws_scale):
    from oracle import losses as ol
    mu = x @ theta + bias                                 # "encoder": [B_local, 8]
    per_sample = (mu ** 2).mean()                         # stands for the image-mean losses (L1 / LSGAN)
    mu_all = mu_all_fn(mu)
    tgt = ol.analytic_hist_target()
    latent = 10.0 * ol.batch_kl(mu_all, n_batch) + 100.0 * ol.corr_loss(mu_all.t()) + 100.0 * ol.HistogramImitation(target=tgt).loss(mu_all)
    return per_sample + latent * ws_scale


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from srgan_amd import dp
    r, w, device = dp.init_from_env("gloo")
    assert (r, w) == (rank, world) and device.type == "cpu" and dp.is_distributed() and dp.world_size() == 2

    # 1. GradReducer with tiny buckets and a missing grad
    dp.BUCKET_BYTES = 64
    params = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(7))]
    params[0].grad = torch.full((5,), float(rank + 1))
    params[1].grad = torch.arange(12.0).view(3, 4) * (rank + 1)
    if rank == 0:
        params[2].grad = torch.ones(7)
    red = dp.GradReducer(params)
    assert len(red._buckets()) == 3
    red.reduce()
    assert torch.allclose(params[0].grad, torch.full((5,), 1.5))
    assert torch.allclose(params[1].grad, torch.arange(12.0).view(3, 4) * 1.5)
    assert torch.allclose(params[2].grad, torch.full((7,), 0.5))

    # 1b. armed reducer: buckets are all-reduced from the post-accumulate-grad hooks DURING the backward (last layer first);
    #     a parameter that gets no gradient leaves its bucket to start()
    dp.BUCKET_BYTES = 64
    w1, w2, w3, unused = (torch.nn.Parameter(torch.full((6,), 0.5)), torch.nn.Parameter(torch.full((20,), 0.25)),
                          torch.nn.Parameter(torch.full((6,), 2.0)), torch.nn.Parameter(torch.ones(4)))
    red = dp.GradReducer([w1, w2, w3, unused])
    red.arm()
    xin = torch.arange(6.0) * (rank + 1)
    ((w1 * xin).sum() * w3.sum() + (w2 ** 2).sum() * (rank + 1)).backward()
    launched = [bk.launched for bk in red._buckets_cache]
    assert len(red._work) >= 2 and not all(launched)            # w1/w2/w3 buckets went out from the hooks, `unused` did not
    local = [p.grad.clone() if p.grad is not None else None for p in (w1, w2, w3)]
    red.start()
    red.finish()
    assert torch.allclose(w1.grad, torch.arange(6.0) * 1.5 * 12.0) and torch.allclose(w2.grad, torch.full((20,), 0.5 * 1.5))
    assert torch.allclose(w3.grad, torch.full((6,), 0.5 * 15.0 * 1.5))
    assert unused.grad is None          # no gradient on any rank in an ARMED pass: stays None, the optimiser skips it
    assert not torch.allclose(local[0], w1.grad)
    # the averaged gradients ARE slices of the persistent flat buckets (nothing is copied back) ...
    for p_ in (w1, w2, w3):
        b_, j_ = red._where[id(p_)]
        assert p_.grad.data_ptr() == red._buckets_cache[b_].views[j_].data_ptr()
    # ... and a second pass without zero_grad accumulates into those slices and reduces them again in place
    red.arm()
    ((w1 * xin).sum() * w3.sum() + (w2 ** 2).sum() * (rank + 1)).backward()
    red.start()
    red.finish()
    assert torch.allclose(w2.grad, torch.full((20,), 0.5 * 1.5 + 0.5 * 1.5))
    # a gradient that arrives for a bucket that has already been sent is an error, not a silent drop
    red.arm()
    w2.grad = None
    (w2 ** 2).sum().backward()
    try:
        (w2 ** 2).sum().backward()
        raise AssertionError("late gradient was accepted")
    except RuntimeError as e:
        assert "already all-reduced" in str(e)
    red.start()
    red.finish()

    # 2. all_gather_rows forward / backward
    x = (torch.arange(6.0).view(3, 2) + 10 * rank).requires_grad_(True)
    g = dp.all_gather_rows(x)
    assert g.shape == (6, 2) and torch.equal(g[3 * rank:3 * rank + 3], x.detach())
    (g * torch.arange(12.0).view(6, 2)).sum().backward()
    assert torch.equal(x.grad, torch.arange(12.0).view(6, 2)[3 * rank:3 * rank + 3])

    # 3. DP gradient == single-process gradient for per-sample-mean + global-batch latent losses
    gen = torch.Generator().manual_seed(0)
    X = torch.randn(8, 5, generator=gen)
    theta0, bias0 = torch.randn(5, 8, generator=gen) * 0.5, torch.randn(8, generator=gen) * 0.1
    theta, bias = theta0.clone().requires_grad_(True), bias0.clone().requires_grad_(True)
    loss = _toy_losses(theta, bias, X[4 * rank:4 * rank + 4], 8, dp.all_gather_rows, float(world))
    loss.backward()
    dp.BUCKET_BYTES = 64 << 20
    dp.GradReducer([theta, bias]).reduce()
    t1, b1 = theta0.clone().requires_grad_(True), bias0.clone().requires_grad_(True)
    _toy_losses(t1, b1, X, 8, lambda m: m, 1.0).backward()
    err = max(float((theta.grad - t1.grad).abs().max()), float((bias.grad - b1.grad).abs().max()))
    out.put((rank, err, float(t1.grad.abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, scale in res:
        assert err <= 1e-5 * scale, (rank, err, scale)


def _worker4(rank, world, port, out):
    """world_size 4: the armed (backward-overlapped) reducer on the toy encoder, 2 of the 8 samples per rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from srgan_amd import dp
    dp.init_from_env("gloo")
    dp.BUCKET_BYTES = 64
    gen = torch.Generator().manual_seed(0)
    X = torch.randn(8, 5, generator=gen)
    theta0, bias0 = torch.randn(5, 8, generator=gen) * 0.5, torch.randn(8, generator=gen) * 0.1
    theta, bias = torch.nn.Parameter(theta0.clone()), torch.nn.Parameter(bias0.clone())
    red = dp.GradReducer([theta, bias])
    red.arm()
    n = 8 // world
    _toy_losses(theta, bias, X[n * rank:n * rank + n], 8, dp.all_gather_rows, float(world)).backward()
    red.start()
    red.finish()
    t1, b1 = theta0.clone().requires_grad_(True), bias0.clone().requires_grad_(True)
    _toy_losses(t1, b1, X, 8, lambda m: m, 1.0).backward()
    err = max(float((theta.grad - t1.grad).abs().max()), float((bias.grad - b1.grad).abs().max()))
    out.put((rank, err, float(t1.grad.abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world_size_4_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, out)) for r in range(4)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in res) == [0, 1, 2, 3]
    for rank, err, scale in res:
        assert err <= 1e-5 * scale, (rank, err, scale)


def _worker8(rank, world, port, out):
    """world_size 8: bucket order and sizes incl. the uneven last bucket, the averaged values of every
    bucket, all_gather_rows in rank order -- and the bf16 message of the bf16 mode (set_bucket_dtype): the all-reduce runs on
    a bf16 buffer, p.grad is bound to the widened fp32 buffer."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from srgan_amd import dp
    dp.init_from_env("gloo")
    assert dp.world_size() == 8
    dp.BUCKET_BYTES = 256                                  # 64 floats per message
    sizes = [40, 24, 64, 10, 30, 7]                        # reversed: [7, 30, 10] | [64] | [24, 40] -> 47, 64, 64 floats
    params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
    red = dp.GradReducer(params)
    layout = [[p.numel() for p in bk.params] for bk in red._buckets_cache]
    gen = torch.Generator().manual_seed(7)
    base = [torch.randn(n, generator=gen) for n in sizes]  # the same on every rank
    for p, b in zip(params, base):
        p.grad = b * (rank + 1)
    red.reduce()
    err32 = max(float((p.grad - b * 4.5).abs().max()) for p, b in zip(params, base))      # mean of 1..8 = 4.5
    bound32 = [p.grad.data_ptr() == red._buckets_cache[red._where[id(p)][0]].views[red._where[id(p)][1]].data_ptr() for p in params]
    # rows in rank order
    x = (torch.arange(4.0).view(2, 2) + 100 * rank).requires_grad_(True)
    g = dp.all_gather_rows(x)
    rows_ok = all(torch.equal(g[2 * r:2 * r + 2], torch.arange(4.0).view(2, 2) + 100 * r) for r in range(world))
    (g * torch.arange(32.0).view(16, 2)).sum().backward()
    rows_ok = rows_ok and torch.equal(x.grad, torch.arange(32.0).view(16, 2)[2 * rank:2 * rank + 2])
    # bf16 messages
    dp.set_bucket_dtype("bf16")
    try:
        for p, b in zip(params, base):
            p.grad = b * (rank + 1)
        red.reduce()
        wire = [bk.wire.dtype for bk in red._buckets_cache]
        grads_fp32 = all(p.grad.dtype == torch.float32 for p in params)
        # every rank's contribution is rounded to bf16 once and the sum is kept in bf16 by the transport: 8 roundings of <= 2^-9
        err16 = max(float(((p.grad - b * 4.5).abs() / (b.abs() * 4.5 + 1e-6)).max()) for p, b in zip(params, base))
        same16 = [p.grad.clone() for p in params]
    finally:
        dp.set_bucket_dtype(None)
    # back to fp32 messages: the bf16 buffers are dropped and the exact average returns
    for p, b in zip(params, base):
        p.grad = b * (rank + 1)
    red.reduce()
    err_back = max(float((p.grad - b * 4.5).abs().max()) for p, b in zip(params, base))
    no_wire = all(bk.wire is None for bk in red._buckets_cache)
    # all ranks hold identical averaged gradients in the bf16 mode too (what keeps the replicas identical)
    flat16 = torch.cat([t.flatten() for t in same16])
    gathered = [torch.zeros_like(flat16) for _ in range(world)]
    dist.all_gather(gathered, flat16)
    identical = all(torch.equal(gathered[0], t) for t in gathered)
    out.put((rank, layout, err32, all(bound32), rows_ok, [str(w) for w in wire], grads_fp32, err16, err_back, no_wire, identical))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world_size_8_gloo_buckets_gather_and_bf16_messages():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, out)) for r in range(8)]
    for p in procs:
        p.start()
    res = [out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(8))
    for rank, layout, err32, bound32, rows_ok, wire, grads_fp32, err16, err_back, no_wire, identical in res:
        assert layout == [[7, 30, 10], [64], [24, 40]], layout          # reverse parameter order, 256-byte messages, uneven first
        assert err32 <= 1e-5 and bound32 and rows_ok, (rank, err32, bound32, rows_ok)
        assert wire == ["torch.bfloat16"] * 3 and grads_fp32
        assert 0 < err16 <= 3e-2, (rank, err16)                         # bf16 on the wire: visible, and bounded
        assert err_back <= 1e-5 and no_wire and identical, (rank, err_back, no_wire, identical)


def test_single_process_is_a_no_op():
    from srgan_amd import dp
    assert not dp.is_distributed() and dp.world_size() == 1 and dp.rank() == 0
    x = torch.ones(2, 3, requires_grad=True)
    assert dp.all_gather_rows(x) is x
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    dp.GradReducer([p]).reduce()
    assert torch.equal(p.grad, torch.ones(3))


def test_bench_launcher_spawns_ranks_before_any_gpu_call():
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): the parent starts the ranks under torch.distributed.run,
    rank 0's JSON line comes back on stdout, exit status 0.  --launch-check stops every rank before GPU work."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["launch_check"] and rec["n_gpus"] == 2 and [x[0] for x in rec["ranks"]] == [0, 1]
    assert [x[1] for x in rec["ranks"]] == [0, 1]          # LOCAL_RANK -> device index
    # a mismatch between --gpus and the environment is refused
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env2,
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0


def _worker_vote(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import time
    from srgan_amd import dp
    from srgan_amd.trainer import _StepGraph
    dp.init_from_env("gloo")
    g = object.__new__(_StepGraph)           # the agreement logic only: no networks, no device
    g.graph, g.key = None, None

    def local_mode(source_image, label):
        if rank == 1:
            raise NotImplementedError("SRGAN_training graph mode: optG is not srgan_amd.optim.Adam")
        return 0
    g._local_mode = local_mode
    t0 = time.time()
    try:
        g.accepts(torch.zeros(1), {})
        res = "returned"
    except NotImplementedError:
        res = "NotImplementedError"
    except RuntimeError as e:
        res = "RuntimeError" if "another rank cannot take this step" in str(e) else "other: " + str(e)
    out.put((rank, res, time.time() - t0))


def test_graph_step_agreement_with_one_failing_rank_raises_on_every_rank():
    """a rank whose ``_local_mode`` raises votes -1 in the collective agreement of ``_StepGraph.accepts``; the failing
    rank re-raises its own error and every OTHER rank raises too, at once -- none of them walks into the eager step's first
    gradient all-reduce to wait for a peer that has left."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_vote, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r, (what, dt)) for r, what, dt in (out.get(timeout=120) for _ in procs))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1][0] == "NotImplementedError" and res[0][0] == "RuntimeError", res
    assert max(dt for _, dt in res.values()) < 30.0, res

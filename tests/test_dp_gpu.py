"""GPU: data-parallel equivalence of the HIP train step.  Two ranks (each half of the batch, sharing the one
GPU of the test box, gloo as the transport) must reproduce the single-process step on the full batch:
same parameters afterwards, rank-averaged losses equal to the full-batch losses.  Exercises the mu all-gather,
the world-size scaling of the global-batch latent losses and the bucketed gradient all-reduce on device tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

pytestmark = pytest.mark.gpu
K, B, STEPS = 2, 4, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _noise_source(rank, world):
    gen = torch.Generator().manual_seed(77)

    def fn(batch, ndim):          # every rank draws the GLOBAL noise and keeps its rows
        full = torch.randn(batch * world, ndim, generator=gen)
        return full[rank * batch:(rank + 1) * batch].clone()
    return fn


def _run(rank, world, out_q=None, kl=0.0, graph=False, recipe=None):
    from oracle import trainer as otrainer
    from tests.common import build_hip_nets
    from srgan_amd.trainer import SRGAN_training
    from srgan_amd import optim as hoptim, ops, dp
    G, D, E = build_hip_nets("T")
    if recipe in ("config3", "config3g"):   # BASELINE configs[3]: 4 domains, global batch 64, bf16 convolutions, 4 ranks
        ops.set_compute_dtype("bf16")       # ("config3g": the same at global batch 16, three steps -- the recorded step's test)
    if recipe == "config2":          # BASELINE configs[2]: pretrained-E recipe (trunk frozen for optE) + bf16 convolutions + DP
        ops.set_compute_dtype("bf16")
        keys = [k_ for k_ in E.state_dict().keys() if not k_.startswith(("fcmean", "fcvar"))]
        E.freeze_melt(keys, "freeze")
    torch.manual_seed(0)
    # Adam with a large eps is nearly linear in the gradient: without it the first steps are lr*sign(g) and
    # rounding-level differences between the two reduction orders are amplified to O(lr) parameter differences,
    # which would hide (or fake) a real data-parallel discrepancy.
    opts = [hoptim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999), eps=1e-2) for net in (G, D, E)]
    if recipe == "config2":
        E.freeze_melt(keys, "melt")
    gb = 64 if recipe == "config3" else (16 if recipe == "config3g" else B)
    sg = SRGAN_training([G, D, E], opts, [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD, KL=kl), K, "cuda",
                        np.eye(4), gb, "mu", 8)
    sg.opt_sche_initialization()
    if graph:
        sg.enable_graph()
    sg.noise_fn = _noise_source(rank, world)
    per = gb // world
    losses = []
    for s in range(STEPS if recipe != "config3" else 2):
        x, label = otrainer.synthetic_batch(gb, 128, 4, seed=300 + s)
        sl = slice(rank * per, (rank + 1) * per)
        lab = {"source": label["source"][sl].cuda(), "target": label["target"][sl]}
        if graph and sg.graph_active and dp.is_distributed():
            sg._graph.graph.trace = []          # host-side order of this replay (checked below)
        losses.append([float(v) for v in sg.train(x[sl].cuda(), lab)])
    from srgan_amd import trainer as htrainer
    if graph and htrainer._StepGraph.fault_hook is not None:
        assert sg._graph is None and not sg.graph_active          # every rank gave the recording up together
    elif graph:
        # data parallel: the recording is cut where a collective starts and where its result is needed -- K x (start, wait) for
        # the discriminator, the mu all-gather, start(E) inside phase 1's backward (round 4), start(G) with wait(E), wait(G),
        # start(G) with its wait: 2K + 6 graph segments
        # round 6: with the C-ABI collectives (SRGAN_DP_COMM=abi) they are captured and the step is ONE graph; the kinds of the
        # captured collectives, in enqueue order, are the cuts of the segmented form
        single = dp.is_distributed() and getattr(sg._graph.graph, "single", False)
        assert single == (dp.is_distributed() and dp.transport() == "abi" and os.environ.get("SRGAN_DP_SINGLE_GRAPH") != "0")
        assert sg.graph_active and len(sg._graph.graph.segments) == (2 * K + 6 if dp.is_distributed() and not single else 1)
        if single:
            assert sg._graph.graph.inline == ["all_reduce", "wait"] * K + ["all_gather", "all_reduce", "all_reduce", "wait", "all_reduce"]
        elif dp.is_distributed():
            _check_overlap_order(sg._graph.graph.trace)
    state = {f"{n}.{k}": v.detach().cpu().numpy().copy() for n, net in (("G", sg.G), ("D", sg.D), ("E", sg.E)) for k, v in net.state_dict().items()}
    terms = {k: float(v) for k, v in sg.loss_terms.items()}
    if out_q is not None:
        out_q.put((rank, losses, state if rank == 0 else None, terms))
    return losses, state, terms


def _check_overlap_order(tr):
    """north_star: "all-reduce ... overlapped ... on a side HIP stream".  The host-side trace of the last replay: every
    discriminator all-reduce i is ENQUEUED (on the communication stream, behind a ``ready`` event) before the segment that holds
    translation i + 1 is launched, and the compute stream is made to wait for its ``done`` event only after that segment; the
    last one is awaited after phase 1's forward segment; G's phase-1 all-reduce is started before and awaited after the segment
    with E's optimiser step.  (Whether the two streams really run side by side is the device's business -- and with gloo the
    all-reduce blocks the host -- so the ORDER of the enqueues is what is asserted.)"""
    kinds = [k for k, _ in tr if k != "segment"]
    assert kinds == ["all_reduce", "wait"] * K + ["all_gather", "all_reduce", "all_reduce", "wait", "all_reduce"], kinds
    pos = {(k, i): n for n, (k, i) in enumerate(tr)}
    starts = [i for k, i in tr if k == "all_reduce"]
    waits = [i for k, i in tr if k == "wait"]
    for d in range(K):                       # discriminator update d: start after segment s, wait after segment s + 1
        s, w = starts[d], waits[d]
        assert w == s + 1 and pos[("all_reduce", s)] < pos[("segment", s + 1)] < pos[("wait", w)], (d, s, w)
    # phase 1 (round 4): start(E) INSIDE the backward -- the segment launched right after it holds the stale-graph generator
    # backward, which therefore runs under E's all-reduce; start(G) (E awaited in the same callable) after that segment,
    # wait(G) one more segment (E's step + E(source)) later
    se, sgen, wg = starts[K], starts[K + 1], waits[K]
    assert sgen == se + 1 and wg == sgen + 1
    assert pos[("all_reduce", se)] < pos[("segment", se + 1)] < pos[("all_reduce", sgen)] < pos[("segment", sgen + 1)] < pos[("wait", wg)]


def _worker(rank, world, port, out_q, kl=0.0, backend="gloo", force=False, graph=False, recipe=None, inject=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SRGAN_DP_DEVICE="0", SRGAN_DP_BACKEND=backend)
    if force:
        os.environ["SRGAN_DP_FORCE"] = "1"
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "style-restricted_gan_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from srgan_amd import dp
    dp.init_from_env()
    assert dp.world_size() == world and dp.is_distributed()
    if inject:           # "<rank>:<stage>": that rank's recording fails before / inside the capture
        from srgan_amd import trainer as htrainer
        bad_rank, bad_stage = inject.split(":")

        def hook(stage):
            if rank == int(bad_rank) and stage == bad_stage:
                raise RuntimeError(f"injected failure {stage} the recording")
        htrainer._StepGraph.fault_hook = staticmethod(hook)
    _run(rank, world, out_q, kl, graph, recipe)
    dist.barrier()
    dist.destroy_process_group()


def _spawn(world, make_args, timeout=240):
    """Start ``world`` rank processes, collect one result per rank and ALWAYS reap them: a rank that dies leaves the others blocked
    in a collective holding the GPU until the process-group timeout, which would wedge the rest of the GPU session."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=make_args(r, port, q)) for r in range(world)]
    try:
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=timeout) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
                if p.is_alive():
                    p.kill()
                    p.join(timeout=10)
    assert [p.exitcode for p in procs] == [0] * world, [p.exitcode for p in procs]
    return res


@pytest.mark.parametrize("kl,graph", [(0.0, False), (0.1, False), (0.0, True)])
def test_two_ranks_equal_one_process(kl, graph):
    """kl=0.1: the conventional-KL term (a SUM over rows, 01/02/03/05 notebooks' 0.1 option) must be pre-scaled by the world
    size like the batch-statistics losses, or E receives 1/ws of its gradient.
    graph=True: both ranks record the step as hipGraph segments and replay them with the collectives issued eagerly between
    the segments (step 0 eager, step 1 records + replays, step 2 replays); same bounds as the eager ranks."""
    ref_losses, ref_state, ref_terms = _run(0, 1, kl=kl)
    res = _spawn(2, lambda r, port, q: (r, 2, port, q, kl, "gloo", False, graph), timeout=240)
    dp_losses = (np.array(res[0][1]) + np.array(res[1][1])) / 2
    ref = np.array(ref_losses)
    # errG / errD are per-sample means -> rank average == full batch; errE's latent part is global on every rank
    if kl == 0.0:
        np.testing.assert_allclose(dp_losses, ref, rtol=1e-3)
    else:      # errE reports the LOCAL KL sum: the rank sum (not the average) of that part equals the full-batch value
        np.testing.assert_allclose(dp_losses[:, :2], ref[:, :2], rtol=1e-3)
        kl_sum = res[0][3]["errE_KL"] + res[1][3]["errE_KL"]
        assert abs(kl_sum - ref_terms["errE_KL"]) <= 1e-3 * abs(ref_terms["errE_KL"]), (kl_sum, ref_terms["errE_KL"])
    for key in ("errE_bKL", "errE_corr", "errE_hist"):
        for r in res:
            assert abs(r[3][key] - ref_terms[key]) <= 1e-3 * max(abs(ref_terms[key]), 1e-3), (key, r[3][key], ref_terms[key])
    for key, v in res[0][2].items():
        d = float(np.abs(v - ref_state[key]).max())
        assert d <= 1e-5, (key, d)        # 4 optimiser steps of at most lr=1e-4 each; observed ~2e-6


@pytest.mark.parametrize("inject", ["1:before", "1:inside"])
def test_failed_recording_on_one_rank_makes_every_rank_fall_back(inject):
    """Rank 1 cannot record the step (fault injected before / inside the recording): the ranks agree on it with one MIN
    all-reduce, BOTH drop graph mode, run that very step eagerly from the inputs staged for the recording and continue eagerly --
    same results as the eager two-rank run (bounds of the test above)."""
    ref_losses, ref_state, ref_terms = _run(0, 1)
    res = _spawn(2, lambda r, port, q: (r, 2, port, q, 0.0, "gloo", False, True, None, inject), timeout=240)
    dp_losses = (np.array(res[0][1]) + np.array(res[1][1])) / 2
    np.testing.assert_allclose(dp_losses, np.array(ref_losses), rtol=1e-3)
    for key, v in res[0][2].items():
        d = float(np.abs(v - ref_state[key]).max())
        assert d <= 1e-5, (key, d)


@pytest.mark.parametrize("transport", ["torch", "abi"])
@pytest.mark.parametrize("graph", [False, True])
def test_rccl_path_one_rank_equals_plain_step(graph, transport):
    """The real RCCL calls on the test box's one GPU: a one-rank ``nccl`` process group with SRGAN_DP_FORCE=1 takes the whole
    data-parallel path -- hook-driven buckets with the G and E reducers armed together, in-place all-reduce of the flat buffers
    on the communication stream (asynchronous: only the ready / done events order it against the compute stream), the mu
    all-gather inside autograd, gradients bound to bucket slices, parameters without gradient left at None -- and must
    reproduce the plain single-process step.  (Two ranks cannot share a device under RCCL; the 2-rank arithmetic is the gloo
    test above.)  graph=True: the same with the step recorded as hipGraph segments, the RCCL all-reduces and the all-gather
    issued eagerly between the segments on every replay.  transport="abi" (SRGAN_DP_COMM=abi): the same exchanges through the
    library's own C-ABI entry points over RCCL (srgan_comm_init with the id from rank 0, srgan_allreduce_bucket,
    srgan_allgather_rows) instead of torch.distributed's."""
    ref_losses, ref_state, ref_terms = _run(0, 1)
    os.environ["SRGAN_DP_COMM"] = transport
    try:
        (rank, losses, state, terms), = _spawn(1, lambda r, port, q: (0, 1, port, q, 0.0, "nccl", True, graph))
    finally:
        os.environ.pop("SRGAN_DP_COMM", None)
    np.testing.assert_allclose(np.array(losses), np.array(ref_losses), rtol=1e-4)
    for key, v in state.items():
        d = float(np.abs(v - ref_state[key]).max())
        assert d <= 1e-6, (key, d)


def test_single_graph_step_equals_segmented_step_bit_for_bit():
    """Round 6: with the C-ABI collectives the recorded data-parallel step is ONE hipGraph -- the bucket
    all-reduces and the mu all-gather are captured on the communication stream (forked / joined by the buckets' events) -- and a
    replay is a single launch with no host-side vote.  One rank over RCCL on the test box's GPU: the parameters after the steps
    are BIT-identical to the segmented form (SRGAN_DP_SINGLE_GRAPH=0: the same kernels and collectives, launched as 2k + 6 graphs
    with host callables between them), and both reproduce the plain single-process step."""
    ref_losses, ref_state, _ = _run(0, 1)
    out = {}
    for single in ("1", "0"):
        os.environ["SRGAN_DP_COMM"] = "abi"
        os.environ["SRGAN_DP_SINGLE_GRAPH"] = single
        try:
            (rank, losses, state, terms), = _spawn(1, lambda r, port, q: (0, 1, port, q, 0.0, "nccl", True, True))
        finally:
            os.environ.pop("SRGAN_DP_COMM", None)
            os.environ.pop("SRGAN_DP_SINGLE_GRAPH", None)
        out[single] = (losses, state)
    assert out["1"][0] == out["0"][0], (out["1"][0], out["0"][0])
    for key, v in out["1"][1].items():
        assert np.array_equal(v, out["0"][1][key]), key
        assert float(np.abs(v - ref_state[key]).max()) <= 1e-6, key
    np.testing.assert_allclose(np.array(out["1"][0]), np.array(ref_losses), rtol=1e-4)


@pytest.mark.parametrize("graph,messages", [(False, "fp32"), (True, "fp32"), (True, "bf16")])
def test_config2_recipe_two_ranks_equal_one_process(graph, messages):
    """BASELINE configs[2] in its stated form at tier-T widths: the pretrained-E recipe (encoder trunk outside optE) AND the bf16
    convolution mode AND two data-parallel ranks, against the single-process step of the same recipe.  graph=True: the ranks
    record and replay graph segments (parameters that receive no gradient must stay without one in the recorded step too).
    messages: what travels in the gradient all-reduces -- "fp32" (SRGAN_DP_BUCKET_DTYPE=fp32) keeps the arithmetic of the
    single-process step to 2e-5; "bf16" is the mode's default (dp.set_bucket_dtype): every rank's gradient is rounded to 8
    mantissa bits before the sum, i.e. an ABSOLUTE error of 2^-9 of the per-rank gradient on the average -- where the ranks'
    contributions cancel that exceeds the average itself, and the (eps = 1e-2, nearly linear) Adam step of such an element
    differs by a fraction of lr.  Bounded per tensor: nothing beyond one lr per step, the bulk far below."""
    from srgan_amd import ops
    try:
        ref_losses, ref_state, ref_terms = _run(0, 1, recipe="config2")
    finally:
        ops.set_compute_dtype("fp32")
    os.environ["SRGAN_DP_BUCKET_DTYPE"] = messages
    try:
        res = _spawn(2, lambda r, port, q: (r, 2, port, q, 0.0, "gloo", False, graph, "config2"), timeout=240)
    finally:
        os.environ.pop("SRGAN_DP_BUCKET_DTYPE", None)
    dp_losses = (np.array(res[0][1]) + np.array(res[1][1])) / 2
    np.testing.assert_allclose(dp_losses, np.array(ref_losses), rtol=1e-3)
    for key, v in res[0][2].items():
        d = np.abs(v - ref_state[key])
        if messages == "fp32":
            assert float(d.max()) <= 2e-5, (key, float(d.max()))
        else:
            assert float(d.max()) <= STEPS * 1.1e-4 and float(np.median(d)) <= 1e-5, (key, float(d.max()), float(np.median(d)))
    trunk = [k for k in ref_state if k.startswith("E.layers")]
    assert trunk and all(np.array_equal(res[0][2][k], ref_state[k]) for k in trunk)      # the frozen-for-optE trunk did not move


@pytest.mark.parametrize("graph,messages,recipe", [(False, "fp32", "config3"), (True, "bf16", "config3g")])
def test_config3_recipe_four_ranks_equal_one_process(graph, messages, recipe):
    """BASELINE configs[3] in its stated form at tier-T widths: 4 domains, global batch 64, bf16 convolution mode, FOUR
    data-parallel ranks (16 images each, sharing the test box's one MI355X over gloo) against the single-process step.
    (False, "fp32"): the ARITHMETIC of four ranks, eager.  (True, "bf16"), VERDICT r4 weak 1c: the combination the
    configuration actually runs -- four ranks recording and replaying graph segments, the weight-gradient kernels writing into
    the bucket slices (dp.grad_slot), bf16 gradient messages (at global batch 16 and three steps, so that a replay happens
    and the test stays short); bounded per tensor like the two-rank bf16-message case."""
    from srgan_amd import ops
    try:
        ref_losses, ref_state, ref_terms = _run(0, 1, recipe=recipe)
    finally:
        ops.set_compute_dtype("fp32")
    os.environ["SRGAN_DP_BUCKET_DTYPE"] = messages
    try:
        res = _spawn(4, lambda r, port, q: (r, 4, port, q, 0.0, "gloo", False, graph, recipe), timeout=300)
    finally:
        os.environ.pop("SRGAN_DP_BUCKET_DTYPE", None)
    if messages == "bf16":
        dp_losses = sum(np.array(r[1]) for r in res) / 4
        np.testing.assert_allclose(dp_losses, np.array(ref_losses), rtol=2e-3)
        for key, v in res[0][2].items():
            d = np.abs(v - ref_state[key])
            assert float(d.max()) <= STEPS * 1.1e-4 and float(np.median(d)) <= 1e-5, (key, float(d.max()), float(np.median(d)))
        return
    dp_losses = sum(np.array(r[1]) for r in res) / 4
    np.testing.assert_allclose(dp_losses, np.array(ref_losses), rtol=1e-3)
    for key in ("errE_bKL", "errE_corr", "errE_hist"):          # global-batch statistics: identical on every rank
        for r in res:
            assert abs(r[3][key] - ref_terms[key]) <= 1e-3 * max(abs(ref_terms[key]), 1e-3), (key, r[3][key], ref_terms[key])
    # bf16 mode rounds x / dy to 8 mantissa bits before every product: a last-bit fp32 difference between the 16-per-rank and
    # the 64-in-one-process summation orders can move a rounded operand by 2^-8, so individual weights differ by a fraction of
    # one (linearised) Adam step lr = 1e-4 (observed: isolated elements up to 6.5e-5, median 1.5e-8) -- an order of magnitude above the fp32 tests' 1e-5, still far
    # below a mis-scaled gradient (ws x or 1/ws x: >= 0.75 lr on every element).  The 1536-element first-layer filter of the
    # discriminator is the noisiest tensor (its gradient sums 16 / 64 images x 4096 positions of bf16-rounded products in a
    # different order per world size): median 5.1e-6 = 0.05 lr after round 3's slab-sum kernels changed the order once more
    worst = 0.0
    for key, v in res[0][2].items():
        d = np.abs(v - ref_state[key])
        worst = max(worst, float(d.max()))
        q999 = float(np.quantile(d, 0.999)) if d.size >= 1000 else float(d.max())
        # (q999: 4.02e-5 seen on the 2352-element first encoder filter, whose "99.9th percentile" is its second-largest element)
        assert float(d.max()) <= 1.5e-4 and q999 <= 6e-5 and float(np.median(d)) <= 1e-5, (key, float(d.max()), q999, float(np.median(d)))


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """``python bench.py --gpus 2`` end to end -- the launcher, two full-width ranks (sharing the test box's one MI355X:
    SRGAN_DP_DEVICE=0, gloo as the transport), the recorded data-parallel step with its collectives started on the communication
    stream between graph segments, max-over-ranks timing, rank 0's JSON line with n_gpus = 2 and weak scaling."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SRGAN_DP_DEVICE="0", SRGAN_DP_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2",
                        "--batch-per-gpu", "8"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["global_batch"] == 16
    assert rec["config"]["parallelism"] == "dp2" and not rec["config"]["graph_fallback"]
    assert "hipGraph segments" in rec["config"]["execution"] and rec["value"] > 0
    assert all(np.isfinite(rec["config"]["losses_last_step"]))


def test_abi_collectives_one_rank_eager_and_captured():
    """The C-ABI collectives (csrc/comm.cpp) without torch.distributed: a one-rank communicator from srgan_comm_unique_id /
    srgan_comm_init, then an in-place fp32 and bf16 bucket all-reduce (average over one rank = identity) and the row gather,
    eagerly on a side stream and CAPTURED into a hipGraph on that stream (fork / join by events inside the capture, the form a
    single-graph data-parallel step takes); replays leave the data intact and RCCL reports no error.  Multi-rank execution of
    the same calls needs more than one device: unmeasured on this pool."""
    import ctypes
    from srgan_amd import _lib
    lib = _lib.load()
    assert lib.srgan_comm_available() == 1
    ident = ctypes.create_string_buffer(128)
    _lib.check(lib.srgan_comm_unique_id(ident), "id")
    comm = ctypes.c_void_p()
    torch.cuda.set_device(0)
    _lib.check(lib.srgan_comm_init(ident, 1, 0, ctypes.byref(comm)), "init")
    n = ctypes.c_int(0)
    _lib.check(lib.srgan_comm_size(comm, ctypes.byref(n)), "size")
    assert n.value == 1
    try:
        g = torch.Generator(device="cuda").manual_seed(3)
        buf = torch.randn(1 << 20, device="cuda", generator=g)
        buf16 = torch.randn(1 << 18, device="cuda", generator=g).to(torch.bfloat16)
        rows = torch.randn(32, 8, device="cuda", generator=g)
        want, want16 = buf.clone(), buf16.clone()
        side = torch.cuda.Stream()

        def exchange():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                st = side.cuda_stream
                _lib.check(lib.srgan_allreduce_bucket(comm, buf.data_ptr(), buf.numel(), 0, 1, st), "allreduce fp32")
                _lib.check(lib.srgan_allreduce_bucket(comm, buf16.data_ptr(), buf16.numel(), 1, 1, st), "allreduce bf16")
                out = torch.empty_like(rows)
                _lib.check(lib.srgan_allgather_rows(comm, rows.data_ptr(), out.data_ptr(), rows.numel(), st), "allgather")
            torch.cuda.current_stream().wait_stream(side)
            return out

        out = exchange()
        torch.cuda.synchronize()
        assert torch.equal(buf, want) and torch.equal(buf16, want16) and torch.equal(out, rows)
        graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(graph, stream=cap):
                buf.mul_(2.0)                      # compute before the exchange, on the capture stream
                out_g = exchange()                 # forked onto the side stream inside the capture, joined again
                buf.add_(1.0)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        exp = want
        for _ in range(3):
            exp = exp * 2.0 + 1.0
        assert torch.allclose(buf, exp) and torch.equal(buf16, want16) and torch.equal(out_g, rows)
    finally:
        _lib.check(lib.srgan_comm_destroy(comm), "destroy")

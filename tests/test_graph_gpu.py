"""GPU: the whole train step replayed as ONE captured hipGraph (SRGAN_training.enable_graph; BASELINE configs[4], the step of
util_notebook.py:696-734) must be BIT-identical to eager execution: same kernels, same order, same inputs -- only the launch
mechanism differs.  Covers the device-side Adam step counter, the staged CPU-generator noise (reference draw order), label
staging, a scheduler lr change and a load_state_dict between replays, and the eager fall-back for another batch shape."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import trainer as otrainer
from tests.common import build_hip_nets

pytestmark = pytest.mark.gpu


def _trainer(tier, batch, k, seed):
    from srgan_amd.trainer import SRGAN_training
    G, D, E = build_hip_nets(tier)
    torch.manual_seed(seed)
    sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), k,
                        "cuda", np.eye(4), batch, "mu", 8)
    sg.opt_sche_initialization()
    return sg


def _steps(sg, batch, size, n, first_seed, hook=None):
    torch.manual_seed(first_seed)      # the step's noise comes from the global CPU generator: same stream for both trainers
    out = []
    for s in range(n):
        if hook is not None:
            hook(sg, s)
        x, label = otrainer.synthetic_batch(batch, size, 4, seed=first_seed + s)
        res = sg.train(x.cuda(), {"source": label["source"].cuda(), "target": label["target"]})
        out.append([float(v) for v in res] + [float(sg.loss_terms[k]) for k in sorted(sg.loss_terms)])
    return np.array(out)


def _state(sg):
    return {f"{n}.{k}": v.detach().clone() for n, net in (("G", sg.G), ("D", sg.D), ("E", sg.E)) for k, v in net.state_dict().items()}


@pytest.mark.parametrize("tier,batch,k,steps,size,dtype", [("T", 4, 3, 5, 128, "fp32"), ("F", 2, 2, 4, 128, "fp32"),
                                                          ("T256", 2, 2, 4, 256, "bf16")])     # configs[4]: 256x256 + bf16 + hipGraph
def test_graph_replay_is_bit_identical_to_eager(tier, batch, k, steps, size, dtype):
    from srgan_amd import ops
    ops.set_compute_dtype(dtype)
    try:
        _graph_vs_eager(tier, batch, k, steps, size)
    finally:
        ops.set_compute_dtype("fp32")


def _graph_vs_eager(tier, batch, k, steps, size):
    def hook(sg, s):
        if s == 3:                 # an epoch boundary: ExponentialLR moves lr on the host; the device record must follow
            sg.scheG.step(), sg.scheD.step(), sg.scheE.step()

    eager = _trainer(tier, batch, k, seed=2)
    ref = _steps(eager, batch, size, steps, 500, hook)
    ref_state = _state(eager)
    ref_steps = {n: [st["step"] for st in opt.state.values()] for n, opt in (("G", eager.optG), ("D", eager.optD), ("E", eager.optE))}

    sg = _trainer(tier, batch, k, seed=2).enable_graph()
    got = _steps(sg, batch, size, steps, 500, hook)
    assert sg.graph_active
    np.testing.assert_array_equal(got, ref)                      # losses and every loss term, all steps
    for key, v in _state(sg).items():
        assert torch.equal(v, ref_state[key]), key               # parameters after the steps: bit-identical
    for n, opt in (("G", sg.optG), ("D", sg.optD), ("E", sg.optE)):
        assert [st["step"] for st in opt.state.values()] == ref_steps[n]     # host-side counters follow the replays
    # the tensors the notebooks read after a step are live views of the graph's memory
    assert sg.target_image.shape == (batch, 3, size, size) and sg.recon_image.shape == (batch, 3, size, size)
    assert torch.isfinite(sg.recon_image).all()


def test_graph_mode_survives_outside_weight_writes_and_other_shapes():
    batch, k = 4, 2
    eager = _trainer("T", batch, k, seed=4)
    sg = _trainer("T", batch, k, seed=4).enable_graph()
    a = _steps(eager, batch, 128, 3, 700)
    b = _steps(sg, batch, 128, 3, 700)
    np.testing.assert_array_equal(a, b)
    # a checkpoint restore between steps (bumps the version counters; the captured step must not multiply stale packed weights)
    for t in (eager, sg):
        sd = {k_: v * 0.5 for k_, v in t.G.state_dict().items()}
        t.G.load_state_dict(sd)
    np.testing.assert_array_equal(_steps(eager, batch, 128, 2, 710), _steps(sg, batch, 128, 2, 710))
    # the last, smaller batch of an epoch runs eagerly and the graph keeps working afterwards
    np.testing.assert_array_equal(_steps(eager, 2, 128, 1, 720), _steps(sg, 2, 128, 1, 720))
    np.testing.assert_array_equal(_steps(eager, batch, 128, 2, 730), _steps(sg, batch, 128, 2, 730))
    for key, v in _state(sg).items():
        assert torch.equal(v, _state(eager)[key]), key


def test_weights_written_between_warmup_and_capture_are_repacked():
    """load_state_dict after the eager warm-up step and BEFORE the step that records the graph: the recording must multiply the
    new weights, not the operands the eager step packed."""
    batch, k = 4, 2
    eager = _trainer("T", batch, k, seed=6)
    sg = _trainer("T", batch, k, seed=6).enable_graph()
    np.testing.assert_array_equal(_steps(eager, batch, 128, 1, 800), _steps(sg, batch, 128, 1, 800))     # warm-up (eager in both)
    for t in (eager, sg):
        t.G.load_state_dict({k_: v * 0.75 for k_, v in t.G.state_dict().items()})
        t.D.load_state_dict({k_: v * 1.25 for k_, v in t.D.state_dict().items()})
    np.testing.assert_array_equal(_steps(eager, batch, 128, 3, 810), _steps(sg, batch, 128, 3, 810))
    assert sg.graph_active


def test_enable_graph_refuses_what_it_cannot_capture():
    from srgan_amd.trainer import SRGAN_training
    G, D, E = build_hip_nets("T")
    opts = [torch.optim.Adam(n.parameters(), lr=1e-4) for n in (G, D, E)]
    sg = SRGAN_training([G, D, E], opts, [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), 1, "cuda", np.eye(4), 4, "mu", 8)
    sg.opt_sche_initialization()
    with pytest.raises(NotImplementedError, match="srgan_amd.optim.Adam"):
        sg.enable_graph()


def test_graph_replay_is_bit_identical_to_eager_at_the_headline_shape():
    """BASELINE configs[1] itself -- full width, 128x128, bs 32, k 5: the shape at which (and only at which) the F(4x4,3x3) trunk
    kernels at batch 32 / 64 / 128, the stride-2 Winograd kernels at their full grids and the merged D / E passes dispatch.
    Step 0 eager, step 1 records and replays, step 2 replays: losses, every loss term and all parameters bit-identical."""
    _graph_vs_eager("F", 32, 5, 3, 128)


def test_optimizer_state_and_frozen_set_changed_between_replays():
    """What a recording bakes in besides shapes and buffers: the moment tensors and step counts of the optimisers
    (``load_state_dict`` replaces them), the set of parameters that take gradients (``freeze_melt``), the loss weights and k.
    Each of these changed between two replays must take effect exactly as in eager execution -- the stale recording is dropped,
    one step runs eagerly, the next records again."""
    import copy
    batch, k = 4, 2
    eager = _trainer("T", batch, k, seed=8)
    sg = _trainer("T", batch, k, seed=8).enable_graph()
    np.testing.assert_array_equal(_steps(eager, batch, 128, 2, 900), _steps(sg, batch, 128, 2, 900))
    saved = [copy.deepcopy(t.optE.state_dict()) for t in (eager, sg)]
    np.testing.assert_array_equal(_steps(eager, batch, 128, 2, 910), _steps(sg, batch, 128, 2, 910))
    assert sg.graph_active
    # 1. optimiser state restored from a checkpoint: older moments and step counts
    for t, sd in zip((eager, sg), saved):
        t.optE.load_state_dict(sd)
    np.testing.assert_array_equal(_steps(eager, batch, 128, 3, 920), _steps(sg, batch, 128, 3, 920))
    assert sg.graph_active
    # 2. the encoder trunk frozen (05-train cell 22 does this around the optimiser construction)
    for t in (eager, sg):
        keys = [k_ for k_ in t.E.state_dict().keys() if not k_.startswith(("fcmean", "fcvar"))]
        t.E.freeze_melt(keys, "freeze")
    np.testing.assert_array_equal(_steps(eager, batch, 128, 3, 930), _steps(sg, batch, 128, 3, 930))
    assert sg.graph_active
    # 3. a loss weight and the number of discriminator updates edited in place
    for t in (eager, sg):
        t.lbd["cycle"] = 2.5
        t.k = 3
    np.testing.assert_array_equal(_steps(eager, batch, 128, 3, 940), _steps(sg, batch, 128, 3, 940))
    assert sg.graph_active
    for key, v in _state(sg).items():
        assert torch.equal(v, _state(eager)[key]), key
    for n, a, b in (("G", sg.optG, eager.optG), ("D", sg.optD, eager.optD), ("E", sg.optE, eager.optE)):
        assert [st["step"] for st in a.state.values()] == [st["step"] for st in b.state.values()], n


def test_failed_recording_in_a_single_process_leaves_consistent_state():
    """A recording that fails without a process group re-raises -- after restoring the host-side optimiser counters (they moved
    while nothing ran), forgetting packed operands whose fill was only recorded, and switching graph mode off, so a caller that
    catches the error continues with a correct eager trainer."""
    from srgan_amd import trainer as htrainer
    batch, k = 4, 2
    eager = _trainer("T", batch, k, seed=9)
    sg = _trainer("T", batch, k, seed=9).enable_graph()
    np.testing.assert_array_equal(_steps(eager, batch, 128, 1, 950), _steps(sg, batch, 128, 1, 950))

    def hook(stage):
        if stage == "inside":
            raise RuntimeError("injected")
    htrainer._StepGraph.fault_hook = staticmethod(hook)
    try:
        counts = [[st["step"] for st in o.state.values()] for o in (sg.optG, sg.optD, sg.optE)]
        x, label = otrainer.synthetic_batch(batch, 128, 4, seed=951)
        rng = torch.get_rng_state()
        with pytest.raises(RuntimeError, match="injected"):
            sg.train(x.cuda(), {"source": label["source"].cuda(), "target": label["target"]})
        torch.set_rng_state(rng)       # the failed step had drawn its noise
        assert sg._graph is None and not sg.graph_active
        assert counts == [[st["step"] for st in o.state.values()] for o in (sg.optG, sg.optD, sg.optE)]
    finally:
        htrainer._StepGraph.fault_hook = None
    np.testing.assert_array_equal(_steps(eager, batch, 128, 2, 951), _steps(sg, batch, 128, 2, 951))
    for key, v in _state(sg).items():
        assert torch.equal(v, _state(eager)[key]), key


def _rerecord_child(q):
    import warnings
    from srgan_amd import ops
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        batch, k = 4, 2
        eager = _trainer("T", batch, k, seed=12)
        sg = _trainer("T", batch, k, seed=12).enable_graph()
        ok = np.array_equal(_steps(eager, batch, 128, 3, 1000), _steps(sg, batch, 128, 3, 1000)) and sg.graph_active
        ops.invalidate_packed()                     # structure epoch moves: the recording's buffers are gone
        a, b = _steps(eager, batch, 128, 1, 1010), _steps(sg, batch, 128, 1, 1010)      # eager in both, right after replays
        ok = ok and np.array_equal(a, b) and not sg.graph_active
        ok = ok and np.array_equal(_steps(eager, batch, 128, 2, 1020), _steps(sg, batch, 128, 2, 1020)) and sg.graph_active
        ops.set_compute_dtype("fp32")               # a compute-mode switch (same mode: only the epoch moves)
        ok = ok and np.array_equal(_steps(eager, batch, 128, 3, 1030), _steps(sg, batch, 128, 3, 1030)) and sg.graph_active
        sg.disable_graph()                          # and eager steps after a recording was dropped by the caller
        ok = ok and np.array_equal(_steps(eager, batch, 128, 2, 1040), _steps(sg, batch, 128, 2, 1040))
        same = all(torch.equal(v, _state(eager)[key]) for key, v in _state(sg).items())
    q.put((bool(ok), bool(same), [str(w.message)[:200] for w in caught if "AccumulateGrad" in str(w.message)]))


def test_rerecord_after_structure_change_and_no_stream_mismatch():
    """The re-record path: replay, structure-epoch change (``invalidate_packed`` / ``set_compute_dtype``), one eager step, a new
    recording, replays -- bit-identical to an all-eager run, and the eager steps that follow a recording must not re-use
    AccumulateGrad nodes made on the capture stream (PyTorch's stream-mismatch warning, emitted once per process: the scenario
    runs in a fresh process so that it would be seen)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rerecord_child, args=(q,))
    p.start()
    try:
        ok, same, warned = q.get(timeout=300)
        p.join(timeout=60)
    finally:
        if p.is_alive():
            p.terminate()
            p.join(timeout=10)
    assert p.exitcode == 0
    assert ok and same
    assert not warned, warned

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

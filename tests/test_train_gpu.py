"""GPU: srgan_amd.SRGAN_training.train (HIP path) against the reference's own trajectories
(tests/golden/train_*.npz, produced by running the imported reference) and against the CPU oracle.
Bar: losses within 1e-3 relative (north_star); parameters after the steps within a few lr."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import trainer as otrainer
from tests.common import build_hip_nets, close, close_params, oracle_params

pytestmark = pytest.mark.gpu

# largest relative loss deviation of the 20-step bf16 trajectories from the fp32 oracle, x 1.5 (measured values: DESIGN.md section 2)
BF16_TRAJ_BOUND_K5 = 8e-4
BF16_TRAJ_BOUND_PRE = 4.5e-4


def run_hip(tier, batch, k, steps, seed, pretrained_e=False, size=128):
    from srgan_amd import optim as hoptim
    from srgan_amd.trainer import SRGAN_training
    G, D, E = build_hip_nets(tier)
    optE = None
    if pretrained_e:   # 05-train cell 22: freeze trunk, Adam(lr=1e-3) over fcmean/fcvar, melt again
        keys = [k_ for k_ in E.state_dict().keys() if not k_.startswith(("fcmean", "fcvar"))]
        E.freeze_melt(keys, "freeze")
        optE = hoptim.Adam(filter(lambda p: p.requires_grad, E.parameters()), lr=1e-3, betas=(0.5, 0.999))
        E.freeze_melt(keys, "melt")
    torch.manual_seed(seed)
    np.random.seed(seed)
    sg = SRGAN_training([G, D, E], [None, None, optE], [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), k,
                        "cuda", np.eye(4), batch, "mu", 8)
    sg.opt_sche_initialization()
    traj = []
    for s in range(steps):
        x, label = otrainer.synthetic_batch(batch, size, 4, seed=100 + s)
        label = {"source": label["source"].cuda(), "target": label["target"]}   # as the notebook: source on device
        errG, errD, errE = sg.train(x.cuda(), label)
        traj.append([float(errG), float(errD), float(errE)])
    return sg, np.array(traj)


@pytest.mark.parametrize("name,k,steps,pre", [("train_T_b4_k2", 2, 3, False), ("train_T_b4_k5", 5, 2, False),
                                              ("train_T_b4_k2_pretrainedE", 2, 2, True),
                                              ("train_T_b4_k2_s10", 2, 10, False)])      # 10 reference steps
def test_train_trajectory_vs_reference_tier_T(golden_dir, name, k, steps, pre):
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    sg, traj = run_hip("T", 4, k, steps, seed=0, pretrained_e=pre)
    close(sg.hi.target, gold["hist_target"], 1e-5, what="hist target")
    np.testing.assert_allclose(traj, gold["losses"], rtol=1e-3)
    lr_e = 1e-3 if pre else 1e-4
    for net_name, net, lr, n_opt in (("G", sg.G, 1e-4, 2 * steps), ("D", sg.D, 1e-4, k * steps), ("E", sg.E, lr_e, steps)):
        for key, v in net.state_dict().items():
            close_params(v, gold[f"{net_name}.{key}"], lr, n_opt, what=f"{net_name}.{key}")


def test_train_trajectory_vs_reference_256(golden_dir):
    """BASELINE configs[4] geometry: 256x256 images, discriminator with five down convs (tiny widths)."""
    gold = np.load(os.path.join(golden_dir, "train_T256_b2_k2.npz"))
    sg, traj = run_hip("T256", 2, 2, 2, seed=0, size=256)
    np.testing.assert_allclose(traj, gold["losses"], rtol=1e-3)
    for net_name, net, lr, n_opt in (("G", sg.G, 1e-4, 4), ("D", sg.D, 1e-4, 4), ("E", sg.E, 1e-4, 2)):
        for key, v in net.state_dict().items():
            close_params(v, gold[f"{net_name}.{key}"], lr, n_opt, what=f"{net_name}.{key}")


def test_train_trajectory_vs_reference_full_size(golden_dir):
    gold = np.load(os.path.join(golden_dir, "train_F_b2_k1.npz"))
    sg, traj = run_hip("F", 2, 1, 2, seed=0)
    np.testing.assert_allclose(traj, gold["losses"], rtol=1e-3)
    for net_name, net in (("G", sg.G), ("D", sg.D), ("E", sg.E)):
        for key, v in net.state_dict().items():
            ck = gold[f"{net_name}_ck.{key}"]
            v = v.detach().double().cpu()
            # Adam's first steps move an element by ~lr * sign(g): where g is within rounding of zero the sign -- and
            # with it up to 2 * lr * steps of that element -- depends on summation order (Winograd vs direct conv,
            # GPU vs CPU).  One such element is budgeted on top of the relative bound.
            flip = 2 * 1e-4 * 4      # 2 * lr * (at most 2 optimiser steps per train step x 2 train steps)
            assert abs(float(v.norm()) - ck[1]) <= 1e-4 * max(ck[1], 1e-6) + flip, key
            np.testing.assert_allclose(v.flatten()[:8].numpy(), ck[2:], atol=flip)


def test_train_step_bf16_compute_mode_full_size(golden_dir):
    """BASELINE configs [2]-[4] run the convolutions on the bf16 MFMA (fp32 accumulation, fp32 storage): the full-size train
    step must track the fp32 reference trajectory within bf16 rounding of the conv operands -- 5e-3 on every loss over two
    steps (round 4, tightened from 1e-2; measured 1.0e-3 with the 16-bit activation storage and the bf16 RGB kernels)."""
    from srgan_amd import ops
    gold = np.load(os.path.join(golden_dir, "train_F_b2_k1.npz"))
    ops.set_compute_dtype("bf16")
    try:
        _, traj = run_hip("F", 2, 1, 2, seed=0)
    finally:
        ops.set_compute_dtype("fp32")
    np.testing.assert_allclose(traj, gold["losses"], rtol=5e-3)
    assert not np.allclose(traj, gold["losses"], rtol=1e-6)      # the mode really changed the arithmetic


TERM_PAIRS = [("errG_dis", "g_dis"), ("errG_class", "g_cls"), ("errG_cycle", "g_cyc"), ("errG_idt", "g_idt"),
              ("errE_bKL", "bkl"), ("errE_corr", "corr"), ("errE_hist", "hist"), ("errG_reg", "g_reg"),
              ("errG_idt_reg", "g_idt_reg")]


def step_vs_oracle(tier, batch, k, seed, batch_seed, size=128, rtol=1e-3, norm_rtol=None):
    """One train step, same seeds on both sides: the three returned losses, every individual loss term, and the parameters
    after the step (element-wise with the Adam sign-flip budget; additionally per-tensor norms when ``norm_rtol`` is given)."""
    PG, PD, PE = oracle_params(tier)
    torch.manual_seed(seed)
    orc = otrainer.SRGANOracle(PG, PD, PE, otrainer.DEFAULT_LBD, k, np.eye(4), batch, "mu", 8)
    x, label = otrainer.synthetic_batch(batch, size, 4, seed=batch_seed)
    ref = [float(v) for v in orc.train(x, label)]
    orc.last_losses = ref

    from srgan_amd.trainer import SRGAN_training
    G, D, E = build_hip_nets(tier)
    torch.manual_seed(seed)
    sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), k,
                        "cuda", np.eye(4), batch, "mu", 8)
    sg.opt_sche_initialization()
    out = [float(v) for v in sg.train(x.cuda(), {"source": label["source"].cuda(), "target": label["target"]})]
    np.testing.assert_allclose(out, ref, rtol=rtol)
    t = {k_: float(v) for k_, v in sg.loss_terms.items()}
    tr = orc.trace
    for a, b in TERM_PAIRS:
        assert abs(t[a] - tr[b]) <= rtol * max(abs(tr[b]), 1e-3), (a, t[a], tr[b])
    for j, name in enumerate(("errD_real", "errD_class", "errD_fake")):      # the LAST discriminator iteration's parts
        want = tr["errD_parts"][-1][j]
        assert abs(t[name] - want) <= rtol * max(abs(want), 1e-3), (name, t[name], want)
    for net, P, n_opt in ((sg.G, orc.G, 2), (sg.D, orc.D, k), (sg.E, orc.E, 1)):
        for key, v in net.state_dict().items():
            close_params(v, P[key], 1e-4, n_opt, what=key)
            if norm_rtol is not None:
                a, b = float(v.detach().double().norm()), float(P[key].detach().double().norm())
                assert abs(a - b) <= norm_rtol * max(b, 1e-6) + 2 * 1e-4 * n_opt, (key, a, b)
    return sg, orc


def test_train_step_vs_oracle_with_loss_terms():
    """Same seeds on both sides; compares every individual loss term of the step."""
    step_vs_oracle("T", 6, 3, seed=5, batch_seed=42)


def test_headline_shape_step_vs_oracle():
    """THE benchmarked workload (BASELINE configs[1]): full-width networks, 128x128, **bs=32, k=5**, one whole train step on
    the HIP path against the CPU oracle on the same seeds.  At this shape -- and only at this shape -- the production dispatch
    runs: F(4x4,3x3) forward / input gradient / weight gradient of the trunk at batch 32, 64 (reconstruction + identity pair)
    and 128 (the k-1 no-grad translations), the stride-2 Winograd kernels at their full grids, the merged D / E passes.
    Losses and every loss term to 1e-3 (north_star), parameters after the step element-wise and by norm."""
    sg, _ = step_vs_oracle("F", 32, 5, seed=3, batch_seed=77, norm_rtol=1e-4)
    assert sg.source_image.shape[0] == 32


def test_headline_shape_step_bf16_vs_oracle():
    """The same workload in the bf16 mode (BASELINE configs [2]-[4] arithmetic at configs[1]'s shape): at bs=32, k=5 and full
    width the residual trunk runs as ops._ResBlockBf16Fn nodes -- LDS-resident-patch bf16 kernels at batch 32 / 64 / 128 with the
    block's intermediates stored as bf16 -- and the step must stay within bf16 rounding of the fp32 CPU oracle on the same seeds:
    1e-2 on the three losses and on every loss term (measured: 4.2e-3)."""
    import os
    from srgan_amd import ops
    from srgan_amd.trainer import SRGAN_training
    PG, PD, PE = oracle_params("F")
    torch.manual_seed(3)
    orc = otrainer.SRGANOracle(PG, PD, PE, otrainer.DEFAULT_LBD, 5, np.eye(4), 32, "mu", 8)
    x, label = otrainer.synthetic_batch(32, 128, 4, seed=77)
    ref = [float(v) for v in orc.train(x, label)]
    ops.set_compute_dtype("bf16")
    try:
        G, D, E = build_hip_nets("F")
        torch.manual_seed(3)
        sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), 5,
                            "cuda", np.eye(4), 32, "mu", 8)
        sg.opt_sche_initialization()
        xb = torch.randn(32, 256, 32, 32, device="cuda")
        with ops.pack_cache():
            w = next(p for n_, p in G.named_parameters() if p.dim() == 4 and tuple(p.shape) == (256, 256, 3, 3))
            sc = torch.ones(32, 256, device="cuda")
            assert ops.res_block_bf16_fusable(xb, w, w, sc, sc)       # the trunk of this step takes the bf16-storage node
        out = [float(v) for v in sg.train(x.cuda(), {"source": label["source"].cuda(), "target": label["target"]})]
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()
    worst = max(abs(a - b) / abs(b) for a, b in zip(out, ref))
    t = {k_: float(v) for k_, v in sg.loss_terms.items()}
    for a, b in TERM_PAIRS:
        e = abs(t[a] - orc.trace[b]) / max(abs(orc.trace[b]), 1e-3)
        worst = max(worst, e)
        assert e <= 1e-2, (a, t[a], orc.trace[b])
    if os.environ.get("SRGAN_TEST_LOG"):
        print("bf16 headline step: losses", out, "oracle", ref, "worst relative deviation", worst)
    np.testing.assert_allclose(out, ref, rtol=1e-2)


def test_config4_full_width_256_step_vs_oracle():
    """BASELINE configs[4]'s REAL dispatch: full-width networks on 256x256 images (64x64x256 trunk maps on F(4x4,3x3) / the
    LDS-resident-patch bf16 kernels, the five-conv discriminator at width 64, 128x128x64 and 256x256x64 stride-2 layers),
    bs 2, k 2, one whole train step against the CPU oracle on the same seeds -- in fp32 (1e-3, north_star) and in the bf16 mode
    the configuration names (1e-2 on the three losses and every loss term against the fp32 oracle)."""
    from srgan_amd import ops
    from srgan_amd.trainer import SRGAN_training
    batch, k, size = 2, 2, 256
    sg, orc = step_vs_oracle("F256", batch, k, seed=13, batch_seed=91, size=size, norm_rtol=1e-4)
    ref = [float(v) for v in orc.last_losses]
    del sg
    torch.cuda.empty_cache()
    x, label = otrainer.synthetic_batch(batch, size, 4, seed=91)
    ops.set_compute_dtype("bf16")
    try:
        G, D, E = build_hip_nets("F256")
        torch.manual_seed(13)
        sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), k,
                            "cuda", np.eye(4), batch, "mu", 8)
        sg.opt_sche_initialization()
        out = [float(v) for v in sg.train(x.cuda(), {"source": label["source"].cuda(), "target": label["target"]})]
    finally:
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()
    np.testing.assert_allclose(out, ref, rtol=1e-2)
    t = {k_: float(v) for k_, v in sg.loss_terms.items()}
    dev = {a: abs(t[a] - orc.trace[b]) / max(abs(orc.trace[b]), 1e-3) for a, b in TERM_PAIRS}
    if os.environ.get("SRGAN_TEST_LOG"):
        print("config4 bf16 step: losses", out, "oracle", ref, "term deviations", dev)
    # single terms of a TWO-image batch after two bf16 discriminator updates are noisier than the headline test's 32-image
    # means (4e-3 there): the identity-regression term moves between 0.7 % and 1.2 % with the summation order of unrelated
    # kernels; 2e-2 on a term, 1e-2 on the three losses
    for a, e in dev.items():
        assert e <= 2e-2, (a, t[a], e)


def _twenty_steps(k, pre=False, batch=4, steps=20, ref=None):
    """``steps`` consecutive train steps at tier-T widths on the real 128x128 geometry against the CPU (fp32) oracle on the same
    seeds, with the epoch boundary of the notebooks in the middle: ``ExponentialLR(gamma=0.95).step()`` on all three optimisers
    after step 10 (util_notebook.py:484-508; the oracle's Adam gets the same factor).  ``pre``: the pretrained-encoder recipe of
    05-train cell 22 (Adam(lr=1e-3) over fcmean / fcvar only).  ``ref``: the oracle's losses if they come from
    tests/golden/traj20_T.npz (tests/golden/make_traj20.py runs the same oracle routine); None runs the oracle here.  The HIP
    side runs in whatever compute mode is set."""
    from srgan_amd import optim as hoptim
    from srgan_amd.trainer import SRGAN_training
    from tests.common import oracle_twenty_steps
    orc = None
    if ref is None:
        orc, ref = oracle_twenty_steps(k, pre, batch, steps)
    G, D, E = build_hip_nets("T")
    optE = None
    if pre:
        keys = [k_ for k_ in E.state_dict().keys() if not k_.startswith(("fcmean", "fcvar"))]
        E.freeze_melt(keys, "freeze")
        optE = hoptim.Adam(filter(lambda p: p.requires_grad, E.parameters()), lr=1e-3, betas=(0.5, 0.999))
        E.freeze_melt(keys, "melt")
    torch.manual_seed(21)
    sg = SRGAN_training([G, D, E], [None, None, optE], [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), k,
                        "cuda", np.eye(4), batch, "mu", 8)
    sg.opt_sche_initialization()
    out = []
    for s in range(steps):
        if s == 10:
            sg.scheG.step(), sg.scheD.step(), sg.scheE.step()
        x, label = otrainer.synthetic_batch(batch, 128, 4, seed=600 + s)
        out.append([float(v) for v in sg.train(x.cuda(), {"source": label["source"].cuda(), "target": label["target"]})])
    assert abs(sg.optG.param_groups[0]["lr"] - 0.95e-4) < 1e-12
    out, ref = np.array(out), np.array(ref)
    rel = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-6)
    if os.environ.get("SRGAN_TEST_LOG"):
        with open(os.environ["SRGAN_TEST_LOG"], "a") as f:
            f.write(f"20-step trajectory k={k} pre={pre}, max relative deviation per step: " + " ".join(f"{v:.1e}" for v in rel.max(1)) + "\n")
    return sg, orc, out, ref


def test_twenty_step_trajectory_vs_oracle_across_a_scheduler_step():
    """fp32: nothing drifts apart over a trajectory -- optimiser state, the stale-graph phase, the RNG order -- beyond fp32
    reduction-order noise amplified by the GAN dynamics."""
    k, steps = 2, 20
    sg, orc, out, ref = _twenty_steps(k)
    np.testing.assert_allclose(out, ref, rtol=1e-3)          # measured on the MI355X: <= 3.8e-5 at every step
    for net, P, n_opt in ((sg.G, orc.G, 2 * steps), (sg.D, orc.D, k * steps), (sg.E, orc.E, steps)):
        for key, v in net.state_dict().items():
            close_params(v, P[key], 1e-4, n_opt, what=key, walk=True)


@pytest.mark.parametrize("name,k,pre,bound", [("k5", 5, False, BF16_TRAJ_BOUND_K5), ("k2_pretrainedE", 2, True, BF16_TRAJ_BOUND_PRE)])
def test_twenty_step_bf16_trajectory_vs_fp32_oracle(golden_dir, name, k, pre, bound):
    """three of the five BASELINE configurations train in the bf16 mode, and the only long trajectory was fp32.
    The same 20 steps (k = 5 as in the headline recipe; the pretrained-encoder recipe of configs[2]) with the convolutions on the
    bf16 MFMA, against the FP32 oracle: the deviation is bf16 rounding of the conv operands (at tier-T widths the layers with
    >= 32 channels) amplified by 20 steps of GAN dynamics.  Bounds = 1.5 x the largest relative deviation measured on the
    MI355X over the 20 steps x 3 losses (round 5: 5.3e-4 for k = 5, 2.8e-4 for the pretrained-encoder recipe; per-step values:
    SRGAN_TEST_LOG).  The oracle's losses come from tests/golden/traj20_T.npz (make_traj20.py: the routine the fp32 test runs live)."""
    from srgan_amd import ops
    gold = np.load(os.path.join(golden_dir, "traj20_T.npz"))
    ops.set_compute_dtype("bf16")
    try:
        _, _, out, ref = _twenty_steps(k, pre, ref=gold[name])
    finally:
        ops.set_compute_dtype("fp32")
    rel = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-6)
    assert float(rel.max()) <= bound, (float(rel.max()), rel.max(1))
    assert float(rel.max()) > 1e-6          # the mode really changed the arithmetic


def test_bs64_step_vs_oracle_tier_T():
    """BASELINE configs[3] batch (bs=64, 4 classes) at tier-T widths on the real 128x128 geometry."""
    step_vs_oracle("T", 64, 2, seed=9, batch_seed=64)


@pytest.mark.parametrize("name,tier,batch,k,steps,pre,size", [
    ("train_T_b4_k2_pretrainedE", "T", 4, 2, 2, True, 128),        # configs[2]: pretrained-E recipe + bf16 convolutions
    ("train_T256_b2_k2", "T256", 2, 2, 2, False, 256)])            # configs[4]: 256x256 geometry + bf16 convolutions
def test_bf16_mode_on_config_recipes_tier_T(golden_dir, name, tier, batch, k, steps, pre, size):
    """The bf16 MFMA compute mode combined with the recipes BASELINE configs[2] / [4] name, against the reference's own fp32
    trajectories within bf16 rounding of the conv operands: 2e-3 (round 4, tightened from 3e-2; measured 1.4e-4 .. 1.9e-4 --
    at tier-T widths only the layers with >= 32 channels multiply in bf16; scratch/bf16_deviation.py prints the numbers)."""
    from srgan_amd import ops
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    ops.set_compute_dtype("bf16")
    try:
        _, traj = run_hip(tier, batch, k, steps, seed=0, pretrained_e=pre, size=size)
    finally:
        ops.set_compute_dtype("fp32")
    np.testing.assert_allclose(traj, gold["losses"], rtol=2e-3)


def test_latent_mode_and_generic_encoder_path():
    """encoded_feature="latent": the style code is the noisy reparametrisation (RNG order matters)."""
    PG, PD, PE = oracle_params("T")
    lbd = dict(otrainer.DEFAULT_LBD)
    torch.manual_seed(11)
    orc = otrainer.SRGANOracle(PG, PD, PE, lbd, 2, np.eye(4), 4, "latent", 8)
    x, label = otrainer.synthetic_batch(4, 128, 4, seed=7)
    ref = [float(v) for v in orc.train(x, label)]
    from srgan_amd.trainer import SRGAN_training
    G, D, E = build_hip_nets("T")
    torch.manual_seed(11)
    sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], lbd, 2, "cuda", np.eye(4), 4, "latent", 8)
    sg.opt_sche_initialization()
    out = [float(v) for v in sg.train(x.cuda(), {"source": label["source"].cuda(), "target": label["target"]})]
    np.testing.assert_allclose(out, ref, rtol=1e-3)


# ---- config 1: conventional SingleGAN (BASELINE configs[0]) on the HIP path --------------------------------
SG_BASE = dict(**{"class": 0.0}, cycle=5.0, idt=5.0, reg=0.5, idt_reg=0.0, KL=0.1, batch_KL=0.0, corr_enc=0.0, hist=0.0)


@pytest.mark.parametrize("name,k,steps,lbd", [("singlegan_T_b8_k1", 1, 3, SG_BASE),
                                              ("singlegan_T_b8_k2_idtreg", 2, 2, dict(SG_BASE, idt_reg=0.5))])
def test_singlegan_trajectory_vs_reference(golden_dir, name, k, steps, lbd):
    from oracle import params
    from srgan_amd import model
    from srgan_amd.trainer import SingleGAN_training
    gold = np.load(os.path.join(golden_dir, name + ".npz"))
    G = model.SingleGenerator(3, 4, 2, 2, 1, "instance", num_con=10)
    G.load_state_dict(params.fill(params.generator_spec(3, 4, 2, 2, 1, 10), 20))
    D = []
    for i in range(2):
        d = model.SingleDiscriminator_original_multi(3, 4, 2, 4, "instance")
        d.load_state_dict(params.fill(params.discriminator_original_spec(3, 4, 2, 4), 21 + i))
        D.append(d.cuda())
    E = model.Encoder_original(3, 8, 4, 4, "instance", 2, "cuda")
    E.load_state_dict(params.fill(params.encoder_original_spec(3, 8, 4, 4, 2), 25))
    G.cuda(), E.cuda()
    torch.manual_seed(0)
    np.random.seed(0)
    sg = SingleGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], dict(lbd), k, "cuda", np.eye(2), 8,
                            (0, 1), 8, "latent", False)
    sg.opt_sche_initialization()
    traj = []
    for s in range(steps):
        x, label = otrainer.synthetic_batch(8, 64, 2, seed=200 + s)
        out = sg.train(x.cuda(), {"source": label["source"].cuda(), "target": label["target"]})
        traj.append([float(v) for v in out])
    np.testing.assert_allclose(np.array(traj), gold["losses"], rtol=1e-3)
    for name_, net, n_opt in (("G", sg.G, 2 * steps), ("D0", sg.D[0], k * steps), ("D1", sg.D[1], k * steps), ("E", sg.E, steps)):
        for key, v in net.state_dict().items():
            close_params(v, gold[f"{name_}.{key}"], 1e-4, n_opt, what=f"{name_}.{key}")


def test_encoder_pretraining_steps_vs_reference(golden_dir):
    """the encoder pre-training job of notebook 04 on the HIP path (Encoder_classifier, cross-entropy on its
    softmax output, Adam(1e-4) with default betas), against the reference's own 3-step trajectory."""
    from oracle import params
    from srgan_amd import losses as hl, model, optim as hoptim
    gold = np.load(os.path.join(golden_dir, "pretrain_T_b8.npz"))
    net = model.Encoder_classifier(3, 8, 4, 4, "instance", 4)
    spec = {k: v for k, v in params.encoder_spec(3, 8, 4, 4, 4).items() if not k.startswith(("fcmean", "fcvar"))}
    net.load_state_dict(params.fill(spec, 2))
    net.cuda()
    opt = hoptim.Adam(net.parameters(), lr=1e-4)
    crit = hl.CrossEntropyLoss()
    out = []
    for s in range(3):
        x, label = otrainer.synthetic_batch(8, 128, 4, seed=400 + s)
        opt.zero_grad()
        y = net(x.cuda())
        loss = crit(y, label["source"].cuda())
        loss.backward()
        opt.step()
        out.append(float(loss))
    np.testing.assert_allclose(out, gold["losses"], rtol=1e-4)
    close(y, gold["last_probs"], 2e-4, what="class probabilities")
    for key, v in net.state_dict().items():
        close_params(v, gold["P." + key], 1e-4, 3, what=key)

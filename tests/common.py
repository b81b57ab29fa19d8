"""Shared helpers for the parity tests (oracle side lives in oracle/)."""
import os

import numpy as np
import torch

from oracle import params

TIER_T = dict(G=dict(nch_in=3, nch=4, reduce=2, num_cls=2, res_num=1, num_con=12),
              D=dict(nch_in=3, nch=4, reduce=2, num_cls=4, n_class=4),
              E=dict(nch_in=3, nch_out=8, nch=4, num_cls=4, num_con=4))
TIER_F = dict(G=dict(nch_in=3, nch=64, reduce=2, num_cls=2, res_num=6, num_con=12),
              D=dict(nch_in=3, nch=64, reduce=2, num_cls=4, n_class=4),
              E=dict(nch_in=3, nch_out=8, nch=64, num_cls=4, num_con=4))


# tiny widths on the 256x256 geometry of BASELINE configs[4] (the discriminator gets a fifth down conv)
TIER_T256 = dict(G=TIER_T["G"], D=dict(nch_in=3, nch=4, reduce=2, num_cls=5, n_class=4), E=TIER_T["E"])


# full widths on the 256x256 geometry (BASELINE configs[4]'s real dispatch: 64x64x256 trunk maps, five-conv discriminator)
TIER_F256 = dict(G=TIER_F["G"], D=dict(nch_in=3, nch=64, reduce=2, num_cls=5, n_class=4), E=TIER_F["E"])


def tier(name):
    return {"T": TIER_T, "T256": TIER_T256, "F256": TIER_F256}.get(name, TIER_F)


def oracle_params(name, seed=0):
    t = tier(name)
    return (params.fill(params.generator_spec(**t["G"]), seed), params.fill(params.discriminator_spec(**t["D"]), seed + 1),
            params.fill(params.encoder_spec(**t["E"]), seed + 2))


def build_hip_nets(name, seed=0, device="cuda"):
    """srgan_amd modules with the reference constructor signatures, loaded with the deterministic fill."""
    from srgan_amd import model
    t = tier(name)
    g, d, e = t["G"], t["D"], t["E"]
    G = model.SingleGenerator(g["nch_in"], g["nch"], g["reduce"], g["num_cls"], g["res_num"], "instance", num_con=g["num_con"])
    D = model.SingleDiscriminator_solo_multi(d["nch_in"], d["nch"], d["reduce"], d["num_cls"], "instance", d["n_class"])
    E = model.Encoder(e["nch_in"], e["nch_out"], e["nch"], e["num_cls"], "instance", e["num_con"], device)
    for net, p in zip((G, D, E), oracle_params(name, seed)):
        net.load_state_dict(p)
        net.to(device)
    return G, D, E


def close(a, b, rtol=1e-4, atol=1e-6, what=""):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max())
    assert err <= atol + rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def close_grad(a, b, tol=2e-4, l2_tol=1e-2, med_tol=1e-2, what=""):
    """Gradient comparison that tolerates activation-mask flips -- and nothing else.

    ReLU / LeakyReLU derivatives are discontinuous: an element whose pre-activation lies within fp32 rounding of zero can take
    the other branch in a different (equally valid) summation order.  One such flip switches the gradient of THAT element on or
    off: the weight-gradient row of its output channel gains or loses one term of its sum over pixels (relative size
    ~1/sqrt(pixels): up to ~2e-2 on the 2x32x32 test maps), and through instance norm (mean over the H*W plane) every element
    of that plane, and everything upstream, moves by ~1/(H*W) of it.  A flip is therefore SPARSE-and-large plus BROAD-and-tiny.
    A wiring or scaling bug (a loss term weighted 2 % wrong, a missing factor on one path) is the opposite: broad and
    proportional.  So: pass if the max-norm error is within ``tol``; otherwise require
      * the relative L2 error within ``l2_tol`` (bounds the sparse part: an indexing bug gives O(1)), AND
      * the MEDIAN element error within ``med_tol`` of the median magnitude (bounds the broad part separately from the
        outliers).
    Measured on the MI355X (SRGAN_TEST_LOG=file): the one test of the suite that takes this branch is the full-width generator
    block on its 2x32x32 input -- a flip in a 16x16 plane of the up path (1/256 of a plane mean) moves every gradient upstream
    of it (max <= 8.3e-3, L2 <= 5.5e-3, median <= 6.2e-3 over 36 tensors), and the weight gradient of the layer that
    owns the flipped activation shows the sparse signature (up_convs.1: max 5.5e-2 on 1.6 % of its elements, median 2.5e-6, L2
    4.1e-3).  The bounds leave 1.6-1.8x on that and are half of round 1's single 2e-2 L2 bound."""
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max())
    if err <= 1e-6 + tol * scale:
        return
    l2 = float((a - b).norm() / max(float(b.norm()), 1e-30))
    med = float((a - b).abs().median() / max(float(b.abs().median()), 1e-30)) if a.numel() >= 32 else 0.0
    frac = float(((a - b).abs() > 1e-6 + tol * scale).double().mean())
    log = os.environ.get("SRGAN_TEST_LOG")
    if log:
        with open(log, "a") as f:
            f.write(f"close_grad fallback {what}: numel {a.numel()} max {err / scale:.3e} l2 {l2:.3e} median {med:.3e} "
                    f"beyond-tol fraction {frac:.3e}\n")
    assert l2 <= l2_tol, f"{what}: max err {err:.3e} (scale {scale:.3e}), rel L2 {l2:.3e}"
    assert med <= med_tol, f"{what}: median err / median magnitude {med:.3e} (broad error: not an activation flip)"


def close_params(a, b, lr, n_opt_steps, what="", walk=False):
    """Parameters after ``n_opt_steps`` Adam steps.  Adam moves every element by about lr*sign(g) in its first
    steps, so an element whose gradient is at rounding-noise level can legitimately end up to 2*lr per step away;
    the bulk of the tensor must agree far more tightly."""
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double().flatten()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double().flatten()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    # |Adam update| <= lr only in step 1; with betas (0.5, 0.999) the bias-corrected ratio m^/sqrt(v^) of step 2 reaches
    # (g1 + 2 g2)/3 / sqrt((g1^2 + g2^2)/2) <= 1.054 (at g2 = 2 g1) and stays below 1.1 in the first handful of steps --
    # observed 4.094e-4 = 2 * lr * 2.047 on one element of G.down_convs.1.weight after 2 steps at bs=32, k=5
    assert float(err.max()) <= 2.2 * lr * n_opt_steps + 1e-6, f"{what}: max err {float(err.max()):.3e}"
    if a.numel() >= 32:      # tiny tensors (a 4-element bias) have no meaningful "bulk"
        # ``walk``: over a long trajectory the elements whose gradient sits at rounding-noise level take independent +-lr
        # steps on the two sides, a random walk of ~lr * sqrt(steps) (measured after 40 steps: median 4.8e-5 = 0.48 lr on the
        # generator's RGB input layer, whose gradients are the noisiest; 0.2 * lr * sqrt(40) = 1.26 lr bounds it)
        bound = 0.2 * lr * (max(n_opt_steps, 1) ** 0.5 if walk else 1.0)
        assert float(err.median()) <= bound, f"{what}: median err {float(err.median()):.3e}"


def oracle_twenty_steps(k, pre, batch=4, steps=20):
    """The CPU oracle over the 20-step tier-T trajectory of tests/test_train_gpu.py (ExponentialLR step after step 10; ``pre``:
    the pretrained-encoder recipe); returns (oracle, losses[steps][3]).  Also run by tests/golden/make_traj20.py."""
    from oracle import trainer as otrainer
    PG, PD, PE = oracle_params("T")
    e_keys = [n for n in PE if n.startswith(("fcmean", "fcvar"))] if pre else None
    torch.manual_seed(21)
    orc = otrainer.SRGANOracle(PG, PD, PE, otrainer.DEFAULT_LBD, k, np.eye(4), batch, "mu", 8,
                               lr=(1e-4, 1e-4, 1e-3 if pre else 1e-4), e_trainable=e_keys)
    ref = []
    for s in range(steps):
        if s == 10:
            for o in (orc.optG, orc.optD, orc.optE):
                o.lr *= 0.95
        x, label = otrainer.synthetic_batch(batch, 128, 4, seed=600 + s)
        ref.append([float(v) for v in orc.train(x, label)])
    return orc, np.array(ref)



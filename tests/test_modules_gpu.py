"""GPU: the srgan_amd nn.Modules (HIP kernels through the C ABI) against
  (a) golden vectors produced by the imported reference (tests/golden/modules_T.npz), and
  (b) the CPU oracle on the same seeded inputs.
Tolerance: 1e-3 relative (BASELINE.json north_star) is the bar; the asserted bound is 2e-4."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import losses as olosses, nets as onets, trainer as otrainer
from tests.common import build_hip_nets, close, close_grad, oracle_params

pytestmark = pytest.mark.gpu
TOL = 2e-4


def pool8(t):
    from srgan_amd import ops
    return F.avg_pool2d(ops.to_nchw(t.detach()).cpu(), 8)


def test_generator_vs_reference_golden(golden_dir):
    gold = np.load(os.path.join(golden_dir, "modules_T.npz"))
    G, _, _ = build_hip_nets("T")
    x, label = otrainer.synthetic_batch(3, 128, 4, seed=11)
    c = torch.cat([olosses.one_hot_rows(label["target"], np.eye(4)), torch.from_numpy(gold["z"])], 1)
    y = G(x.cuda(), c.cuda())
    assert tuple(y.shape) == (3, 3, 128, 128)
    wy = torch.linspace(-1, 1, y.numel()).view(3, 3, 128, 128).cuda()
    (y * wy).sum().backward()
    close(pool8(y), gold["G_y_pool8"], TOL, what="G out")
    assert abs(float(y.detach().double().sum()) - float(gold["G_y_sum"])) <= TOL * float(gold["G_y_abs"])
    for k, p in G.named_parameters():
        close(p.grad, gold["G_grad." + k], TOL, 1e-5, what="G grad " + k)


def test_discriminator_vs_reference_golden(golden_dir):
    gold = np.load(os.path.join(golden_dir, "modules_T.npz"))
    _, D, _ = build_hip_nets("T")
    x, _ = otrainer.synthetic_batch(3, 128, 4, seed=11)
    xd = x.cuda().requires_grad_(True)
    (o1, o2), (c1, c2) = D(xd)
    assert tuple(o1.shape) == (3, 1, 7, 7) and tuple(o2.shape) == (3, 1, 3, 3) and tuple(c1.shape) == (3, 4)
    s = (o1 ** 2).sum() + (o2 * 0.5).sum() + (c1 * torch.arange(4.0).cuda()).sum() + (c2 ** 2).sum()
    s.backward()
    for name, t in (("D_o1", o1), ("D_o2", o2), ("D_c1", c1), ("D_c2", c2)):
        close(t.reshape(gold[name].shape), gold[name], TOL, what=name)
    close(pool8(xd.grad), gold["D_dx_pool8"], TOL, what="D dx")
    for k, p in D.named_parameters():
        close(p.grad, gold["D_grad." + k], TOL, 1e-6, what="D grad " + k)


def test_encoder_vs_reference_golden(golden_dir):
    gold = np.load(os.path.join(golden_dir, "modules_T.npz"))
    _, _, E = build_hip_nets("T")
    x, _ = otrainer.synthetic_batch(3, 128, 4, seed=11)
    xe = x.cuda().requires_grad_(True)
    torch.manual_seed(3)                      # the reparametrisation noise comes from the CPU generator
    code, mu, logvar, cls, none = E(xe)
    assert none is None
    s = (mu * torch.linspace(0.5, 1.5, mu.numel()).view_as(mu).cuda()).sum() + (logvar ** 2).sum() + cls.sum() + code.sum()
    s.backward()
    for name, t in (("E_code", code), ("E_mu", mu), ("E_logvar", logvar), ("E_cls", cls)):
        close(t, gold[name], TOL, what=name)
    close(pool8(xe.grad), gold["E_dx_pool8"], TOL, what="E dx")
    for k, p in E.named_parameters():
        close(p.grad, gold["E_grad." + k], TOL, 1e-6, what="E grad " + k)


def test_full_width_generator_block_vs_oracle():
    """Real channel widths (64/128/256, vector-path kernels) on a small image against the oracle."""
    from srgan_amd import model
    from oracle import params
    spec = params.generator_spec(3, 64, 2, 2, 2, 12)
    P = params.fill(spec, 3)
    G = model.SingleGenerator(3, 64, 2, 2, 2, "instance", num_con=12)
    G.load_state_dict(P)
    G.cuda()
    x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(1)) * 2 - 1
    c = torch.randn(2, 12, generator=torch.Generator().manual_seed(2))
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    yr = onets.generator(Pr, x, c)
    w = torch.randn(yr.shape, generator=torch.Generator().manual_seed(3))
    (yr * w).sum().backward()
    cd = c.cuda().requires_grad_(True)
    y = G(x.cuda(), cd)
    (y * w.cuda()).sum().backward()
    close(y, yr, TOL, what="G out")
    for k, p in G.named_parameters():
        close_grad(p.grad, Pr[k].grad, TOL, what="G grad " + k)


def test_full_width_encoder_discriminator_vs_oracle():
    from srgan_amd import model
    from oracle import params
    # E at real widths on a 64x64 image (30->15->7->3->1 maps), D at 128x128 with nch=16
    Pe = params.fill(params.encoder_spec(3, 8, 32, 4, 4), 5)
    E = model.Encoder(3, 8, 32, 4, "instance", 4, "cuda")
    E.load_state_dict(Pe)
    E.cuda()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)) * 2 - 1
    eps = torch.randn(2, 8, generator=torch.Generator().manual_seed(9))
    Pr = {k: v.clone().requires_grad_(True) for k, v in Pe.items()}
    xr = x.clone().requires_grad_(True)
    code_r, mu_r, lv_r, cls_r, _ = onets.encoder(Pr, xr, noise=eps)
    (mu_r.sum() + (lv_r ** 2).sum() + cls_r.sum()).backward()
    xd = x.cuda().requires_grad_(True)
    _, mu, lv, cls, _ = E(xd)
    (mu.sum() + (lv ** 2).sum() + cls.sum()).backward()
    close(mu, mu_r, TOL, what="mu")
    close(lv, lv_r, TOL, what="logvar")
    close_grad(xd.grad, xr.grad, TOL, what="E dx")
    for k, p in E.named_parameters():
        close_grad(p.grad, Pr[k].grad, TOL, what="E grad " + k)

    Pd = params.fill(params.discriminator_spec(3, 16, 2, 4, 4), 6)
    D = model.SingleDiscriminator_solo_multi(3, 16, 2, 4, "instance", 4)
    D.load_state_dict(Pd)
    D.cuda()
    x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(2)) * 2 - 1
    Pr = {k: v.clone().requires_grad_(True) for k, v in Pd.items()}
    (o1r, o2r), (c1r, c2r) = onets.discriminator(Pr, x, 4)
    ((o1r ** 2).sum() + o2r.sum() + (c1r ** 2).sum() + (c2r ** 2).sum()).backward()
    (o1, o2), (c1, c2) = D(x.cuda())
    ((o1 ** 2).sum() + o2.sum() + (c1 ** 2).sum() + (c2 ** 2).sum()).backward()
    close(o1, o1r, TOL, what="o1")
    close(c2, c2r, TOL, what="c2")
    for k, p in D.named_parameters():
        close_grad(p.grad, Pr[k].grad, TOL, what="D grad " + k)


def test_config1_modules_vs_oracle():
    """SingleGAN (config 1) networks: Encoder_original (CBIN-conditioned) and the per-domain D."""
    from srgan_amd import model
    from oracle import params
    Pe = params.fill(params.encoder_original_spec(3, 8, 8, 4, 2), 7)
    E = model.Encoder_original(3, 8, 8, 4, "instance", 2, "cuda")
    E.load_state_dict(Pe)
    E.cuda()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)) * 2 - 1
    c = torch.eye(2)[[0, 1]]
    eps = torch.randn(2, 8, generator=torch.Generator().manual_seed(9))
    Pr = {k: v.clone().requires_grad_(True) for k, v in Pe.items()}
    code_r, mu_r, lv_r = onets.encoder_original(Pr, x, c, noise=eps)
    (mu_r.sum() + lv_r.sum()).backward()
    code, mu, lv = E(x.cuda(), c.cuda())
    (mu.sum() + lv.sum()).backward()
    close(mu, mu_r, TOL)
    for k, p in E.named_parameters():
        close_grad(p.grad, Pr[k].grad, TOL, what="E_original grad " + k)
    Pd = params.fill(params.discriminator_original_spec(3, 8, 2, 4), 8)
    D = model.SingleDiscriminator_original_multi(3, 8, 2, 4, "instance")
    D.load_state_dict(Pd)
    D.cuda()
    outs_r = onets.discriminator_original(Pd, x)
    outs = D(x.cuda())
    assert [tuple(o.shape) for o in outs] == [(2, 1, 3, 3), (2, 1, 1, 1)]
    for o, r in zip(outs, outs_r):
        close(o, r, TOL)


@pytest.mark.parametrize("leaf_input", [False, True])
def test_second_scale_on_a_side_stream_changes_no_bit(leaf_input):
    """model._second_scale: discriminator2 (+ its heads) runs on a side stream, forward and -- replayed by autograd -- backward.
    Same kernels, same operands, same accumulation order of the two input gradients: outputs, input gradient and every
    parameter gradient bit-identical to the single-stream run, for a leaf input (its two gradients meet behind a view node of
    the caller's stream) and for a non-leaf one (the generator's output in the trainer); full width, batch 8."""
    from srgan_amd import model
    _, D, _ = build_hip_nets("F")
    x0, _ = otrainer.synthetic_batch(8, 128, 4, seed=21)
    saved = model._PARALLEL_SCALES
    runs = {}
    try:
        for par in (False, True):
            model._PARALLEL_SCALES = par
            for p in D.parameters():
                p.grad = None
            x = x0.cuda().requires_grad_(True)
            xin = x if leaf_input else x * 1.0
            (o1, o2), (c1, c2) = D(xin)
            ((o1 ** 2).mean() + (o2 * 0.5).sum() + (c1 * torch.arange(4.0).cuda()).sum() + (c2 ** 2).sum()).backward()
            torch.cuda.synchronize()
            runs[par] = ([t.detach().clone() for t in (o1, o2, c1, c2, x.grad)], {k: p.grad.clone() for k, p in D.named_parameters()})
            del o1, o2, c1, c2, xin, x          # the graph goes, and with it the parameters' AccumulateGrad nodes of this pass
    finally:
        model._PARALLEL_SCALES = saved
    for a, b in zip(runs[False][0], runs[True][0]):
        assert torch.equal(a, b)
    for k in runs[False][1]:
        assert torch.equal(runs[False][1][k], runs[True][1][k]), k
    assert model._side_streams                     # the fork really happened


def test_reparametrize_kernels_equal_the_elementwise_chain():
    """Encoder.reparametrize (model.py:459-463) as one launch forward and one backward against the reference's chain of
    elementwise passes (every product and sum rounded on its own in both), value and both gradients."""
    from srgan_amd import model
    g0 = torch.Generator().manual_seed(5)
    mu, logvar, eps, gy = (torch.randn(32, 8, generator=g0).cuda() for _ in range(4))
    logvar = logvar * 3
    a, b = mu.clone().requires_grad_(True), logvar.clone().requires_grad_(True)
    out = model._ReparamFn.apply(a, b, eps)
    out.backward(gy)
    ar, br = mu.clone().requires_grad_(True), logvar.clone().requires_grad_(True)
    std = torch.exp(0.5 * br)
    ref = eps * std + ar
    ref.backward(gy)
    close(out, ref, 1e-6)                 # (expf here, ATen's exp there: an ulp apart at most)
    assert torch.equal(a.grad, ar.grad)
    close(b.grad, br.grad, 1e-6)
    # against float64: not further from the exact value than the chain is
    exact = eps.double() * torch.exp(0.5 * logvar.double()) + mu.double()
    assert float((out.detach().double() - exact).abs().max()) <= 1.5 * float((ref.detach().double() - exact).abs().max()) + 1e-12


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_encoder_shortcut_on_a_side_stream_changes_no_bit(mode):
    """model._side_branch (round 5): the shortcut of every encoder block (AvgPool2d -> 1x1 conv, model.py:409-411) runs on the
    side stream, forward and -- replayed by autograd -- backward.  Same kernels, same operands, the same engine-ordered sum of the
    block input's two gradients: outputs, input gradient and every parameter gradient bit-identical to the single-stream run,
    in both compute modes (bf16: with the blocks' 16-bit activations)."""
    from srgan_amd import model, ops
    _, _, E = build_hip_nets("F")
    x0, _ = otrainer.synthetic_batch(8, 128, 4, seed=22)
    saved = model._PARALLEL_SHORTCUT
    runs = {}
    ops.set_compute_dtype(mode)
    try:
        for par in (False, True):
            model._PARALLEL_SHORTCUT = par
            for p in E.parameters():
                p.grad = None
            x = x0.cuda().requires_grad_(True)
            with ops.pack_cache():
                _, mu, logvar, cls, _ = E(x * 1.0)
                ((mu ** 2).mean() + (logvar * 0.5).sum() + (cls * torch.arange(float(cls.shape[1])).cuda()).sum()).backward()
            torch.cuda.synchronize()
            runs[par] = ([t.detach().clone() for t in (mu, logvar, cls, x.grad)], {k: p.grad.clone() for k, p in E.named_parameters()})
            del mu, logvar, cls, x
    finally:
        model._PARALLEL_SHORTCUT = saved
        ops.set_compute_dtype("fp32")
        ops.invalidate_packed()
    for a, b in zip(runs[False][0], runs[True][0]):
        assert torch.equal(a, b)
    for k in runs[False][1]:
        assert torch.equal(runs[False][1][k], runs[True][1][k]), k
    assert model._side_streams

#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING the reference in the build container.

Run from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (read-only); never runs on the GPU box.  Writes only numbers
(.npz / .json) next to this file -- no reference source, bytecode or pickled modules.

What is imported from the reference: pyfiles/model.py (networks), pyfiles/util.py
(losses, class_encode), pyfiles/util_notebook.py (SRGAN_training).  Obstacles handled
outside the reference: a stub ``prdc`` module, and optimisers with
torch-1.4 arithmetic that write through ``p.data`` (modern torch.optim.Adam bumps the
version counters and SRGAN_training.train raises at util_notebook.py:689).
"""
import json
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import params as oparams  # noqa: E402  (build-owned deterministic fill)

_stub = types.ModuleType("prdc")
_stub.compute_prdc = lambda *a, **k: None
sys.modules["prdc"] = _stub
import matplotlib  # noqa: E402
matplotlib.use("Agg")
sys.path.insert(0, "/root/reference/pyfiles")
import model as ref_model            # noqa: E402
import util as ref_util              # noqa: E402
import util_notebook as ref_nb       # noqa: E402


class LegacyAdam(torch.optim.Optimizer):
    """torch==1.4.0 Adam arithmetic, parameters updated through .data (no version bump)."""

    def __init__(self, params, lr=1e-4, betas=(0.5, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def step(self, closure=None):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p.data)
                    st["exp_avg_sq"] = torch.zeros_like(p.data)
                st["step"] += 1
                g = p.grad.data
                st["exp_avg"].mul_(b1).add_(g, alpha=1 - b1)
                st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = st["exp_avg_sq"].sqrt() / math.sqrt(1 - b2 ** st["step"]) + group["eps"]
                p.data.addcdiv_(st["exp_avg"], denom, value=-group["lr"] / (1 - b1 ** st["step"]))


def shapes_of(net):
    return [[k, list(v.shape)] for k, v in net.state_dict().items()]


def load_fill(net, seed):
    sd = net.state_dict()
    filled = {k: torch.from_numpy(oparams.fill_array(k, tuple(v.shape), seed)) for k, v in sd.items()}
    net.load_state_dict(filled)
    return net


def pool8(t):
    return torch.nn.functional.avg_pool2d(t, 8).detach().numpy()


def build_nets(tier, seed=0):
    if tier in ("T", "T256"):      # tiny widths, real 128x128 (256x256: D with num_cls=5, BASELINE configs[4]) geometry
        G = ref_model.SingleGenerator(3, 4, 2, 2, 1, "instance", num_con=12)
        D = ref_model.SingleDiscriminator_solo_multi(3, 4, 2, 5 if tier == "T256" else 4, "instance", 4)
        E = ref_model.Encoder(3, 8, 4, 4, "instance", 4, "cpu")
    else:                # notebook configuration (05-train cell 13/20)
        G = ref_model.SingleGenerator(3, 64, 2, 2, 6, "instance", num_con=12)
        D = ref_model.SingleDiscriminator_solo_multi(3, 64, 2, 4, "instance", 4)
        E = ref_model.Encoder(3, 8, 64, 4, "instance", 4, "cpu")
    return load_fill(G, seed), load_fill(D, seed + 1), load_fill(E, seed + 2)


def synthetic_batch(batch, size, n_class, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    src = torch.randint(0, n_class, (batch,), generator=g)
    tgt = (src + torch.randint(1, n_class, (batch,), generator=g)) % n_class
    return x, {"source": src, "target": tgt}


LBD = {"class": 1.0, "cycle": 5.0, "idt": 5.0, "reg": 0.5, "idt_reg": 0.5, "KL": 0.0,
       "batch_KL": 10.0, "corr_enc": 100.0, "hist": 100.0}


def golden_shapes():
    out = {}
    for tier in ("T", "F"):
        G, D, E = build_nets(tier)
        out[tier] = {"G": shapes_of(G), "D": shapes_of(D), "E": shapes_of(E)}
    out["D_original_64"] = shapes_of(ref_model.SingleDiscriminator_original_multi(3, 64, 2, 4, "instance"))
    out["E_original_64"] = shapes_of(ref_model.Encoder_original(3, 8, 64, 4, "instance", 2, "cpu"))
    out["G_cfg1"] = shapes_of(ref_model.SingleGenerator(3, 64, 2, 2, 6, "instance", num_con=10))
    # PyTorch default init under a fixed seed (weights_init is a no-op): per-key [sum, |sum|] checksums
    torch.manual_seed(7)
    nets = dict(G=ref_model.SingleGenerator(3, 8, 2, 2, 2, "instance", num_con=12),
                D=ref_model.SingleDiscriminator_solo_multi(3, 8, 2, 4, "instance", 4),
                E=ref_model.Encoder(3, 8, 8, 4, "instance", 4, "cpu"))
    out["init_seed7"] = {n: [[k, float(v.double().sum()), float(v.double().abs().sum())] for k, v in net.state_dict().items()]
                         for n, net in nets.items()}
    with open(os.path.join(HERE, "shapes.json"), "w") as f:
        json.dump(out, f)


def golden_modules():
    """Tier-T module forward outputs + all parameter gradients of a fixed scalar."""
    G, D, E = build_nets("T")
    x, label = synthetic_batch(3, 128, 4, seed=11)
    gen = torch.Generator().manual_seed(5)
    z = torch.randn(3, 8, generator=gen)
    c = torch.cat([ref_util.class_encode(label["target"], "cpu", np.eye(4)), z], 1)
    out = {}
    # G
    y = G(x, c)
    wy = torch.linspace(-1, 1, y.numel()).view_as(y)
    (y * wy).sum().backward()
    out["G_y_pool8"] = pool8(y)
    out["G_y_sum"] = np.float64(y.double().sum())
    out["G_y_abs"] = np.float64(y.double().abs().sum())
    for k, p in G.named_parameters():
        out["G_grad." + k] = p.grad.numpy().copy()
    # D
    xd = x.clone().requires_grad_(True)
    (o1, o2), (c1, c2) = D(xd)
    s = (o1 ** 2).sum() + (o2 * 0.5).sum() + (c1 * torch.arange(4.0)).sum() + (c2 ** 2).sum()
    s.backward()
    out.update(D_o1=o1.detach().numpy(), D_o2=o2.detach().numpy(), D_c1=c1.detach().numpy(), D_c2=c2.detach().numpy())
    out["D_dx_pool8"] = pool8(xd.grad)
    for k, p in D.named_parameters():
        out["D_grad." + k] = p.grad.numpy().copy()
    # E
    xe = x.clone().requires_grad_(True)
    torch.manual_seed(3)
    code, mu, logvar, cls, _ = E(xe)
    s = (mu * torch.linspace(0.5, 1.5, mu.numel()).view_as(mu)).sum() + (logvar ** 2).sum() + cls.sum() + code.sum()
    s.backward()
    torch.manual_seed(3)
    eps = torch.FloatTensor(mu.size()).normal_()
    out.update(E_code=code.detach().numpy(), E_mu=mu.detach().numpy(), E_logvar=logvar.detach().numpy(),
               E_cls=cls.detach().numpy(), E_eps=eps.numpy())
    out["E_dx_pool8"] = pool8(xe.grad)
    for k, p in E.named_parameters():
        out["E_grad." + k] = p.grad.numpy().copy()
    out["x_seed"], out["z"] = np.int64(11), z.numpy()
    np.savez_compressed(os.path.join(HERE, "modules_T.npz"), **out)


def golden_losses():
    out = {}
    torch.manual_seed(0)
    hi = ref_util.histogram_imitation("cpu")
    out["hist_target_seed0"] = hi.target.detach().numpy()
    torch.manual_seed(1234)
    mu = torch.randn(32, 8)
    mu2 = (1.5 * torch.sin(0.37 * torch.arange(256.0))).reshape(32, 8)
    for name, m in (("randn1234", mu), ("sin", mu2)):
        m = m.clone().requires_grad_(True)
        n_batch = 32
        var = torch.var(m, dim=0) * n_batch / (n_batch - 1)
        mean = torch.mean(m, dim=0)
        bkl = -0.5 * torch.sum(1 + torch.log(var) - mean ** 2 - var)
        corr = ref_util.corrcoef_loss(m.T, "cpu")
        hist = hi.loss(m)
        gb, = torch.autograd.grad(bkl, m, retain_graph=True)
        gc, = torch.autograd.grad(corr, m, retain_graph=True)
        gh, = torch.autograd.grad(hist, m)
        out[f"{name}_mu"] = m.detach().numpy()
        out[f"{name}_vals"] = np.array([float(bkl), float(corr), float(hist)], dtype=np.float64)
        out[f"{name}_dbkl"], out[f"{name}_dcorr"], out[f"{name}_dhist"] = gb.numpy(), gc.numpy(), gh.numpy()
    # LSGAN / class-MSE known answers
    g = torch.Generator().manual_seed(7)
    o1, o2 = torch.randn(4, 1, 7, 7, generator=g), torch.randn(4, 1, 3, 3, generator=g)
    q1 = torch.softmax(torch.randn(4, 4, generator=g), 1)
    q2 = torch.softmax(torch.randn(4, 4, generator=g), 1)
    lab = torch.tensor([0, 3, 1, 2])
    mse = nn.MSELoss()
    out["ls_o1"], out["ls_o2"], out["ls_q1"], out["ls_q2"], out["ls_lab"] = o1.numpy(), o2.numpy(), q1.numpy(), q2.numpy(), lab.numpy()
    out["ls_vals"] = np.array([float(ref_util.get_loss_D([o1, o2], 1., mse, "cpu")),
                               float(ref_util.get_loss_D([o1, o2], 0., mse, "cpu")),
                               float(ref_util.get_domainloss_D([q1, q2], ref_util.class_encode(lab, "cpu", np.eye(4)), mse))])
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)


def run_train(tier, batch, k, steps, seed, pretrained_e=False, size=128):
    """Drive the reference SRGAN_training.train and record what it returns + final params."""
    G, D, E = build_nets(tier)
    optE = None
    if pretrained_e:   # 05-train cell 22: only fcmean/fcvar are in optE (lr 1e-3), then melted
        keys = [k_ for k_ in E.state_dict().keys() if not k_.startswith(("fcmean", "fcvar"))]
        E.freeze_melt(keys, "freeze")
        optE = LegacyAdam(filter(lambda p: p.requires_grad, E.parameters()), lr=1e-3, betas=(0.5, 0.999))
        E.freeze_melt(keys, "melt")
    torch.manual_seed(seed)
    np.random.seed(seed)
    sg = ref_nb.SRGAN_training([G, D, E], [LegacyAdam(G.parameters()), LegacyAdam(D.parameters()),
                                           optE if optE is not None else LegacyAdam(E.parameters())],
                               [nn.MSELoss(), nn.MSELoss()], dict(LBD), k, "cpu", np.eye(4), batch, "mu", 8)
    sg.opt_sche_initialization()
    losses = []
    for s in range(steps):
        x, label = synthetic_batch(batch, size, 4, seed=100 + s)
        errG, errD, errE = sg.train(x, label)
        losses.append([float(errG), float(errD), float(errE)])
    out = {"losses": np.array(losses, dtype=np.float64), "hist_target": sg.hi.target.detach().numpy()}
    for name, net in (("G", sg.G), ("D", sg.D), ("E", sg.E)):
        for k_, v in net.state_dict().items():
            v = v.detach().double()
            if tier in ("T", "T256"):
                out[f"{name}.{k_}"] = v.float().numpy()
            else:
                out[f"{name}_ck.{k_}"] = np.array([float(v.sum()), float(v.norm())] + v.flatten()[:8].tolist())
    return out


def golden_train():
    np.savez_compressed(os.path.join(HERE, "train_T_b4_k2.npz"), **run_train("T", 4, 2, 3, seed=0))
    np.savez_compressed(os.path.join(HERE, "train_T_b4_k5.npz"), **run_train("T", 4, 5, 2, seed=0))
    np.savez_compressed(os.path.join(HERE, "train_T_b4_k2_pretrainedE.npz"), **run_train("T", 4, 2, 2, seed=0, pretrained_e=True))
    np.savez_compressed(os.path.join(HERE, "train_F_b2_k1.npz"), **run_train("F", 2, 1, 2, seed=0))
    golden_train_256()


def golden_train_long():
    """a REFERENCE trajectory longer than three steps (10 steps, tier T, k = 2; losses only + final
    parameters), so that the oracle -- whose own 20-step runs are the yardstick of the bf16 trajectory tests -- is pinned to
    the reference over more optimiser steps than the other fixtures cover (20 generator, 20 discriminator, 10 encoder)."""
    np.savez_compressed(os.path.join(HERE, "train_T_b4_k2_s10.npz"), **run_train("T", 4, 2, 10, seed=0))


def golden_train_256():
    np.savez_compressed(os.path.join(HERE, "train_T256_b2_k2.npz"), **run_train("T256", 2, 2, 2, seed=0, size=256))


def run_singlegan(k, steps, seed, lbd, batch=8):
    """Config 1 (notebook 01): conventional SingleGAN, 64x64, 2 domains, per-domain D list, Encoder_original."""
    G = load_fill(ref_model.SingleGenerator(3, 4, 2, 2, 1, "instance", num_con=2 + 8), 20)
    D = [load_fill(ref_model.SingleDiscriminator_original_multi(3, 4, 2, 4, "instance"), 21 + i) for i in range(2)]
    E = load_fill(ref_model.Encoder_original(3, 8, 4, 4, "instance", 2, "cpu"), 25)
    torch.manual_seed(seed)
    np.random.seed(seed)
    sg = ref_nb.SingleGAN_training([G, D, E], [LegacyAdam(G.parameters()), None, LegacyAdam(E.parameters())],
                                   [nn.MSELoss(), nn.MSELoss()], dict(lbd), k, "cpu", np.eye(2), 8, (0, 1), batch, "latent", False)
    sg.opt_sche_initialization()
    losses = []
    for s in range(steps):
        x, label = synthetic_batch(batch, 64, 2, seed=200 + s)
        errG, errD, errE = sg.train(x, label)
        losses.append([float(errG), float(errD), float(errE)])
    out = {"losses": np.array(losses, dtype=np.float64)}
    for name, net in (("G", sg.G), ("D0", sg.D[0]), ("D1", sg.D[1]), ("E", sg.E)):
        for k_, v in net.state_dict().items():
            out[f"{name}.{k_}"] = v.detach().float().numpy()
    return out


def golden_singlegan():
    base = {"class": 0.0, "cycle": 5.0, "idt": 5.0, "reg": 0.5, "idt_reg": 0.0, "KL": 0.1, "batch_KL": 0.0, "corr_enc": 0.0, "hist": 0.0}
    np.savez_compressed(os.path.join(HERE, "singlegan_T_b8_k1.npz"), **run_singlegan(1, 3, 0, base))
    ext = dict(base, idt_reg=0.5)
    np.savez_compressed(os.path.join(HERE, "singlegan_T_b8_k2_idtreg.npz"), **run_singlegan(2, 2, 0, ext))


def golden_pretrain():
    """Encoder pre-training job (notebook 04 cells 18/22): Encoder_classifier + CrossEntropyLoss on its softmax output,
    Adam(lr=1e-4, default betas).  Tier-T widths, 3 steps, batch 8."""
    net = ref_model.Encoder_classifier(3, 8, 4, 4, "instance", 4)
    sd = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(oparams.fill_array(k, tuple(v.shape), 2)) for k, v in sd.items()})
    opt = LegacyAdam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    crit = nn.CrossEntropyLoss()
    losses, outs = [], None
    for s in range(3):
        x, label = synthetic_batch(8, 128, 4, seed=400 + s)
        opt.zero_grad()
        y = net(x)
        loss = crit(y, label["source"])
        loss.backward()
        opt.step()
        losses.append(float(loss))
        outs = y.detach().numpy()
    out = {"losses": np.array(losses), "last_probs": outs}
    for k_, v in net.state_dict().items():
        out["P." + k_] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "pretrain_T_b8.npz"), **out)


def golden_inference():
    """get_samples (pyfiles/util_notebook.py:858-949), the multimodal inference path: one source image, every target class,
    a list of latent codes pushed through G in chunks of `batch`, E re-encoding each output.  Tier-T networks."""
    G, _, E = build_nets("T")
    torch.manual_seed(3)
    dataset = [(torch.rand(3, 128, 128) * 2 - 1, int(i % 4)) for i in range(3)]
    latent = np.random.RandomState(7).randn(5, 8).astype(np.float32)
    data, label = ref_nb.get_samples(G, E, dataset, 1, latent=latent, classes=(0, 1, 2, 3), ref_label=np.eye(4), ndim=8,
                                     image_type="tensor", batch=2, device="cpu")
    out = {"latent": latent, "source": data["source"].numpy(), "source_label": np.asarray(label["source"]),
           "images": np.stack([dataset[i][0].numpy() for i in range(3)]), "labels": np.array([d[1] for d in dataset])}
    for c in range(4):
        out[f"target.{c}"] = data["target"][c].numpy()
        out[f"mu.{c}"] = np.concatenate(label["latent"][c], axis=0)
    pil, _ = ref_nb.get_samples(G, E, dataset, 1, latent=latent[:2], classes=(0, 1), ref_label=np.eye(4), ndim=8,
                                image_type="pil", batch=32, device="cpu")
    out["pil.source"] = np.asarray(pil["source"])
    out["pil.target.1"] = np.stack([np.asarray(im) for im in pil["target"][1]])
    np.savez_compressed(os.path.join(HERE, "inference_T.npz"), **out)


def golden_plot():
    """get_output_and_plot (pyfiles/util_notebook.py:738-846), the sample sheet the train notebooks draw every third of an
    epoch (05-train cell 24): the seven G_transformation calls in their order (CPU-generator noise), then a matplotlib
    figure.  Stored: every panel's title and 8-bit image, read back from the figure the reference returns."""
    G, D, E = build_nets("T")
    sg = ref_nb.SRGAN_training([G, D, E], [LegacyAdam(G.parameters()), LegacyAdam(D.parameters()), LegacyAdam(E.parameters())],
                               [nn.MSELoss(), nn.MSELoss()], dict(LBD), 1, "cpu", np.eye(4), 4, "mu", 8)
    torch.manual_seed(3)
    dataset = [(torch.rand(3, 128, 128) * 2 - 1, int(i % 4)) for i in range(3)]
    names = ["male, smiling", "male, not smiling", "female, smiling", "female, not smiling"]     # 05-train cell 7
    torch.manual_seed(5)
    with torch.no_grad():
        fig = ref_nb.get_output_and_plot(sg, dataset, 1, [(0, 1, 2, 3), names], 3, "cpu")
    titles, panels, slots = [], [], []
    for ax in fig.axes:
        titles.append(ax.get_title())
        panels.append(np.asarray(ax.images[0].get_array()))
        g = ax.get_subplotspec().get_geometry()
        slots.append([g[0], g[1], g[2]])
    # every 4th pixel of every panel (random-noise images do not compress: 1.3 MB in full)
    np.savez_compressed(os.path.join(HERE, "plot_T.npz"), titles=np.array(titles), panels=np.stack(panels).astype(np.uint8)[:, ::4, ::4],
                        slots=np.array(slots), figsize=np.array(fig.get_size_inches()),
                        image_checksum=np.array([float(d[0].double().sum()) for d in dataset]),      # the test re-draws them (seed 3)
                        labels=np.array([d[1] for d in dataset]))


def golden_facedataset():
    """File selection / split / label logic of FaceDataset (pyfiles/dataset.py:58-124) on a synthetic label set.
    The reference class still uses ``np.int`` (removed from numpy): the alias is restored for the import only."""
    import pickle
    import tempfile
    np.int = int
    import dataset as ref_dataset
    rng = np.random.default_rng(5)
    files = []
    for f in range(3):
        n = 90
        names = np.array(["%06d.jpg" % (f * 1000 + i) for i in rng.permutation(n)])
        attrs = rng.choice(["1", "-1"], size=(n, 5))
        files.append(np.concatenate([names[:, None], attrs], axis=1))
    cases = [
        dict(dataset_label={"class": [1, 2], "delete": [], "existed": []}, classes=(0, 1, 2, 3), train_num=20, val_num=5, test_num=5),
        dict(dataset_label={"class": [3], "delete": [4], "existed": [5]}, classes=(0, 1), train_num=2000, val_num=4, test_num=3),
        dict(dataset_label={"class": [2, 4], "delete": [1], "existed": []}, classes=(0, 1, 2, 3), train_num=7, val_num=2, test_num=6),
    ]
    out = {"label_files": [f.tolist() for f in files], "cases": []}
    with tempfile.TemporaryDirectory() as tmp:
        lab = os.path.join(tmp, "labels") + os.sep
        os.makedirs(lab)
        for i, f in enumerate(files):
            with open(os.path.join(lab, "part%d.pkl" % i), "wb") as fh:
                pickle.dump(f, fh)
        for case in cases:
            rec = {k: (list(v) if isinstance(v, tuple) else v) for k, v in case.items()}
            rec["splits"] = {}
            for dt in ("train", "val", "test"):
                ds = ref_dataset.FaceDataset("IMG/", lab, None, case["dataset_label"], case["classes"], dt, case["train_num"],
                                             case["val_num"], case["test_num"])
                rec["splits"][dt] = {"images": list(ds.images), "labels": [int(v) for v in ds.labels]}
            out["cases"].append(rec)
    out["class_label_3"] = [list(t) for t in ref_dataset.get_class_label(3)]
    with open(os.path.join(HERE, "facedataset.json"), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "facedataset":
        golden_facedataset()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "inference":
        golden_inference()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "plot":
        golden_plot()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "trainlong":
        golden_train_long()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "train256":
        golden_train_256()
        sys.exit(0)
    golden_shapes()
    golden_modules()
    golden_losses()
    golden_train()
    golden_train_long()
    golden_singlegan()
    golden_pretrain()
    golden_facedataset()
    golden_inference()
    golden_plot()
    print("golden fixtures written to", HERE)

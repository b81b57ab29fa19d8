"""CPU restatement of the SRGAN train step (k x update_D + two-phase update_GandE).

TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows ``SRGAN_training`` in the reference, pyfiles/util_notebook.py:
  * G_transformation  :510-561      * update_D        :563-594
  * update_GandE      :596-694      * UnrolledUpdate  :696-728   * train :730-734
and the optimiser it builds (:484-508, ``optim.Adam(lr, betas=(0.5, 0.999))`` with the
torch==1.4.0 arithmetic pinned by Docker/requirements.txt:8).

Semantics that are easy to lose and are kept on purpose:
  1. the graph of ``target_image`` built in the LAST update_D is back-propagated in phase 1,
     G is stepped, and the SAME graph is back-propagated again in phase 2: saved activations
     are old, weights read at backward time are new.  ``Adam14`` therefore writes through
     ``.data`` (no autograd version bump), which is what torch 1.4's optimiser did.
  2. the "unrolled" restore of D is a no-op (state_dict aliases live tensors) -> not restated.
  3. corr / hist terms are nested under ``batch_KL > 0``.
  4. batch-KL applies n/(n-1) on top of the unbiased variance, n = constructor batch size.
  5. errE returned is a reporting sum (cycle/idt terms included), not the optimised errE.
  6. RNG order on the CPU default generator: k x randn(B, ndim); phase 1: normal_ x2;
     phase 2: normal_ x3 (Appendix B.3).
"""
import math

import torch

from . import nets, losses


class Adam14:
    """Adam exactly as torch 1.4 computed it, updating ``p.data`` in place.

    m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ;
    p <- p - lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).   Params with grad None are skipped.
    """

    def __init__(self, params, lr=1e-4, betas=(0.5, 0.999), eps=1e-8):
        self.params = list(params)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state = {}

    def step(self):
        b1, b2 = self.betas
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.setdefault(id(p), {"t": 0, "m": torch.zeros_like(p.data), "v": torch.zeros_like(p.data)})
            st["t"] += 1
            g = p.grad.data
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = st["v"].sqrt() / math.sqrt(1 - b2 ** st["t"]) + self.eps
            p.data.addcdiv_(st["m"], denom, value=-self.lr / (1 - b1 ** st["t"]))


def _leafify(P):
    return {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}


def _zero(P):
    for p in P.values():
        p.grad = None


DEFAULT_LBD = dict(**{"class": 1.0}, cycle=5.0, idt=5.0, reg=0.5, idt_reg=0.5, KL=0.0,
                   batch_KL=10.0, corr_enc=100.0, hist=100.0)   # 05-train notebook, cell 16


class SRGANOracle:
    def __init__(self, PG, PD, PE, lbd, k, ref_label, n_batch, encoded_feature="mu", ndim=8,
                 lr=(1e-4, 1e-4, 1e-4), hist_target=None, e_trainable=None):
        self.G, self.D, self.E = _leafify(PG), _leafify(PD), _leafify(PE)
        self.lbd, self.k = dict(lbd), int(k)
        self.ref_label, self.n_batch = ref_label, n_batch
        self.encoded_feature, self.ndim = encoded_feature, ndim
        self.n_class = len(ref_label)
        e_params = [p for n, p in self.E.items() if e_trainable is None or n in e_trainable]
        self.optG = Adam14(self.G.values(), lr[0])
        self.optD = Adam14(self.D.values(), lr[1])
        self.optE = Adam14(e_params, lr[2])
        self.hi = None
        if self.lbd["hist"] > 0:   # draws randn(100000,1) unless a target is injected
            self.hi = losses.HistogramImitation(target=hist_target)
        self.trace = {}

    # -- helpers -----------------------------------------------------------------------
    def _onehot(self, label):
        return losses.one_hot_rows(label, self.ref_label)

    def _enc(self, x):
        return nets.encoder(self.E, x)      # draws normal_() from the default generator

    def _translate(self, target_label, src, use_encoder=False, ref_image=None):
        if use_encoder:
            info = list(self._enc(ref_image))
            z = info[0] if self.encoded_feature == "latent" else info[1]
        else:
            z = torch.randn(src.shape[0], self.ndim)
            info = z
        c = torch.cat([self._onehot(target_label), z], 1)
        return nets.generator(self.G, src, c), info

    # -- update_D ----------------------------------------------------------------------
    def update_D(self):
        _zero(self.D)
        self.target_image, self.c_rand = self._translate(self.label["target"], self.source)
        out, cls = nets.discriminator(self.D, self.source, self.n_class)
        real = losses.lsgan(out, 1.0)
        dom = losses.class_mse(cls, self._onehot(self.label["source"]))
        out_f, _ = nets.discriminator(self.D, self.target_image.detach(), self.n_class)
        fake = losses.lsgan(out_f, 0.0)
        errD = real + dom * self.lbd["class"] + fake
        errD.backward()
        self.trace.setdefault("errD_iters", []).append(float(errD))
        self.trace.setdefault("errD_parts", []).append((float(real), float(dom), float(fake)))
        if self.capture_grads:
            self.trace.setdefault("gradD", []).append({k: p.grad.clone() for k, p in self.D.items()})
        self.optD.step()
        return errD.detach()

    # -- update_GandE ------------------------------------------------------------------
    def update_GandE(self):
        L, tr = self.lbd, self.trace
        _zero(self.G)
        _zero(self.E)
        src, lab = self.source, self.label
        # phase 1 -----------------------------------------------------------------
        recon, enc_info = self._translate(lab["source"], self.target_image, True, src)
        out, cls = nets.discriminator(self.D, self.target_image, self.n_class)
        g_dis = losses.lsgan(out, 1.0)
        g_cls = losses.class_mse(cls, self._onehot(lab["target"]))
        g_cyc = (src - recon).abs().mean()
        errG = g_dis + g_cls * L["class"] + g_cyc * L["cycle"]
        errE = 0.0
        errE_report = g_cyc * L["cycle"]
        tr.update(g_dis=float(g_dis), g_cls=float(g_cls), g_cyc=float(g_cyc))
        _, mu, logvar, _, _ = enc_info
        if L["KL"] > 0:
            kl = losses.conventional_kl(mu, logvar)
            errE = errE + kl * L["KL"]
            errE_report = errE_report + kl * L["KL"]
            tr["kl"] = float(kl)
        if L["idt"] > 0:
            idt, _ = self._translate(lab["source"], src, True, src)
            g_idt = (src - idt).abs().mean()
            errG = errG + g_idt * L["idt"]
            errE_report = errE_report + g_idt * L["idt"]
            tr["g_idt"] = float(g_idt)
        if L["batch_KL"] > 0:
            bkl = losses.batch_kl(mu, self.n_batch)
            errE = errE + bkl * L["batch_KL"]
            errE_report = errE_report + bkl * L["batch_KL"]
            tr["bkl"] = float(bkl)
            if L["corr_enc"] > 0:
                corr = losses.corr_loss(mu.t())
                errE = errE + corr * L["corr_enc"]
                errE_report = errE_report + corr * L["corr_enc"]
                tr["corr"] = float(corr)
            if L["hist"] > 0:
                hist = self.hi.loss(mu)
                errE = errE + hist * L["hist"]
                errE_report = errE_report + hist * L["hist"]
                tr["hist"] = float(hist)
        tr["mu"] = mu.detach().clone()
        errG.backward(retain_graph=True)
        if torch.is_tensor(errE):
            errE.backward(retain_graph=True)
        if self.capture_grads:
            tr["gradG_p1"] = {k: p.grad.clone() for k, p in self.G.items()}
            tr["gradE_p1"] = {k: (None if p.grad is None else p.grad.clone()) for k, p in self.E.items()}
        self.optG.step()
        self.optE.step()
        # phase 2 (G only) --------------------------------------------------------
        _zero(self.G)
        _zero(self.E)
        _, t_enc, _, _, _ = self._enc(self.target_image)          # stale graph of target_image
        g_reg = (self.c_rand - t_enc).abs().mean()
        errG_ex = g_reg * L["reg"]
        tr["g_reg"] = float(g_reg)
        if L["idt_reg"] * L["idt"] > 0:
            idt_rand, info = self._translate(lab["source"], src, True, src)
            src_c = info[1]
            _, idt_enc, _, _, _ = self._enc(idt_rand)
            g_idt_reg = (src_c - idt_enc).abs().mean()
            errG_ex = errG_ex + g_idt_reg * L["idt_reg"] * (L["idt"] / L["cycle"])
            tr["g_idt_reg"] = float(g_idt_reg)
        errG_ex.backward()
        if self.capture_grads:
            tr["gradG_p2"] = {k: p.grad.clone() for k, p in self.G.items()}
        self.optG.step()
        return (errG + errG_ex).detach(), errE_report.detach()

    # -- train -------------------------------------------------------------------------
    def train(self, source_image, label, capture_grads=False):
        self.capture_grads = capture_grads
        self.trace = {}
        self.source, self.label = source_image, label
        errD0 = None
        for i in range(self.k):
            e = self.update_D()
            if i == 0:
                errD0 = e
        errG, errE = self.update_GandE()
        return [errG, errD0, errE]


def synthetic_batch(batch, size, n_class, seed=0):
    """CelebA-shaped synthetic batch: x~U(-1,1), target != source."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    src = torch.randint(0, n_class, (batch,), generator=g)
    tgt = (src + torch.randint(1, n_class, (batch,), generator=g)) % n_class
    return x, {"source": src, "target": tgt}


# =============================================================================================
# Conventional SingleGAN (config 1 of BASELINE.json: notebook 01, 64x64, 2 domains, CPU plumbing)
# =============================================================================================
SINGLEGAN_LBD = dict(**{"class": 0.0}, cycle=5.0, idt=5.0, reg=0.5, idt_reg=0.0, KL=0.1,
                     batch_KL=0.0, corr_enc=0.0, hist=0.0)      # util_notebook.py:12-16 preset "conventionalKL"


class SingleGANOracle:
    """CPU restatement of ``SingleGAN_training`` (pyfiles/util_notebook.py:76-417), per-domain discriminators
    (``singleD=False``): D is a LIST of SingleDiscriminator_original_multi parameter dicts, one per class, each
    seeing the boolean-masked sub-batch of its domain (:225-249, :286-293); the encoder is the CBIN-conditioned
    ``Encoder_original(x, onehot(label))`` (:170-171); phase 2's identity-regression path uses a RANDOM latent
    (:356).  ``update_D`` returns the LAST domain's errD (:251).  RNG order on the CPU default generator:
    k x randn(B, ndim); phase 1: normal_ (recon), normal_ (identity); phase 2: normal_ (E(target)), then if
    idt_reg*idt > 0: randn(B, ndim), normal_."""

    def __init__(self, PG, PD_list, PE, lbd, k, ref_label, ndim, classes, n_batch, encoded_feature="latent",
                 lr=(1e-4, 1e-4, 1e-4)):
        self.G, self.E = _leafify(PG), _leafify(PE)
        self.D = [_leafify(p) for p in PD_list]
        self.lbd, self.k = dict(lbd), int(k)
        self.ref_label, self.ndim, self.classes = ref_label, ndim, tuple(classes)
        self.n_batch, self.encoded_feature = n_batch, encoded_feature
        self.optG = Adam14(self.G.values(), lr[0])
        self.optD = [Adam14(d.values(), lr[1]) for d in self.D]
        self.optE = Adam14(self.E.values(), lr[2])
        self.hi = losses.HistogramImitation() if self.lbd["hist"] > 0 else None
        self.trace = {}

    def _onehot(self, label):
        return losses.one_hot_rows(label, self.ref_label)

    def _translate(self, target_label, src, use_encoder=False, ref_image=None):
        if use_encoder:
            info = list(nets.encoder_original(self.E, ref_image, self._onehot(target_label)))
            z = info[0] if self.encoded_feature == "latent" else info[1]
        else:
            z = torch.randn(src.shape[0], self.ndim)
            info = z
        c = torch.cat([self._onehot(target_label), z], 1)
        return nets.generator(self.G, src, c), info

    def update_D(self):
        self.target_image, self.c_rand = self._translate(self.label["target"], self.source)
        errD = None
        for i in self.classes:
            _zero(self.D[i])
            errD = 0
            real = self.source[self.label["source"] == i]
            if real.shape[0] != 0:
                errD = errD + losses.lsgan(nets.discriminator_original(self.D[i], real), 1.0)
            fake = self.target_image[self.label["target"] == i].detach()
            if fake.shape[0] != 0:
                errD = errD + losses.lsgan(nets.discriminator_original(self.D[i], fake), 0.0)
            errD.backward()
            self.optD[i].step()
            self.trace.setdefault("errD", []).append(float(errD))
        return errD.detach()

    def update_GandE(self):
        L, tr = self.lbd, self.trace
        _zero(self.G)
        _zero(self.E)
        src, lab = self.source, self.label
        recon, enc_info = self._translate(lab["source"], self.target_image, True, src)
        errG = 0
        for i in self.classes:
            fake = self.target_image[lab["target"] == i]
            if fake.shape[0] != 0:
                errG = errG + losses.lsgan(nets.discriminator_original(self.D[i], fake), 1.0) / len(self.classes)
        tr["g_dis"] = float(errG)
        g_cyc = (src - recon).abs().mean()
        errG = errG + g_cyc * L["cycle"]
        errE = 0
        errE_report = g_cyc * L["cycle"]
        tr["g_cyc"] = float(g_cyc)
        _, mu, logvar = enc_info
        if L["KL"] > 0:
            kl = losses.conventional_kl(mu, logvar)
            errE = errE + kl * L["KL"]
            errE_report = errE_report + kl * L["KL"]
            tr["kl"] = float(kl)
        if L["idt"] > 0:
            idt, _ = self._translate(lab["source"], src, True, src)
            g_idt = (src - idt).abs().mean()
            errG = errG + g_idt * L["idt"]
            errE_report = errE_report + g_idt * L["idt"]
            tr["g_idt"] = float(g_idt)
        if L["batch_KL"] > 0:
            bkl = losses.batch_kl(mu, self.n_batch)
            errE = errE + bkl * L["batch_KL"]
            errE_report = errE_report + bkl * L["batch_KL"]
            if L["corr_enc"] > 0:
                corr = losses.corr_loss(mu.t())
                errE = errE + corr * L["corr_enc"]
                errE_report = errE_report + corr * L["corr_enc"]
            if L["hist"] > 0:
                hist = self.hi.loss(mu)
                errE = errE + hist * L["hist"]
                errE_report = errE_report + hist * L["hist"]
        errG.backward(retain_graph=True)
        errE.backward(retain_graph=True)
        self.optG.step()
        self.optE.step()
        _zero(self.G)
        _zero(self.E)
        _, t_enc, _ = nets.encoder_original(self.E, self.target_image, self._onehot(lab["target"]))
        g_reg = (self.c_rand - t_enc).abs().mean()
        errG_ex = g_reg * L["reg"]
        tr["g_reg"] = float(g_reg)
        if L["idt_reg"] * L["idt"] > 0:
            idt_rand, src_c = self._translate(lab["source"], src, False)
            _, idt_enc, _ = nets.encoder_original(self.E, idt_rand, self._onehot(lab["source"]))
            g_idt_reg = (src_c - idt_enc).abs().mean()
            errG_ex = errG_ex + g_idt_reg * L["idt_reg"] * (L["idt"] / L["cycle"])
            tr["g_idt_reg"] = float(g_idt_reg)
        errG_ex.backward()
        self.optG.step()
        return (errG + errG_ex).detach(), errE_report.detach()

    def train(self, source_image, label):
        self.trace = {}
        self.source, self.label = source_image, label
        errD0 = None
        for i in range(self.k):
            e = self.update_D()
            if i == 0:
                errD0 = e
        errG, errE = self.update_GandE()
        return [errG, errD0, errE]

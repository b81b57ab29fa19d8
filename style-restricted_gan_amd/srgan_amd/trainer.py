"""``SRGAN_training``: the Style-Restricted GAN train step on the MI355X HIP path.

Same constructor / method signatures and return values as the reference class
(pyfiles/util_notebook.py:419-734); ``sg.train(source_image, label)`` drops into the loop of
``05-train_Style-Restricted_GAN.ipynb``.  What differs is HOW the step is executed:

* every network op and loss is a hand-written gfx950 kernel (``srgan_amd.ops``);
* result-preserving work savings of SURVEY.md Appendix B.2: the G forwards of the first k-1
  D-updates record no graph; D's weight gradients are not computed in phase 1 and E's not in
  phase 2 (they are discarded by the reference's ``zero_grad`` calls); the encoder trunk runs once
  for the two phase-1 ``E(source)`` calls; errG and errE are back-propagated in one pass;
* under ``torch.distributed`` (one process per GPU) gradients are averaged with bucketed RCCL
  all-reduce (``srgan_amd.dp``) and the batch-statistics losses see the all-gathered mu.

Semantics kept on purpose: stale ``target_image`` graph re-used after
``optG.step()`` with weights read at backward time; the no-op "unrolled" restore of D; corr/hist
nested under batch_KL; the n/(n-1) double correction with the constructor's batch size; errE returned
is the reporting sum; CPU-generator RNG order (k x randn, then normal_ x2, then normal_ x3).
"""
import contextlib
import gc
import os

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from . import _lib, dp, ops
from .losses import class_encode, get_domainloss_D, get_loss_D, histogram_imitation
from .model import SingleGenerator, _cpu_normal_like, host_to_device
from .optim import Adam

__all__ = ["SRGAN_training", "SingleGAN_training"]


@contextlib.contextmanager
def _frozen(params):
    """Temporarily mark parameters as not requiring grad (skips their weight-gradient kernels)."""
    flags = [(p, p.requires_grad) for p in params]
    for p, _ in flags:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p, f in flags:
            p.requires_grad_(f)


class SRGAN_training():
    """
    net            : [G, D, E] nn.Modules          opt : [optG, optD, optE] (None -> fused HIP Adam)
    criterion      : [criterion, criterion_class]   (nn.MSELoss in every reference notebook)
    lbd            : dict of loss weights: class, cycle, idt, reg, idt_reg, KL, batch_KL, corr_enc, hist
    unrolled_k     : number of D updates per G/E update
    device         : "cuda" / torch.device           ref_label : np.array (class_num, dim), usually one-hot
    batch_size     : GLOBAL batch size (n of the batch-KL correction)
    encoded_feature: "latent" or "mu"                ndim : style-code dimension
    """

    def __init__(self, net, opt, criterion, lbd, unrolled_k, device, ref_label,
                 batch_size=64, encoded_feature="latent", ndim=8):
        self.G, self.D, self.E = net[0].to(device), net[1].to(device), net[2].to(device)
        self.optG, self.optD, self.optE = opt[0], opt[1], opt[2]
        self.scheG, self.scheD, self.scheE = None, None, None
        self.criterion, self.criterion_class = criterion
        self.lbd = lbd
        self.k = unrolled_k
        self.device = device
        self.ref_label = ref_label
        self.n_batch = batch_size
        self.encoded_feature = encoded_feature
        self.ndim = ndim
        self.source_image = None
        self.target_image = None
        self.recon_image = None
        self.label = None
        self.c_rand = None
        self.enc_info = None
        self.target_cenc = None
        if lbd["hist"] > 0:
            self.hi = histogram_imitation(device)
        self.loss_terms = {}           # last step's individual loss values (device scalars)
        self._reducers = {}
        self.noise_fn = torch.randn    # style noise source (CPU default generator, as the reference); tests may inject
        self._graph = None             # hipGraph mode (enable_graph)
        self._g_active = False         # True while the step body reads its inputs from the static device buffers
        ref = np.asarray(ref_label)
        self._ref_is_onehot = ref.ndim == 2 and ref.shape[0] == ref.shape[1] and np.array_equal(ref, np.eye(ref.shape[0]))

    # ------------------------------------------------------------------------------------------
    def opt_sche_initialization(self, lr=[0.0001, 0.0001, 0.0001]):
        """Create the missing optimisers (Adam lr, betas=(0.5,0.999)) and ExponentialLR(gamma=0.95)
        schedulers for G, D, E   (util_notebook.py:484-508)."""
        lr_G, lr_D, lr_E = lr
        if self.optG is None:
            self.optG = Adam(self.G.parameters(), lr=lr_G, betas=(0.5, 0.999))
        self.scheG = optim.lr_scheduler.ExponentialLR(self.optG, gamma=0.95)
        if self.optD is None:
            self.optD = Adam(self.D.parameters(), lr=lr_D, betas=(0.5, 0.999))
        self.scheD = optim.lr_scheduler.ExponentialLR(self.optD, gamma=0.95)
        if self.optE is None:
            self.optE = Adam(self.E.parameters(), lr=lr_E, betas=(0.5, 0.999))
        self.scheE = optim.lr_scheduler.ExponentialLR(self.optE, gamma=0.95)
        return

    # ------------------------------------------------------------------------------------------
    # helpers
    def _cached(self, kind, which, make):
        """Per-label-tensor cache (labels are fixed during a step; class_encode forces a host sync)."""
        cache = self.__dict__.setdefault("_label_cache", {})
        lab = self.label[which]
        hit = cache.get((kind, which))
        if hit is None or hit[0] is not lab:
            hit = cache[(kind, which)] = (lab, make(lab))
        return hit[1]

    def _onehot(self, which):
        if self._g_active:
            return self._graph.onehot(which)
        return self._cached("onehot", which, lambda lab: class_encode(lab, self.device, self.ref_label))

    def _label_dev(self, which):
        if self._g_active:
            return self._graph.lab[which]
        return self._cached("index", which,
                            lambda lab: host_to_device(torch.as_tensor(lab).to(dtype=torch.int64), self.device))

    def _noise(self, kind, nb):
        """Next style / reparametrisation noise [nb, ndim] of the step, on the device.  All of it comes from the CPU default
        generator in the reference's order: k x ``randn`` (util_notebook.py:554), then ``normal_`` per
        encoder call (model.py:461).  Graph mode: the draws were made up front (same order) and staged; this hands out the
        next slice of the static buffer."""
        if self._g_active:
            return self._graph.next_noise()
        cpu = self.noise_fn(nb, self.ndim) if kind == "randn" else torch.FloatTensor(nb, self.ndim).normal_()
        return host_to_device(cpu, self.device)

    def _noise_kinds(self):
        """The step's noise draws in order (fused execution paths)."""
        L = self.lbd
        return (["randn"] * self.k + ["normal"] * (2 if L["idt"] > 0 else 1)
                + ["normal"] * (3 if L["idt_reg"] * L["idt"] > 0 else 1))

    @staticmethod
    def _opt_params(opt):
        return [p for g in opt.param_groups for p in g["params"]]

    def _step(self, opt):
        """optimiser step + one launch that re-packs the cached conv operands of the weights it just changed"""
        opt.step()
        ops.refresh_packed(self._opt_params(opt))

    def _reducer(self, name, opt):
        red = self._reducers.get(name)
        if red is None:
            red = self._reducers[name] = dp.GradReducer(self._opt_params(opt))
            ops._sink_alloc = dp.grad_slot      # the weight-gradient kernels of a recorded step write into the bucket slices
        return red

    def _reduce_arm(self, name, opt):
        """before the backward whose gradients `opt` will consume: buckets are all-reduced as they complete"""
        if dp.is_distributed():
            self._reducer(name, opt).arm()

    def _reduce_start(self, name, opt):
        if not dp.is_distributed():
            return None
        red = self._reducer(name, opt)
        red.start()
        return red

    def _encode(self, image, feat=None, noise=None):
        """E(image) -> 5-list; with a precomputed trunk feature only the heads run.  The reparametrisation noise is
        drawn here (CPU default generator, as model.py:461) unless the caller pre-drew it to fix the RNG order."""
        E = dp.unwrap(self.E)
        if feat is None and noise is not None and hasattr(E, "features"):
            feat = E.features(image)
        if feat is not None and hasattr(E, "features"):
            mu, logvar = E.fcmean(feat), E.fcvar(feat)
            eps = noise if noise is not None else self._noise("normal", mu.shape[0])
            return [E.reparam_with(mu, logvar, eps), mu, logvar, E.fcclass(feat), None]
        return list(self.E(image))

    def _fused_paths(self):
        """The batched / fused execution needs this package's own modules (feature / logit entry points)."""
        return (hasattr(dp.unwrap(self.D), "forward_logits") and self._ref_is_onehot
                and hasattr(dp.unwrap(self.E), "features") and self._class_is_mse())

    def _class_is_mse(self):
        """The fused class-loss kernel is softmax + MSE (nn.MSELoss in every reference notebook, 05-train cell 13); any other
        criterion_class takes the generic path through the criterion itself."""
        c = self.criterion_class
        return isinstance(c, nn.MSELoss) and getattr(c, "reduction", "mean") == "mean"

    def _d_losses(self, image, gan_target, class_which, want_class):
        """LSGAN (+ class MSE) of D(image) through the fused head/loss kernels."""
        D = dp.unwrap(self.D)
        if hasattr(D, "forward_logits") and self._ref_is_onehot and self._class_is_mse():
            outs, logits = D.forward_logits(image)
            gan = get_loss_D(outs, gan_target, self.criterion, self.device)
            cls = None
            if want_class:
                lab = self._label_dev(class_which)
                w = 1.0 / len(logits)
                cls = 0.0
                for z in logits:
                    cls = cls + ops.softmax_mse(z, lab, w)[0]
            return gan, cls
        outs, probs = self.D(image)
        gan = get_loss_D(outs, gan_target, self.criterion, self.device)
        cls = get_domainloss_D(probs, self._onehot(class_which), self.criterion_class) if want_class else None
        return gan, cls

    # ------------------------------------------------------------------------------------------
    def G_transformation(self, target_label, source_image, encoder=False, ref_image=None, _enc_info=None):
        """Translate ``source_image`` to ``target_label`` with a style code from E(ref_image) (encoder=True) or
        N(0, I) noise drawn on the CPU default generator   (util_notebook.py:510-561).
        Returns (image, info): info = [latent, mu, logvar, class_output, None] or the random latent."""
        if encoder:
            info = _enc_info if _enc_info is not None else self._encode(ref_image)
            latent, mu = info[0], info[1]
            if self.encoded_feature == "latent":
                latent_vector = latent
            elif self.encoded_feature == "mu":
                latent_vector = mu
        else:
            latent_vector = self._noise("randn", source_image.shape[0])
            info = latent_vector
        if isinstance(target_label, str):
            class_vector = self._onehot(target_label)
        else:
            class_vector = class_encode(target_label, self.device, self.ref_label)
        class_vector = torch.cat([class_vector, latent_vector], 1)
        target_image = self.G(source_image, class_vector)
        return target_image, info

    # ------------------------------------------------------------------------------------------
    def update_D(self, _fake=None, _next_fake=None, _defer=False):
        """One discriminator update (util_notebook.py:563-594); returns errD.
        Data parallel: ``_fake`` hands in a pre-computed translation, ``_next_fake`` is work to run UNDER this update's gradient
        all-reduce (the next translation: G does not change during the discriminator loop), ``_defer`` leaves the wait for the
        all-reduce and the optimiser step to ``_finish_D()`` so that the caller can put more independent work in between."""
        self._finish_D()
        self.D.zero_grad()
        if _fake is None:
            self.target_image, self.c_rand = self.G_transformation("target", self.source_image, False)
        else:
            self.target_image, self.c_rand = _fake

        if self._fused_paths():
            # real and fake halves through D as ONE batch (per-sample network: exact; 2x the rows per GEMM launch)
            B = self.source_image.shape[0]
            outs, logits = dp.unwrap(self.D).forward_logits(ops.cat_batch([self.source_image, self.target_image.detach()]))
            # LSGAN(real, 1) + class MSE * lbd + LSGAN(fake, 0) over both scales, values and gradients: one launch
            errD, parts = ops.d_losses(outs, logits, self._label_dev("source"), B, 1., 0., self.lbd["class"])
            errD_real, errD_class, errD_fake = parts[0], parts[1], parts[2]
        else:
            errD_real, errD_class = self._d_losses(self.source_image, 1., "source", True)
            errD_fake, _ = self._d_losses(self.target_image.detach(), 0., None, False)
            errD = errD_real + errD_class * self.lbd["class"] + errD_fake
        self._reduce_arm("D", self.optD)
        with ops.fused_param_grads(not dp.hooks_need_live_grads(), self.device):
            errD.backward()
        self._d_pending = self._reduce_start("D", self.optD) or True
        dp.launch_pending()                                         # recorded step: the all-reduces start here ...
        nxt = _next_fake() if _next_fake is not None else None      # ... and this runs under them
        if not _defer:
            self._finish_D()
        self.loss_terms.update(errD_real=errD_real.detach(), errD_class=errD_class.detach(), errD_fake=errD_fake.detach())
        if _next_fake is not None:
            return errD, nxt
        return errD

    def _finish_D(self):
        """Wait for the discriminator's gradient all-reduce (if any) and take its optimiser step, if an ``update_D`` left them."""
        red = self.__dict__.get("_d_pending")
        if red is None:
            return
        self._d_pending = None
        if red is not True:
            red.finish()
        self._step(self.optD)

    # ------------------------------------------------------------------------------------------
    def update_GandE(self):
        """Generator + encoder update in two phases (util_notebook.py:596-694); returns [errG, errE_output]."""
        L = self.lbd
        self.G.zero_grad()
        self.E.zero_grad()
        E = dp.unwrap(self.E)
        src = self.source_image
        ws = dp.world_size()

        # ---------------- phase 1: G and E ----------------
        optE_ids = {id(p) for p in self._opt_params(self.optE)}
        e_unused = [p for p in self.E.parameters() if p.requires_grad and id(p) not in optE_ids]
        with _frozen(e_unused):
            feat = E.features(src) if hasattr(E, "features") else None
            source_enc_info = self._encode(src, feat)
            pair = feat is not None and L["idt"] > 0
            # Recorded data-parallel step (round 4): phase 1 reaches the kept target_image graph LAST in its backward (autograd
            # runs the youngest nodes first and that graph was built in the last discriminator update) -- after E's gradients are
            # final.  The pass is cut there: phase 1 reads target_image through a detached alias, the first backward stops at the
            # alias, E's all-reduce is started on the communication stream, and the stale graph is back-propagated from the
            # alias's gradient UNDER it (the same two contributions summed in the same order: the alias's gradient is what the
            # engine's input buffer would hold).
            cut = pair and dp.recording() and self.target_image.requires_grad
            t_img = self.target_image.detach().requires_grad_(True) if cut else self.target_image
            if pair:
                # the reference calls E(source) a second time for the identity path: same weights, same input -> same
                # mu; only the reparametrisation noise is drawn again (keeps the CPU RNG sequence identical).  The
                # reconstruction G(target_image, c) and the identity G(source, c') then run as ONE batch through G.
                idt_info = self._encode(src, feat)
                pick = 0 if self.encoded_feature == "latent" else 1
                oh = self._onehot("source")
                c_pair = torch.cat([torch.cat([oh, source_enc_info[pick]], 1), torch.cat([oh, idt_info[pick]], 1)], 0)
                both = self.G(ops.cat_batch([t_img, src]), c_pair)
                nb = src.shape[0]
                recon_image, identity_image = both[:nb], both[nb:]
            else:
                recon_image, _ = self.G_transformation("source", self.target_image, True, src, _enc_info=source_enc_info)
            self._finish_D()                               # (data parallel: the last D update's all-reduce ran under the passes above)
            fused = self._fused_paths()
            # loss terms as (device scalar, weight) lists: the optimised sums and the two reported sums are each ONE launch
            # (ops.lincomb) instead of one launch per python-level `+` / `*`
            g_terms, e_terms, rep_terms = [], [], []
            with _frozen(list(self.D.parameters())):       # D's weight grads would be discarded
                if fused:
                    outs, logits = dp.unwrap(self.D).forward_logits(t_img)
                    d_total, parts = ops.d_losses(outs, logits, self._label_dev("target"), self.target_image.shape[0], 1., 0., L["class"])
                    errG_dis, errG_class = parts[0], parts[1]
                    g_terms.append((d_total, 1.0))
                else:
                    errG_dis, errG_class = self._d_losses(t_img, 1., "target", True)
                    g_terms += [(errG_dis, 1.0), (errG_class, L["class"])]
            errG_cycle = ops.l1_mean(src, recon_image, 1.0)
            g_terms.append((errG_cycle, L["cycle"]))
            rep_terms.append((errG_cycle, L["cycle"]))
            terms = dict(errG_dis=errG_dis, errG_class=errG_class, errG_cycle=errG_cycle)

            _, mu, logvar, _, _ = source_enc_info
            if L["KL"] > 0:
                # a SUM over the rows of the batch (util_notebook.py:630-634): under data parallelism each rank holds the
                # sum over ITS rows and the gradient all-reduce averages, so the optimised term is pre-scaled by the world
                # size (as the batch-statistics losses below); the reported value stays the local sum
                errE_KL = ops.kl_normal(mu, logvar)
                e_terms.append((errE_KL, L["KL"] * ws if ws > 1 else L["KL"]))
                rep_terms.append((errE_KL, L["KL"]))
                terms["errE_KL"] = errE_KL

            if L["idt"] > 0:
                if not pair:
                    identity_image, _ = self.G_transformation("source", src, True, src)
                errG_idt = ops.l1_mean(src, identity_image, 1.0)
                g_terms.append((errG_idt, L["idt"]))
                rep_terms.append((errG_idt, L["idt"]))
                terms["errG_idt"] = errG_idt

            if L["batch_KL"] > 0:
                w_corr = L["corr_enc"] if L["corr_enc"] > 0 else 0.0
                w_hist = L["hist"] if L["hist"] > 0 else 0.0
                target = self.hi.target if w_hist > 0 else torch.full((50,), 0.02, device=mu.device)
                mu_all = dp.all_gather_rows(mu)
                total, parts, _ = ops.latent_losses(mu_all, self.n_batch, target, L["batch_KL"], w_corr, w_hist)
                # every rank evaluates the global-batch loss; its local rows' gradient must be SUMMED over ranks
                # while the all-reduce averages -> pre-scale by the world size
                e_terms.append((total, float(ws) if ws > 1 else 1.0))
                rep_terms.append((total, 1.0))
                terms.update(errE_bKL=parts[0], errE_corr=parts[1], errE_hist=parts[2])

            total_p1 = ops.lincomb(g_terms + e_terms)
            with torch.no_grad():
                errG = ops.lincomb(g_terms)
                errE_output = ops.lincomb(rep_terms)
            self._reduce_arm("G", self.optG)
            self._reduce_arm("E", self.optE)
            # the generator's weights are reached twice in this pass (reconstruction / identity graph and the kept target_image
            # graph): their weight-gradient kernels add the second contribution themselves (ops.fused_param_grads)
            with ops.fused_param_grads(not dp.hooks_need_live_grads(), self.device):
                total_p1.backward(retain_graph=True)     # target_image's graph is needed again in phase 2
            redE = None
            if cut:
                # E's gradients are final: flatten its buckets and start their all-reduce (a cut of the recording), then run the
                # stale-graph generator backward under it; the generator's parameters take their second contribution on top of
                # the first (ops.fused_param_grads(seed=True): the weight-gradient kernels add into p.grad)
                redE = self._reduce_start("E", self.optE)
                dp.launch_pending()
                with ops.fused_param_grads(True, self.device, seed=True):
                    self.target_image.backward(t_img.grad, retain_graph=True)
        # Data parallel: E's buckets go out first, G's behind them on the same communication stream; E's optimiser step, its
        # repack and phase 2's E(source) forward (which needs the new E, not the new G) then run UNDER G's all-reduce.  The two
        # Adam steps touch disjoint parameters, so their order does not matter.
        if redE is None:
            redE = self._reduce_start("E", self.optE)
        redG = self._reduce_start("G", self.optG)
        dp.launch_pending(then_wait=redE)
        if redE is not None:
            redE.finish()
        self._step(self.optE)
        do_idt_reg = L["idt_reg"] * L["idt"] > 0
        early_info = None
        if redG is not None and do_idt_reg and self._fused_paths():
            nb = src.shape[0]
            early_noise = [self._noise("normal", nb) for _ in range(3)]     # reference order: E(target_image), E(source), E(idt_random_image)
            with _frozen(list(self.E.parameters())), torch.no_grad():
                early_info = self._encode(src, noise=early_noise[1])
        if redG is not None:
            redG.finish()
        self._step(self.optG)

        # ---------------- phase 2: G only ----------------
        self.G.zero_grad()
        self.E.zero_grad()
        with _frozen(list(self.E.parameters())):           # only optG steps: E's weight grads are discarded
            if do_idt_reg and self._fused_paths():
                # reference order of the noise draws: E(target_image), E(source), E(idt_random_image)
                nb = src.shape[0]
                if early_info is not None:                  # (data parallel: drawn and encoded under G's all-reduce, above)
                    (n1, n2, n3), info = early_noise, early_info
                else:
                    n1, n2, n3 = (self._noise("normal", nb) for _ in range(3))
                    with torch.no_grad():                   # its gradient only reaches E's parameters
                        info = self._encode(src, noise=n2)
                idt_random_image, info = self.G_transformation("source", src, True, src, _enc_info=info)
                feats = E.features(ops.cat_batch([self.target_image, idt_random_image]))    # ONE batch through E
                target_cenc = self._encode(None, feats[:nb], n1)[1]
                idt_cenc_rand = self._encode(None, feats[nb:], n3)[1]
                errG_reg = ops.l1_mean(self.c_rand, target_cenc, 1.0)
                errG_idt_reg = ops.l1_mean(info[1], idt_cenc_rand, 1.0)
                errG_ex = ops.lincomb([(errG_reg, L["reg"]), (errG_idt_reg, L["idt_reg"] * (L["idt"] / L["cycle"]))])
                terms["errG_reg"], terms["errG_idt_reg"] = errG_reg, errG_idt_reg
            else:
                _, target_cenc, _, _, _ = self._encode(self.target_image)
                errG_reg = ops.l1_mean(self.c_rand, target_cenc, 1.0)
                errG_ex = errG_reg * L["reg"]
                terms["errG_reg"] = errG_reg
                if do_idt_reg:
                    with torch.no_grad():                   # its gradient only reaches E's parameters
                        info = self._encode(src)
                    idt_random_image, info = self.G_transformation("source", src, True, src, _enc_info=info)
                    source_c_rand = info[1]
                    _, idt_cenc_rand, _, _, _ = self._encode(idt_random_image)
                    errG_idt_reg = ops.l1_mean(source_c_rand, idt_cenc_rand, 1.0)
                    errG_ex = errG_ex + errG_idt_reg * (L["idt_reg"] * (L["idt"] / L["cycle"]))
                    terms["errG_idt_reg"] = errG_idt_reg
            self._reduce_arm("G", self.optG)
            with ops.fused_param_grads(not dp.hooks_need_live_grads(), self.device):
                errG_ex.backward()
        redG = self._reduce_start("G", self.optG)
        if redG is not None:
            redG.finish()
        self._step(self.optG)

        self.recon_image = recon_image.detach()
        self.loss_terms.update({k: v.detach() for k, v in terms.items()})
        with torch.no_grad():
            errG = ops.lincomb([(errG, 1.0), (errG_ex, 1.0)])
        return [errG, errE_output.detach()]

    # ------------------------------------------------------------------------------------------
    def UnrolledUpdate(self):
        """k discriminator updates then one G/E update (util_notebook.py:696-728).  The reference then
        "restores" D from ``paramD = self.D.state_dict()``, which aliases the live tensors -- a no-op that
        is not replayed here.  Returns [errorG, errorD(first iteration), errorE]."""
        errorD = None
        src, k = self.source_image, self.k
        nb = src.shape[0]
        # G does not change during the k discriminator updates, so all k translations are computed up front, with the
        # noise drawn in the reference's order (k x randn(B, ndim) on the CPU generator).  Only the LAST translation's
        # graph is ever back-propagated (phases 1 and 2); the first k-1 run as ONE no-grad batch of (k-1)*B images.
        noises = [self._noise("randn", nb) for _ in range(k)]
        oh = self._onehot("target")
        if dp.is_distributed():
            # Data parallel: every discriminator update ends in a gradient all-reduce, and the only work of the loop that does
            # not depend on it is the NEXT translation (G is fixed during the loop).  So the translations run one at a time, each
            # under the previous update's all-reduce; the last update's all-reduce runs under the encoder / generator forward
            # passes of phase 1 (update_GandE calls _finish_D() right before it needs the updated discriminator).
            def translate(i):
                if i < k - 1:
                    with torch.no_grad():
                        return self.G(src, torch.cat([oh, noises[i]], 1)), noises[i]
                return self.G(src, torch.cat([oh, noises[i]], 1)), noises[i]       # the graph that phases 1 and 2 back-propagate

            fake = translate(0)
            for i in range(k):
                nxt = (lambda j=i + 1: translate(j)) if i + 1 < k else None
                out = self.update_D(_fake=fake, _next_fake=nxt, _defer=(i == k - 1))
                errD, fake = out if nxt is not None else (out, None)
                if i == 0:
                    errorD = errD.detach()
            errorG, errorE = self.update_GandE()
            return [errorG, errorD, errorE]
        fakes = []
        if k > 1 and isinstance(dp.unwrap(self.G), SingleGenerator):     # per-sample network: batching is exact
            # the kernels address an activation with 32-bit byte offsets (< 4 GiB per tensor): translations are batched in
            # groups whose widest activation (the first conv's nch planes at full resolution) stays under that, e.g.
            # 256x256, B=64, k=5 runs as two groups of 2*B images instead of one of 4*B
            Gm = dp.unwrap(self.G)
            widest = max(int(Gm.down_convs[0].weight.shape[0]), 3) * src.shape[2] * src.shape[3] * 4
            per_group = max(1, min(k - 1, (((1 << 32) - (1 << 26)) // widest) // nb))
            with torch.no_grad():
                for lo in range(0, k - 1, per_group):
                    zs = noises[lo:min(lo + per_group, k - 1)]
                    c_all = torch.cat([torch.cat([oh, z], 1) for z in zs], 0)
                    imgs = self.G(ops.cat_batch([src] * len(zs)) if len(zs) > 1 else src, c_all)
                    fakes += [(imgs[i * nb:(i + 1) * nb], z) for i, z in enumerate(zs)]
        elif k > 1:
            with torch.no_grad():
                fakes = [(self.G(src, torch.cat([oh, z], 1)), z) for z in noises[:-1]]
        fakes.append((self.G(src, torch.cat([oh, noises[-1]], 1)), noises[-1]))
        for i in range(k):
            errD = self.update_D(_fake=fakes[i])
            fakes[i] = None
            if i == 0:
                errorD = errD.detach()
        errorG, errorE = self.update_GandE()
        return [errorG, errorD, errorE]

    def train(self, source_image, label):
        self.label = label
        self.loss_terms = {}
        g = self._graph
        if g is not None and g.accepts(source_image, label):
            return g.run(source_image, label)
        if g is not None:
            g.before_eager_step()
        self._g_active = False
        self.source_image = ops.to_nhwc(source_image)
        with ops.pack_cache():        # weights only change at self._step() inside this scope
            error = self.UnrolledUpdate()
        if g is not None:
            g.note_eager_step(source_image, label)
        return error

    # ------------------------------------------------------------------------------------------
    # hipGraph mode (BASELINE configs[4]: "hipGraph step"; the step captured is util_notebook.py:696-734)
    def enable_graph(self):
        """From now on ``train()`` replays the whole step -- k discriminator updates, both generator / encoder phases, the
        optimiser steps and weight repacks, ~1900 launches -- as ONE captured hipGraph.  The first call with a given input
        shape still runs eagerly (it creates the optimiser state, packed operands and workspaces a capture must not
        re-create), the second captures and replays, later ones only stage the step's inputs (image batch, labels, the
        CPU-generator noise drawn in the reference's order) into static device buffers and launch the graph.  Inputs of
        another shape (an epoch's last partial batch) run eagerly.  Results are bit-identical to eager execution.
        Under a process group the recording is cut where a collective starts and where its result is needed (2k + 6 graph
        segments; between them the all-reduces are enqueued on the communication stream and run UNDER the following segment --
        the next translation, phase 1's forward passes, E's optimiser step: ``_Recording``, ``dp.launch_pending``); if any rank
        fails to record, all ranks drop graph mode together and continue eagerly
        (``SRGAN_DP_GRAPH=0`` refuses graph mode under a process group altogether)."""
        self._graph = _StepGraph(self)
        return self

    def disable_graph(self):
        if self._graph is not None:
            self._graph.before_eager_step()
        self._graph = None
        self._g_active = False

    @property
    def graph_active(self):
        return self._graph is not None and self._graph.graph is not None


class _StepGraph:
    """Static inputs, capture and replay of one ``SRGAN_training`` step (see ``SRGAN_training.enable_graph``)."""

    def __init__(self, sg):
        self.sg = sg
        reason = self.unsupported_reason()
        if reason:
            raise NotImplementedError("SRGAN_training.enable_graph: " + reason)
        self.key = None          # input signature the eager warm-up ran with
        self.graph = None
        self.x = self.noise = self.out = None
        self.lab = {}
        self._onehot = {}
        self._noise_i = 0
        self._versions = None
        self.table = None
        self._epoch = -1
        self._keep = None
        self._baked = None       # fingerprint of everything the recording holds by value or by pointer (see _fingerprint)
        self._steps = None       # host-side optimiser step counts the device records stand at after the last replay
        self._graph_ran = False  # the previous step ran on the capture stream (recorded or replayed)
        self._force_segmented = False   # a single-graph recording failed on some rank: record segments from here on
        self._last_single = False       # the recording being made / made last captured its collectives (one graph)

    fault_hook = None            # tests: callable(stage) with stage in {"before", "inside"}; raising makes the recording fail

    def unsupported_reason(self):
        sg = self.sg
        if not sg._fused_paths():
            return "needs this package's own G / D / E modules, a one-hot ref_label and nn.MSELoss criteria"
        for name in ("optG", "optD", "optE"):
            if not isinstance(getattr(sg, name), Adam):
                return f"{name} is not srgan_amd.optim.Adam (call opt_sche_initialization(), or pass that class)"
        if dp.is_distributed() and os.environ.get("SRGAN_DP_GRAPH") == "0":
            return "SRGAN_DP_GRAPH=0: data-parallel steps run eagerly (hook-driven all-reduce under the backward)"
        return None

    @staticmethod
    def _key(source_image, label):
        def dev(t):
            return t.device.type if torch.is_tensor(t) else "host"
        return (tuple(source_image.shape), dev(label["source"]), dev(label["target"]))

    def _fingerprint(self):
        """Everything a recording bakes in besides the input shape and the buffers ``ops.structure_epoch()`` watches: the
        optimiser objects, their hyper-parameters and moment tensors (``load_state_dict`` replaces them), which parameters
        take gradients (``freeze_melt`` / ``requires_grad_``), k, the loss weights, the criteria, the style-code source, the
        label table.  A replay with any of these changed would silently keep training the OLD configuration."""
        sg = self.sg
        fp = [sg.k, sg.n_batch, sg.ndim, sg.encoded_feature, tuple(sorted((k, float(v)) for k, v in sg.lbd.items())),
              type(sg.criterion), type(sg.criterion_class), np.asarray(sg.ref_label, dtype=np.float64).tobytes()]
        for opt in (sg.optG, sg.optD, sg.optE):
            fp.append(id(opt))
            for g in getattr(opt, "param_groups", ()):
                fp.append((tuple(g["betas"]), float(g["eps"])))
                for p in g["params"]:
                    st = opt.state.get(p) or {}
                    m, v = st.get("exp_avg"), st.get("exp_avg_sq")
                    fp.append((id(p), m.data_ptr() if m is not None else 0, v.data_ptr() if v is not None else 0))
        fp.append(tuple((id(p), p.data_ptr(), p.requires_grad) for p in self._all_params()))
        return tuple(fp)

    def _opt_steps(self):
        sg = self.sg
        return [[int(opt.state[p]["step"]) if len(opt.state.get(p) or {}) else -1 for g in opt.param_groups for p in g["params"]]
                for opt in (sg.optG, sg.optD, sg.optE)]

    def _drop(self):
        self.graph, self.key, self._keep, self._baked, self._steps = None, None, None, None, None
        self.x = self.noise = self.out = None

    def _local_mode(self, source_image, label):
        """2: replay the recording, 1: record (the eager warm-up of this input shape has run), 0: eager."""
        if self.graph is not None:
            stale = self._epoch != ops.structure_epoch()
            # a buffer the recording points at was replaced (compute-mode switch, invalidate_packed, optimiser re-seeded), or
            # something it baked in changed (optimiser state loaded, requires_grad toggled, lbd / k edited), or the host-side
            # step counts are not where the last replay left them (load_state_dict): forget the graph; this step runs eagerly
            # with the new state (which also re-seeds the device-side Adam records), the next one records again
            if not stale and (self._baked != self._fingerprint() or self._steps != self._opt_steps()):
                stale = True
                reason = self.unsupported_reason()
                if reason:
                    raise NotImplementedError("SRGAN_training graph mode: " + reason)
            if stale:
                self._drop()
        if self.key is None or self._key(source_image, label) != self.key or not source_image.is_cuda:
            return 0
        return 2 if self.graph is not None else 1

    def accepts(self, source_image, label):
        """Replay / record this step (True) or run it eagerly (False).  Under a process group the decision is COLLECTIVE (one
        MIN all-reduce of the local answer on the host-side control group): a rank replaying index-ordered flat buckets beside
        a rank running the hook-ordered eager step -- or recording, with its extra agreement all-reduce -- would issue different
        collectives.  If any rank has to record, every rank records again; if any rank runs eagerly, all do."""
        failure = None
        try:
            mode = self._local_mode(source_image, label)
        except NotImplementedError as e:
            # A rank that can no longer take graph mode must still join the agreement below -- raising before the collective would
            # leave the other ranks blocked in it.  It votes -1, a value no healthy rank uses (a vote of 0 read as
            # "eager" on the peers, which then entered the eager step and blocked in its first gradient all-reduce until the dead
            # rank's connection timed out); on a negative minimum EVERY rank raises, at the same point of its program.
            failure, mode = e, -1
            if not dp.is_distributed():
                raise
        if dp.is_distributed() and mode == 2 and failure is None and getattr(self.graph, "single", False):
            # Single-graph data-parallel step (round 6): the collectives are INSIDE the recording, so a replay is one launch with
            # no host-side exchange at all -- graph-or-eager was decided collectively when the step was recorded, and every
            # condition that drops a recording (_local_mode) is a program-state change an SPMD program makes on every rank at the
            # same step.  A rank that has to re-record alone votes below while its peers replay: it blocks there until the control
            # group's timeout and raises (dp.control_group); the segmented mode keeps the per-step vote.
            return True
        if dp.is_distributed():
            agreed = dp.all_min(mode)
            if failure is not None:
                raise failure
            if agreed < 0:
                raise RuntimeError("SRGAN_training graph mode: another rank cannot take this step (see its own error); no rank "
                                   "runs it")
            if agreed < 2 and self.graph is not None:
                if agreed == 1:
                    key = self.key
                    self._drop()           # another rank re-records: record again with it (same segment structure)
                    self.key = key
            mode = agreed
        return mode >= 1

    def before_eager_step(self):
        """An eager step follows a recorded / replayed one: the autograd graph the step keeps on purpose (``target_image`` is
        back-propagated twice) pins AccumulateGrad nodes made on the CAPTURE stream; re-used by an eager backward they would
        synchronise across streams on every gradient (PyTorch warns about the mismatch).  Drop those graphs first."""
        if not self._graph_ran:
            return
        self._graph_ran = False
        sg = self.sg
        sg.target_image = sg.recon_image = sg.c_rand = sg.enc_info = sg.target_cenc = None
        sg.loss_terms = {}
        for opt in (sg.optG, sg.optD, sg.optE):
            opt._keep_alive = None
        gc.collect()

    def note_eager_step(self, source_image, label):
        if self.key is None and source_image.is_cuda:
            self.key = self._key(source_image, label)
        # the eager step left every packed operand fresh: writes from here on (load_state_dict, copy_) must be noticed
        self._versions = sum(p._version for p in self._all_params())
        if self.graph is not None:      # an eager step of this trainer (another batch shape) moved the optimisers legitimately
            self._steps = self._opt_steps()

    # -- what the captured body reads -----------------------------------------------------------------------------
    def onehot(self, which):
        hit = self._onehot.get(which)
        if hit is None:              # once per step (inside the graph): rows of the device copy of ref_label
            hit = self._onehot[which] = self.table.index_select(0, self.lab[which])
        return hit

    def next_noise(self):
        t = self.noise[self._noise_i]
        self._noise_i += 1
        return t

    # -- staging (outside the graph, stream-ordered before the replay) ---------------------------------------------
    def _stage(self, source_image, label):
        sg = self.sg
        dev = source_image.device
        nb = source_image.shape[0]
        kinds = sg._noise_kinds()
        draws = [sg.noise_fn(nb, sg.ndim) if kd == "randn" else torch.FloatTensor(nb, sg.ndim).normal_() for kd in kinds]
        host_noise = torch.stack([d.to(torch.float32) for d in draws]).pin_memory()
        x = ops.to_nhwc(source_image)
        if self.x is None:
            self.x = torch.empty_like(x)
            self.noise = torch.empty(len(kinds), nb, sg.ndim, dtype=torch.float32, device=dev)
            self.lab = {w: torch.empty(nb, dtype=torch.int64, device=dev) for w in ("source", "target")}
            self.table = torch.tensor(np.asarray(sg.ref_label), dtype=torch.float32).to(dev)
        self.x.copy_(x)
        self.noise.copy_(host_noise, non_blocking=True)
        for w in ("source", "target"):
            lab = torch.as_tensor(label[w]).detach().to(dtype=torch.int64)
            if not lab.is_cuda:
                lab = lab.pin_memory()
            self.lab[w].copy_(lab.view(-1), non_blocking=True)

    def _all_params(self):
        sg = self.sg
        return [p for net in (sg.G, sg.D, sg.E) for p in net.parameters()]

    def _guard_packed(self):
        """Weights written behind the graph's back (load_state_dict, copy_: they bump the autograd version, the raw-pointer
        Adam does not) would leave the captured step multiplying stale packed operands: re-pack before replaying."""
        v = sum(p._version for p in self._all_params())
        if self._versions is not None and v != self._versions:
            ops.refresh_packed(self._all_params(), force=True)
        self._versions = v

    def _capture(self):
        sg = self.sg
        lib_prof_off()
        # Every autograd graph of the eager warm-up must be gone before recording: a parameter's AccumulateGrad node lives as
        # long as any graph that points at it, and remembers the stream it was created on.  The step keeps its last graph
        # alive on purpose (target_image is back-propagated twice), so the nodes of the eager step -- made on the caller's
        # stream -- would be re-used by the recorded step and pull that stream into the capture as an unjoined fork.
        sg.target_image = sg.recon_image = sg.c_rand = None
        sg.loss_terms = {}
        sg.__dict__.get("_label_cache", {}).clear()
        for opt in (sg.optG, sg.optD, sg.optE):
            opt._keep_alive = None
        gc.collect()
        hook = type(self).fault_hook
        if hook is not None:
            hook("before")
        ops.mark_singles_stale()          # their pack launches belong to the recording (ops.refresh_packed)
        g = _Recording(self.x.device, single=not self._force_segmented)
        self._last_single = g.single
        epoch0 = ops.structure_epoch()    # a buffer replaced DURING the recording (the weight-gradient arena grown
        sg._g_active = True               # between two segments) would leave earlier segments pointing at freed memory
        self._noise_i, self._onehot = 0, {}
        sg.source_image = self.x
        try:
            with g:
                with ops.pack_cache(refresh_on_entry=False):
                    err = sg.UnrolledUpdate()
                if hook is not None:
                    hook("inside")
                self.out = torch.stack([e.detach().reshape(()) for e in err])
        finally:
            sg._g_active = False
        if self._noise_i != self.noise.shape[0]:
            raise RuntimeError(f"captured step consumed {self._noise_i} noise draws, staged {self.noise.shape[0]}")
        if ops.structure_epoch() != epoch0:
            raise RuntimeError("a workspace the step points at was replaced while the step was being recorded "
                               "(ops.structure_epoch moved): the recording is dropped")
        self.graph = g
        self.terms = dict(sg.loss_terms)
        self._epoch = ops.structure_epoch()
        self._keep = ops.graph_keepalive() + [t for opt in (sg.optG, sg.optD, sg.optE) for t in opt.graph_keepalive()]
        self._baked = self._fingerprint()

    def _abandon(self, err, snap, keep_graph_mode=False):
        """Some rank could not record the step: every rank drops graph mode together.  THIS step still runs -- eagerly, from the
        inputs already staged for the recording (same image batch, labels and pre-drawn noise, so the CPU generator is where an
        eager run would have left it) -- and later steps take the ordinary eager path."""
        import warnings
        sg = self.sg
        for opt, c in zip((sg.optG, sg.optD, sg.optE), snap):
            opt.restore_host_counters(c)          # the recording advanced them without running anything
        warnings.warn("SRGAN_training: the step could not be recorded as " + ("ONE hipGraph with captured collectives" if keep_graph_mode
                                                                              else "hipGraph segments") + " on every rank"
                      + (f" (this rank: {type(err).__name__}: {err})" if err is not None else "")
                      + ("; this step runs eagerly, the next one is recorded as segments" if keep_graph_mode else "; running eagerly from here on"))
        self.graph = self._keep = self._baked = self._steps = None
        self._noise_i, self._onehot = 0, {}
        sg.target_image = sg.recon_image = sg.c_rand = None
        sg.loss_terms = {}
        sg.__dict__.get("_label_cache", {}).clear()
        # host-side caches may describe device buffers whose filling launch was only RECORDED (a packed operand or a repack table
        # first made while recording): forget them all, the eager step re-packs from the weights
        ops.invalidate_packed()
        sg.source_image = self.x
        sg._g_active = True
        try:
            with ops.pack_cache():
                out = sg.UnrolledUpdate()
        finally:
            sg._g_active = False
            if not keep_graph_mode:
                sg._graph = None
        return out

    def run(self, source_image, label):
        sg = self.sg
        self._stage(source_image, label)
        self._guard_packed()
        for opt in (sg.optG, sg.optD, sg.optE):
            opt.sync_device()
        if self.graph is None:
            err = None
            snap = [opt.host_counters() for opt in (sg.optG, sg.optD, sg.optE)]
            try:
                self._capture()      # records; the host-side optimiser counters advanced while recording
            except Exception as e:   # noqa: BLE001 -- under a process group the ranks first agree on what happened
                if not dp.is_distributed():
                    # leave nothing half-advanced behind for a caller that catches this: the counters moved without anything
                    # having run, caches may describe operands whose fill was only recorded, and graph mode stays off
                    for opt, c in zip((sg.optG, sg.optD, sg.optE), snap):
                        opt.restore_host_counters(c)
                    ops.invalidate_packed()
                    self._drop()
                    sg._graph = None
                    sg.target_image = sg.recon_image = sg.c_rand = None
                    sg.loss_terms = {}
                    raise
                err = e
            if dp.is_distributed() and not dp.all_agree(err is None):
                # some rank could not record the step: EVERY rank gives the recording up together (a rank replaying segments
                # beside a rank running eagerly would issue the same collectives in another order) and runs eagerly.  If the
                # attempt was the single-graph form (collectives captured), the segmented form is tried at the next step before
                # graph mode is given up: this step runs eagerly, the next one records segments.
                retry = self._last_single and not self._force_segmented
                self._force_segmented = self._force_segmented or self._last_single
                return self._abandon(err, snap, keep_graph_mode=retry)
            self._deltas = [opt.counters_since(c) for opt, c in zip((sg.optG, sg.optD, sg.optE), snap)]
        else:
            for opt, d in zip((sg.optG, sg.optD, sg.optE), self._deltas):
                opt.advance_host(d)
        self.graph.replay()
        self._graph_ran = True
        self._steps = self._opt_steps()
        sg.source_image = self.x
        sg.loss_terms = dict(self.terms)
        out = self.out.clone()       # the static result vector is overwritten by the next replay
        return [out[0], out[1], out[2]]


class _Recording:
    """One train step as a chain of hipGraph segments with host callables between them.

    Single process: one segment -- the whole step is one graph launch.  Data parallel: ``srgan_amd.dp`` ends the segment being
    recorded wherever the step starts a collective (``dp.launch_pending``: the bucket all-reduces after a backward; the mu
    all-gather) and wherever it needs the result (``GradReducer.finish``), and hands over a host callable; ``replay()`` launches
    segment, callable, segment, ... on the caller's stream.  A "start" callable enqueues the in-place all-reduces of the flat
    buckets on the communication stream behind a ``ready`` event and returns; the segments launched next -- the work the trainer
    placed between start and finish: the next translation during the discriminator loop, phase 1's encoder / generator forward
    under the last discriminator all-reduce, E's optimiser step and phase 2's E(source) under G's -- run UNDER the collectives;
    the "wait" callable makes the compute stream wait for the ``done`` events right before the optimiser-step segment.  Nothing
    of RCCL is captured and no stream is forked inside a capture: the collectives stay ordinary eager calls on static buffers in
    a fixed order.  2k + 6 segments per step (k x (start, wait), the all-gather, start(E) inside phase 1's backward, start(G) + wait(E), wait(G),
    start(G) + wait(G): a wait that follows its start directly rides in the same callable).  All segments allocate from one private pool and are replayed in recording order, so a tensor made in one segment
    is valid in the following ones."""

    def __init__(self, device, single=True):
        self.device = device
        # Round 6: with the C-ABI collectives (dp.set_transport("abi") / SRGAN_DP_COMM=abi) a collective is one
        # RCCL enqueue on a stream -- capturable -- so the step is recorded as ONE graph: where the segmented form cuts, the host
        # callable runs INSIDE the capture (the communication stream forks off the capture stream at the bucket's `ready` event
        # and joins it again at `done`), and a replay is a single launch.  SRGAN_DP_SINGLE_GRAPH=0 keeps the segments.
        self.single = bool(single and dp.is_distributed() and dp.transport() == "abi" and os.environ.get("SRGAN_DP_SINGLE_GRAPH") != "0")
        self.inline = []           # single-graph form: kinds of the captured collectives, in enqueue order
        self.segments = []         # [(CUDAGraph, callable or None)]
        self.trace = None          # tests: a list -> the host-side order of the last replay: ("segment", i) / (comm kind, i)
        self.pending = []          # flat gradient buckets laid out since the last cut (dp.GradReducer)
        self._cur = None
        self._pool = None
        self._stream = None
        self._stream_ctx = None
        # other threads of a data-parallel process (the process group's watchdog) may touch HIP while this one records
        self._mode = "thread_local" if dp.is_distributed() else "global"

    def _begin(self):
        self._cur = torch.cuda.CUDAGraph()
        self._cur.capture_begin(pool=self._pool, capture_error_mode=self._mode)

    def cut(self, comm):
        if self.single:
            comm()                 # enqueued into the capture (fork to / join from the communication stream)
            self.inline.append(getattr(comm, "kind", "comm"))
            return
        self._cur.capture_end()
        self.segments.append((self._cur, comm))
        self._begin()

    def __enter__(self):
        torch.cuda.synchronize(self.device)
        gc.collect()
        torch.cuda.empty_cache()
        self._pool = torch.cuda.graph_pool_handle()
        self._stream = torch.cuda.Stream(self.device)
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        self._stream_ctx = torch.cuda.stream(self._stream)
        self._stream_ctx.__enter__()
        dp.set_recorder(self if dp.is_distributed() else None)
        self._begin()
        return self

    def __exit__(self, *exc):
        dp.set_recorder(None)
        try:
            self._cur.capture_end()
            self.segments.append((self._cur, None))
        finally:
            self._cur = None
            self._stream_ctx.__exit__(*exc)
        if exc[0] is None and self.pending:
            raise RuntimeError("recorded step left gradient buckets without their all-reduce (GradReducer.finish() not called)")
        return False

    def replay(self):
        trace = self.trace
        if trace is not None:
            trace.clear()
        with torch.no_grad():          # the collectives write buffers (views made inside autograd functions) in place
            for i, (graph, comm) in enumerate(self.segments):
                graph.replay()
                if trace is not None:
                    trace.append(("segment", i))
                if comm is not None:
                    comm()
                    if trace is not None:
                        trace.append((getattr(comm, "kind", "comm"), i))


def lib_prof_off():
    _lib.load().srgan_prof_enable(0)     # HIP-event brackets cannot be recorded into a capture


def _select_rows(x, mask):
    """x[mask] along the batch dimension of an NHWC-dense (logical NCHW) tensor, keeping it NHWC-dense."""
    mask = torch.as_tensor(mask).to(x.device)
    return x.permute(0, 2, 3, 1)[mask].permute(0, 3, 1, 2)


class SingleGAN_training():
    """Conventional SingleGAN trainer on the HIP path -- BASELINE.json configs[0] (notebook 01).

    Same constructor / method signatures as the reference class (pyfiles/util_notebook.py:76-417):
      net = [G, D, E] with D a LIST of per-domain discriminators when ``singleD=False`` (each applied to the
      boolean-masked sub-batch of its domain) or one SingleDiscriminator_solo_multi when ``singleD=True``;
      E is the CBIN-conditioned ``Encoder_original(x, onehot(label))``.
    Kept semantics: ``update_D`` returns the LAST domain's errD; the per-domain D optimisers are always rebuilt by
    ``opt_sche_initialization``; phase 2's identity-regression path draws a RANDOM latent; stale-graph phase 2;
    the no-op "unrolled" restore.  Networks are used where the caller put them (the reference does not move them).
    """

    def __init__(self, net, opt, criterion, lbd, unrolled_k, device, ref_label, ndim,
                 classes, batch_size=64, encoded_feature="latent", singleD=False):
        self.G, self.D, self.E = net[0], net[1], net[2]
        self.optG, self.optD, self.optE = opt[0], opt[1], opt[2]
        self.scheG, self.scheD, self.scheE = None, None, None
        self.criterion, self.criterion_class = criterion
        self.lbd = lbd
        self.k = unrolled_k
        self.device = device
        self.ref_label = ref_label
        self.n_batch = batch_size
        self.encoded_feature = encoded_feature
        self.ndim = ndim
        self.classes = classes
        self.singleD = singleD
        self.source_image = None
        self.target_image = None
        self.recon_image = None
        self.label = None
        self.c_rand = None
        self.enc_info = None
        self.target_cenc = None
        if lbd["hist"] > 0:
            self.hi = histogram_imitation(device)
        self.loss_terms = {}
        self.noise_fn = torch.randn

    def opt_sche_initialization(self, lr=[0.0001, 0.0001, 0.0001]):
        lr_G, lr_D, lr_E = lr
        if self.optG is None:
            self.optG = Adam(self.G.parameters(), lr=lr_G, betas=(0.5, 0.999))
        self.scheG = optim.lr_scheduler.ExponentialLR(self.optG, gamma=0.95)
        if self.singleD:
            if self.optD is None:
                self.optD = Adam(self.D.parameters(), lr=lr_D, betas=(0.5, 0.999))
            self.scheD = optim.lr_scheduler.ExponentialLR(self.optD, gamma=0.95)
        else:
            self.optD = []
            self.scheD = []
            for i in self.classes:
                self.optD.append(Adam(self.D[i].parameters(), lr=lr_D, betas=(0.5, 0.999)))
                self.scheD.append(optim.lr_scheduler.ExponentialLR(self.optD[i], gamma=0.95))
        if self.optE is None:
            self.optE = Adam(self.E.parameters(), lr=lr_E, betas=(0.5, 0.999))
        self.scheE = optim.lr_scheduler.ExponentialLR(self.optE, gamma=0.95)
        return

    def _onehot(self, label):
        return class_encode(label, self.device, self.ref_label)

    def G_transformation(self, target_label, source_image, encoder=False, ref_image=None):
        class_vector = self._onehot(target_label)
        if encoder:
            latent, mu, logvar = self.E(ref_image, class_vector)
            info = [latent, mu, logvar]
            latent_vector = latent if self.encoded_feature == "latent" else mu
        else:
            latent_vector = host_to_device(self.noise_fn(source_image.shape[0], self.ndim), self.device)
            info = latent_vector
        target_image = self.G(source_image, torch.cat([class_vector, latent_vector], 1))
        return target_image, info

    def _mask(self, which, i):
        return (torch.as_tensor(self.label[which]) == i)

    def update_D(self):
        self.target_image, self.c_rand = self.G_transformation(self.label["target"], self.source_image, False)
        if self.singleD:
            self.D.zero_grad()
            output, output_class = self.D(self.source_image)
            errD_real = get_loss_D(output, 1., self.criterion, self.device)
            errD_class = get_domainloss_D(output_class, self._onehot(self.label["source"]), self.criterion_class)
            errD = errD_real + errD_class * self.lbd["class"]
            output, _ = self.D(self.target_image.detach())
            errD = errD + get_loss_D(output, 0., self.criterion, self.device)
            errD.backward()
            self.optD.step()
            return errD
        errD = None
        for i in self.classes:
            errD = 0
            self.D[i].zero_grad()
            m_real, m_fake = self._mask("source", i), self._mask("target", i)
            if int(m_real.sum()) != 0:
                errD = errD + get_loss_D(self.D[i](_select_rows(self.source_image, m_real)), 1., self.criterion, self.device)
            if int(m_fake.sum()) != 0:
                fake = _select_rows(self.target_image.detach(), m_fake)
                errD = errD + get_loss_D(self.D[i](fake), 0., self.criterion, self.device)
            errD.backward()
            self.optD[i].step()
        return errD

    def update_GandE(self):
        L = self.lbd
        self.G.zero_grad()
        self.E.zero_grad()
        src = self.source_image
        recon_image, source_enc_info = self.G_transformation(self.label["source"], self.target_image, True, src)
        d_params = list(self.D.parameters()) if self.singleD else [p for d in self.D for p in d.parameters()]
        errG = 0
        with _frozen(d_params):                      # D's weight gradients are discarded by the next zero_grad
            if self.singleD:
                output, output_class = self.D(self.target_image)
                errG_dis = get_loss_D(output, 1., self.criterion, self.device)
                errG_class = get_domainloss_D(output_class, self._onehot(self.label["target"]), self.criterion_class)
                errG = errG + errG_dis + errG_class * L["class"]
            else:
                for i in self.classes:
                    m = self._mask("target", i)
                    if int(m.sum()) != 0:
                        out = self.D[i](_select_rows(self.target_image, m))
                        errG = errG + get_loss_D(out, 1., self.criterion, self.device) / len(self.classes)
        errG_cycle = ops.l1_mean(src, recon_image, 1.0)
        errG = errG + errG_cycle * L["cycle"]
        errE = 0
        errE_output = errG_cycle * L["cycle"]
        terms = dict(errG_cycle=errG_cycle)
        _, mu, logvar = source_enc_info
        if L["KL"] > 0:
            errE_KL = ops.kl_normal(mu, logvar)
            errE = errE + errE_KL * L["KL"]
            errE_output = errE_output + errE_KL * L["KL"]
            terms["errE_KL"] = errE_KL
        if L["idt"] > 0:
            identity_image, _ = self.G_transformation(self.label["source"], src, True, src)
            errG_idt = ops.l1_mean(src, identity_image, 1.0)
            errG = errG + errG_idt * L["idt"]
            errE_output = errE_output + errG_idt * L["idt"]
            terms["errG_idt"] = errG_idt
        if L["batch_KL"] > 0:
            w_corr = L["corr_enc"] if L["corr_enc"] > 0 else 0.0
            w_hist = L["hist"] if L["hist"] > 0 else 0.0
            target = self.hi.target if w_hist > 0 else torch.full((50,), 0.02, device=mu.device)
            total, parts, _ = ops.latent_losses(mu, self.n_batch, target, L["batch_KL"], w_corr, w_hist)
            errE = errE + total
            errE_output = errE_output + total
            terms.update(errE_bKL=parts[0], errE_corr=parts[1], errE_hist=parts[2])
        (errG + errE).backward(retain_graph=True)
        self.optG.step()
        self.optE.step()

        self.G.zero_grad()
        self.E.zero_grad()
        with _frozen(list(self.E.parameters())):
            _, target_cenc, _ = self.E(self.target_image, self._onehot(self.label["target"]))
            errG_reg = ops.l1_mean(self.c_rand, target_cenc, 1.0)
            errG_ex = errG_reg * L["reg"]
            terms["errG_reg"] = errG_reg
            if L["idt_reg"] * L["idt"] > 0:
                idt_random_image, source_c_rand = self.G_transformation(self.label["source"], src, False)
                _, idt_cenc_rand, _ = self.E(idt_random_image, self._onehot(self.label["source"]))
                errG_idt_reg = ops.l1_mean(source_c_rand, idt_cenc_rand, 1.0)
                errG_ex = errG_ex + errG_idt_reg * (L["idt_reg"] * (L["idt"] / L["cycle"]))
                terms["errG_idt_reg"] = errG_idt_reg
            errG_ex.backward()
        self.optG.step()
        self.recon_image = recon_image.detach()
        self.loss_terms.update({k: v.detach() for k, v in terms.items()})
        return [errG.detach() + errG_ex.detach(), errE_output.detach()]

    def UnrolledUpdate(self):
        errorD = None
        for i in range(self.k):
            errD = self.update_D()
            if i == 0:
                errorD = errD.detach()
        errorG, errorE = self.update_GandE()
        return [errorG, errorD, errorE]

    def train(self, source_image, label):
        self.source_image = ops.to_nhwc(source_image)
        self.label = label
        self.loss_terms = {}
        return self.UnrolledUpdate()

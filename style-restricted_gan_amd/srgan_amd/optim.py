"""Fused Adam on the HIP path with the arithmetic of the reference's pinned torch==1.4.0
(``optim.Adam(lr, betas=(0.5, 0.999))`` built at util_notebook.py:498-507; formula SURVEY.md F.6).

Parameters are updated through raw device pointers: the autograd version counters are NOT bumped,
which is what lets phase 2 of ``update_GandE`` back-propagate a graph recorded before the step
(SURVEY.md Appendix C-1).  It is a ``torch.optim.Optimizer`` so ``ExponentialLR`` can drive ``lr``.

The step counter and the hyper-parameters of a parameter cohort live in a small DEVICE record
(``ops.adam_state_new``): ``step()`` launches "t += 1; derive the bias corrections; update" without
baking t or lr into the launch, so a train step captured into a hipGraph advances the optimiser on every
replay.  The host keeps ``state[p]["step"]`` in step for ``state_dict()`` / resume (``advance_host`` after a
replay) and pushes ``lr`` changes made by a scheduler to the device record (``sync_device``).
"""
import struct

import torch

from . import ops


class _Cohort:
    """Parameters of one group that share a step count: one pointer table, one device-side state record."""
    __slots__ = ("state", "table", "steps", "lr", "n", "max_numel")

    def __init__(self):
        self.state = self.table = None
        self.steps = -1            # completed steps the device record stands at
        self.lr = None
        self.n = self.max_numel = 0


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._cohorts = {}

    def _cohort(self, gi, params, group, steps_done):
        key = (gi,) + tuple(id(p) for p in params)
        co = self._cohorts.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if co is None or co.steps != steps_done:
            # first use, or the host-side count moved (load_state_dict / resume): (re)seed the device record
            if capturing:
                raise RuntimeError("srgan_amd.optim.Adam: an optimiser cohort first appears (or was re-seeded) inside a hipGraph "
                                   "capture -- run one eager train step with the same parameter set first")
            if co is not None:
                ops.bump_structure_epoch()       # a captured step points at the record being replaced
            if co is None:
                co = self._cohorts[key] = _Cohort()
                co.table = torch.empty(40 * len(params), dtype=torch.uint8, device=params[0].device)
            b1, b2 = group["betas"]
            co.state = ops.adam_state_new(params[0].device, group["lr"], b1, b2, group["eps"], steps_done)
            co.lr, co.steps = group["lr"], steps_done
            co.n, co.max_numel = len(params), max(p.numel() for p in params)
        if co.lr != group["lr"]:
            if capturing:
                raise RuntimeError("srgan_amd.optim.Adam: lr changed inside a hipGraph capture")
            ops.adam_state_set_lr(co.state, group["lr"])
            co.lr = group["lr"]
        return co

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            batches = {}                 # completed step count -> [(p, g, m, v)]: one multi-tensor launch per count
            for p in group["params"]:
                if p.grad is None:       # torch 1.4 skips parameters that received no gradient
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.data.is_contiguous():
                    raise RuntimeError("srgan_amd.optim.Adam needs contiguous parameters")
                batches.setdefault(int(st["step"]), []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
                st["step"] += 1
            for steps_done, items in batches.items():
                params = [it[0] for it in items]
                co = self._cohort(gi, params, group, steps_done)
                rows = []
                for p, g, m, v in items:
                    rows.extend((p.data.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()))
                # the gradient pointers change from step to step (and are the capture-time ones inside a graph): the table is
                # rewritten by every step, in stream order, through kernel arguments
                ops.upload_small(struct.pack(f"{len(rows)}q", *rows), params[0].device, out=co.table)
                ops.adam_multi_dev_(co.table, co.n, co.max_numel, co.state)
                co.steps = steps_done + 1
                self._keep_alive = items                     # until the next step: the launch reads them asynchronously
            ops.mark_stale(group["params"])                  # parameters were written through raw pointers
        return loss

    # -- hipGraph support --------------------------------------------------------------------------------------------
    def sync_device(self):
        """Push a scheduler's lr change to the device records (call between steps, outside any capture)."""
        for key, co in self._cohorts.items():
            lr = self.param_groups[key[0]]["lr"]
            if co.lr != lr:
                ops.adam_state_set_lr(co.state, lr)
                co.lr = lr

    def graph_keepalive(self):
        return [t for co in self._cohorts.values() for t in (co.state, co.table)]

    def host_counters(self):
        """Snapshot of the host-side step counters (per-parameter ``state[p]["step"]`` and the cohorts')."""
        return ({id(p): self.state[p]["step"] for g in self.param_groups for p in g["params"] if p in self.state and len(self.state[p])},
                {key: co.steps for key, co in self._cohorts.items()})

    def restore_host_counters(self, snap):
        per_param, per_cohort = snap
        for g in self.param_groups:
            for p in g["params"]:
                if id(p) in per_param:
                    self.state[p]["step"] = per_param[id(p)]
        for key, co in self._cohorts.items():
            if key in per_cohort:
                co.steps = per_cohort[key]

    def counters_since(self, snap):
        """What moved between ``snap`` (``host_counters()``) and now -- i.e. what ONE run of the step recorded in between does to
        the host-side counters: per parameter and per cohort (a cohort or parameter that did not take part stays out)."""
        per_param, per_cohort = snap
        now_p, now_c = self.host_counters()
        return ({pid: n - per_param.get(pid, 0) for pid, n in now_p.items() if n != per_param.get(pid, 0)},
                {key: n - per_cohort[key] for key, n in now_c.items() if key in per_cohort and n != per_cohort[key]})

    def advance_host(self, delta):
        """A replayed graph ran its optimiser steps on the device: move the host-side counters by what the recording moved
        (``counters_since``) -- exactly the parameters and cohorts that stepped while it was recorded."""
        per_param, per_cohort = delta
        for group in self.param_groups:
            for p in group["params"]:
                d = per_param.get(id(p))
                if d:
                    self.state[p]["step"] += d
        for key, d in per_cohort.items():
            self._cohorts[key].steps += d

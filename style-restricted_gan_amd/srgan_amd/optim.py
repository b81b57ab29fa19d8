"""Fused Adam on the HIP path with the arithmetic of the reference's pinned torch==1.4.0
(``optim.Adam(lr, betas=(0.5, 0.999))`` built at util_notebook.py:498-507; formula SURVEY.md F.6).

Parameters are updated through raw device pointers: the autograd version counters are NOT bumped,
which is what lets phase 2 of ``update_GandE`` back-propagate a graph recorded before the step
(SURVEY.md Appendix C-1).  It is a ``torch.optim.Optimizer`` so ``ExponentialLR`` can drive ``lr``.
"""
import torch

from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            batches = {}                 # step count -> [(p, g, m, v)]: one multi-tensor launch per count
            for p in group["params"]:
                if p.grad is None:       # torch 1.4 skips parameters that received no gradient
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.data.is_contiguous():
                    raise RuntimeError("srgan_amd.optim.Adam needs contiguous parameters")
                batches.setdefault(st["step"], []).append((p.data, g, st["exp_avg"], st["exp_avg_sq"]))
            for step, items in batches.items():
                if len(items) == 1:
                    pd, g, m, v = items[0]
                    ops.adam_step_(pd, g, m, v, group["lr"], b1, b2, group["eps"], step)
                    continue
                rows = []
                for pd, g, m, v in items:
                    rows.extend((pd.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), pd.numel()))
                host = torch.tensor(rows, dtype=torch.int64).pin_memory()
                table = host.to(items[0][0].device, non_blocking=True)
                ops.adam_multi_step_(table, len(items), max(pd.numel() for pd, _, _, _ in items), group["lr"], b1, b2,
                                     group["eps"], step)
                self._keep_alive = (host, table, items)     # until the next step: the launch reads them asynchronously
            ops.mark_stale(group["params"])                  # parameters were written through raw pointers
        return loss

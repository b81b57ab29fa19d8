"""Which parameter gradients of the full-width generator move most in the bf16 mode, and why (VERDICT r4 weak 1a / ADVICE r4).
The body of tests/test_ops_gpu.py::test_generator_bf16_activation_storage_vs_fp32_tensors with a per-tensor table: relative L2
error against the exact-fp32 mode of the bf16-storage run (e_s) and of the fp32-tensor bf16 run (e_c), and for e_s its split into
the broad part (median element error / median magnitude) and the sparse part (fraction of elements beyond 5 % of the tensor's max,
and the L2 error that remains once those elements are left out).  Run with SRGAN_HIP_LIB=scratch/libsrgan_exp.so and
SRGAN_NO_RGBIN16=1 / SRGAN_NO_RGBOUT16=1 to put the RGB layers back on the fp32 kernels."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops, model
torch.manual_seed(0)
G = model.SingleGenerator(3, 64, 2, 2, 1, "instance", num_con=12).cuda()
x = (torch.rand(2, 3, 16, 128) * 2 - 1).cuda()
c = torch.cat([torch.eye(4)[torch.tensor([1, 3])], torch.randn(2, 8)], 1).cuda()
gy = torch.randn(2, 3, 16, 128, generator=torch.Generator().manual_seed(5)).cuda()
params = [p for p in G.parameters()]
def run(mode, storage):
    ops.set_compute_dtype(mode); ops.STORAGE_BF16 = storage
    try:
        for p in params: p.grad = None
        with ops.pack_cache():
            y = G(x, c); (y * gy).sum().backward()
        return y.detach().clone(), [p.grad.detach().clone() for p in params]
    finally:
        ops.STORAGE_BF16 = True; ops.set_compute_dtype("fp32"); ops.invalidate_packed()
y32, g32 = run("fp32", True); yc, gc = run("bf16", False); ys, gs = run("bf16", True)
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-12))
print("output: storage", rel(ys, y32), "fp32 tensors", rel(yc, y32))
rows = []
for (name, _), a, b, r in zip(G.named_parameters(), gs, gc, g32):
    d = (a - r).abs().double().flatten(); rr = r.abs().double().flatten()
    med = float(d.median() / max(float(rr.median()), 1e-30)) if d.numel() >= 32 else float("nan")
    big = d > 0.05 * float(rr.max())
    rest = float(d[~big].norm() / (r.double().flatten()[~big].norm() + 1e-30)) if int((~big).sum()) else float("nan")
    rows.append((rel(a, r), rel(b, r), med, float(big.double().mean()), rest, name, tuple(r.shape)))
for e_s, e_c, med, frac, rest, name, shp in sorted(rows, reverse=True)[:14]:
    print(f"{name:44s} {str(shp):18s} e_s {e_s:.4f} e_c {e_c:.4f} | median/median {med:.4f} sparse frac {frac:.4f} l2 without them {rest:.4f}")

# ---- the mechanism: ReLU-mask flips.  (a) fraction of activation zeros that differ between the fp32 and the bf16 run at every
# norm + ReLU of the generator; (b) the same gradients in the exact-fp32 mode when only the INPUT and the WEIGHTS are rounded to
# bf16 once (a 2^-9 perturbation of the operands of the first use, fp32 arithmetic everywhere): what a perturbation of that
# size does to the gradients of this network, whatever produced it.
acts = {}
def hook(name):
    def f(mod, inp, out): acts.setdefault(name, []).append((out.detach().float() > 0).cpu())
    return f
hs = [m.register_forward_hook(hook(f"down_cnorms.{i}")) for i, m in enumerate(G.down_cnorms)] + \
     [m.register_forward_hook(hook(f"up_norms.{i}")) for i, m in enumerate(G.up_norms)]
run("fp32", True); run("bf16", True)
for h in hs: h.remove()
for k, (a, b) in acts.items():
    print(f"mask flips {k}: {float((a != b).double().mean()):.5f} of {a.numel()} elements")
def bf16r(t): return t.to(torch.bfloat16).to(torch.float32)
keep = [p.data.clone() for p in params]
x_keep = x.clone()
for p in params: p.data.copy_(bf16r(p.data))
x.copy_(bf16r(x))
yp, gp = run("fp32", True)
for p, k in zip(params, keep): p.data.copy_(k)
x.copy_(x_keep)
print("fp32 arithmetic, input and weights rounded to bf16 once: output", rel(yp, y32))
rows = sorted(((rel(a, r), name) for (name, _), a, r in zip(G.named_parameters(), gp, g32)), reverse=True)[:6]
for e, name in rows: print(f"   {name:44s} {e:.4f}")
ratios = sorted(((rel(a, r) / max(rel(b, r), 1e-9), rel(a, r), rel(b, r), name) for (name, _), a, b, r in zip(G.named_parameters(), gs, gp, g32)), reverse=True)
print("largest e_s / e_p:", [(round(q, 3), round(es, 4), round(ep, 4), n) for q, es, ep, n in ratios[:6]])
print("largest e_s - 1.3 e_p:", max(es - 1.3 * ep for _, es, ep, _ in ratios))

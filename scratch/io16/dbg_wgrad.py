"""Debug: 16-bit weight gradient of a generic layer against the fp32-tensor path of the bf16 mode."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops
torch.manual_seed(0)
n, ci, co, h, w = 2, 64, 128, 30, 30
x = torch.randn(n, ci, h, w).cuda().contiguous(memory_format=torch.channels_last)
wt = (torch.randn(co, ci, 3, 3) / 24).cuda()
gy = torch.randn(n, co, h, w).cuda().contiguous(memory_format=torch.channels_last)
ops.set_compute_dtype("bf16")
with ops.pack_cache():
    xa = x.to(torch.bfloat16).requires_grad_(True); wa = wt.clone().requires_grad_(True)
    ya = ops.conv2d_io(xa, wa, 1, ops.PAD_REFLECT, True)
    ya.backward(gy.to(torch.bfloat16))
    xb = x.to(torch.bfloat16).float().requires_grad_(True); wb = wt.clone().requires_grad_(True)
    yb = ops.conv2d(xb, wb, None, 1, 1, ops.PAD_REFLECT)
    yb.backward(gy.to(torch.bfloat16).float())
a, b = wa.grad, wb.grad
print("max", float((a - b).abs().max()), "scale", float(b.abs().max()))
d = (a - b).abs()
print("per-o err", d.amax(dim=(1, 2, 3))[:16].tolist())
print("per-i err", d.amax(dim=(0, 2, 3))[:16].tolist())
print("per-tap err", d.amax(dim=(0, 1)).tolist())
print("a[0,0]", a[0, 0].tolist(), "b[0,0]", b[0, 0].tolist())
print("ratio", (a / b).flatten()[:8].tolist())
bad = (d > 1e-3 * float(b.abs().max()))
print("bad fraction", float(bad.float().mean()))
bo = bad.any(dim=3).any(dim=2)          # [o][i]
print("bad by o (count of i):", bo.sum(1).tolist())
print("bad by i (count of o):", bo.sum(0).tolist())

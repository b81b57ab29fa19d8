"""bf16-mode trunk convolution (conv_halo16.hip) against the staged implicit GEMM it replaces (SRGAN_NO_HALO16=1 in another run)."""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops
ops.set_compute_dtype("bf16")
dev = torch.device("cuda", 0)
C = int(os.environ.get("C", 256)); S = int(os.environ.get("S", 32))
for B in (32, 64, 128):
    x = ops.to_nhwc(torch.randn(B, C, S, S, device=dev))
    w = (torch.randn(C, C, 3, 3, device=dev) / 48).requires_grad_(True)
    gy = ops.to_nhwc(torch.randn(B, C, S, S, device=dev))
    xg = x.clone().requires_grad_(True)
    with ops.pack_cache():
        def fwd(): return ops.conv2d(x, w.detach(), None, 1, 1)
        y = ops.conv2d(xg, w, None, 1, 1)
        def bwd(): return torch.autograd.grad(y, [xg], gy, retain_graph=True)
        for name, fn in (("fwd", fwd), ("dgrad", bwd)):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): fn()
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / 20 * 1e3
            fl = 2.0 * B * S * S * C * C * 9
            print(f"C={C} S={S} B={B:4d} {name:6s} {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s  ({fl / us / 1e6 / 2500:.3f} of 2.5 PF)")

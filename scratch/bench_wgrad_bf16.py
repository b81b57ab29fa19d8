"""Trunk weight gradient (256 -> 256, 3x3, 32x32 maps) in bf16 mode: wgrad kernel + slab reduce, whole-call time by HIP events."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops
ops.set_compute_dtype(os.environ.get("DT", "bf16"))
for B in (32, 64, 128):
    ci, h, k, co = 256, 32, 3, 256
    x = torch.randn(B, h, h, ci, device="cuda").permute(0, 3, 1, 2)
    w = torch.randn(co, ci, k, k, device="cuda") / 48
    gy = torch.randn(B, h, h, co, device="cuda").permute(0, 3, 1, 2)
    desc = ops._conv_desc(B, h, h, ci, h, h, co, k, k, 1, 1, 0, w)
    dw = torch.empty_like(w)
    fl = 2.0 * B * h * h * co * k * k * ci
    fn = lambda: ops._run_conv_wgrad(desc, x, gy, dw, None)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 10 * 1e3
    print(f"B={B:4d} wgrad call {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")

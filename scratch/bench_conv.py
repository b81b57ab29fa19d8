"""Micro-benchmark of the implicit-GEMM kernels on the SRGAN layer shapes (B=32)."""
import sys, os, ctypes, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops, _lib
lib = _lib.load()
if os.environ.get("DT") == "bf16":
    ops.set_compute_dtype("bf16")
B = int(os.environ.get("B", "32"))
REP = int(os.environ.get("REP", "10"))
shapes = [  # name, Cin, H, Cout, k, s, p
    ("G.res 256->256 k3 @32", 256, 32, 256, 3, 1, 1),
    ("G.down1 64->128 k4s2 @128", 64, 128, 128, 4, 2, 1),
    ("G.down2 128->256 k4s2 @64", 128, 64, 256, 4, 2, 1),
    ("D.c2 64->128 k4s2 @64", 64, 64, 128, 4, 2, 1),
    ("D.c3 128->256 k4s2 @32", 128, 32, 256, 4, 2, 1),
    ("D.c4 256->512 k4s2 @16", 256, 16, 512, 4, 2, 1),
    ("Er.l3a 512->512 k3 @7 reflect", 512, 7, 512, 3, 1, 1),
    ("Er.l3b 512->1024 k3 @7 reflect", 512, 7, 1024, 3, 1, 1),
    ("E.l0 64->128 k3 @62", 64, 62, 128, 3, 1, 1),
    ("E.l2 256->512 k3 @15", 256, 15, 512, 3, 1, 1),
    ("Er.l0a 64->64 k3 @62 reflect", 64, 62, 64, 3, 1, 1),
    ("Er.l0b 64->128 k3 @62 reflect", 64, 62, 128, 3, 1, 1),
    ("Er.l1a 128->128 k3 @31 reflect", 128, 31, 128, 3, 1, 1),
    ("Er.l1b 128->256 k3 @31 reflect", 128, 31, 256, 3, 1, 1),
    ("Er.l2a 256->256 k3 @15 reflect", 256, 15, 256, 3, 1, 1),
    ("Er.l2b 256->512 k3 @15 reflect", 256, 15, 512, 3, 1, 1),
    ("G.first 3->64 k7 @128", 3, 128, 64, 7, 1, 3),
    ("G.last 64->3 k7 @128", 64, 128, 3, 7, 1, 3),
]
only = os.environ.get("ONLY")
def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REP): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / REP
for name, ci, h, co, k, s, p in shapes:
    if only and only not in name: continue
    x = torch.randn(B, h, h, ci, device="cuda").permute(0, 3, 1, 2)
    w = torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5
    pm = 1 if "reflect" in name else 0
    y = ops.conv2d(x, w, None, s, p, pm)
    gy = torch.randn_like(y)
    ho = y.shape[2]
    desc = ops._conv_desc(B, h, h, ci, ho, ho, co, k, k, s, p, pm, w)
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * B * ho * ho * co * k * k * ci
    t_f = timeit(lambda: ops._run_conv_fwd(desc, x, w, None, y, 0, 0.0))
    t_d = timeit(lambda: ops._run_conv_dgrad(desc, gy, w, dx))
    t_w = timeit(lambda: ops._run_conv_wgrad(desc, x, gy, dw, None))
    print(f"{name:28s} GFLOP {fl/1e9:7.2f} | fwd {t_f*1e3:7.1f} us {fl/t_f/1e9:6.1f} TF | dgrad {t_d*1e3:7.1f} us {fl/t_d/1e9:6.1f} TF | wgrad {t_w*1e3:7.1f} us {fl/t_w/1e9:6.1f} TF")

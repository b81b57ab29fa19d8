import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops
def timeit(fn, rep=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rep): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / rep
B=32
for name, h, co, k, s, p in [("G.first 3->64 k7", 128, 64, 7, 1, 3), ("D.first 3->64 k4s2", 128, 64, 4, 2, 1), ("E.first 3->64 k7s2p1", 128, 64, 7, 2, 1)]:
    x = torch.randn(B, h, h, 3, device="cuda").permute(0, 3, 1, 2)
    w = torch.randn(co, 3, k, k, device="cuda") / (3*k*k)**0.5
    y = ops.conv2d(x, w, None, s, p); gy = torch.randn_like(y); ho = y.shape[2]
    desc = ops._conv_desc(B, h, h, 3, ho, ho, co, k, k, s, p, 0, w)
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    fl = 2.0 * B * ho * ho * co * k * k * 3
    tf = timeit(lambda: ops._run_conv_fwd(desc, x, w, None, y, 0, 0.0))
    td = timeit(lambda: ops._run_conv_dgrad(desc, gy, w, dx))
    tw = timeit(lambda: ops._run_conv_wgrad(desc, x, gy, dw, None))
    print(f"{name:24s} GFLOP {fl/1e9:6.2f} | fwd {tf*1e3:7.1f} us {fl/tf/1e9:5.1f} TF | dgrad {td*1e3:7.1f} us {fl/td/1e9:5.1f} TF | wgrad {tw*1e3:7.1f} us {fl/tw/1e9:5.1f} TF")

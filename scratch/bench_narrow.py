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
for B in (32, 64):
    x = torch.randn(B, 128, 128, 64, device="cuda").permute(0, 3, 1, 2)
    w = torch.randn(3, 64, 7, 7, device="cuda") / 56
    y = ops.conv2d(x, w, None, 1, 3)
    gy = torch.randn_like(y)
    desc = ops._conv_desc(B, 128, 128, 64, 128, 128, 3, 7, 7, 1, 3, 0, w)
    dw = torch.empty_like(w)
    fl = 2.0 * B * 128 * 128 * 3 * 49 * 64
    tf = timeit(lambda: ops._run_conv_fwd(desc, x, w, None, y, 0, 0.0))
    tw = timeit(lambda: ops._run_conv_wgrad(desc, x, gy, dw, None))
    print(f"B={B} narrow fwd {tf*1e3:7.1f} us {fl/tf/1e9:5.1f} TF | wgrad {tw*1e3:7.1f} us {fl/tw/1e9:5.1f} TF")

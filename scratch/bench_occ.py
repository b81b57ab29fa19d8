import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops
REP=10
def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REP): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / REP
for B, co in [(32,128),(32,256),(32,384),(32,512),(64,256),(16,256),(8,256)]:
    ci, h, k = 256, 32, 3
    x = torch.randn(B, h, h, ci, device="cuda").permute(0, 3, 1, 2)
    w = torch.randn(co, ci, k, k, device="cuda") / 48
    y = ops.conv2d(x, w, None, 1, 1)
    desc = ops._conv_desc(B, h, h, ci, h, h, co, k, k, 1, 1, 0, w)
    fl = 2.0 * B * h * h * co * k * k * ci
    t = timeit(lambda: ops._run_conv_fwd(desc, x, w, None, y, 0, 0.0))
    tiles = (B*h*h//128) * ((co+127)//128)
    print(f"B={B} Cout={co} tiles={tiles} ({tiles/256:.2f}/CU) fwd {t*1e3:7.1f} us {fl/t/1e9:6.1f} TF")

import sys, os, ctypes
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops, _lib
lib = _lib.load()
def kernel_time(fn, reps=6):
    fn(); torch.cuda.synchronize()
    lib.srgan_prof_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.srgan_prof_enable(0)
    tot=0; n=0
    for kid in range(lib.srgan_prof_num_kernels()):
        ms, c, fl = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
        lib.srgan_prof_collect(kid, ctypes.byref(ms), ctypes.byref(c), ctypes.byref(fl))
        tot+=ms.value; n+=c.value
    return tot/n*1e3
for B in (8, 16, 32, 64, 128, 256):
    ci, h, k, co = 256, 32, 3, 256
    x = torch.randn(B, h, h, ci, device="cuda").permute(0, 3, 1, 2)
    w = torch.randn(co, ci, k, k, device="cuda") / 48
    gy = torch.randn(B, h, h, co, device="cuda").permute(0, 3, 1, 2)
    desc = ops._conv_desc(B, h, h, ci, h, h, co, k, k, 1, 1, 0, w)
    dw = torch.empty_like(w)
    fl = 2.0 * B * h * h * co * k * k * ci
    t = kernel_time(lambda: ops._run_conv_wgrad(desc, x, gy, dw, None))
    M = B*h*h
    splits = min(-(-1024//36), -(-M//256)); rps = -(-(-(-M//splits))//32)*32; splits = -(-M//rps)
    print(f"B={B:4d} M={M:7d} splits={splits:3d} ktiles/block={rps//32:4d} blocks={36*splits:5d} kernel {t:8.1f} us  {fl/t/1e6:6.1f} TF")

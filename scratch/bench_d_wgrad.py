"""Weight gradient of the discriminator's 4x4 stride-2 layers at batch 64 (real + fake): direct vs batch-split Winograd."""
import sys, os, ctypes
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops, _lib
lib = _lib.load()
def kernel_time(fn, reps=8):
    fn(); torch.cuda.synchronize()
    lib.srgan_prof_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.srgan_prof_enable(0)
    out = {}
    for kid in range(lib.srgan_prof_num_kernels()):
        ms, c, fl = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
        lib.srgan_prof_collect(kid, ctypes.byref(ms), ctypes.byref(c), ctypes.byref(fl))
        if c.value: out[lib.srgan_prof_kernel_name(kid).decode()] = round(ms.value / c.value * 1e3, 1)
    return out
for (B, ci, h, co) in ((64, 64, 64, 128), (64, 128, 32, 256), (64, 256, 16, 512), (32, 64, 128, 128), (64, 32, 32, 64), (64, 64, 16, 128)):
    x = torch.randn(B, h, h, ci, device="cuda").permute(0, 3, 1, 2)
    w = torch.randn(co, ci, 4, 4, device="cuda") / 32
    gy = torch.randn(B, h // 2, h // 2, co, device="cuda").permute(0, 3, 1, 2)
    desc = ops._conv_desc(B, h, h, ci, h // 2, h // 2, co, 4, 4, 2, 1, 0, w)
    dw = torch.empty_like(w)
    fl = 2.0 * B * (h // 2) ** 2 * co * 16 * ci
    print(f"B={B} {ci}->{co} {h}^2  {fl/1e9:.2f} GF:", kernel_time(lambda: ops._run_conv_wgrad(desc, x, gy, dw, None)), flush=True)

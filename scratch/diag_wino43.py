"""Per-phase shader cycles of wino43_kernel's main loop from a diagnostic build (conv_wino43.hip compiled with -DW43_DIAG into
scratch/libsrgan_diag.so: s_memtime stamps; the kernel returns after the loop and writes workgroup 17's counters into the
output buffer).  Run with SRGAN_HIP_LIB=scratch/libsrgan_diag.so."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops
B = 32
x = torch.randn(B, 32, 32, 256, device="cuda").permute(0, 3, 1, 2)
w = torch.randn(256, 256, 3, 3, device="cuda") / 48
for rep in range(3):
    y = ops.conv2d(x, w, None, 1, 1)
    torch.cuda.synchronize()
v = y.permute(0, 2, 3, 1).reshape(-1)[:64].view(8, 8).cpu()
print("wave      loop      copy  multiply   barrier | per chunk: loop copy multiply barrier")
for wv in range(8):
    t = v[wv].tolist(); nk = t[4]
    print(f"{wv} {t[0]:9.0f} {t[1]:9.0f} {t[2]:9.0f} {t[3]:9.0f}   |" + " ".join(f"{a/nk:8.0f}" for a in t[:4]))

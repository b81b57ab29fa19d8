"""Per-phase shader cycles of the Winograd main loop from a diagnostic build (scratch/libsrgan_diag.so: s_memtime stamps, the
kernel returns after the loop and writes the counters of workgroup 17 into the output buffer)."""
import sys, os, ctypes
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
print("wave  loop   store   m0(16)  m1a(8)  barrier  tail(8)   per chunk: loop store m0 m1a bar tail")
for wv in range(8):
    t = v[wv].tolist(); nk = t[6]
    print(f"{wv} cls{int(t[7])} {t[0]:8.0f} {t[1]:7.0f} {t[2]:7.0f} {t[3]:7.0f} {t[4]:7.0f} {t[5]:7.0f}   |" + " ".join(f"{a/nk:7.0f}" for a in t[:6]))

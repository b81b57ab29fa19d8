"""Per-phase shader cycles of wino_wgrad_kernel's main loop from a diagnostic build (s_memtime stamps; the kernel returns
after the loop and writes workgroup 17's counters into the split-K slab = the head of the conv workspace)."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops
B = 32
x = torch.randn(B, 32, 32, 256, device="cuda").permute(0, 3, 1, 2)
w = (torch.randn(256, 256, 3, 3, device="cuda") / 48).requires_grad_(True)
seen = []
orig = ops._conv_ws
def spy(desc, dev):
    r = orig(desc, dev); seen.append(r[0]); return r
ops._conv_ws = spy
y = ops.conv2d(x, w, None, 1, 1)
y.backward(torch.randn_like(y))
torch.cuda.synchronize()
ws = seen[-1]
v = ws.view(torch.float32).reshape(-1)[:64].view(8, 8).cpu()
print("wave      loop   g0-3+store  g4+loads   g5-6   barrier   tail | per chunk")
for wv in range(8):
    t = v[wv].tolist(); nk = t[6]
    print(f"{wv} {t[0]:9.0f} {t[1]:9.0f} {t[2]:9.0f} {t[3]:9.0f} {t[4]:9.0f} {t[5]:9.0f} |" + " ".join(f"{a/nk:7.0f}" for a in t[:6]))

"""Phase stamps of rgbout_conv_kernel (build with -DRGBOUT_EXP=8, or 8 + ablation bits, and point SRGAN_HIP_LIB at it): per workgroup
start / prologue done / each 16-channel chunk done / end on the 100 MHz wall clock, and which XCD / CU ran it."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "style-restricted_gan_amd"))
from srgan_amd import ops

B = int(os.environ.get("B", 32))
x = torch.randn(B, 64, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(3, 64, 7, 7, device="cuda") * 0.02
for _ in range(3):
    y = ops.conv2d(x, w, None, 1, 3)
torch.cuda.synchronize()
raw = y.detach().permute(0, 2, 3, 1).contiguous().view(-1).view(torch.int64).cpu().numpy()
n = B * 8 * 2
t = raw[: n * 8].reshape(n, 8)
t0 = t[:, 0].min()
us = (t[:, :7] - t0) / 100.0
xcc = (t[:, 7] >> 32) & 0xF
hw = t[:, 7] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
print("workgroups", n, " kernel span %.1f us" % us[:, 6].max())
print("start  : min %.1f median %.1f max %.1f" % (us[:, 0].min(), np.median(us[:, 0]), us[:, 0].max()))
d = np.diff(us, axis=1)
for k, name in enumerate(["prologue", "chunk0", "chunk1", "chunk2", "chunk3", "epilogue"]):
    print("%-9s: median %.2f us  min %.2f  max %.2f" % (name, np.median(d[:, k]), d[:, k].min(), d[:, k].max()))
print("total per workgroup: median %.2f" % np.median(us[:, 6] - us[:, 0]))
first = us[:, 0] < 5.0
for name, sel in (("first round", first), ("second round", ~first)):
    if sel.any():
        print("  %-12s n %3d  start median %.1f  prologue %.2f  chunks %s  total %.2f" % (
            name, int(sel.sum()), np.median(us[sel, 0]), np.median(d[sel, 0]), " ".join("%.2f" % np.median(d[sel, k]) for k in range(1, 5)),
            np.median(us[sel, 6] - us[sel, 0])))
print("workgroups starting in the first 5 us:", int(first.sum()), " distinct (xcc, se, cu):", len(set(zip(xcc.tolist(), se.tolist(), cu.tolist()))))
per_xcc = [int((xcc == k).sum()) for k in range(8)]
print("per XCD:", per_xcc)

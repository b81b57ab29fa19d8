"""Achieved HBM rate of the instance-norm passes at one fixed shape (run under rocprofv3 --kernel-trace --stats, one shape per
process: SHAPE="B,C,H"): forward + backward of ops.instance_norm_act, 20 times."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops
B, C, H = (int(v) for v in os.environ.get("SHAPE", "32,64,128").split(","))
x = torch.randn(B, H, H, C, device="cuda").permute(0, 3, 1, 2).requires_grad_(True)
sc = (torch.rand(B, C, device="cuda") + 0.5).requires_grad_(True); sh = torch.randn(B, C, device="cuda").requires_grad_(True)
gy = torch.randn(B, H, H, C, device="cuda").permute(0, 3, 1, 2)
for _ in range(20):
    y = ops.instance_norm_act(x, sc, sh, None, ops.ACT_RELU)
    y.backward(gy)
torch.cuda.synchronize()
print("tensor MB", B * C * H * H * 4 / 1e6)

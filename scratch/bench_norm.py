"""Slab instance-norm kernels on the trunk tensor (B x 32 x 32 x 256): forward and backward launch time."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops
for B in (32, 64, 128):
    x = torch.randn(B, 32, 32, 256, device="cuda").permute(0, 3, 1, 2).requires_grad_(True)
    sc = torch.rand(B, 256, device="cuda") + 0.5; sh = torch.randn(B, 256, device="cuda")
    res = torch.randn(B, 32, 32, 256, device="cuda").permute(0, 3, 1, 2)
    gy = torch.randn(B, 32, 32, 256, device="cuda").permute(0, 3, 1, 2)
    def fwd():
        return ops.instance_norm_act(x, sc, sh, None, ops.ACT_RELU)
    y = fwd(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    n = 30
    with torch.no_grad():
        ev[0].record()
        for _ in range(n): fwd()
        ev[1].record()
    y = fwd()
    ev[2].record()
    for _ in range(n): y.backward(gy, retain_graph=True)
    ev[3].record(); torch.cuda.synchronize()
    print(f"B={B}: fwd {ev[0].elapsed_time(ev[1])/n*1e3:.1f} us  bwd {ev[2].elapsed_time(ev[3])/n*1e3:.1f} us (includes autograd host time)", flush=True)

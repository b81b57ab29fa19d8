"""Parity of the F(4x4,2x2) kernels on one geometry: forward, input gradient (with and without the mask epilogue's twin), against
PyTorch-CPU.  Usage: check.py N I H W O"""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch, torch.nn.functional as F
from srgan_amd import ops
n, i, h, w, o = [int(a) for a in sys.argv[1:6]]
torch.manual_seed(0)
x = torch.randn(n, i, h, w); wt = torch.randn(o, i, 4, 4) / (i * 16) ** 0.5
yr = F.conv2d(x, wt, None, 2, 1)
gy = torch.randn_like(yr)
dxr = F.conv_transpose2d(gy, wt, None, 2, 1)
xd = x.cuda().contiguous(memory_format=torch.channels_last); wd = wt.cuda()
y = ops.conv2d(xd, wd, None, 2, 1)
torch.cuda.synchronize()
e = (y.cpu() - yr).abs()
print("fwd max err", float(e.max()), "scale", float(yr.abs().max()))
if float(e.max()) > 1e-3:
    bad = (e > 1e-3).nonzero()
    print(" bad count", len(bad), "first", bad[:5].tolist(), "last", bad[-5:].tolist())
desc = ops._conv_desc(n, h, w, i, h // 2, w // 2, o, 4, 4, 2, 1, 0, wd)
gyd = gy.cuda().contiguous(memory_format=torch.channels_last)
dx = torch.empty_like(xd)
ops._run_conv_dgrad(desc, gyd, wd, dx)
torch.cuda.synchronize()
e = (dx.cpu() - dxr).abs()
print("dgrad max err", float(e.max()), "scale", float(dxr.abs().max()))
if float(e.max()) > 1e-3:
    bad = (e > 1e-3).nonzero()
    print(" bad count", len(bad), "first", bad[:5].tolist(), "last", bad[-5:].tolist())
e = (y.cpu() - yr).abs()
print("fwd after dgrad max err", float(e.max()))
if os.environ.get("MAP"):
    e = (y.cpu() - yr).abs()
    bad = (e > 1e-3)
    print("bad per channel:", [int(c) for c in bad.sum(dim=(0, 2, 3)).nonzero().flatten()][:64])
    bp = bad.sum(dim=(0, 1))
    print("bad per (y%4, x%4):")
    for a in range(4):
        print([int(bp[a::4, b::4].sum()) for b in range(4)])
    print("bad per tile x index:", [int(bad[..., 4*t:4*t+4].sum()) for t in range(bad.shape[-1] // 4)])
    print("bad per tile y index:", [int(bad[:, :, 4*t:4*t+4].sum()) for t in range(bad.shape[-2] // 4)])
    print("bad per image:", [int(bad[b].sum()) for b in range(bad.shape[0])])
if os.environ.get("SENT"):
    dx.fill_(1000.0)
    ops._run_conv_dgrad(desc, gyd, wd, dx)
    torch.cuda.synchronize()
    d_ = dx.cpu()
    bad = (d_ - dxr).abs() > 1e-3
    print("dgrad bad", int(bad.sum()), "of which still sentinel", int((d_[bad] == 1000.0).sum()))
    idx = bad.nonzero()[:6]
    for i_ in idx:
        i_ = tuple(int(v) for v in i_)
        print(i_, float(d_[i_]), float(dxr[i_]))
    # is the bad value the right value of some OTHER element?
    i0 = tuple(int(v) for v in idx[0]); v0 = float(d_[i0])
    near = ((dxr - v0).abs() < 1e-4).nonzero()[:5]
    print("value", v0, "found in reference at", near.tolist())

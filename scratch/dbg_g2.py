import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "style-restricted_gan_amd"))
import torch, numpy as np, torch.nn.functional as F
from srgan_amd import model, ops
from oracle import params, nets as onets
def rel(a, b): return float((a.detach().cpu().double()-b.detach().double()).abs().max()/max(float(b.detach().abs().max()),1e-30))
spec = params.generator_spec(3, 64, 2, 2, 2, 12)
P = params.fill(spec, 3)
G = model.SingleGenerator(3, 64, 2, 2, 2, "instance", num_con=12); G.load_state_dict(P); G.cuda()
x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(1)) * 2 - 1
c = torch.randn(2, 12, generator=torch.Generator().manual_seed(2))
Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
taps_r = []
def T(t, name): t.retain_grad(); taps_r.append((name, t)); return t
h = x
for i in range(3):
    h = T(F.conv2d(h, Pr[f"down_convs.{i}.weight"], None, 1 if i==0 else 2, 3 if i==0 else 1), f"dconv{i}")
    h = T(torch.relu(onets.cbin(h, c, Pr, f"down_cnorms.{i}")), f"dnorm{i}")
for j in range(2):
    r = T(F.conv2d(h, Pr[f"resBlocks.{j}.c1.weight"], None, 1, 1), f"res{j}c1")
    r = T(torch.relu(onets.cbin(r, c, Pr, f"resBlocks.{j}.cn1")), f"res{j}n1")
    r = T(F.conv2d(r, Pr[f"resBlocks.{j}.c2.weight"], None, 1, 1), f"res{j}c2")
    h = T(onets.cbin(r, c, Pr, f"resBlocks.{j}.cn2") + h, f"res{j}out")
for i in range(2):
    h = T(F.conv_transpose2d(h, Pr[f"up_convs.{i}.weight"], None, 2, 1), f"up{i}")
    h = T(torch.relu(onets.inorm(h)), f"upn{i}")
h = T(F.conv2d(h, Pr["up_convs.2.weight"], None, 1, 3), "last")
yr = torch.tanh(h)
w = torch.randn(yr.shape, generator=torch.Generator().manual_seed(3))
(yr * w).sum().backward()
taps = []
def hook(name):
    def f(mod, inp, out):
        t = out[0] if isinstance(out, tuple) else out
        t.retain_grad(); taps.append((name, t))
    return f
for i in range(3):
    G.down_convs[i].register_forward_hook(hook(f"dconv{i}")); G.down_cnorms[i].register_forward_hook(hook(f"dnorm{i}"))
for j in range(2):
    b = G.resBlocks[j]
    b.c1.register_forward_hook(hook(f"res{j}c1")); b.cn1.register_forward_hook(hook(f"res{j}n1")); b.c2.register_forward_hook(hook(f"res{j}c2")); b.cn2.register_forward_hook(hook(f"res{j}out"))
for i in range(2):
    G.up_convs[i].register_forward_hook(hook(f"up{i}")); G.up_norms[i].register_forward_hook(hook(f"upn{i}"))
G.up_convs[2].register_forward_hook(hook("last"))
y = G(x.cuda(), c.cuda())
(y * w.cuda()).sum().backward()
dr = dict(taps_r)
for name, t in taps:
    print(f"{name:10s} val {rel(t, dr[name]):.2e} grad {rel(t.grad, dr[name].grad):.2e}  shape {tuple(t.shape)}")
print("---- standalone norm on captured tensors")
d = dict(taps)
xin = d["up1"].detach().clone(); gy = d["upn1"].grad.detach().clone()
xs = xin.clone().requires_grad_(True)
ys = ops.instance_norm_act(xs, None, None, None, 1, 0.0); ys.backward(gy)
xc = ops.to_nchw(xin).cpu().requires_grad_(True)
yc = torch.relu(F.instance_norm(xc)); yc.backward(ops.to_nchw(gy).cpu())
print("standalone val", rel(ys, yc), "dx", rel(xs.grad, xc.grad), "in-graph dx vs cpu", rel(d["up1"].grad, xc.grad))
print("x stats: mean", float(xin.mean()), "std", float(xin.std()), "min var over (n,c)", float(ops.to_nchw(xin).var(dim=(2,3)).min()))
print("gy dense?", ops.is_nhwc_dense(gy), gy.stride(), "x", xin.stride())
print("---- flip count")
a = ops.to_nchw(d["up1"].grad).cpu().double(); b = dr["up1"].grad.double()
e = (a-b).abs(); print("n elems", e.numel(), "n > 1e-4*max:", int((e > 1e-4*b.abs().max()).sum()), "max err", float(e.max()), "max", float(b.abs().max()))
print("---- dw exactness")
def rnd(*s, seed=0): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
x = rnd(2,64,32,32,seed=1).requires_grad_(True); w = (rnd(128,64,4,4,seed=2)/32).requires_grad_(True)
y = F.conv2d(x,w,None,2,1); gy = rnd(*y.shape,seed=3); y.backward(gy)
xd = x.detach().cuda().requires_grad_(True); wd = w.detach().cuda().requires_grad_(True)
yd = ops.conv2d(xd,wd,None,2,1); yd.backward(gy.cuda())
A = wd.grad.cpu(); B = w.grad
print("max|a-b|", float((A-B).abs().max()), "a", A.flatten()[:4].tolist(), "b", B.flatten()[:4].tolist(), "equal:", bool((A==B).all()))

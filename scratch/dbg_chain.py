import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "style-restricted_gan_amd"))
import torch, numpy as np, torch.nn.functional as F
from srgan_amd import ops
def rnd(*s, seed=0): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
def rel(a, b): return float((a.detach().cpu().double()-b.detach().double()).abs().max()/b.detach().abs().max())
x = rnd(2,128,16,16,seed=1).requires_grad_(True)
wt = (rnd(128,64,4,4,seed=2)/np.sqrt(128*4)).requires_grad_(True)
wc = (rnd(3,64,7,7,seed=3)/np.sqrt(64*49)).requires_grad_(True)
a = F.conv_transpose2d(x, wt, None, 2, 1); a.retain_grad()
b = torch.relu(F.instance_norm(a)); b.retain_grad()
c = F.conv2d(b, wc, None, 1, 3); c.retain_grad()
y = torch.tanh(c)
w = rnd(*y.shape, seed=4)
(y*w).sum().backward()
xd = x.detach().cuda().requires_grad_(True); wtd = wt.detach().cuda().requires_grad_(True); wcd = wc.detach().cuda().requires_grad_(True)
ad = ops.conv_transpose2d(xd, wtd, 2, 1); ad.retain_grad()
bd = ops.instance_norm_act(ad, None, None, None, 1, 0.0); bd.retain_grad()
cd = ops.conv2d(bd, wcd, None, 1, 3); cd.retain_grad()
yd = ops.tanh(cd)
(yd*w.cuda()).sum().backward()
for n_, p, q in [("a",ad,a),("b",bd,b),("c",cd,c),("y",yd,y)]: print(n_, rel(p,q))
for n_, p, q in [("dc",cd.grad,c.grad),("db",bd.grad,b.grad),("da",ad.grad,a.grad),("dx",xd.grad,x.grad),("dwt",wtd.grad,wt.grad),("dwc",wcd.grad,wc.grad)]:
    print(n_, rel(p,q), float(p.abs().max()), float(q.abs().max()))

import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "style-restricted_gan_amd"))
import torch, numpy as np, torch.nn.functional as F
from srgan_amd import ops
def rnd(*s, seed=0): return torch.randn(*s, generator=torch.Generator().manual_seed(seed))
def rel(a, b): return float((a.detach().cpu().double()-b.detach().double()).abs().max()/b.abs().max())
# conv 64->3 k7 at 32x32
for (n,i,h,o,k,s,p) in [(2,64,32,3,7,1,3),(2,128,16,64,4,2,1),(2,3,32,64,7,1,3),(2,64,32,128,4,2,1),(2,256,8,256,3,1,1)]:
    x = rnd(n,i,h,h,seed=1).requires_grad_(True); w = (rnd(o,i,k,k,seed=2)/np.sqrt(i*k*k)).requires_grad_(True)
    y = F.conv2d(x,w,None,s,p); gy = rnd(*y.shape,seed=3); y.backward(gy)
    xd = x.detach().cuda().requires_grad_(True); wd = w.detach().cuda().requires_grad_(True)
    yd = ops.conv2d(xd,wd,None,s,p); yd.backward(gy.cuda())
    print("conv",(n,i,h,o,k,s,p),"y",rel(yd,y),"dx",rel(xd.grad,x.grad),"dw",rel(wd.grad,w.grad))
for (n,ci,h,co) in [(2,128,16,64),(2,256,8,128)]:
    x = rnd(n,ci,h,h,seed=1).requires_grad_(True); w = (rnd(ci,co,4,4,seed=2)/np.sqrt(ci*4)).requires_grad_(True)
    y = F.conv_transpose2d(x,w,None,2,1); gy = rnd(*y.shape,seed=3); y.backward(gy)
    xd = x.detach().cuda().requires_grad_(True); wd = w.detach().cuda().requires_grad_(True)
    yd = ops.conv_transpose2d(xd,wd,2,1); yd.backward(gy.cuda())
    print("convT",(n,ci,h,co),"y",rel(yd,y),"dx",rel(xd.grad,x.grad),"dw",rel(wd.grad,w.grad))
for shape in [(2,64,32,32),(2,128,16,16),(2,256,8,8)]:
    x = (rnd(*shape,seed=1)*2+0.5).requires_grad_(True)
    y = torch.relu(F.instance_norm(x)); gy = rnd(*shape,seed=4); y.backward(gy)
    xd = x.detach().cuda().requires_grad_(True)
    yd = ops.instance_norm_act(xd,None,None,None,1,0.0); yd.backward(gy.cuda())
    print("IN",shape,"y",rel(yd,y),"dx",rel(xd.grad,x.grad))

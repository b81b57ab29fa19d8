import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "style-restricted_gan_amd"))
import torch, numpy as np
from srgan_amd import model, ops
from oracle import params, nets as onets
spec = params.generator_spec(3, 64, 2, 2, 2, 12)
P = params.fill(spec, 3)
G = model.SingleGenerator(3, 64, 2, 2, 2, "instance", num_con=12); G.load_state_dict(P); G.cuda()
x = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(1)) * 2 - 1
c = torch.randn(2, 12, generator=torch.Generator().manual_seed(2))
Pr = {k: v.clone().double().requires_grad_(True) for k, v in P.items()}
yr = onets.generator(Pr, x.double(), c.double())
w = torch.randn(yr.shape, generator=torch.Generator().manual_seed(3))
(yr * w.double()).sum().backward()
Pf = {k: v.clone().requires_grad_(True) for k, v in P.items()}
yf = onets.generator(Pf, x, c)
(yf * w).sum().backward()
y = G(x.cuda(), c.cuda())
(y * w.cuda()).sum().backward()
print("out err hip", float((y.cpu().double()-yr).abs().max()), "cpu32", float((yf.double()-yr).abs().max()))
for k, p in G.named_parameters():
    r = Pr[k].grad
    s = float(r.abs().max())
    print(f"{k:40s} hip {float((p.grad.cpu().double()-r).abs().max())/s:.2e}  cpu32 {float((Pf[k].grad.double()-r).abs().max())/s:.2e}")

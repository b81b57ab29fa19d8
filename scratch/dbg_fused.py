import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import numpy as np, torch
from srgan_amd import ops
os.environ["SRGAN_WINOGRAD_THRESHOLD_SCALE"] = "0"
n, c, o = 4, 64, 64
g = torch.Generator().manual_seed(1)
x = torch.randn(n, c, 32, 32, generator=g).cuda(); wt = (torch.randn(o, c, 3, 3, generator=g) / np.sqrt(c * 9)).cuda()
sc = (torch.rand(n, c, generator=g) + 0.5).cuda(); sh = (torch.randn(n, c, generator=g) * 0.3).cuda()
with ops.pack_cache(), torch.no_grad():
    y1 = ops.instance_norm_act_conv(x, sc, sh, wt, ops.ACT_RELU, 0.0, 1e-5)
    h = ops.instance_norm_act(x, sc, sh, None, ops.ACT_RELU, 0.0, 1e-5)
    y2 = ops.conv2d(h, wt, None, 1, 1)
d = (y1 - y2).abs()
print("max diff", float(d.max()), "scale", float(y2.abs().max()), "frac nonzero", float((d > 0).float().mean()))
idx = torch.nonzero(d > 1e-3)
print(idx[:10].tolist(), idx.shape)

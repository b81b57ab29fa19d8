"""Which aten ops launch the small elementwise kernels of a train step (torch profiler, shapes recorded)."""
import sys, os, collections
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch, bench
from torch.profiler import profile, ProfilerActivity
sg = bench.build_trainer(128, 32, 5, torch.device("cuda"))
def batch(s):
    x, src, tgt = bench.synthetic_batch(32, 128, 4, seed=s)
    return x.cuda(), {"source": src.cuda(), "target": tgt}
b = [batch(0), batch(1)]
for s in range(2): sg.train(*b[s])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    sg.train(*b[0]); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::mul", "aten::fill_", "aten::zero_", "aten::copy_", "aten::sum", "aten::div", "aten::zeros", "aten::cat", "aten::clone", "aten::contiguous", "aten::neg", "aten::exp", "aten::pow", "aten::sub"):
        shp = str(e.input_shapes[:2]) if e.input_shapes else ""
        cnt[(e.name, shp)] += 1
for k, v in cnt.most_common(45): print(v, k)

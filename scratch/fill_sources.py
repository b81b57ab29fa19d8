"""Which autograd nodes / forward ops enclose the small fill / add / copy launches of one eager train step (torch profiler: the
chain of cpu_parent events of every aten::fill_ / zero_ / zeros / add_ / add / mul / copy_ / cat)."""
import os, sys, collections
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
B = 32
sg = bench.build_trainer(128, B, 5, torch.device("cuda"))
batches = []
for s in range(3):
    x, src, tgt = bench.synthetic_batch(B, 128, 4, seed=s)
    batches.append((x.cuda(), {"source": src.cuda(), "target": tgt}))
sg.train(*batches[0]); sg.train(*batches[1]); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    sg.train(*batches[2]); torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::mul", "aten::copy_", "aten::cat", "aten::exp", "aten::sum", "aten::ones_like", "aten::zeros")
agg = collections.Counter()
for ev in prof.events():
    if ev.name not in want:
        continue
    chain, p = [], ev.cpu_parent
    while p is not None:
        chain.append(p.name)
        p = p.cpu_parent
    if any(c in want for c in chain):        # counted at its outermost aten op
        continue
    top = " < ".join(c[:60] for c in chain[:3]) or "(top level)"
    agg[(ev.name, str(ev.input_shapes)[:50], top)] += 1
for (name, shp, top), n in agg.most_common(70):
    print("%4d %-14s %-50s %s" % (n, name, shp, top))

"""Which Python lines of one eager train step issue the small ATen launches (fill_, add, copy_, cat ...)?"""
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    sg.train(*batches[2]); torch.cuda.synchronize()
agg = collections.Counter()
shapes = {}
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::copy_", "aten::cat", "aten::mul", "aten::sum", "aten::clone", "aten::zeros", "aten::zeros_like", "aten::contiguous"):
        st = [f for f in (ev.stack or []) if "srgan_amd" in f or "bench.py" in f]
        key = (ev.name, st[0] if st else ((ev.stack or ["?"])[0]))
        agg[key] += 1
        shapes.setdefault(key, str(ev.input_shapes)[:80])
for (name, where), n in agg.most_common(60):
    print("%4d %-16s %-90s %s" % (n, name, where[-90:], shapes[(name, where)]))

"""Which Python call sites of one eager train step issue the small ATen launches?  Wraps the usual suspects and counts callers."""
import os, sys, collections, traceback
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
import bench
B = 32
sg = bench.build_trainer(128, B, 5, torch.device("cuda"))
batches = []
for s in range(3):
    x, src, tgt = bench.synthetic_batch(B, 128, 4, seed=s)
    batches.append((x.cuda(), {"source": src.cuda(), "target": tgt}))
sg.train(*batches[0]); sg.train(*batches[1]); torch.cuda.synchronize()
counts = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "srgan_amd" in fr.filename or fr.filename.endswith("bench.py"):
            return "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
    return "?"
def wrap(obj, name, label):
    orig = getattr(obj, name)
    def w(*a, **k):
        t = a[0] if a and torch.is_tensor(a[0]) else None
        cuda = (t is not None and t.is_cuda) or any(torch.is_tensor(v) and v.is_cuda for v in a) or "cuda" in str(k.get("device", ""))
        if cuda or label in ("zeros", "cat", "full", "ones", "stack"):
            counts[(label, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, w)
for n in ("zero_", "fill_", "copy_", "clone", "contiguous", "add_", "add", "mul", "mul_", "sub", "__add__", "__mul__", "__sub__", "__rmul__", "__radd__", "sum", "mean", "exp", "to", "detach_"):
    wrap(torch.Tensor, n, n)
for n in ("zeros", "cat", "full", "ones", "stack", "zeros_like", "empty_like"):
    wrap(torch, n, n)
sg.train(*batches[2]); torch.cuda.synchronize()
for (label, where), n in counts.most_common(70):
    if label in ("empty_like",):
        continue
    print("%4d %-12s %s" % (n, label, where))

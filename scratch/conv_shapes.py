"""Geometry of every convolution descriptor one train step builds (count per shape): which layers the generic kernels serve."""
import sys, os, collections
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
import bench
from srgan_amd import ops
if os.environ.get("SRGAN_DTYPE") == "bf16":
    ops.set_compute_dtype("bf16")
B = int(os.environ.get("B", 32)); S = int(os.environ.get("SIZE", 128))
sg = bench.build_trainer(S, B, 5, torch.device("cuda"))
x, src, tgt = bench.synthetic_batch(B, S, 4, seed=0)
batch = (x.cuda(), {"source": src.cuda(), "target": tgt})
sg.train(*batch); torch.cuda.synchronize()
seen = collections.Counter()
orig = ops._conv_desc
def spy(n, hi, wi, i, ho, wo, o, kh, kw, stride, pad, pad_mode, weight):
    seen[(n, hi, wi, i, ho, wo, o, kh, kw, stride, pad, pad_mode)] += 1
    return orig(n, hi, wi, i, ho, wo, o, kh, kw, stride, pad, pad_mode, weight)
ops._conv_desc = spy
sg.train(*batch); torch.cuda.synchronize()
print("   n   hi   wi    i   ho   wo    o kh kw s p pm  descs      GF")
for k, c in sorted(seen.items(), key=lambda kv: -(2.0 * kv[0][0] * kv[0][4] * kv[0][5] * kv[0][6] * kv[0][3] * kv[0][7] * kv[0][8])):
    n, hi, wi, i, ho, wo, o, kh, kw, s, p, pm = k
    gf = 2.0 * n * max(ho * wo, hi * wi // (s * s) if ho > hi else 0) * o * i * kh * kw / 1e9
    print(f"{n:4d} {hi:4d} {wi:4d} {i:4d} {ho:4d} {wo:4d} {o:4d} {kh:2d} {kw:2d} {s} {p} {pm:2d} {c:6d} {gf:8.2f}")

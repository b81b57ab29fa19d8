"""Per-(kernel, FLOP count) table of the GEMM launches of one train step: which layers run below the fleet average."""
import sys, os, ctypes, collections
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
import bench
from srgan_amd import _lib
lib = _lib.load()
if os.environ.get("SRGAN_DTYPE") == "bf16":
    from srgan_amd import ops
    ops.set_compute_dtype("bf16")
B = 32
sg = bench.build_trainer(128, B, 5, torch.device("cuda"))
batches = []
for s in range(2):
    x, src, tgt = bench.synthetic_batch(B, 128, 4, seed=s)
    batches.append((x.cuda(), {"source": src.cuda(), "target": tgt}))
sg.train(*batches[0]); torch.cuda.synchronize()
_shapes = collections.Counter()
if os.environ.get("SHAPES"):
    from srgan_amd import ops as _ops
    _orig = _ops._conv_desc
    def _rec(n, hi, wi, i, ho, wo, o, kh, kw, stride, pad, pad_mode, weight):
        _shapes[(n, hi, wi, i, ho, wo, o, kh, kw, stride, pad)] += 1
        return _orig(n, hi, wi, i, ho, wo, o, kh, kw, stride, pad, pad_mode, weight)
    _ops._conv_desc = _rec
lib.srgan_prof_enable(1)
sg.train(*batches[1]); torch.cuda.synchronize()
lib.srgan_prof_enable(0)
agg = collections.defaultdict(list)
for i in range(lib.srgan_prof_num_slots()):
    kid, ms, fl = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
    _lib.check(lib.srgan_prof_slot(i, ctypes.byref(kid), ctypes.byref(ms), ctypes.byref(fl)), "slot")
    agg[(lib.srgan_prof_kernel_name(kid.value).decode(), round(fl.value / 1e9, 2))].append(ms.value)
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in agg.values())
print(f"GEMM launches total {tot:.2f} ms")
for (name, gf), v in rows[:int(os.environ.get("ROWS", "60"))]:
    print(f"{name:32s} {gf:8.2f} GF x{len(v):3d}  avg {1e3*sum(v)/len(v):8.1f} us  {gf/ (sum(v)/len(v)):7.1f} TF   total {sum(v):6.2f} ms")

for k, c in sorted(_shapes.items(), key=lambda kv: -kv[1]):
    n, hi, wi, i, ho, wo, o, kh, kw, st, pad = k
    print(f"desc x{c:3d}  n {n} in {hi}x{wi}x{i} out {ho}x{wo}x{o} k {kh}x{kw} s{st} p{pad}  {2e-9*n*ho*wo*o*kh*kw*i:8.2f} GF")

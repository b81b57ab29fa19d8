"""bf16 mode, graph replay, full-width step: N steps -- finite losses, no memory growth, losses track the fp32 run."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import numpy as np, torch
import bench
from srgan_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = 32
dev = torch.device("cuda", 0)
def run(dt):
    ops.set_compute_dtype(dt)
    sg = bench.build_trainer(128, B, 5, dev)
    sg.enable_graph()
    torch.manual_seed(123)
    out, mem = [], []
    for s in range(N):
        x, src, tgt = bench.synthetic_batch(B, 128, 4, seed=s % 16)
        r = sg.train(x.to(dev), {"source": src.to(dev), "target": tgt})
        if s % 20 == 0 or s == N - 1:
            out.append([float(v) for v in r])
            torch.cuda.synchronize(); mem.append(round(torch.cuda.memory_allocated() / 2**20))
    ops.invalidate_packed()
    return np.array(out), mem
a, ma = run("bf16")
b, mb = run("fp32")
ops.set_compute_dtype("fp32")
print("bf16 finite:", bool(np.isfinite(a).all()), "mem MiB:", ma[1], "->", ma[-1])
print("fp32 finite:", bool(np.isfinite(b).all()), "mem MiB:", mb[1], "->", mb[-1])
rel = np.abs(a - b) / np.abs(b)
for i in range(len(a)):
    print(i * 20 if i < len(a) - 1 else N - 1, ["%.4f" % v for v in a[i]], ["%.4f" % v for v in b[i]], "rel", ["%.3f" % v for v in rel[i]])

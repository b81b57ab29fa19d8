"""Soak: N steps of the full-width step in graph mode vs eager -- identical losses at every step, no memory growth."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import numpy as np, torch
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
def run(graph):
    sg = bench.build_trainer(128, B, 5, dev)
    if graph:
        sg.enable_graph()
    torch.manual_seed(123)
    out, mem = [], []
    for s in range(N):
        x, src, tgt = bench.synthetic_batch(B, 128, 4, seed=s)
        r = sg.train(x.to(dev), {"source": src.to(dev), "target": tgt})
        out.append([float(v) for v in r])
        if s in (5, N - 1):
            torch.cuda.synchronize(); mem.append(torch.cuda.memory_allocated() / 2**20)
        if s == 60:                      # an epoch boundary
            sg.scheG.step(); sg.scheD.step(); sg.scheE.step()
    return np.array(out), mem, sg
a, ma, sga = run(False)
del sga; torch.cuda.empty_cache()
b, mb, sgb = run(True)
print("steps", N, "identical:", np.array_equal(a, b), "max |diff|", float(np.abs(a - b).max()))
print("eager mem MiB (step 5, last):", [round(v) for v in ma], " graph mem MiB:", [round(v) for v in mb])
print("last losses", a[-1].tolist())

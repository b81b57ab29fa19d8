"""Host time of one replayed train step (hipGraphLaunch of ~1 250 kernel nodes + the Python around it) against its GPU time: K steps
enqueued back to back without synchronising, wall time of the enqueue loop, then of the drain.  DT=fp32|bf16."""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
import bench
from srgan_amd import ops
dev = torch.device("cuda", 0)
dt = os.environ.get("DT", "bf16")
ops.set_compute_dtype(dt)
B = int(os.environ.get("B", "32"))
sg = bench.build_trainer(128, B, 5, dev)
batches = []
for s in range(4):
    x, src, tgt = bench.synthetic_batch(B, 128, 4, seed=s)
    batches.append((x.to(dev), {"source": src.to(dev), "target": tgt}))
if os.environ.get("GRAPH", "1") == "1":
    sg.enable_graph()
step = lambda b: sg.train(*b)
for i in range(4):
    step(batches[i % 4])
torch.cuda.synchronize()
K = 20
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(K):
        step(batches[i % 4])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{dt} batch {B}: enqueue {1e3 * (t1 - t0) / K:.2f} ms per step on the host, total {1e3 * (t2 - t0) / K:.2f} ms per step")

# the raw replay alone (no staging, no counters): hipGraphLaunch of the recorded step, K times back to back
rec = sg._graph.graph
segs = [g for g, _ in rec.segments]
print(f"{len(segs)} segment(s)")
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(K):
        for g in segs:
            g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{dt}: raw replay: enqueue {1e3 * (t1 - t0) / K:.2f} ms per step on the host, total {1e3 * (t2 - t0) / K:.2f} ms per step")
# one replay with an idle GPU in front and behind: the host cost of the launch call itself
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for g in segs:
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{dt}: one replay on an idle GPU: the call returns after {1e3 * (t1 - t0):.2f} ms, done after {1e3 * (t2 - t0):.2f} ms")

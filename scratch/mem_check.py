"""Device memory must not grow from step to step (pack cache, workspaces, autograd graphs)."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch, bench
sg = bench.build_trainer(128, 32, 5, torch.device("cuda"))
def batch(s):
    x, src, tgt = bench.synthetic_batch(32, 128, 4, seed=s)
    return x.cuda(), {"source": src.cuda(), "target": tgt}
for s in range(3): sg.train(*batch(s))
torch.cuda.synchronize(); a0, r0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
for s in range(3, 33): sg.train(*batch(s))
torch.cuda.synchronize(); a1, r1 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
print(f"allocated {a0/2**30:.2f} -> {a1/2**30:.2f} GiB, reserved {r0/2**30:.2f} -> {r1/2**30:.2f} GiB, peak {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
assert a1 <= a0 * 1.02 + (64 << 20), "memory grows"

"""Cost of one kernel node in a replayed hipGraph: a chain of N tiny dependent kernels (x += 1 on 64 floats) captured once and
replayed; and the same chain with a 1 M-element tensor (a ~5 us kernel) for the exposed-gap part."""
import torch, time
torch.cuda.init()
for numel in (64, 1 << 20):
    x = torch.zeros(numel, device="cuda")
    for n in (200, 1000):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): x.add_(1.0)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(n): x.add_(1.0)
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): g.replay()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 10
        # the kernel alone, eager back to back
        a.record()
        for _ in range(n): x.add_(1.0)
        b.record(); torch.cuda.synchronize()
        te = a.elapsed_time(b)
        print(f"numel {numel:8d}  nodes {n:5d}  replay {t*1e3/n:6.2f} us per node   eager {te*1e3/n:6.2f} us per launch")

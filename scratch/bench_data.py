"""Throughput of the GPU input transform (crop + PIL-exact resize + flip + ToTensor + MinMax) on CelebA-shaped batches,
beside the same transform through Pillow on one host core."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import numpy as np, torch
from srgan_amd.data import GpuTransform
from oracle import preprocess as opre
B = 32
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, 218, 178, 3), dtype=np.uint8)).pin_memory()
t = GpuTransform()
flips = t.draw_flips(B)
for _ in range(3): t(x, flips=flips)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n): t(x, flips=flips)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
xd = x.cuda(); a.record()
for _ in range(n): t(xd, flips=flips)
b.record(); torch.cuda.synchronize()
print(f"GPU transform, batch {B}: {dt*1e3:.3f} ms per batch incl. pinned H2D copy = {B/dt:.0f} images/s; device-resident input: {a.elapsed_time(b)/n:.3f} ms")
xn = x.numpy()
t0 = time.perf_counter()
for i in range(B): opre.transform_pil(xn[i], int(flips[i]))
dc = time.perf_counter() - t0
print(f"Pillow + numpy on one host core: {dc*1e3:.1f} ms per batch = {B/dc:.0f} images/s")

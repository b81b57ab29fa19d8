"""Random geometries of the layers the F(4x4,2x2) kernels serve (4x4 / stride-2 / pad-1, output map a multiple of 4, output channels
a multiple of 64 for the direction, reduce channels a multiple of 32): forward, input gradient (with and without the LeakyReLU mask
epilogue through a two-conv chain) and weight gradient against PyTorch-CPU, with the Winograd thresholds forced off and on.
Usage: fuzz.py <cases> <seed>"""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import numpy as np, torch, torch.nn.functional as F
from srgan_amd import ops
n_cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(seed)
torch.set_num_threads(16)
worst = 0.0
fails = []
for case in range(n_cases):
    ci = int(rng.choice([32, 64, 64, 96, 128, 256])); co = int(rng.choice([64, 64, 128, 192, 256]))
    h, w = 8 * int(rng.randint(1, 9)), 8 * int(rng.randint(1, 9))
    n = int(rng.randint(1, 40))
    if n * ci * h * w > 6_000_000:
        continue
    forced = bool(rng.rand() < 0.7)
    os.environ.pop("SRGAN_WINOGRAD_THRESHOLD_SCALE", None)
    if forced:
        os.environ["SRGAN_WINOGRAD_THRESHOLD_SCALE"] = "0"
    g = torch.Generator().manual_seed(case)
    x = torch.randn(n, ci, h, w, generator=g).requires_grad_(True)
    wt = (torch.randn(co, ci, 4, 4, generator=g) / (ci * 16) ** 0.5).requires_grad_(True)
    yr = F.conv2d(x, wt, None, 2, 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd, wd = x.detach().cuda().requires_grad_(True), wt.detach().cuda().requires_grad_(True)
    packed = bool(rng.rand() < 0.5)
    ops.invalidate_packed()
    ctx = ops.pack_cache() if packed else __import__("contextlib").nullcontext()
    with ctx:
        y = ops.conv2d(xd, wd, None, 2, 1)
        y.backward(gy.cuda())
    ops.invalidate_packed()
    for name, a, b, tol in (("y", y, yr, 2e-5), ("dx", xd.grad, x.grad, 2e-5), ("dw", wd.grad, wt.grad, 5e-5)):
        err = float((a.detach().cpu().double() - b.detach().double()).abs().max()); sc = float(b.detach().abs().max())
        worst = max(worst, err / sc)
        if err > 1e-6 + tol * sc:
            fails.append((case, name, (n, ci, h, w, co), forced, packed, err / sc))
print(f"{n_cases} cases, worst relative error {worst:.2e}, over tolerance: {fails}")

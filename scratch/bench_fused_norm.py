"""Kernel time of cn1+ReLU -> c2 on the trunk tensor: fused (norm writes V) vs chain (slab norm + input transform), cold inputs."""
import sys, os, ctypes
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops, _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NBUF = 10
xs = [torch.randn(B, 32, 32, 256, device="cuda").permute(0, 3, 1, 2) for _ in range(NBUF)]
w = torch.randn(256, 256, 3, 3, device="cuda") / 48
sc = torch.rand(B, 256, device="cuda") + 0.5; sh = torch.randn(B, 256, device="cuda")
def timeit(fn, n=40):
    for i in range(4): fn(xs[i % NBUF])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(xs[i % NBUF])
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
with ops.pack_cache(), torch.no_grad():
    t_norm = timeit(lambda x: ops.instance_norm_act(x, sc, sh, None, ops.ACT_RELU))
    hs = [ops.instance_norm_act(x, sc, sh, None, ops.ACT_RELU) for x in xs]
    t_conv = timeit(lambda x: ops.conv2d(x, w, None, 1, 1))
    t_chain = timeit(lambda x: ops.conv2d(ops.instance_norm_act(x, sc, sh, None, ops.ACT_RELU), w, None, 1, 1))
    t_fused = timeit(lambda x: ops.instance_norm_act_conv(x, sc, sh, w, ops.ACT_RELU))
print(f"B={B}: norm {t_norm:.1f} us, conv (transform+multiply) {t_conv:.1f} us, chain {t_chain:.1f} us, fused {t_fused:.1f} us (host-launch bound if small)")

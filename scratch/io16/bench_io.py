"""igemm16_kernel on the encoder's layers by I/O type (bf16 mode, batch B): forward time with fp32 / bf16 tensors on either side.
With the experiment build: SRGAN_IG16_EXP ablation bits (1 no source loads, 2 no weight loads, 4 no MFMAs, 8 no epilogue, 16 no K loop)."""
import sys, os
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops
ops.set_compute_dtype("bf16")
B = int(os.environ.get("B", "64")); REP = int(os.environ.get("REP", "20"))
shapes = [("Er.l0a 64->64 @62", 64, 62, 64), ("Er.l0b 64->128 @62", 64, 62, 128), ("Er.l1b 128->256 @31", 128, 31, 256), ("Er.l2b 256->512 @15", 256, 15, 512)]
def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REP): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / REP * 1e3
with ops.pack_cache():
    for name, ci, h, co in shapes:
        w = (torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5)
        x32 = torch.randn(B, h, h, ci, device="cuda").permute(0, 3, 1, 2)
        x16 = x32.to(torch.bfloat16)
        fl = 2.0 * B * h * h * co * 9 * ci
        with torch.no_grad():
            r = []
            for xin, o16 in ((x32, False), (x32, True), (x16, False), (x16, True)):
                r.append(timeit(lambda: ops.conv2d_io(xin, w, 1, ops.PAD_REFLECT, o16)))
        print(f"{name:22s} {fl/1e9:6.2f} GF | f32->f32 {r[0]:6.1f} | f32->bf16 {r[1]:6.1f} | bf16->f32 {r[2]:6.1f} | bf16->bf16 {r[3]:6.1f} us  ({fl/r[3]/1e6:.0f} TF)", flush=True)

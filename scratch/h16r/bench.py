"""halo16r_kernel at batch 32 / 64 / 128 through the C ABI entry (srgan_halo16_conv), every I/O type: 20 calls each."""
import os, sys, ctypes
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops, _lib
ops.set_compute_dtype("bf16")
dev = torch.device("cuda", 0)
lib = _lib.load()
C, S = 256, 32
w = torch.randn(C, C, 3, 3, device=dev) / 48
for B in [int(b) for b in os.environ.get("BATCHES", "32,64,128").split(",")]:
    d = ops._conv_desc(B, S, S, C, S, S, C, 3, 3, 1, 1, 0, w)
    nb = lib.srgan_conv2d_packed_bytes(ctypes.byref(d), 0, 0)
    packed = torch.empty(nb, dtype=torch.uint8, device=dev)
    _lib.check(lib.srgan_conv2d_pack(ctypes.byref(d), 0, 0, w.data_ptr(), packed.data_ptr(), nb, None), "pack")
    for in16, out16 in ((1, 1), (0, 1), (1, 0)):
        x = torch.randn(B, S, S, C, device=dev).to(torch.bfloat16 if in16 else torch.float32)
        y = torch.empty(B, S, S, C, device=dev, dtype=torch.bfloat16 if out16 else torch.float32)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            _lib.check(lib.srgan_halo16_conv(ctypes.byref(d), 0, x.data_ptr(), in16, packed.data_ptr(), None, y.data_ptr(), out16, st), "conv")
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            _lib.check(lib.srgan_halo16_conv(ctypes.byref(d), 0, x.data_ptr(), in16, packed.data_ptr(), None, y.data_ptr(), out16, st), "conv")
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        fl = 2.0 * B * S * S * C * C * 9
        print(f"B={B:4d} in16={in16} out16={out16} {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s ({fl / us / 1e6 / 2500:.3f})")

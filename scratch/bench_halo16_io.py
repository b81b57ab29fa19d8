"""halo16_kernel (residual-trunk 3x3 conv, bf16 mode) by I/O types: fp32 / bf16 source x fp32 / bf16 destination (+ fp32 skip)."""
import sys, os, ctypes
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops, _lib
lib = _lib.load()
ops.set_compute_dtype("bf16")
B = int(os.environ.get("B", 32)); H = int(os.environ.get("H", 32)); C = 256; REP = int(os.environ.get("REP", 20))
w = (torch.randn(C, C, 3, 3, device="cuda") / 48).requires_grad_(True)
d = ops._conv_desc(B, H, H, C, H, H, C, 3, 3, 1, 1, ops.PAD_ZERO, w)
x32 = torch.randn(B, H, H, C, device="cuda"); x16 = x32.to(torch.bfloat16)
y32 = torch.empty_like(x32); y16 = torch.empty_like(x16); res = torch.randn_like(x32)
st = torch.cuda.current_stream().cuda_stream
with ops.pack_cache():
    hit, _ = ops._packed(d, w, 0, ops.ACT_NONE)
    fl = 2.0 * B * H * H * C * C * 9
    for name, src, s16, r, dst, d16 in (("fp32->fp32", x32, 0, None, y32, 0), ("fp32->fp32+res", x32, 0, res, y32, 0), ("fp32->bf16", x32, 0, None, y16, 1),
                                        ("bf16->bf16", x16, 1, None, y16, 1), ("bf16->fp32", x16, 1, None, y32, 0), ("bf16->fp32+res", x16, 1, res, y32, 0)):
        def run():
            _lib.check(lib.srgan_halo16_conv(ctypes.byref(d), 0, src.data_ptr(), s16, hit.buf.data_ptr(), r.data_ptr() if r is not None else None,
                                             dst.data_ptr(), d16, st), "halo16")
        run(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(REP): run()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / REP
        print(f"{name:16s} {t*1e3:7.1f} us  {fl/t/1e9:6.1f} TF/s")

"""Instance-norm slab kernels with 16-bit I/O on the residual-trunk shape (N x 32 x 32 x 256): time and algorithmic GB/s."""
import sys, os, ctypes
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from srgan_amd import ops, _lib
lib = _lib.load()
REP = 20
def timeit(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REP): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / REP * 1e-3
for B, HW, C in ((32, 1024, 256), (64, 1024, 256), (128, 1024, 256), (32, 4096, 128), (32, 16384, 64), (16, 4096, 256)):
    n = B * HW * C
    x32 = torch.randn(B, HW, C, device="cuda"); x16 = x32.bfloat16()
    g32 = torch.randn_like(x32); g16 = g32.bfloat16(); res = torch.randn_like(x32)
    y32 = torch.empty_like(x32); y16 = torch.empty_like(x16)
    sc = torch.rand(B, C, device="cuda") + 0.5; sh = torch.randn(B, C, device="cuda")
    mean = torch.empty(B * C, device="cuda"); rstd = torch.empty_like(mean)
    dsc = torch.empty(B, C, device="cuda"); dsh = torch.empty_like(dsc)
    nb = lib.srgan_instnorm_workspace(B, HW, C); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr() if t is not None else None
    def fwd(x, x16f, r, y, y16f):
        return lambda: _lib.check(lib.srgan_instnorm_fwd_io(P(x), x16f, P(sc), P(sh), P(r), P(y), y16f, P(mean), P(rstd), B, HW, C, 1e-5, 1, 0.0, P(ws), nb, st), "f")
    def bwd(x, x16f, g, g16f, dx, dx16f):
        return lambda: _lib.check(lib.srgan_instnorm_bwd_io(P(x), x16f, P(g), g16f, P(sc), P(sh), P(mean), P(rstd), P(dx), dx16f, P(dsc), P(dsh), B, HW, C, 1, 0.0, P(ws), nb, st), "b")
    rows = [("fwd bf16->bf16", fwd(x16, 1, None, y16, 1), 4), ("fwd bf16->fp32+res", fwd(x16, 1, res, y32, 0), 10), ("fwd fp32->fp32", fwd(x32, 0, None, y32, 0), 8),
            ("fwd fp32->bf16", fwd(x32, 0, None, y16, 1), 6),
            ("bwd x16 g16 -> dx16", bwd(x16, 1, g16, 1, y16, 1), 6), ("bwd x16 g32 -> dx16", bwd(x16, 1, g32, 0, y16, 1), 8), ("bwd x32 g32 -> dx32", bwd(x32, 0, g32, 0, y32, 0), 12)]
    two = "" if lib.srgan_instnorm_slab_applicable(B, HW, C) else " (two-pass: x read twice)"
    print(f"N={B} HW={HW} C={C}{two}")
    for name, fn, bpe in rows:
        t = timeit(fn)
        print(f"   {name:22s} {t*1e6:7.1f} us  {n*bpe/t/1e12:5.2f} TB/s (one read of each input, one write)")

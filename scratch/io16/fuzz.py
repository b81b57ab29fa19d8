"""Random geometries of the stride-1 convolutions with 16-bit I/O (bf16 mode): python scratch/io16/fuzz.py <n> <seed>.
Every dtype pair, forward + input gradient + weight gradient against the fp32 convolution of the bf16-rounded operands, and the
bit-identity with the fp32-tensor path on rounded operands (tests/test_ops_gpu.py::test_generic_conv_io_every_dtype_pair)."""
import os, sys, random
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from tests import test_ops_gpu as t
from srgan_amd import ops
n, seed = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed)
bad = skipped = 0
for i in range(n):
    ci = rng.choice([64, 128, 192, 256, 512])
    co = rng.choice([64, 128, 192, 256, 320, 512])
    h, w = rng.randint(2, 40), rng.randint(2, 40)
    nb = rng.randint(1, 6)
    if nb * h * w * max(ci, co) > 3_000_000:
        nb = 1
    pm = rng.choice(["reflect", "zeros"])
    case = (nb, ci, co, h, w, pm)
    in16, out16 = rng.random() < 0.5, rng.random() < 0.5
    try:
        t.test_generic_conv_io_every_dtype_pair(ops, case, in16, out16)
    except BaseException as e:
        if "skip" in type(e).__name__.lower() or "Skipped" in str(type(e)):
            skipped += 1
            continue
        bad += 1
        print("FAIL", case, in16, out16, str(e)[:200], flush=True)
print("done: %d cases, %d skipped, failures: %d" % (n, skipped, bad))

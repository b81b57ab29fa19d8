"""One-off: many more random bf16-mode conv geometries than the committed test (python scratch/fuzz_conv_bf16.py <n> <seed>)."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from tests import test_ops_gpu as t
from srgan_amd import ops
n, seed = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for case in t._random_bf16_conv_cases(n, seed):
    try:
        t.test_conv2d_bf16_random_geometries(ops, case)
    except Exception as e:
        bad += 1
        print("FAIL", case, str(e)[:200], flush=True)
print("done, failures:", bad)

"""One-off: many more random conv geometries than the committed test (python scratch/fuzz_conv.py <n> <seed>)."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from tests import test_ops_gpu as t
from srgan_amd import ops
n, seed = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for mode in ("default", "forced"):
    if mode == "forced":
        os.environ["SRGAN_WINOGRAD_THRESHOLD_SCALE"] = "0"
    for case in t._random_conv_cases(n, seed):
        try:
            t.test_conv2d_random_geometries(ops, case, mode)
        except Exception as e:
            bad += 1
            print("FAIL", mode, case, str(e)[:200], flush=True)
    os.environ.pop("SRGAN_WINOGRAD_THRESHOLD_SCALE", None)
print("done, failures:", bad)

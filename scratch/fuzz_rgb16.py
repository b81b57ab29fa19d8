"""One-off: the bf16 mode's 7x7 RGB layers (ops.conv2d_act_io with k = 7: rgbin16 / rgbout16 / rgb_wgrad16 kernels, 64-channel side
fp32 or bf16) on random map sizes and batches, each product against the fp32 convolution of the bf16-rounded operands
(python scratch/fuzz_rgb16.py <n> <seed>)."""
import os, sys, random
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
from tests import test_ops_gpu as t
from srgan_amd import ops
n, seed = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed)
bad = 0
for it in range(n):
    nb = rng.choice([1, 2, 3, 5, 8])
    h = 16 * rng.randint(2, 12)
    w = 32 * rng.randint(2, 8)
    layer = rng.choice(["in", "out"])
    wide = rng.choice([False, True])
    try:
        t.test_rgb_layers_with_a_bf16_64_channel_side(ops, nb, h, w, layer, wide)
    except Exception as e:
        bad += 1
        print("FAIL", (nb, h, w, layer, wide), str(e)[:300], flush=True)
print("done, failures:", bad)

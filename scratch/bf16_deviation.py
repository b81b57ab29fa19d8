"""Measured deviation of the bf16 mode from the reference's fp32 trajectories (the bounds asserted in tests/test_train_gpu.py):
tier-T recipes (configs[2] / [4]) and the full-size two-step trajectory."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import numpy as np
from tests.test_train_gpu import run_hip
from srgan_amd import ops
gd = os.path.join(_R, "tests", "golden")
cases = [("train_T_b4_k2", "T", 4, 2, 3, False, 128), ("train_T_b4_k2_pretrainedE", "T", 4, 2, 2, True, 128),
         ("train_T256_b2_k2", "T256", 2, 2, 2, False, 256), ("train_F_b2_k1", "F", 2, 1, 2, False, 128)]
for name, tier, batch, k, steps, pre, size in cases:
    gold = np.load(os.path.join(gd, name + ".npz"))["losses"]
    ops.set_compute_dtype("bf16")
    try:
        _, traj = run_hip(tier, batch, k, steps, seed=0, pretrained_e=pre, size=size)
    finally:
        ops.set_compute_dtype("fp32")
    dev = np.abs(np.asarray(traj) - gold) / np.abs(gold)
    print(f"{name:28s} max relative deviation {dev.max():.3e}   per loss {np.round(dev.max(0), 5)}")

"""Writes tests/golden/traj20_T.npz: the CPU oracle's losses over the 20-step tier-T trajectories that
tests/test_train_gpu.py::test_twenty_step_bf16_trajectory_vs_fp32_oracle compares the bf16 mode with (k = 5; the pretrained-encoder
recipe with k = 2), so that the GPU test does not spend 7 minutes of its box time in the CPU oracle.  The oracle is the pinned
restatement (tests/test_oracle_golden.py); the same function runs it live for the fp32 trajectory test.
    python tests/golden/make_traj20.py            (about 8 minutes on 8 cores)"""
import os, sys
import numpy as np
import torch
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _R)
from tests.common import oracle_twenty_steps


if __name__ == "__main__":
    torch.set_num_threads(8)
    out = {}
    for name, k, pre in (("k5", 5, False), ("k2_pretrainedE", 2, True)):
        out[name] = oracle_twenty_steps(k, pre)[1]
        print(name, out[name][-1])
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traj20_T.npz"), **out)

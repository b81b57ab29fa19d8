"""Writes tests/golden/bench_first_step.json: the CPU oracle's [errG, errD, errE] for bench.py's FIRST step -- the same
default-initialised full-width networks (seed 0), the same histogram target, the same synthetic batch and the same
CPU-generator noise stream as `python bench.py --gpus 1` (BASELINE configs[1]: 128x128, bs=32, k=5, fp32).  bench.py
compares its own first step with these numbers on every run, so the headline workload itself is parity-checked by the
driver's bench command, not only by the test-suite.

Needs neither a GPU nor the reference: the oracle (oracle/, pinned to the reference by tests/test_oracle_golden.py) is the
checker.  Run from the repo root:  python tests/golden/make_bench_first_step.py  (about 2 minutes on 8 cores).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "style-restricted_gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import trainer as ot  # noqa: E402


def first_step(size, batch, k):
    G, D, E = bench.build_nets(size, "cpu")                  # torch.manual_seed(0) + default init, as the bench
    PG, PD, PE = ({n: v.detach().clone() for n, v in net.state_dict().items()} for net in (G, D, E))
    orc = ot.SRGANOracle(PG, PD, PE, bench.LBD, k, np.eye(4), batch, "mu", 8)     # draws the histogram target next, as the trainer
    torch.manual_seed(1000)                                  # rank 0's noise stream
    x, src, tgt = bench.synthetic_batch(batch, size, 4, seed=10_000)
    t0 = time.time()
    out = [float(v) for v in orc.train(x, {"source": src, "target": tgt})]
    return out, time.time() - t0


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 1)
    configs = {}
    for size, batch, k in ((128, 32, 5),):
        losses, dt = first_step(size, batch, k)
        configs[f"{size}x{size}_b{batch}_k{k}"] = {"losses": losses, "oracle_seconds": round(dt, 1)}
        print(size, batch, k, losses, f"{dt:.1f}s")
    rec = {"what": "CPU-oracle losses [errG, errD, errE] of bench.py's first train step (seed 0 default init, rank 0)",
           "band": 2e-3, "band_note": "north_star: CPU-reference loss parity within 1e-3 relative; 2x head-room for the "
                                      "hist*100-dominated errE whose terms partly cancel",
           "torch": torch.__version__, "configs": configs}
    with open(os.path.join(ROOT, "tests", "golden", "bench_first_step.json"), "w") as f:
        json.dump(rec, f, indent=1)

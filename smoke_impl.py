"""Body of __graft_entry__.smoke(): one tiny SRGAN train step on cuda:0, checked against the CPU oracle."""
import numpy as np
import torch
import torch.nn as nn


def run():
    from oracle import trainer as otrainer
    from tests.common import build_hip_nets, oracle_params
    from srgan_amd.trainer import SRGAN_training
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    torch.cuda.set_device(0)
    PG, PD, PE = oracle_params("T")
    torch.manual_seed(1)
    orc = otrainer.SRGANOracle(PG, PD, PE, otrainer.DEFAULT_LBD, 2, np.eye(4), 4, "mu", 8)
    x, label = otrainer.synthetic_batch(4, 128, 4, seed=3)
    ref = [float(v) for v in orc.train(x, label)]
    G, D, E = build_hip_nets("T")
    torch.manual_seed(1)
    sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), 2,
                        "cuda", np.eye(4), 4, "mu", 8)
    sg.opt_sche_initialization()
    out = [float(v) for v in sg.train(x.cuda(), {"source": label["source"].cuda(), "target": label["target"]})]
    np.testing.assert_allclose(out, ref, rtol=1e-3)
    print("smoke ok: errG/errD/errE HIP", out, "oracle", ref)


if __name__ == "__main__":
    run()

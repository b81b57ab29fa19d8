"""Debug: 2 ranks over gloo on one GPU; eager vs injected-failure fallback: per-step losses and parameter checksums of rank 0."""
import os, sys, socket
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import numpy as np, torch, torch.multiprocessing as mp, torch.nn as nn

def work(rank, world, port, q, graph, inject):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      SRGAN_DP_DEVICE="0", SRGAN_DP_BACKEND="gloo")
    if inject: os.environ["SRGAN_TEST_FAIL_CAPTURE"] = inject
    from srgan_amd import dp
    dp.init_from_env()
    from tests import test_dp_gpu as T
    from oracle import trainer as otrainer
    from tests.common import build_hip_nets
    from srgan_amd.trainer import SRGAN_training
    from srgan_amd import optim as hoptim
    G, D, E = build_hip_nets("T")
    torch.manual_seed(0)
    opts = [hoptim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4, betas=(0.5, 0.999), eps=1e-2) for net in (G, D, E)]
    sg = SRGAN_training([G, D, E], opts, [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), 2, "cuda", np.eye(4), 4, "mu", 8)
    sg.opt_sche_initialization()
    if graph: sg.enable_graph()
    sg.noise_fn = T._noise_source(rank, world)
    out = []
    for s in range(3):
        x, label = otrainer.synthetic_batch(4, 128, 4, seed=300 + s)
        sl = slice(rank * 2, rank * 2 + 2)
        l = [float(v) for v in sg.train(x[sl].cuda(), {"source": label["source"][sl].cuda(), "target": label["target"][sl]})]
        cs = [float(sum(p.detach().double().abs().sum() for p in net.parameters())) for net in (sg.G, sg.D, sg.E)]
        steps = [int(next(iter(o.state.values()))["step"]) if len(o.state) else -1 for o in (sg.optG, sg.optD, sg.optE)]
        out.append((l, cs, steps, float(torch.rand(1))))
        if rank == 0 and s == 1: print('TERMS', 'graph' if graph else 'eager', inject, {k: round(float(v), 6) for k, v in sg.loss_terms.items()}, flush=True)
    if rank == 0: q.put(out)
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    res = {}
    for name, graph, inject in (("eager", False, None), ("fallback", True, "1:before"), ("fallback0", True, "0:before")):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        q = ctx.Queue()
        ps = [ctx.Process(target=work, args=(r, 2, port, q, graph, inject)) for r in range(2)]
        [p.start() for p in ps]
        res[name] = q.get(timeout=90)
        [p.join() for p in ps]
    for name, r in res.items():
        for s, (l, cs, steps, rnd) in enumerate(r):
            print(name, s, ["%.6f" % v for v in l], ["%.6f" % v for v in cs], steps, "%.6f" % rnd)

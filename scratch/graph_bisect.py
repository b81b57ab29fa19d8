"""Which part of the train step breaks hipGraph capture?  One stage per process: python scratch/graph_bisect.py <stage>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "style-restricted_gan_amd")]
import faulthandler; faulthandler.enable()
import numpy as np, torch, torch.nn as nn
from srgan_amd import ops, model, optim
stage = sys.argv[1]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def capture(fn, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    print(stage, "captured", flush=True)
    g.replay(); g.replay()
    torch.cuda.synchronize()
    print(stage, "replayed ok", out if not torch.is_tensor(out) else float(out.float().sum()), flush=True)


if stage == "upload":
    dst = torch.zeros(4096, dtype=torch.uint8, device=dev)
    capture(lambda: ops.upload_small(bytes(range(256)) * 12, dev, out=dst))
elif stage == "upload_small":
    dst = torch.zeros(4096, dtype=torch.uint8, device=dev)
    capture(lambda: ops.upload_small(bytes(range(128)), dev, out=dst))
elif stage == "conv_fwd":
    w = torch.randn(64, 64, 3, 3, device=dev); x = ops.to_nhwc(torch.randn(2, 64, 32, 32, device=dev))
    capture(lambda: ops.conv2d(x, w, None, 1, 1))
elif stage == "conv_fwd_bwd":
    w = torch.randn(64, 64, 3, 3, device=dev, requires_grad=True); x = ops.to_nhwc(torch.randn(2, 64, 32, 32, device=dev))
    def f():
        w.grad = None
        ops.conv2d(x, w, None, 1, 1).square().mean().backward()
        return w.grad
    capture(f)
elif stage == "adam":
    p = torch.nn.Parameter(torch.randn(1000, device=dev)); q = torch.nn.Parameter(torch.randn(50, 3, device=dev))
    opt = optim.Adam([p, q], lr=1e-3)
    def f():
        p.grad = torch.ones_like(p); q.grad = torch.ones_like(q)
        opt.step()
        return p
    capture(f)
elif stage in ("G_fwd", "G_fwd_bwd", "G_fwd_bwd_nomulti"):
    if stage.endswith("nomulti"):
        os.environ["SRGAN_NO_CBIN_MULTI"] = "1"
    G = model.SingleGenerator(3, 8, 2, 2, 1, "instance", num_con=12).to(dev)
    x = ops.to_nhwc(torch.randn(2, 3, 64, 64, device=dev)); c = torch.randn(2, 12, device=dev)
    def f():
        if stage == "G_fwd":
            with torch.no_grad():
                return G(x, c)
        for p in G.parameters():
            p.grad = None
        with ops.pack_cache(refresh_on_entry=False):
            G(x, c).square().mean().backward()
        return G.down_convs[0].weight.grad
    capture(f)
elif stage == "D_fwd_bwd":
    D = model.SingleDiscriminator_solo_multi(3, 8, 2, 4, "instance", 4).to(dev)
    x = ops.to_nhwc(torch.randn(2, 3, 128, 128, device=dev))
    def f():
        for p in D.parameters():
            p.grad = None
        outs, logits = D.forward_logits(x)
        (sum(ops.mse_const(o, 1.0, 0.5) for o in outs) + sum(ops.softmax_mse(z, torch.zeros(2, dtype=torch.int64, device=dev), 0.5)[0] for z in logits)).backward()
        return D.last_layer1.weight.grad
    capture(f)
elif stage == "E_fwd_bwd":
    E = model.Encoder(3, 8, 8, 4, "instance", 4, dev).to(dev)
    x = ops.to_nhwc(torch.randn(2, 3, 128, 128, device=dev))
    def f():
        for p in E.parameters():
            p.grad = None
        feat = E.features(x)
        mu = E.fcmean(feat)
        tot, parts, _ = ops.latent_losses(mu, 2, torch.full((50,), 0.02, device=dev), 10.0, 100.0, 100.0)
        tot.backward()
        return mu
    capture(f)
elif stage == "step":
    from oracle import trainer as otrainer
    from tests.common import build_hip_nets
    from srgan_amd.trainer import SRGAN_training
    G, D, E = build_hip_nets("T")
    sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                        "cuda", np.eye(4), 4, "mu", 8)
    sg.opt_sche_initialization()
    sg.enable_graph()
    for s in range(3):
        x, label = otrainer.synthetic_batch(4, 128, 4, seed=s)
        print([float(v) for v in sg.train(x.cuda(), {"source": label["source"].cuda(), "target": label["target"]})], flush=True)
    print("step ok")
elif stage.startswith("part_"):
    from oracle import trainer as otrainer
    from tests.common import build_hip_nets
    from srgan_amd.trainer import SRGAN_training
    G, D, E = build_hip_nets("T")
    sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), 1,
                        "cuda", np.eye(4), 4, "mu", 8)
    sg.opt_sche_initialization()
    sg.enable_graph()
    x, label = otrainer.synthetic_batch(4, 128, 4, seed=0)
    lab = {"source": label["source"].cuda(), "target": label["target"]}
    sg.train(x.cuda(), lab)           # eager warm-up
    gr = sg._graph
    gr._stage(x.cuda(), lab)
    sg.label = lab
    sg.target_image = sg.recon_image = sg.c_rand = None
    sg.loss_terms = {}
    import gc; gc.collect()

    def body():
        sg._g_active = True
        gr._noise_i, gr._onehot = 0, {}
        sg.source_image = gr.x
        with ops.pack_cache(refresh_on_entry=False):
            oh = sg._onehot("target")
            z = sg._noise("randn", 4)
            if stage == "part_onehot":
                return oh.sum() + z.sum()
            fake = (sg.G(gr.x, torch.cat([oh, z], 1)), z)
            if stage == "part_gfwd":
                return fake[0].sum()
            errD = sg.update_D(_fake=fake)
            if stage == "part_D":
                return errD.detach()
            if stage == "part_GE":
                out = sg.update_GandE()
                return out[0]
    capture(body, warm=0)
elif stage == "after":
    from oracle import trainer as otrainer
    from tests.common import build_hip_nets
    from srgan_amd.trainer import SRGAN_training
    G, D, E = build_hip_nets("T")
    sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], dict(otrainer.DEFAULT_LBD), 2,
                        "cuda", np.eye(4), 4, "mu", 8)
    sg.opt_sche_initialization()
    sg.enable_graph()
    def run(b, seed):
        x, label = otrainer.synthetic_batch(b, 128, 4, seed=seed)
        out = sg.train(x.cuda(), {"source": label["source"].cuda(), "target": label["target"]})
        torch.cuda.synchronize()
        print(b, sg.graph_active, [float(v) for v in out], flush=True)
    for s in range(3):
        run(4, s)
    mode = sys.argv[2] if len(sys.argv) > 2 else "small"
    if mode == "small":
        run(2, 10)
    elif mode == "same_eager":
        g = sg._graph; sg._graph = None
        run(4, 10)
        sg._graph = g
    run(4, 11)
    run(4, 12)
    print("after ok")

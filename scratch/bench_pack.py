"""Time of the per-optimiser-step weight repack (ops.refresh_packed -> pack_multi_kernel) per network, fp32 and bf16 mode."""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
import bench
from srgan_amd import ops
dev = torch.device("cuda", 0)
for dt in ("fp32", "bf16"):
    ops.set_compute_dtype(dt)
    sg = bench.build_trainer(128, 32, 5, dev)
    x, src, tgt = bench.synthetic_batch(32, 128, 4, seed=1)
    with ops.pack_cache():
        sg.label = {"source": src.to(dev), "target": tgt}
        sg.loss_terms = {}
        sg.source_image = ops.to_nhwc(x.to(dev))
        sg.UnrolledUpdate()
        torch.cuda.synchronize()
        for name, opt in (("G", sg.optG), ("D", sg.optD), ("E", sg.optE)):
            params = sg._opt_params(opt)
            for _ in range(2): ops.refresh_packed(params, force=True)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): ops.refresh_packed(params, force=True)
            b.record(); torch.cuda.synchronize()
            print(f"{dt} repack {name}: {a.elapsed_time(b) / 10 * 1e3:8.1f} us   cached operands {len(ops._pack_cache)}")
    ops.invalidate_packed()
ops.set_compute_dtype("fp32")

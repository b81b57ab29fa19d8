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

# per-operand detail (DETAIL=1): every cached operand of the last mode packed alone, largest first
if os.environ.get("DETAIL"):
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
            rows = []
            for key, h in list(ops._pack_cache.items()):
                for _ in range(2): ops._pack_one(h)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10): ops._pack_one(h)
                b.record(); torch.cuda.synchronize()
                d = h.desc
                rows.append((a.elapsed_time(b) * 100, h.weight.numel(), h.buf.numel() * h.buf.element_size(), h.kind, d.I, d.O, d.kh, d.stride))
            rows.sort(reverse=True)
            print(f"{dt}: {len(rows)} operands, sum of single packs {sum(r[0] for r in rows):.0f} us")
            for us, nw, nb, kind, i, o, k, s in rows[:40]:
                print(f"  {us:7.1f} us  weights {nw * 4 / 1e6:7.2f} MB -> operand {nb / 1e6:7.2f} MB  kind {kind}  {i:4d} -> {o:4d} k{k} s{s}")
        ops.invalidate_packed()
    ops.set_compute_dtype("fp32")

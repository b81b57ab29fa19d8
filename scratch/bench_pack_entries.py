"""Which operand makes the per-optimiser-step repack (pack_multi_kernel, 256 workgroups per operand) slow?  Every operand of one
network packed ALONE through the multi-pack launch (a one-record table), largest first; NET=G|D|E, DT=fp32|bf16."""
import os, sys, ctypes
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "style-restricted_gan_amd"))
import torch
import bench
from srgan_amd import ops, _lib
dev = torch.device("cuda", 0)
dt = os.environ.get("DT", "bf16")
ops.set_compute_dtype(dt)
lib = _lib.load()
sg = bench.build_trainer(128, 32, 5, dev)
x, src, tgt = bench.synthetic_batch(32, 128, 4, seed=1)
with ops.pack_cache():
    sg.label = {"source": src.to(dev), "target": tgt}
    sg.loss_terms = {}
    sg.source_image = ops.to_nhwc(x.to(dev))
    sg.UnrolledUpdate()
    torch.cuda.synchronize()
    opt = {"G": sg.optG, "D": sg.optD, "E": sg.optE}[os.environ.get("NET", "D")]
    ids = {id(p) for p in sg._opt_params(opt)}
    nb = lib.srgan_pack_entry_bytes()
    rows = []
    whole = bytearray()
    for key, h in ops._pack_cache.items():
        if key[0] not in ids or h.weight is None:
            continue
        rec = (ctypes.c_char * nb)()
        if lib.srgan_conv2d_pack_entry(ctypes.byref(h.desc), h.kind, h.act, ops._ptr(h.weight), ops._ptr(h.buf), ctypes.byref(rec)) != 0:
            continue
        whole += bytes(rec)
        tab = ops.upload_small(bytes(rec), dev)
        for _ in range(3):
            lib.srgan_conv2d_pack_multi(ops._ptr(tab), 1, ops._stream())
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            lib.srgan_conv2d_pack_multi(ops._ptr(tab), 1, ops._stream())
        b.record(); torch.cuda.synchronize()
        d = h.desc
        rows.append((a.elapsed_time(b) * 50, h.kind, d.I, d.O, d.kh, d.stride, d.Hi, h.weight.numel() * 4 / 1e6, h.buf.numel() / 1e6))
    n = len(rows)
    tab = ops.upload_small(bytes(whole), dev)
    for _ in range(3):
        lib.srgan_conv2d_pack_multi(ops._ptr(tab), n, ops._stream())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        lib.srgan_conv2d_pack_multi(ops._ptr(tab), n, ops._stream())
    b.record(); torch.cuda.synchronize()
    print(f"{dt} {os.environ.get('NET', 'D')}: {n} operands in one launch {a.elapsed_time(b) * 50:.1f} us; alone, sum {sum(r[0] for r in rows):.0f} us")
    for us, kind, i, o, k, s, hi, wmb, omb in sorted(rows, reverse=True):
        print(f"  {us:7.1f} us  kind {kind}  {i:4d} -> {o:4d} k{k} s{s} @ {hi:3d}   weights {wmb:6.2f} MB -> operand {omb:6.2f} MB")
ops.set_compute_dtype("fp32")

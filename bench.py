#!/usr/bin/env python3
"""Benchmark of the SRGAN G+D+E train step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" = one ``SRGAN_training.train`` call (k=5 D updates + the two-phase G/E update) on one synthetic
CelebA-shaped batch.  Workload at N=1 = BASELINE.json configs[1]: SRGAN-nopretrain, 128x128, bs=32/GPU, fp32,
notebook hyper-parameters (05-train cells 13/16).  Weak scaling: 32 images per GPU, global batch 32*N.
Prints ONE JSON line on rank 0 with ``roofline`` (dominant kernel, HIP-event timed inside the timed region)
and, at N=1, ``cpu_baseline`` (the CPU oracle timed on the host cores on a bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "style-restricted_gan_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402

LBD = {"class": 1.0, "cycle": 5.0, "idt": 5.0, "reg": 0.5, "idt_reg": 0.5, "KL": 0.0,
       "batch_KL": 10.0, "corr_enc": 100.0, "hist": 100.0}          # 05-train cell 16
GFLOP_PER_IMAGE = {128: 412.46, 256: 1677.69}                        # SURVEY.md 8d (k=5, E trainable)
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2516.6}                                         # MI355X dense matrix peak, MI355X_MICROARCH.md


def synthetic_batch(batch, size, n_class, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    src = torch.randint(0, n_class, (batch,), generator=g)
    tgt = (src + torch.randint(1, n_class, (batch,), generator=g)) % n_class
    return x, src, tgt


def build_trainer(size, global_batch, k, device):
    from srgan_amd import model
    from srgan_amd.trainer import SRGAN_training
    torch.manual_seed(0)             # identical replicas on every rank (default PyTorch init, as the reference)
    np.random.seed(0)
    d_cls = 4 if size == 128 else 5
    G = model.SingleGenerator(3, 64, 2, 2, 6, "instance", num_con=12)
    D = model.SingleDiscriminator_solo_multi(3, 64, 2, d_cls, "instance", 4)
    E = model.Encoder(3, 8, 64, 4, "instance", 4, device)
    sg = SRGAN_training([G, D, E], [None, None, None], [nn.MSELoss(), nn.MSELoss()], dict(LBD), k, device,
                        np.eye(4), global_batch, "mu", 8)
    sg.opt_sche_initialization()
    return sg


def host_cores():
    """Threads we may really use: the affinity mask, capped at the GPU box's CPU share for one GPU (16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("SRGAN_BENCH_CPU_THREADS", "16"))))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(size, k):
    """The CPU oracle (oracle/, a port of the reference's PyTorch-CPU path) on a bounded sample."""
    from oracle import params as op, trainer as ot
    cores = host_cores()
    torch.set_num_threads(cores)
    spec = (op.generator_spec(3, 64, 2, 2, 6, 12), op.discriminator_spec(3, 64, 2, 4, 4), op.encoder_spec(3, 8, 64, 4, 4))
    PG, PD, PE = (op.fill(s, i) for i, s in enumerate(spec))
    torch.manual_seed(0)
    batch = 8
    orc = ot.SRGANOracle(PG, PD, PE, LBD, k, np.eye(4), batch, "mu", 8)
    orc_w = ot.SRGANOracle(PG, PD, PE, LBD, 1, np.eye(4), 2, "mu", 8)
    x2, l2 = ot.synthetic_batch(2, size, 4, seed=2)
    log(f"cpu baseline: warm-up on {cores} threads")
    orc_w.train(x2, l2)                                   # warm-up (thread pool, allocator), k=1, 2 images
    log("cpu baseline: timed steps")
    steps, dt = 0, 0.0
    while dt < 12.0 and steps < 6:                        # bounded sample: ~10-30 s of CPU work
        x, label = ot.synthetic_batch(batch, size, 4, seed=3 + steps)
        t0 = time.perf_counter()
        orc.train(x, label)
        dt += time.perf_counter() - t0
        steps += 1
    return {"value": round(batch * steps / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} full train steps (k={k}, fp32, same networks/losses) of the CPU oracle at batch {batch} "
                      f"({size}x{size}) after a k=1 batch-2 warm-up; {dt:.1f} s on {cores} threads"}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed PMC summary (profiles/r*_pmc_traffic.json, collected
    with separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same bench command), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        entry = json.load(open(files[-1]))["kernels"].get(kernel_name)
        return entry["hbm_bytes_per_launch"] if entry else None
    except (OSError, ValueError, KeyError):
        return None


GD_GFLOP_PER_IMAGE = {128: 60.54, 256: 243.86}        # (3 F_G - f_G) + (3 F_D - f_D), SURVEY.md 8d


def micro_gd(args):
    """One generator forward + full backward and one discriminator forward + full backward on a batch of 32 (inputs do not
    require gradients, so the first-layer input gradients are skipped, as in the SURVEY formula).  Single GPU."""
    from srgan_amd import _lib, model, ops
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    device = torch.device("cuda", 0)
    _lib.load()
    if args.dtype == "bf16":
        ops.set_compute_dtype("bf16")
    torch.manual_seed(0)
    B, S = args.batch_per_gpu, args.size
    G = model.SingleGenerator(3, 64, 2, 2, 6, "instance", num_con=12).to(device)
    D = model.SingleDiscriminator_solo_multi(3, 64, 2, 4 if S == 128 else 5, "instance", 4).to(device)
    x = (torch.rand(B, 3, S, S, device=device) * 2 - 1)
    c = torch.cat([torch.eye(4, device=device)[torch.randint(0, 4, (B,), device=device)], torch.randn(B, 8, device=device)], 1)

    def step():
        for p in list(G.parameters()) + list(D.parameters()):
            p.grad = None
        with ops.pack_cache():
            G(x, c).square().mean().backward()
            outs, cls = D(x)
            (sum(o.square().mean() for o in outs) + sum(q.square().mean() for q in cls)).backward()

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    gf = GD_GFLOP_PER_IMAGE.get(S)
    value = B / dt
    peak = PEAK_TFLOPS["f32" if args.dtype == "f32" else "bf16"]
    tf = value * gf / 1e3 if gf else None
    print(json.dumps({
        "metric": f"images/sec G+D forward-backward, {S}x{S} bs={B} (micro-benchmark, SURVEY 8d)", "value": round(value, 2),
        "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt, 3),
        "higher_is_better": True, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "one G forward + backward and one D forward + backward (all parameter gradients), default PyTorch "
                               "init, first-layer input gradients skipped", "gflop_per_image_algorithmic": gf},
        "roofline": {"bound": "mfma", "achieved": round(tf, 2) if tf else None, "peak": peak, "unit": "TFLOP/s",
                     "frac": round(tf / peak, 4) if tf else None,
                     "note": "whole micro-benchmark: algorithmic FLOPs (SURVEY 8d) x images/s against the dense MFMA peak of the "
                             "dtype; includes norm / pointwise / reduction kernels; Winograd layers execute 2.25x fewer MFMA FLOPs "
                             "than counted, so the algorithmic fraction can exceed 1"}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 (default, BASELINE configs[1]) or the bf16 MFMA compute mode of configs [2]-[4] (never the headline)")
    ap.add_argument("--micro", choices=["gd"], default=None,
                    help="gd: the 'G+D forward-backward' micro-benchmark of SURVEY.md 8d (north_star's >= 70 %% MFMA-roofline target) "
                         "instead of the full train step")
    args = ap.parse_args()
    if args.micro == "gd":
        return micro_gd(args)

    from srgan_amd import _lib, dp
    rank, world, device = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    lib = _lib.load()
    if args.dtype == "bf16":
        from srgan_amd import ops as _ops
        _ops.set_compute_dtype("bf16")
    B = args.batch_per_gpu
    sg = build_trainer(args.size, B * world, args.k, device)
    torch.manual_seed(1000 + rank)           # per-rank noise stream after identical construction

    batches = []
    for s in range(args.steps + args.warmup):
        x, src, tgt = synthetic_batch(B, args.size, 4, seed=10_000 * (rank + 1) + s)
        batches.append((x.to(device), {"source": src.to(device), "target": tgt}))   # inputs resident in HBM
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: inputs resident, warm-up x{args.warmup}")
    for s in range(args.warmup):
        sg.train(*batches[s])
        torch.cuda.synchronize()
        log(f"warm-up step {s} done")
    barrier()
    lib.srgan_prof_enable(1)
    t0 = time.perf_counter()
    last = None
    for s in range(args.warmup, args.warmup + args.steps):
        last = sg.train(*batches[s])
    barrier()
    elapsed = time.perf_counter() - t0
    lib.srgan_prof_enable(0)
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s")
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel by accumulated HIP-event time inside the timed region
    best = None
    kernels = {}
    for kid in range(lib.srgan_prof_num_kernels()):
        ms, n, fl = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
        _lib.check(lib.srgan_prof_collect(kid, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), "prof_collect")
        if n.value == 0:
            continue
        name = lib.srgan_prof_kernel_name(kid).decode()
        kernels[name] = {"launches": n.value, "total_ms": round(ms.value, 3), "avg_us": round(1e3 * ms.value / n.value, 2),
                         "tflops": round(fl.value / (ms.value * 1e-3) / 1e12, 2)}
        if best is None or ms.value > best[1]:
            best = (name, ms.value, n.value, fl.value)

    if rank == 0:
        images = B * world * args.steps
        value = images / elapsed
        peak = PEAK_TFLOPS["f32" if args.dtype == "f32" else "bf16"]
        name, ms, n, fl = best
        achieved = fl / (ms * 1e-3) / 1e12
        # Winograd F(2x2,3x3) kernels issue 16 multiplies per 36 algorithmic ones: the algorithmic rate can exceed the
        # MFMA peak, so the rate of the FLOPs actually issued on the matrix pipe is reported next to it
        wino = name.startswith("wino")
        wino43 = name.startswith("wino43")
        executed = achieved / (4.0 if wino43 else 2.25) if wino else achieved
        out = {
            "metric": "images/sec G+D+E train step, CelebA 128x128 bs=32/GPU" if (args.size == 128 and B == 32) else
                      f"images/sec G+D+E train step, CelebA {args.size}x{args.size} bs={B}/GPU",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SRGAN-nopretrain (03-train) G+D+E train step, {args.size}x{args.size}, "
                                   f"bs={B}/GPU, k={args.k}, " + ("fp32" if args.dtype == "f32" else "bf16 MFMA conv forward / input gradient, "
                                   "fp32 storage, weight gradients, norms, losses, Adam") + ", E trainable (BASELINE configs[1])",
                       "global_batch": B * world, "unrolled_k": args.k, "parallelism": f"dp{world}",
                       "gflop_per_image_algorithmic": GFLOP_PER_IMAGE.get(args.size),
                       "step_tflops_algorithmic": round(value * GFLOP_PER_IMAGE.get(args.size, 0) / 1e3, 2),
                       "losses_last_step": [round(float(v), 4) for v in last]},
            "roofline": {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": pmc_traffic(name),
                         "executed_mfma_tflops": round(executed, 2), "executed_frac": round(executed / peak, 4),
                         "algorithm": ("Winograd F(4x4,3x3): the multiply kernel issues 4x fewer MFMA FLOPs than the algorithmic count "
                                       "(its input transform is the separate HBM-bound wino43_input_kernel, listed in all_gemm_kernels), so the "
                                       "ALGORITHMIC rate asked for in `achieved` exceeds the MFMA peak (frac > 1); executed_frac is the "
                                       "utilisation of the matrix pipe") if wino43 else
                                      ("Winograd F(2x2,3x3) / F(3x3,2x2): 2.25x fewer MFMA FLOPs than the algorithmic count, so the "
                                       "ALGORITHMIC rate asked for in `achieved` can exceed the MFMA peak (frac > 1); "
                                       "executed_frac is the utilisation of the matrix pipe") if wino
                                      else "implicit GEMM: executed = algorithmic FLOPs",
                         "launches": n, "avg_launch_us": round(1e3 * ms / n, 2),
                         "traffic_note": "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC, profiles/r*_pmc_traffic.json)",
                         "note": "achieved = sum of algorithmic conv FLOPs (2*N*Ho*Wo*O*kh*kw*I) of this kernel's launches / "
                                 "sum of their HIP-event durations inside the timed region (rank 0)",
                         "all_gemm_kernels": kernels},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.size, args.k)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

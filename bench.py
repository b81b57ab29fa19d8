#!/usr/bin/env python3
"""Benchmark of the SRGAN G+D+E train step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 from a bare shell: the parent starts N fresh rank processes (``python -m torch.distributed.run``, one per GPU,
RCCL) BEFORE touching the GPU itself, forwards rank 0's JSON line and exits with their status.  Already under
torch.distributed.run (WORLD_SIZE set): runs as that rank.

A "step" = one ``SRGAN_training.train`` call (k=5 D updates + the two-phase G/E update) on one synthetic
CelebA-shaped batch.  Workload at N=1 = BASELINE.json configs[1]: SRGAN-nopretrain, 128x128, bs=32/GPU, fp32,
notebook hyper-parameters (05-train cells 13/16).  Weak scaling: 32 images per GPU, global batch 32*N.
Prints ONE JSON line on rank 0 with ``roofline`` (dominant kernel, HIP-event timed) and, at N=1, ``cpu_baseline``
(the CPU oracle timed on the host cores on a bounded sample) and ``micro_gd`` (north_star's G+D forward-backward target).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
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
GFLOP_PER_IMAGE_FROZEN_E = {128: 399.34, 256: 1621.70}               # SURVEY.md 8d (config 3: encoder trunk frozen)
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2516.6}                         # MI355X dense matrix peak, MI355X_MICROARCH.md
FIRST_STEP_FIXTURE = os.path.join(ROOT, "tests", "golden", "bench_first_step.json")


def synthetic_batch(batch, size, n_class, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    src = torch.randint(0, n_class, (batch,), generator=g)
    tgt = (src + torch.randint(1, n_class, (batch,), generator=g)) % n_class
    return x, src, tgt


def build_nets(size, device):
    """The three networks of 05-train cell 13 with PyTorch's default initialisation under seed 0 (the reference's
    ``weights_init`` is a no-op).  Parameter holders only: constructing them needs no GPU."""
    from srgan_amd import model
    torch.manual_seed(0)             # identical replicas on every rank
    np.random.seed(0)
    d_cls = 4 if size == 128 else 5
    G = model.SingleGenerator(3, 64, 2, 2, 6, "instance", num_con=12)
    D = model.SingleDiscriminator_solo_multi(3, 64, 2, d_cls, "instance", 4)
    E = model.Encoder(3, 8, 64, 4, "instance", 4, device)
    return G, D, E


TRUNK_KEYS_NOT_FROZEN = ("fcmean.weight", "fcmean.bias", "fcvar.weight", "fcvar.bias")


def build_trainer(size, global_batch, k, device, pretrained_e=False):
    from srgan_amd.optim import Adam
    from srgan_amd.trainer import SRGAN_training
    G, D, E = build_nets(size, device)
    opt_e = None
    if pretrained_e:
        # BASELINE configs[2] (05-train cells 14-16): the classifier keys are frozen, only fcmean / fcvar train, with an
        # externally built Adam(lr=1e-3).  The real .pth is a git-LFS pointer, so the "pretrained" trunk is the default init.
        keys = [k_ for k_ in E.state_dict().keys() if k_ not in TRUNK_KEYS_NOT_FROZEN]
        E.freeze_melt(keys, "freeze")
        E.to(device)
        opt_e = Adam([p for p in E.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999))
    sg = SRGAN_training([G, D, E], [None, None, opt_e], [nn.MSELoss(), nn.MSELoss()], dict(LBD), k, device,
                        np.eye(4), global_batch, "mu", 8)
    sg.opt_sche_initialization()
    return sg


def host_cores():
    """Threads we may really use: the affinity mask, bounded by the cgroup CPU quota when the box sets one
    (``SRGAN_BENCH_CPU_THREADS`` overrides)."""
    if os.environ.get("SRGAN_BENCH_CPU_THREADS"):
        return max(1, int(os.environ["SRGAN_BENCH_CPU_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(size, k, batch, timed_steps=2):
    """The CPU oracle (oracle/, a port of the reference's PyTorch-CPU path) on a bounded sample of the SAME workload: full-width
    networks, the metric's batch size, k D-updates -- 1 warm-up step OF THE SAME SHAPE (it pays the oneDNN primitive creation of
    every batch-32 / 64 / 128 convolution, the thread pool and the allocator) + 2 timed steps, the MEDIAN reported next to both
    samples.  ~25 s per step on 16 cores of the GPU box's host: ~80 s in all."""
    from oracle import params as op, trainer as ot
    cores = host_cores()
    torch.set_num_threads(cores)
    spec = (op.generator_spec(3, 64, 2, 2, 6, 12), op.discriminator_spec(3, 64, 2, 4 if size == 128 else 5, 4),
            op.encoder_spec(3, 8, 64, 4, 4))
    PG, PD, PE = (op.fill(s, i) for i, s in enumerate(spec))
    torch.manual_seed(0)
    orc = ot.SRGANOracle(PG, PD, PE, LBD, k, np.eye(4), batch, "mu", 8)
    samples = []
    for step in range(1 + timed_steps):
        x, label = ot.synthetic_batch(batch, size, 4, seed=2 + step)
        log(f"cpu baseline: {'warm-up' if step == 0 else f'timed step {step}'} at batch {batch} on {cores} threads")
        t0 = time.perf_counter()
        orc.train(x, label)
        dt = time.perf_counter() - t0
        if step:
            samples.append(dt)
        else:
            warm = dt
    med = float(np.median(samples))
    return {"value": round(batch / med, 4), "unit": "images/sec", "cores": cores, "cpu_model": cpu_model(),
            "kind": "port", "seconds_per_step": [round(s, 2) for s in samples], "warmup_seconds": round(warm, 2),
            "images_per_sec_samples": [round(batch / s, 4) for s in samples],
            "sample": f"1 warm-up + {timed_steps} timed, median: full train steps (k={k}, fp32, same networks / losses / batch "
                      f"size: {batch} images of {size}x{size}) of the CPU oracle, warm-up step of the same shape; "
                      f"{sum(samples):.1f} s timed on {cores} threads"}


def kernel_source_sha():
    """sha256[:16] over the kernel sources (csrc/*.hip, *.h, *.cpp, Makefile) and the host files that choose the launches: what a
    committed PMC summary must have been collected on for its bytes to describe THIS build (the GPU box has no .git)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pk = os.path.join(ROOT, "style-restricted_gan_amd")
    files = sorted(glob.glob(os.path.join(pk, "csrc", "*.hip")) + glob.glob(os.path.join(pk, "csrc", "*.h")) +
                   glob.glob(os.path.join(pk, "csrc", "*.cpp")) + [os.path.join(pk, "csrc", "Makefile")] +
                   [os.path.join(pk, "srgan_amd", n) for n in ("ops.py", "model.py", "trainer.py")])
    for p in files:
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_name, dtype="f32"):
    """-> (bytes per launch or None, source file, current): HBM bytes per launch of `kernel_name` from the committed PMC summary
    (profiles/r*_pmc_traffic.json, or r*_bf16_pmc_traffic.json for --dtype bf16; separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes over this same bench command, scratch/collect_evidence.sh).  `current` is True only when the summary records the
    kernel_source_sha of THIS tree: then it is this build's traffic; otherwise the line carries it as `traffic_replayed` and
    `traffic` stays null."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))
                   if ("_bf16_" in os.path.basename(f)) == (dtype == "bf16"))
    if not files:
        return None, None, False
    try:
        doc = json.load(open(files[-1]))
        entry = doc["kernels"].get(kernel_name)
        return ((entry["hbm_bytes_per_launch"] if entry else None), os.path.relpath(files[-1], ROOT),
                doc.get("kernel_source_sha") == kernel_source_sha())
    except (OSError, ValueError, KeyError):
        return None, None, False


GD_GFLOP_PER_IMAGE = {128: 60.54, 256: 243.86}        # (3 F_G - f_G) + (3 F_D - f_D), SURVEY.md 8d


HBM_PEAK_TBS = 8.0                                                   # HBM3E spec (6.3 TB/s measured for a float4 copy), MI355X_MICROARCH.md
HBM_KERNELS = ("in_stats_partial", "in_apply", "in_bwd_partial", "in_bwd_apply", "in_fwd_slab", "in_bwd_slab", "in_fwd_slab_v", "in_bwd_slab_vz")


def is_hbm_kernel(name):
    """Kernels whose launch brackets carry ALGORITHMIC BYTES instead of FLOPs (csrc/norm.hip: the instance-norm passes)."""
    return name in HBM_KERNELS


def executed_divisor(name):
    """Algorithmic conv FLOPs / MFMA FLOPs actually issued: Winograd F(4x4,3x3) multiplies 36 values per 16 outputs x 9 taps
    (4x fewer), F(4x4,2x2) 25 per 16 x 4 (2.56x fewer), F(2x2,3x3) / F(3x3,2x2) 16 per 36 (2.25x fewer, before the waste of ragged
    edge tiles); 1 for the direct forms."""
    if name.startswith("wino43"):
        return 4.0
    if name.startswith("wino42"):
        return 2.56          # F(4x4,2x2): 25 multiplies per 16 outputs x 4 taps
    if name.startswith("wino"):
        return 2.25
    return 1.0


def micro_gd_run(size, batch, dtype, steps, warmup, device):
    """One generator forward + full backward and one discriminator forward + full backward on a batch (inputs do not
    require gradients, so the first-layer input gradients are skipped, as in the SURVEY formula).  Single GPU."""
    from srgan_amd import model, ops
    torch.manual_seed(0)
    B, S = batch, size
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

    for _ in range(max(warmup, 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gf = GD_GFLOP_PER_IMAGE.get(S)
    value = B / dt
    peak = PEAK_TFLOPS["f32" if dtype == "f32" else "bf16"]
    tf = value * gf / 1e3 if gf else None
    # MFMA FLOPs the GEMM-class launches of one micro step actually ISSUE (algorithmic / Winograd divisor), from two more steps
    # under the library's HIP-event brackets: against the UN-instrumented step time above
    from srgan_amd import _lib
    lib = _lib.load()
    lib.srgan_prof_enable(1)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    lib.srgan_prof_enable(0)
    issued, counted, gemm_ms = 0.0, 0.0, 0.0
    for kid in range(lib.srgan_prof_num_kernels()):
        ms, n, fl = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
        _lib.check(lib.srgan_prof_collect(kid, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), "prof_collect")
        if n.value and not is_hbm_kernel(lib.srgan_prof_kernel_name(kid).decode()):
            issued += fl.value / executed_divisor(lib.srgan_prof_kernel_name(kid).decode()) / 2
            counted += fl.value / 2
            gemm_ms += ms.value / 2
    issued_tf = issued / dt / 1e12
    return {
        "metric": f"images/sec G+D forward-backward, {S}x{S} bs={B} (micro-benchmark, SURVEY 8d)", "value": round(value, 2),
        "unit": "images/sec", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * dt, 3),
        "higher_is_better": True, "dtype": dtype, "data": "synthetic",
        "config": {"workload": "one G forward + backward and one D forward + backward (all parameter gradients), default PyTorch "
                               "init, first-layer input gradients skipped", "gflop_per_image_algorithmic": gf},
        "roofline": {"bound": "mfma", "algorithmic_tflops": round(tf, 2) if tf else None, "peak": peak, "unit": "TFLOP/s",
                     "algorithmic_frac": round(tf / peak, 4) if tf else None,
                     "issued_tflops": round(issued_tf, 2), "issued_frac": round(issued_tf / peak, 4),
                     "issued_frac_over_gemm_kernel_time": round(issued / (gemm_ms * 1e-3) / 1e12 / peak, 4) if gemm_ms else None,
                     "gemm_kernel_ms_per_step": round(gemm_ms, 3),
                     "gflop_per_image_in_gemm_launches": round(counted / B / 1e9, 2),
                     "issued_note": "MFMA FLOPs the conv launches of one micro step ISSUE (algorithmic / 4 on F(4x4,3x3), / 2.25 on "
                                    "F(2x2,3x3) and F(3x3,2x2), / 1 on the direct forms; the VALU head kernels are not counted) / "
                                    "the micro step's wall time / peak: the utilisation of the matrix pipe over the whole "
                                    "forward-backward, always <= 1",
                     "note": "whole micro-benchmark: algorithmic FLOPs (SURVEY 8d) x images/s against the dense MFMA peak of the "
                             "dtype; includes norm / pointwise / reduction kernels; the Winograd layers issue 2.25-4x fewer MFMA "
                             "FLOPs than counted, so this ALGORITHMIC fraction can exceed 1 (north_star's >= 0.70 target is "
                             "stated on algorithmic FLOPs)"}}


def bf16_sub_record(args):
    """`bench.py --dtype bf16` on the same workload in a child process -> {value, ms_per_step, dominant kernel and its issued
    fraction of the bf16 matrix peak, whole-step issued fraction}: the bf16 mode next to the fp32 headline, timed by the same run."""
    log("bf16 mode: the same workload through --dtype bf16 (child process)")
    cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "bf16", "--steps", "20", "--warmup", "5", "--size", str(args.size),
           "--batch-per-gpu", str(args.batch_per_gpu), "--k", str(args.k), "--graph", args.graph, "--no-cpu-baseline", "--no-micro"]
    if args.pretrained_e:
        cmd.append("--pretrained-e")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"child exited {r.returncode}", "stderr_tail": r.stderr[-400:]}
        d = json.loads(lines[-1])
    except Exception as e:      # noqa: BLE001 -- the fp32 line must still be printed
        return {"error": f"{type(e).__name__}: {e}"}
    rf = d.get("roofline", {})
    return {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
            "warmup": d["warmup"], "dtype": "bf16", "execution": d["config"]["execution"],
            "dominant_kernel": rf.get("kernel"), "dominant_kernel_frac": rf.get("frac"), "dominant_kernel_avg_launch_us": rf.get("avg_launch_us"),
            "peak_tflops": rf.get("peak"), "step_executed_frac": rf.get("step", {}).get("executed_frac"),
            "step_executed_frac_over_gemm_kernel_time": rf.get("step", {}).get("executed_frac_over_gemm_kernel_time"),
            "instance_norm_passes": (d.get("roofline_hbm") or {}).get("instance_norm_passes"),
            "losses_last_step": d["config"].get("losses_last_step"),
            "note": "bench.py --dtype bf16 on the same workload (bf16 MFMA convolutions, 16-bit activations where the kernels take "
                    "them, fp32 statistics / losses / Adam): never the headline value"}


def micro_gd(args):
    from srgan_amd import _lib, ops
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    device = torch.device("cuda", 0)
    _lib.load()
    if args.dtype == "bf16":
        ops.set_compute_dtype("bf16")
    print(json.dumps(micro_gd_run(args.size, args.batch_per_gpu, args.dtype, args.steps, args.warmup, device)))


# ------------------------------------------------------------------------------------------------------------------
# multi-GPU launcher: runs in the parent BEFORE any GPU call
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """Start n rank processes of this script under torch.distributed.run and return their exit status.  The parent has
    not initialised HIP (``import torch`` does not), so nothing that owns a GPU context is ever forked or re-exec'ed."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC only on this pool (RCCL needs it)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or n
    env.setdefault("OMP_NUM_THREADS", str(max(1, cores // n)))   # host threads per rank: the CPU-generator noise, not a pool fight
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"launching {n} ranks: {' '.join(cmd[1:])}")
    return subprocess.run(cmd, env=env).returncode


def launch_check(rank, world, device):
    """--launch-check: prove the rendezvous (every rank reports in) and stop before any GPU work."""
    ranks = [None] * world
    if world > 1:
        dist.all_gather_object(ranks, (rank, int(os.environ.get("LOCAL_RANK", "0")), str(device)))
    else:
        ranks = [(rank, 0, str(device))]
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": sorted(ranks),
                          "backend": dist.get_backend() if world > 1 else None,
                          "omp_num_threads": os.environ.get("OMP_NUM_THREADS")}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def first_step_check(first, args, world):
    """The first (un-warmed) step's [errG, errD, errE] against the CPU oracle's values for the SAME seeds, networks and
    batch (tests/golden/bench_first_step.json, written by tests/golden/make_bench_first_step.py).  Only the configuration
    the fixture was generated for is compared; everything else must at least be finite."""
    vals = [float(v) for v in first]
    from srgan_amd import _lib
    if _lib.ab("SRGAN_BENCH_NO_CHECK"):              # ablation builds with deliberately wrong kernels (make exp, scratch/ only)
        return {"hip": vals, "oracle": None, "check": "DISABLED (SRGAN_BENCH_NO_CHECK next to an experiment build)"}
    if not all(np.isfinite(vals)):
        raise SystemExit(f"bench: non-finite losses in the first step: {vals}")
    if not os.path.exists(FIRST_STEP_FIXTURE):
        return {"hip": vals, "oracle": None}
    fx = json.load(open(FIRST_STEP_FIXTURE))
    key = f"{args.size}x{args.size}_b{args.batch_per_gpu}_k{args.k}"
    same = (world == 1 and args.dtype == "f32" and not args.pretrained_e and key in fx["configs"])
    if not same:
        return {"hip": vals, "oracle": None}
    ref = fx["configs"][key]["losses"]
    rel = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(vals, ref))
    band = fx["band"]
    if rel > band:
        raise SystemExit(f"bench: first-step losses {vals} differ from the CPU oracle's {ref} by {rel:.2e} (> {band})")
    return {"hip": [round(v, 5) for v in vals], "oracle": ref, "max_rel_err": float(f"{rel:.3e}"), "band": band}


def _trainer_has_graph():
    from srgan_amd.trainer import SRGAN_training
    return hasattr(SRGAN_training, "enable_graph")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # (the counts every evidence line under profiles/ was measured with)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-micro", action="store_true", help="skip the G+D forward-backward sub-record of the N=1 line")
    ap.add_argument("--no-bf16", action="store_true", help="skip the bf16-mode sub-record of the N=1 fp32 line")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 (default, BASELINE configs[1]) or the bf16 MFMA compute mode of configs [2]-[4] (never the headline)")
    ap.add_argument("--pretrained-e", action="store_true",
                    help="BASELINE configs[2] recipe: encoder trunk frozen, fcmean/fcvar on an external Adam(1e-3)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step as a captured hipGraph (auto: when the trainer supports it)")
    ap.add_argument("--micro", choices=["gd"], default=None,
                    help="gd: the 'G+D forward-backward' micro-benchmark of SURVEY.md 8d (north_star's >= 70 %% MFMA-roofline target) "
                         "instead of the full train step")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: every rank reports in, rank 0 prints a JSON line, nobody touches the GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))            # parent: no GPU call has happened in this process
    if args.micro == "gd":
        return micro_gd(args)

    from srgan_amd import dp
    if args.launch_check:
        ws = int(os.environ.get("WORLD_SIZE", "1"))
        if ws > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            with dp._stdout_to_stderr():         # gloo's connection banner must not precede the JSON line on stdout
                dist.init_process_group("gloo", rank=int(os.environ.get("RANK", "0")), world_size=ws)
        if ws != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ws}")
        return launch_check(int(os.environ.get("RANK", "0")), ws, "unopened")

    from srgan_amd import _lib
    rank, world, device = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    if world > 1:
        torch.set_num_threads(max(1, host_cores() // world))
    lib = _lib.load()
    exit_code = 0
    if args.dtype == "bf16":
        from srgan_amd import ops as _ops
        _ops.set_compute_dtype("bf16")
    B = args.batch_per_gpu
    # which physical device every rank drives (the driver checks that N ranks saw N devices)
    me = (rank, str(device), torch.cuda.get_device_properties(device).name, getattr(torch.cuda.get_device_properties(device), "uuid", None))
    me = (me[0], me[1], me[2], str(me[3]) if me[3] is not None else None)
    devices = [me]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, me, group=dp.control_group())
        devices = sorted(devices)
    sg = build_trainer(args.size, B * world, args.k, device, args.pretrained_e)
    torch.manual_seed(1000 + rank)           # per-rank noise stream after identical construction

    use_graph = args.graph != "off" and _trainer_has_graph()
    if args.graph == "on" and not use_graph:
        raise SystemExit("--graph on: this trainer has no hipGraph mode")
    warm = max(args.warmup, 2 if use_graph else 1)      # step 0 is the oracle-checked one; a graph needs one more to be captured
    batches = []
    for s in range(args.steps + warm):
        x, src, tgt = synthetic_batch(B, args.size, 4, seed=10_000 * (rank + 1) + s)
        batches.append((x.to(device), {"source": src.to(device), "target": tgt}))   # inputs resident in HBM
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # step 0 (eager, un-warmed): checked against the CPU oracle's losses for the same seeds
    log(f"rank {rank}/{world}: inputs resident; first step (checked against the oracle fixture)")
    if use_graph:
        try:
            sg.enable_graph()            # step 0 still runs eagerly (it creates optimiser state / packed operands); step 1 captures
        except NotImplementedError as e:
            if args.graph == "on":
                raise SystemExit(f"--graph on: {e}")
            log(f"hipGraph mode not available here, running eagerly: {e}")
            use_graph = False
    first = [float(v) for v in sg.train(*batches[0])]
    check = first_step_check(first, args, world) if rank == 0 else None
    log(f"warm-up x{args.warmup}" + (" (captures the hipGraph)" if use_graph else ""))
    for s in range(1, warm):
        sg.train(*batches[s])
        torch.cuda.synchronize()
        log(f"warm-up step {s} done")
    graphed = bool(use_graph and getattr(sg, "graph_active", False))
    graph_fallback = bool(use_graph and not graphed)
    _rec = getattr(getattr(sg, "_graph", None), "graph", None) if graphed else None
    graph_segments = len(_rec.segments) if _rec is not None else None
    graph_single = bool(_rec is not None and getattr(_rec, "single", False))
    if graph_fallback and (world == 1 or args.graph == "on"):
        # enable_graph() was accepted but the step is not replaying a recording after the warm-up: the line would silently
        # describe another execution mode -- refuse instead of warning.  (Under a process group with --graph auto the ranks'
        # agreed fall-back to eager launches is kept -- a first multi-rank RCCL run should still produce its number -- and the
        # line says so in "graph_fallback" / "execution".)
        raise SystemExit("bench: hipGraph mode was requested and available, but the warm-up did not leave a recorded step behind "
                         "(the trainer fell back to eager launches); run with --graph off to measure that")
    barrier()
    prof_live = not graphed                   # HIP events cannot bracket kernels inside a graph replay
    if prof_live:
        lib.srgan_prof_enable(1)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    last = None
    marks[0].record()
    for i, s in enumerate(range(warm, warm + args.steps)):
        last = sg.train(*batches[s])
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if prof_live:
        lib.srgan_prof_enable(0)
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s")
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    if world > 1:
        t = torch.tensor([elapsed, median_ms], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, median_ms = float(t[0].item()), float(t[1].item())
    last = [float(v) for v in last]
    if not all(np.isfinite(last)):
        raise SystemExit(f"bench: non-finite losses after the timed region: {last}")

    prof_steps = args.steps
    if not prof_live:
        # same kernels, same shapes, eager launches: per-kernel HIP-event durations for the roofline record
        prof_steps = min(args.steps, 3)
        if hasattr(sg, "disable_graph"):
            sg.disable_graph()
        torch.cuda.synchronize()
        lib.srgan_prof_enable(1)
        for s in range(prof_steps):
            sg.train(*batches[1 + s])
        torch.cuda.synchronize()
        lib.srgan_prof_enable(0)

    # per-kernel totals of the HIP-event brackets
    kernels = {}
    hbm = {}
    best = None
    for kid in range(lib.srgan_prof_num_kernels()):
        ms, n, fl = ctypes.c_double(), ctypes.c_longlong(), ctypes.c_double()
        _lib.check(lib.srgan_prof_collect(kid, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), "prof_collect")
        if n.value == 0:
            continue
        name = lib.srgan_prof_kernel_name(kid).decode()
        if is_hbm_kernel(name):
            hbm[name] = {"launches": n.value, "total_ms": round(ms.value, 3), "avg_us": round(1e3 * ms.value / n.value, 2),
                         "algorithmic_mb_per_launch": round(fl.value / n.value / 1e6, 2),
                         "achieved_tb_s": round(fl.value / (ms.value * 1e-3) / 1e12, 3), "_ms": ms.value, "_bytes": fl.value}
            continue
        div = executed_divisor(name)
        kernels[name] = {"launches": n.value, "total_ms": round(ms.value, 3), "avg_us": round(1e3 * ms.value / n.value, 2),
                         "algorithmic_tflops": round(fl.value / (ms.value * 1e-3) / 1e12, 2),
                         "executed_tflops": round(fl.value / div / (ms.value * 1e-3) / 1e12, 2),
                         "_ms": ms.value, "_fl": fl.value}
        if fl.value > 0 and (best is None or ms.value > best[1]):
            best = (name, ms.value, n.value, fl.value)

    if rank == 0:
        images = B * world * args.steps
        value = images / elapsed
        peak = PEAK_TFLOPS["f32" if args.dtype == "f32" else "bf16"]
        name, ms, n, fl = best
        div = executed_divisor(name)
        algorithmic = fl / (ms * 1e-3) / 1e12
        executed = algorithmic / div
        # the layer a Winograd F(4x4,3x3) multiply belongs to also runs its HBM-bound transform kernel
        partner = {"wino43_kernel": "wino43_input_kernel", "wino43_wgrad_kernel": "wino43_dy_kernel"}.get(name)
        layer = None
        if partner and partner in kernels:
            lms = ms + kernels[partner]["_ms"]
            layer = {"kernels": [partner, name], "total_ms": round(lms, 3),
                     "executed_tflops": round(fl / div / (lms * 1e-3) / 1e12, 2),
                     "frac": round(fl / div / (lms * 1e-3) / 1e12 / peak, 4),
                     "algorithmic_tflops": round(fl / (lms * 1e-3) / 1e12, 2)}
        step_exec = sum(kk["_fl"] / executed_divisor(nm) for nm, kk in kernels.items()) / prof_steps      # FLOPs per step
        step_alg_launched = sum(kk["_fl"] for kk in kernels.values()) / prof_steps       # algorithmic FLOPs the launches carry
        gemm_ms = sum(kk["_ms"] for kk in kernels.values()) / prof_steps
        step_ms_mean = 1e3 * elapsed / args.steps
        for kk in kernels.values():
            kk.pop("_ms"), kk.pop("_fl")
        traffic, traffic_src, traffic_current = pmc_traffic(name, args.dtype)
        switches = _lib.active_switches()
        gflop_img = (GFLOP_PER_IMAGE_FROZEN_E if args.pretrained_e else GFLOP_PER_IMAGE).get(args.size)
        out = {
            "metric": "images/sec G+D+E train step, CelebA 128x128 bs=32/GPU" if (args.size == 128 and B == 32) else
                      f"images/sec G+D+E train step, CelebA {args.size}x{args.size} bs={B}/GPU",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_ms_mean, 3), "ms_per_step_median": round(median_ms, 3),
            "value_at_median_step": round(B * world / (median_ms * 1e-3), 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SRGAN-{'full, pretrained-E recipe (05-train)' if args.pretrained_e else 'nopretrain (03-train)'} "
                                   f"G+D+E train step, {args.size}x{args.size}, "
                                   f"bs={B}/GPU, k={args.k}, " + ("fp32" if args.dtype == "f32" else "bf16 MFMA convolutions (forward, input and weight gradient; "
                                   "fp32 accumulation), fp32 tensors in HBM, fp32 norms, losses, Adam") +
                                   (", E trunk frozen (BASELINE configs[2])" if args.pretrained_e else ", E trainable (BASELINE configs[1])"),
                       "global_batch": B * world, "unrolled_k": args.k, "parallelism": f"dp{world}",
                       "graph_fallback": graph_fallback,
                       # (keyed on the data-parallel CODE PATH, not on the rank count: SRGAN_DP_FORCE=1 drives it with one rank)
                       "backend": (dist.get_backend() if dp.is_distributed() else None),
                       "collectives": (dp.transport() if dp.is_distributed() else None),
                       "graph_segments": graph_segments,
                       "ranks_seen": (dist.get_world_size() if dp.is_distributed() else 1), "devices": devices,
                       "gradient_message_dtype": (dp.bucket_dtype() if dp.is_distributed() else None),
                       "kernel_source_sha": kernel_source_sha(),
                       "experiment_switches": switches,
                       "execution": ("eager launches (hipGraph recording FAILED on some rank: all ranks fell back)" if graph_fallback else
                                     "eager launches" if not graphed else "hipGraph replay of the captured step" if not dp.is_distributed() else
                                     ("ONE hipGraph per step with the RCCL collectives (C ABI: srgan_allreduce_bucket / srgan_allgather_rows) "
                                      "captured on the communication stream, no host-side vote per step") if graph_single else
                                     f"{graph_segments} hipGraph segments replayed with the {'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend()} "
                                     "all-reduces started on the communication stream between them (running under the next segment)"),
                       "gflop_per_image_algorithmic": gflop_img,
                       "step_tflops_algorithmic": round(value * (gflop_img or 0) / 1e3, 2),
                       # the contract's figure counts every convolution of the reference's step; the launches carry less because
                       # phase 1 shares one E(source) trunk pass (result-preserving): both are stated
                       "step_tflop_algorithmic_contract": round(B * (gflop_img or 0) / 1e3, 3),
                       "step_tflop_algorithmic_launched": round(step_alg_launched / 1e12, 3),
                       "step_tflops_algorithmic_launched": round(step_alg_launched / 1e12 / (step_ms_mean * 1e-3), 2),
                       "losses_first_step_vs_oracle": check,
                       "losses_last_step": [round(v, 4) for v in last]},
            "roofline": {"bound": "mfma", "kernel": name, "achieved": round(executed, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(executed / peak, 4),
                         "traffic": traffic if traffic_current else None,
                         "traffic_replayed": None if traffic_current else traffic,
                         "traffic_note": "no PMC entry for this kernel under profiles/" if traffic is None else
                                         (f"HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC) from {traffic_src}, collected "
                                          "over this same command in separate --pmc passes" +
                                          (" ON THIS BUILD (the summary records this tree's kernel_source_sha)" if traffic_current else
                                           " on an EARLIER build (kernel_source_sha differs): replayed, not this build's traffic")),
                         "achieved_note": "achieved / frac = MFMA FLOPs this kernel actually ISSUES (algorithmic conv FLOPs / "
                                          f"{div:g}) / HIP-event duration, against the dense matrix peak: the utilisation of the matrix pipe",
                         "algorithmic_tflops": round(algorithmic, 2), "algorithmic_over_peak": round(algorithmic / peak, 4),
                         "algorithmic_note": "direct-convolution FLOPs (2*N*Ho*Wo*O*kh*kw*I, SURVEY 8d) of this kernel's launches / "
                                             "their duration; exceeds the peak for the Winograd kernels because they skip multiplies",
                         "launches": n, "avg_launch_us": round(1e3 * ms / n, 2),
                         "layer": layer,
                         "step": {"executed_mfma_tflop_per_step": round(step_exec / 1e12, 3),
                                  "executed_frac": round(step_exec / (step_ms_mean * 1e-3) / 1e12 / peak, 4),
                                  "executed_frac_over_gemm_kernel_time": round(step_exec / (gemm_ms * 1e-3) / 1e12 / peak, 4),
                                  "gemm_kernel_ms_per_step": round(gemm_ms, 3),
                                  "note": "all GEMM-class launches of a step: issued MFMA FLOPs / step time (and / their own "
                                          "summed duration) / peak"},
                         "measured_over": (f"{prof_steps} eager steps after the timed region (HIP events cannot bracket kernels "
                                           "inside a hipGraph replay; same kernels and shapes)") if not prof_live else
                                          "the timed region (HIP events on the launch stream, rank 0)",
                         "all_gemm_kernels": kernels},
        }
        if hbm:
            # the largest HBM-bound kernel of the step: algorithmic bytes per launch / duration / 8 TB/s
            top = max(hbm, key=lambda k_: hbm[k_]["_ms"])
            tot_ms = sum(v["_ms"] for v in hbm.values())
            tot_b = sum(v["_bytes"] for v in hbm.values())
            for v in hbm.values():
                v.pop("_ms"), v.pop("_bytes")
            out["roofline_hbm"] = {"bound": "hbm", "kernel": top, "achieved": round(1e3 * hbm[top]["achieved_tb_s"], 1), "peak": 1e3 * HBM_PEAK_TBS,
                                   "unit": "GB/s", "frac": round(hbm[top]["achieved_tb_s"] / HBM_PEAK_TBS, 4),
                                   "launches": hbm[top]["launches"], "avg_launch_us": hbm[top]["avg_us"],
                                   "algorithmic_mb_per_launch": hbm[top]["algorithmic_mb_per_launch"],
                                   "instance_norm_passes": {"ms_per_step": round(tot_ms / prof_steps, 3),
                                                            "achieved_tb_s": round(tot_b / (tot_ms * 1e-3) / 1e12, 3),
                                                            "note": "all instance-norm passes: the two-pass and slab kernels of csrc/norm.hip and (round 5) the "
                                                                    "fused norm + Winograd-transform kernels of the trunk (in_fwd_slab_v, in_bwd_slab_vz: "
                                                                    "tensor read(s) + 2.25x transform image(s) written): bytes each pass must move / "
                                                                    "summed duration"},
                                   "measured_over": out["roofline"]["measured_over"], "kernels": hbm}
        if world == 1 and not args.no_micro and args.size in GD_GFLOP_PER_IMAGE:
            log("micro benchmark: G+D forward-backward")
            del sg
            torch.cuda.empty_cache()
            out["micro_gd"] = micro_gd_run(args.size, B, args.dtype, 10, 3, device)
        if world == 1 and args.dtype == "f32" and not args.no_bf16 and not args.no_micro and not switches:
            # the bf16 mode (BASELINE configs[2]-[4]) measured by the driver's own command -- the same workload
            # through `--dtype bf16` in a child process (its own 5 + 20 steps, graph replay), reduced to the numbers that matter
            out["bf16"] = bf16_sub_record(args)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.size, args.k, B)
        print(json.dumps(out))
        if switches:
            log(f"experiment build / switches active ({switches}): the line above is NOT a result of the product library")
            exit_code = 3
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()

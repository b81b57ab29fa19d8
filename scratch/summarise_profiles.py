"""Turns the rocprofv3 outputs of one evidence run (gpurun_out/ev/{stats,fetch,write}) into the files kept under
profiles/: the kernel-stats CSV as is, and r<NN>_pmc_traffic.json = per-launch HBM bytes per GEMM kernel
(2 * FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md's HBM section; separate PMC passes)."""
import csv, glob, json, os, re, shutil, sys, collections

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ev = os.path.join(root, "gpurun_out", os.environ.get("EV_OUT", "ev"))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02_final"
pmc_tag = tag[:-6] if tag.endswith("_final") else tag      # r02_final -> r02_pmc_*.json; r02_bf16 -> r02_bf16_pmc_*.json


def prof_name(n):
    n = n.replace("void ", "").replace("srgan::", "")
    m = re.match(r"igemm_kernel<(\d+), (\d+), (\d+), (\d+), (true|false)(?:, (?:true|false))?>", n)
    if m:
        return "igemm_kernel<%s,%s,%s,%s,%s>" % (*m.groups()[:4], "vec" if m.group(5) == "true" else "gen")
    m = re.match(r"wgrad_kernel<\d+, \d+, \d+, \d+, (true|false)", n)
    if m:
        return "wgrad_kernel<%s>" % ("vec" if m.group(1) == "true" else "gen")
    if n.startswith("wino42_kernel"):
        return "wino42_kernel"
    m = re.match(r"wino_wgrad_kernel<(\d)>", n)
    if m:
        return "wino_wgrad_kernel" if m.group(1) == "0" else "wino_wgrad_kernel<4x4s2>"
    m = re.match(r"wino_kernel<(\d)(?:, (?:true|false))?>", n)
    if m:
        return "wino_kernel" if m.group(1) == "0" else "wino_kernel<4x4s2>"
    if n.startswith("rgbin_conv_kernel") or n.startswith("rgbin16_conv_kernel"):
        return "rgbin_conv_kernel"
    if n.startswith("rgb_wgrad_kernel") or n.startswith("rgb_wgrad16_kernel"):      # (round 6: the bf16 mode's kernel; one profile name)
        return "rgb_wgrad_kernel"
    if "rgbout_conv_kernel" in n or "rgbout16_conv_kernel" in n:
        return "rgbout_conv_kernel"
    if n.startswith("wino43_wgrad_kernel"):
        return "wino43_wgrad_kernel"
    if n.startswith("wino43_dy_kernel"):
        return "wino43_dy_kernel"
    if n.startswith("wino43_input_kernel"):
        return "wino43_input_kernel"
    if n.startswith("wino43_kernel"):
        return "wino43_kernel"
    if n.startswith("igemm16_kernel"):
        return "igemm16_kernel"
    # the instance-norm passes (HBM-bound: their traffic against the tensor bytes is the point)
    for k in ("in_stats_partial", "in_apply_pow2", "in_bwd_partial", "in_bwd_apply_pow2", "in_fwd_slab_v16", "in_bwd_slab_vz",
              "in_fwd_slab", "in_bwd_slab"):
        if n.startswith(k):
            return {"in_apply_pow2": "in_apply", "in_bwd_apply_pow2": "in_bwd_apply"}.get(k, k)
    if "halo16e_kernel" in n:        # (round 6: the encoder's large-map layers in the bf16 mode)
        return "halo16e_kernel"
    if "halo16t_kernel" in n:
        return "halo16t_kernel"
    if "halo16s2_wgrad_kernel" in n:
        return "halo16s2_wgrad_kernel"
    if "halo16s_kernel" in n:
        return "halo16s_kernel"
    if "halo16_wgrad_kernel" in n:
        return "halo16_wgrad_kernel"
    if "halo16_kernel" in n or "halo16r_kernel" in n:      # (round 6: the 256-channel layers run halo16r_kernel; one profile name)
        return "halo16_kernel"
    return None


def newest(pattern):
    """gpurun merges every run's files into gpurun_out/: take the most recent match."""
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    if not f:
        raise SystemExit("no file matches " + pattern)
    return f[-1]


def counter(sub, name):
    f = newest(os.path.join(ev, sub, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = prof_name(r["Kernel_Name"])
        if k and r["Counter_Name"] == name:
            agg[k].append(float(r["Counter_Value"]))
    return agg


def evidence_sha():
    """kernel_source_sha of the tree the evidence run executed (bench.py prints it in config): bench.py reports a PMC summary as
    THIS build's traffic only when it matches the running tree's."""
    p = os.path.join(ev, "bench_plain.json")
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])["config"]["kernel_source_sha"]
    except (OSError, ValueError, KeyError, IndexError):
        return None


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
out = {"_doc": "Per-launch averages of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py "
               "--steps 5 --warmup 1 --graph off --no-cpu-baseline --no-micro`; hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE (KB*1024): on gfx950 "
               "FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact.",
       "kernel_source_sha": evidence_sha(), "kernels": {}}
for k in sorted(fetch):
    f, w = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / max(len(write[k]), 1)
    out["kernels"][k] = {"launches": len(fetch[k]), "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open(os.path.join(root, "profiles", pmc_tag + "_pmc_traffic.json"), "w"), indent=1)

# matrix-pipe utilisation per GEMM kernel: SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (= 64 x #MFMA for the fp32
# 32x32x2 instruction); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so GUI/8 is the kernel's length in shader cycles
if glob.glob(os.path.join(ev, "mfma", "*", "*counter_collection.csv")):
    busy, gui = counter("mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), counter("mfma", "GRBM_GUI_ACTIVE")
    util = {"_doc": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over `python3 bench.py --steps 5 --warmup 1 "
                    "--graph off --no-cpu-baseline --no-micro`; per kernel: sum over launches of MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8). "
                    "Launches shorter than ~0.3 ms over-count GUI cycles (MI355X_MICROARCH.md, DVFS), so small kernels read low.",
            "kernel_source_sha": evidence_sha(), "kernels": {}}
    for k in sorted(busy):
        b, g = sum(busy[k]), sum(gui[k])
        util["kernels"][k] = {"launches": len(busy[k]), "mfma_busy_cycles": int(b), "gui_active": int(g),
                              "mfma_utilisation": round(b / (1024.0 * g / 8.0), 4) if g else None}
    json.dump(util, open(os.path.join(root, "profiles", pmc_tag + "_pmc_mfma.json"), "w"), indent=1)
stats = newest(os.path.join(ev, "stats", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(root, "profiles", tag + "_bench_steps5_kernel_stats.csv"))
for src, dst in (("bench_plain.json", tag + "_bench_steps20.json"), ("bench_eager.json", tag + "_bench_steps20_eager.json"),
                 ("bench_under_rocprof.json", tag + "_bench_steps5_under_rocprof.json")):
    p = os.path.join(ev, src)
    if os.path.exists(p):
        line = open(p).read().strip().splitlines()[-1]
        json.loads(line)
        open(os.path.join(root, "profiles", dst), "w").write(line + "\n")
print("wrote profiles/", tag)

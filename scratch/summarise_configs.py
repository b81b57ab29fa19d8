"""Copies the per-configuration evidence of scratch/collect_configs.sh (gpurun_out/cfg) to profiles/<round>_<tag>_bench_steps20.json,
..._bench_steps5_under_rocprof.json and ..._bench_steps5_kernel_stats.csv.  Usage: python scratch/summarise_configs.py r04"""
import glob, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(root, "gpurun_out", "cfg")
for f in sorted(glob.glob(os.path.join(src, "c*_*.json"))):
    tag = os.path.basename(f)[:-5]
    if tag.endswith(".under_rocprof"):
        continue
    line = open(f).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(root, "profiles", f"{rnd}_{tag}_bench_steps20.json"), "w").write(line + "\n")
    u = os.path.join(src, tag + ".under_rocprof.json")
    if os.path.exists(u):
        open(os.path.join(root, "profiles", f"{rnd}_{tag}_bench_steps5_under_rocprof.json"), "w").write(open(u).read().strip().splitlines()[-1] + "\n")
    st = sorted(glob.glob(os.path.join(src, tag + ".stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if st:
        shutil.copy(st[-1], os.path.join(root, "profiles", f"{rnd}_{tag}_bench_steps5_kernel_stats.csv"))
    d = json.loads(line)
    print(f"{tag:10s} {d['value']:8.1f} images/s {d['ms_per_step']:7.2f} ms/step")

# the one-rank data-parallel rehearsals (bench line + its log: the RCCL banner in the log is the only proof the path ran in round 5;
# since round 6 the line itself carries backend / collectives / graph_segments)
for f in sorted(glob.glob(os.path.join(src, "dp1_*.json"))):
    tag = os.path.basename(f)[:-5]
    line = open(f).read().strip().splitlines()[-1]
    d = json.loads(line)
    open(os.path.join(root, "profiles", f"{rnd}_{tag}_bench_steps20.json"), "w").write(line + "\n")
    err = os.path.join(src, tag + ".err")
    if os.path.exists(err):
        shutil.copy(err, os.path.join(root, "profiles", f"{rnd}_{tag}_bench_steps20.log"))
    print(f"{tag:18s} {d['value']:8.1f} images/s {d['ms_per_step']:7.2f} ms/step  {d['config'].get('graph_segments')} segment(s), {d['config'].get('collectives')}")

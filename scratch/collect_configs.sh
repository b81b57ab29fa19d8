#!/bin/bash
# Single-GPU evidence for the other BASELINE configurations: per configuration the un-profiled bench line
# (graph replay, 20 steps) and the rocprofv3 --kernel-trace --stats summary of the same command (5 steps).  Run on the GPU box;
# scratch/summarise_configs.py (here) copies the results to profiles/<round>_<tag>_*.
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/cfg
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-micro "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag bench FAILED"; return 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag.stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-micro "$@" > $O/$tag.under_rocprof.json 2> $O/$tag.stats.err || { echo "$tag stats FAILED"; return 1; }
  find $O/$tag.stats -name "*kernel_trace.csv" -delete
  python3 -c "import json,sys; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('%-12s %8.1f images/s %7.2f ms/step  %s' % ('$tag', d['value'], d['ms_per_step'], d['config']['workload'][:70]))"
}
run c2_bf16 --pretrained-e --dtype bf16 &&
run c2_fp32 --pretrained-e &&
run c3_bf16 --batch-per-gpu 64 --dtype bf16 &&
run c3_fp32 --batch-per-gpu 64 &&
run c4_bf16 --size 256 --batch-per-gpu 16 --dtype bf16 &&
run c4_fp32 --size 256 --batch-per-gpu 16
# the data-parallel code path on ONE rank over RCCL (real collectives on the communication stream, 2k + 6 graph segments), and eager
dp1() {
  tag=$1; shift
  SRGAN_DP_FORCE=1 "$@" python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-micro > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; return 1; }
  python3 -c "import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('%-12s %8.1f images/s %7.2f ms/step  %s' % ('$tag', d['value'], d['ms_per_step'], d['config']['execution'][:70]))"
}
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-micro > $O/c1_fp32_plain.json 2> $O/c1_fp32_plain.err && python3 -c "import json; d=json.loads(open('$O/c1_fp32_plain.json').read().strip().splitlines()[-1]); print('%-12s %8.1f images/s %7.2f ms/step' % ('c1 plain', d['value'], d['ms_per_step']))"
dp1 dp1_graph env
dp1 dp1_eager env SRGAN_DP_GRAPH=0
# round 6: the C-ABI collectives -- ONE graph per step with the collectives captured, and the same transport with segments
dp1 dp1_abi_single env SRGAN_DP_COMM=abi
dp1 dp1_abi_segments env SRGAN_DP_COMM=abi SRGAN_DP_SINGLE_GRAPH=0

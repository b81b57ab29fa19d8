#!/bin/bash
# Idle time between kernels of the replayed step: kernel trace of a short bench run, then scratch/gap_analysis.py on it.
# GAP_FLAGS (optional): extra bench.py flags, e.g. "--dtype bf16".
R=$(cd "$(dirname "$0")/.." && pwd)
E=$R/gpurun_out/gap
rm -rf $E && mkdir -p $E
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $E/tr -- python3 $R/bench.py $GAP_FLAGS --steps 6 --warmup 2 --no-cpu-baseline --no-micro > $E/bench.json 2> $E/err.txt || exit 1
T=$(find $E/tr -name "*kernel_trace.csv" | head -1)
MS=$(python3 -c "import json,sys; print(json.loads([l for l in open('$E/bench.json') if l.startswith('{')][-1])['ms_per_step'])")
python3 $R/scratch/gap_analysis.py $T $MS 4 > $E/gaps.txt 2>&1
cp $T $E/trace.csv; gzip -f $E/trace.csv; find $E/tr -name "*kernel_trace.csv" -delete
cat $E/gaps.txt

#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/bwn
cd /tmp && export TMPDIR=/tmp
for s in ${SHAPES:-32,64,128 32,128,64 32,256,32 128,64,128}; do
  export SHAPE=$s; rm -rf $R/gpurun_out/bwn/s$s
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bwn/s$s -- python3 $R/scratch/bw_norm.py > $R/gpurun_out/bwn/s$s.out 2> $R/gpurun_out/bwn/s$s.err
  f=$(ls $R/gpurun_out/bwn/s$s/*/*kernel_stats.csv | head -1)
  echo "== $s $(cat $R/gpurun_out/bwn/s$s.out | tail -1)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "srgan::" in r["Name"] and int(r["Calls"]) >= 20:
        print("  %-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done

#!/bin/bash
# Same-box comparison of several library builds: scratch/ab_lib3.sh <dtype> <a.so> <b.so> ...  (round-robin, three passes)
R=$(cd "$(dirname "$0")/.." && pwd)
dt=$1; shift
for pass in 1 2 3; do
  for lib in "$@"; do
    export SRGAN_HIP_LIB=$R/$lib
    python3 $R/bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-micro 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$dt $lib', d['value'], d['ms_per_step'])" || exit 1
  done
done

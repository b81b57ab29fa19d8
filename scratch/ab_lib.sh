#!/bin/bash
# Same-box A/B of two builds of the library: scratch/ab_lib.sh <old.so> <new.so> [dtypes...]  (alternating, two passes)
R=$(cd "$(dirname "$0")/.." && pwd)
old=$1; new=$2; shift; shift
for dt in ${@:-f32 bf16}; do
for pass in 1 2; do
  for v in new old; do
    if [ $v = new ]; then export SRGAN_HIP_LIB=$R/$new; else export SRGAN_HIP_LIB=$R/$old; fi
    python3 $R/bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-micro 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$dt $v', d['value'], d['ms_per_step'])" || exit 1
  done
done
done

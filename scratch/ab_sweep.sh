#!/bin/bash
# Same-box sweep of one integer experiment switch: scratch/ab_sweep.sh <ENV_NAME> "<values, '-' = unset>" [dtypes...]
R=$(cd "$(dirname "$0")/.." && pwd)
export SRGAN_HIP_LIB=$R/scratch/libsrgan_exp.so
name=$1; vals=$2; shift 2
for dt in ${@:-f32 bf16}; do
for pass in 1 2; do
  for v in $vals; do
    if [ "$v" = "-" ]; then unset $name; else export $name=$v; fi
    python3 $R/bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-micro --no-bf16 2>/dev/null < /dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$dt $name=$v', d['value'], d['ms_per_step'])" || exit 1
  done
done
done

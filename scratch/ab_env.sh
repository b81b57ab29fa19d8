#!/bin/bash
# Same-box A/B of one experiment-build switch: scratch/ab_env.sh <ENV_NAME> [dtypes...]   (alternating, two passes; "on" = switch set)
R=$(cd "$(dirname "$0")/.." && pwd)
export SRGAN_HIP_LIB=$R/scratch/libsrgan_exp.so
name=$1; shift
for dt in ${@:-f32 bf16}; do
for pass in 1 2; do
  for v in on off; do
    if [ $v = on ]; then export $name=1; else unset $name; fi
    python3 $R/bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-micro 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$dt $name=$v', d['value'], d['ms_per_step'])" || exit 1
  done
done
done

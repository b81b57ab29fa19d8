#!/bin/bash
# Same-box A/B of the encoder shortcut on the side stream (experiment build, alternating, two passes, fp32 then bf16).
R=$(cd "$(dirname "$0")/.." && pwd)
export SRGAN_HIP_LIB=$R/scratch/libsrgan_exp.so
for dt in f32 bf16; do
for pass in 1 2; do
  for v in on off; do
    if [ $v = off ]; then export SRGAN_NO_PARALLEL_SHORTCUT=1; else unset SRGAN_NO_PARALLEL_SHORTCUT; fi
    python3 $R/bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-micro 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$dt $v', d['value'], d['ms_per_step'])" || exit 1
  done
done
done

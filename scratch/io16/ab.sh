#!/bin/bash
# Same-box A/B of the encoder's 16-bit activations (bf16 mode): experiment build, alternating, two passes.
R=$(cd "$(dirname "$0")/../.." && pwd)
export SRGAN_HIP_LIB=$R/scratch/libsrgan_exp.so
mkdir -p $R/gpurun_out/io
for pass in 1 2; do
  for v in on off; do
    if [ $v = off ]; then export SRGAN_NO_CONV_IO16=1; else unset SRGAN_NO_CONV_IO16; fi
    python3 $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-micro 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'])" || exit 1
  done
done

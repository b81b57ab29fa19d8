#!/bin/bash
# The other BASELINE configurations in graph mode on one box (headline line first): fp32 / bf16 each.
R=$(cd "$(dirname "$0")/.." && pwd)
run() { python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-micro "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %8.1f images/s %7.2f ms' % ('$*', d['value'], d['ms_per_step']))"; }
run
run --dtype bf16
run --pretrained-e
run --pretrained-e --dtype bf16
run --batch-per-gpu 64
run --batch-per-gpu 64 --dtype bf16
run --size 256 --batch-per-gpu 16
run --size 256 --batch-per-gpu 16 --dtype bf16
SRGAN_DP_FORCE=1 run
SRGAN_DP_FORCE=1 SRGAN_DP_GRAPH=0 run

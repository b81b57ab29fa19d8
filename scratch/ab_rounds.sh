#!/bin/bash
# same-box A/B of three trees: the end of round 2 (scratch/r2tree = 81ec157), the middle of round 3 (scratch/r3atree = f698ae3,
# before the RGB-head kernel / slab sums / forked second scale / ...) and HEAD; alternating, two passes
R=$(cd "$(dirname "$0")/.." && pwd)
for pass in 1 2; do
  for t in scratch/r2tree scratch/r3atree .; do
    (cd $R/$t && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-micro 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])")
  done
done

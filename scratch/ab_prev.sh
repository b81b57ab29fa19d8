#!/bin/bash
# same-box A/B of scratch/prevtree (a `git archive` of an earlier commit, built in place) against this tree; alternating, two passes
#   scratch/ab_prev.sh [dtypes...]
R=$(cd "$(dirname "$0")/.." && pwd)
for dt in ${@:-f32 bf16}; do
for pass in 1 2; do
  for t in scratch/prevtree .; do
    (cd $R/$t && python3 bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-micro --no-bf16 2>/dev/null < /dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$dt $t', d['value'], d['ms_per_step'])") || exit 1
  done
done
done

#!/bin/bash
# same-box A/B: the end of round 4 (scratch/r4tree = 97ce276, `git worktree add --detach scratch/r4tree 97ce276 && make -C
# scratch/r4tree/style-restricted_gan_amd/csrc -j8`) against HEAD, alternating, two passes; fp32 headline and the bf16 mode
R=$(cd "$(dirname "$0")/.." && pwd)
for pass in 1 2; do
  for t in scratch/r4tree .; do
    for f in "" "--dtype bf16"; do
      (cd $R/$t && python bench.py $f --steps 20 --warmup 5 --no-cpu-baseline --no-micro 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', '$f', d['value'], d['ms_per_step'])")
    done
  done
done

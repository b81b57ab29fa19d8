#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of scratch/h16r/bench.py for every ablation library
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/h16r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in $R/scratch/h16r/lib_*.so; do
  n=$(basename $lib .so); n=${n#lib_}
  SRGAN_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$n -- python3 $R/scratch/h16r/bench.py > $O/bench_$n.log 2>&1 || { tail -5 $O/bench_$n.log; exit 1; }
  f=$(find $O/p_$n -name "*kernel_stats.csv" | head -1)
  echo "== H16R_EXP=$n"; grep -i "halo16r" $f | awk -F, '{printf "%s calls %s avg %.1f us\n", substr($1,1,90), $2, $4/1000}'
  find $O/p_$n -name "*kernel_trace.csv" -delete
done

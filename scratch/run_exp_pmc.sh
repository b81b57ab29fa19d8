#!/bin/bash
# SQ cycle counters of the Winograd kernel for the experimental builds (cycles vs wall: separates DVFS from stalls)
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for e in 0 ${EXPS:-1 2 3 4 5 6 7}; do
  if [ $e != 0 ]; then export SRGAN_HIP_LIB=$R/scratch/libsrgan_exp$e.so; fi
  ONLY=G.res REP=3 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmcx/e$e -- python3 $R/scratch/bench_conv.py > /dev/null 2>&1
done

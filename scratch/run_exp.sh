#!/bin/bash
# times the G.res layer with experimental builds of the Winograd kernel (scratch/libsrgan_exp*.so)
cd "$(dirname "$0")/.."
echo "base"; ONLY=G.res python scratch/bench_conv.py 2>/dev/null | cut -c1-120
for e in ${EXPS:-1 2 3 4}; do echo "exp $e"; SRGAN_HIP_LIB=$PWD/scratch/libsrgan_exp$e.so ONLY=G.res python scratch/bench_conv.py 2>/dev/null | cut -c1-120; done

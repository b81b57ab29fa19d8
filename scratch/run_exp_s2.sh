#!/bin/bash
# ablation builds of wino_kernel (-DWINO_EXP=n, scratch/libsrgan_exp<n>.so) on the 4x4 / stride-2 generator layers
cd "$(dirname "$0")/.."
echo "base"; ONLY=G.down B=${B:-32} python scratch/bench_conv.py 2>/dev/null | cut -c1-150
for e in ${EXPS:-4 5 6 7}; do echo "exp $e"; SRGAN_HIP_LIB=$PWD/scratch/libsrgan_exp$e.so ONLY=G.down B=${B:-32} python scratch/bench_conv.py 2>/dev/null | cut -c1-150; done

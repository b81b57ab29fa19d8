#!/bin/bash
# per-layer times of the WINO42_EXP ablation builds (scratch/wino42/build_ablations.sh) next to the plain exp build
mkdir -p gpurun_out/r5b
out=gpurun_out/r5b/ablations.log; : > $out
for n in 0 "$@"; do
  lib=scratch/wino42/lib_$n.so; [ $n = 0 ] && lib=scratch/libsrgan_exp.so
  for only in G.down1 G.down2; do
    echo -n "EXP=$n " >> $out
    ONLY=$only REP=20 SRGAN_HIP_LIB=$lib timeout -k 10 120 python scratch/bench_conv.py 2>/dev/null | grep GFLOP >> $out
  done
done
cat $out

#!/bin/bash
mkdir -p gpurun_out/r5c
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d_fwd_bwd or random_geometries or conv_transpose2d or leaky_relu_backward_in_the_consumers_input_gradient or fused_leaky" > gpurun_out/r5c/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5c/tests.log
for only in G.down1 G.down2 D.c2; do
  ONLY=$only REP=20 timeout -k 10 120 python scratch/bench_conv.py 2>/dev/null | grep GFLOP
done
B=64 ONLY=D.c REP=20 timeout -k 10 120 python scratch/bench_conv.py 2>/dev/null | grep GFLOP
echo "no-epilogue:"; ONLY=G.down REP=20 SRGAN_HIP_LIB=scratch/wino42/lib_16.so timeout -k 10 120 python scratch/bench_conv.py 2>/dev/null | grep GFLOP
SRGAN_HIP_LIB=scratch/wino42/lib_128.so timeout -k 10 120 python scratch/wino42/diag.py 2>/dev/null | head -20

#!/bin/bash
# quick pass: conv parity cases + the per-layer times
mkdir -p gpurun_out/r5h
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d_fwd_bwd or random_geometries or conv_transpose2d or leaky_relu_backward_in_the_consumers_input_gradient or fused_leaky" > gpurun_out/r5h/tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r5h/tests.log
for only in G.down1 G.down2; do ONLY=$only REP=20 timeout -k 10 120 python scratch/bench_conv.py 2>/dev/null | grep GFLOP; done
B=64 ONLY=D.c REP=20 timeout -k 10 120 python scratch/bench_conv.py 2>/dev/null | grep GFLOP

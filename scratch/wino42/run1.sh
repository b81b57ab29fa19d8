#!/bin/bash
# first GPU pass of the F(4x4,2x2) kernels: parity of the stride-2 conv cases, then per-layer times against F(3x3,2x2)
set -o pipefail
mkdir -p gpurun_out/r5a
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv2d_fwd_bwd or random_geometries or conv_transpose2d or leaky_relu_backward_in_the_consumers_input_gradient" > gpurun_out/r5a/tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r5a/tests.log
tail -5 gpurun_out/r5a/tests.log
for only in G.down1 G.down2 D.c2 D.c3; do
  ONLY=$only REP=20 timeout -k 10 120 python scratch/bench_conv.py >> gpurun_out/r5a/bench_new.log 2>&1
  ONLY=$only REP=20 SRGAN_HIP_LIB=scratch/libsrgan_exp.so SRGAN_NO_WINOGRAD42=1 timeout -k 10 120 python scratch/bench_conv.py >> gpurun_out/r5a/bench_old.log 2>&1
  B=64 ONLY=$only REP=20 timeout -k 10 120 python scratch/bench_conv.py >> gpurun_out/r5a/bench_new_b64.log 2>&1
  B=64 ONLY=$only REP=20 SRGAN_HIP_LIB=scratch/libsrgan_exp.so SRGAN_NO_WINOGRAD42=1 timeout -k 10 120 python scratch/bench_conv.py >> gpurun_out/r5a/bench_old_b64.log 2>&1
done
echo NEW; cat gpurun_out/r5a/bench_new.log; echo OLD; cat gpurun_out/r5a/bench_old.log
echo NEW64; cat gpurun_out/r5a/bench_new_b64.log; echo OLD64; cat gpurun_out/r5a/bench_old_b64.log

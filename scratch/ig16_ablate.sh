#!/bin/bash
# igemm16_kernel timing ablations (exp build): per SRGAN_IG16_EXP value the kernel's average duration on one layer
R=$(cd "$(dirname "$0")/.." && pwd)
only=$1; shift
for e in "$@"; do
  bash $R/scratch/prof_conv.sh ab_$e "$only" DT=bf16 B=64 SRGAN_HIP_LIB=$R/scratch/libsrgan_exp.so SRGAN_IG16_EXP=$e SRGAN_NO_SPLITK=${NOSPLIT:-} 2>&1 | grep -E "^==|igemm16"
done

#!/bin/bash
# builds scratch/wino42/lib_<n>.so = the exp library with -DWINO42_EXP=<n> (timing ablations of wino42_kernel, WRONG results):
#   bit 0 (1)  only the first chunk is gathered      bit 1 (2)  no transform arithmetic      bit 2 (4)  no LDS stores of V
#   bit 3 (8)  filter fragments of chunk 0 only      bit 4 (16) no epilogue                  bit 5 (32) no MFMAs
set -e
cd "$(dirname "$0")/../../style-restricted_gan_amd/csrc"
make exp -j8 > /dev/null
for n in "$@"; do
  rm -rf build_exp_$n; cp -rp build_exp build_exp_$n; rm -f build_exp_$n/conv_wino42.o
  make exp EXPFLAGS=-DWINO42_EXP=$n EXPDIR=$PWD/build_exp_$n EXPOUT=$PWD/../../scratch/wino42/lib_$n.so > /dev/null
  rm -rf build_exp_$n
done
ls -la ../../scratch/wino42/*.so

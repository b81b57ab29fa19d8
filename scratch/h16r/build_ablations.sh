#!/bin/bash
# builds scratch/h16r/lib_<n>.so = the exp library with -DH16R_EXP=<n> (timing ablations of halo16r_kernel, WRONG results):
#   bit 0 (1) no result stores   bit 1 (2) no filter-fragment loads in the loop   bit 2 (4) no activation-fragment reads in the loop
#   bit 3 (8) no MFMAs           bit 4 (16) no halo streaming in the loop
set -e
cd "$(dirname "$0")/../../style-restricted_gan_amd/csrc"
make exp -j8 > /dev/null
for n in "$@"; do
  rm -rf build_exp_$n; cp -rp build_exp build_exp_$n; rm -f build_exp_$n/conv_halo16.o
  make exp EXPFLAGS=-DH16R_EXP=$n EXPDIR=$PWD/build_exp_$n EXPOUT=$PWD/../../scratch/h16r/lib_$n.so > /dev/null
  rm -rf build_exp_$n
done
ls -la ../../scratch/h16r/*.so

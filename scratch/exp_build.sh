#!/bin/bash
# what-if builds of the conv kernels: scratch/_exp/libstp_exp{1,2}.so (see STP_EXP in conv_igemm.hip / conv_wgrad.hip)
set -e
cd "$(dirname "$0")/.."
C=segmentation_training_pipeline_amd/csrc
for n in ${EXPS:-1 2 3}; do
  for f in conv_igemm conv_wgrad; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Iinclude -DSTP_EXP=$n -c $C/$f.hip -o scratch/_exp/${f}_exp$n.o &
  done
  wait
  objs=$(ls $C/_obj/*.o | grep -v -e conv_igemm.o -e conv_wgrad.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/_exp/libstp_exp$n.so scratch/_exp/conv_igemm_exp$n.o scratch/_exp/conv_wgrad_exp$n.o $objs
done
ls -la scratch/_exp/*.so

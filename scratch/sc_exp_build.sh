#!/bin/bash
# what-if builds of the small-channel kernels: scratch/_exp/libstp_sc_exp<n>.so (see STP_EXP in conv_sc.hip)
set -e
cd "$(dirname "$0")/.."
C=segmentation_training_pipeline_amd/csrc
mkdir -p scratch/_exp
for n in ${EXPS:-11 12 13}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Iinclude -DSTP_EXP=$n -c $C/conv_sc.hip -o scratch/_exp/conv_sc_exp$n.o &
done
wait
for n in ${EXPS:-11 12 13}; do
  objs=$(ls $C/_obj/*.o | grep -v -e "conv_sc\.o" -e "\.f16\.o" -e "_prev\.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/_exp/libstp_sc_exp$n.so scratch/_exp/conv_sc_exp$n.o $objs
done
ls -la scratch/_exp/*.so

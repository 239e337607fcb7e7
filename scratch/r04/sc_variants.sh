#!/bin/bash
# what-if builds of the small-channel kernels with extra -D flags: scratch/_exp/libstp_sc_exp<tag>.so (loaded by scratch/sc_bench.py under EXP=<tag>)
#   usage: sc_variants.sh tag1:"-DSC_RING=6" tag2:"-DSC_RING=8 -DSC_WPE_SMALL=5" ...
set -e
cd "$(dirname "$0")/../.."
C=segmentation_training_pipeline_amd/csrc
mkdir -p scratch/_exp
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  for f in conv_sc conv_sc_lean; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Iinclude $flags -c $C/$f.hip -o scratch/_exp/${f}_exp$tag.o &
  done
done
wait
for spec in "$@"; do
  tag=${spec%%:*}
  objs=$(ls $C/_obj/*.o | grep -v -e "conv_sc\.o" -e "conv_sc_lean\.o" -e "\.f16\.o" -e "_prev\.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/_exp/libstp_sc_exp$tag.so scratch/_exp/conv_sc_exp$tag.o scratch/_exp/conv_sc_lean_exp$tag.o $objs
done
ls -la scratch/_exp/*.so

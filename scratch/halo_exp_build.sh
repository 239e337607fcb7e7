#!/bin/bash
# scratch builds of conv_halo.hip: scratch/_exp/libstp_halo_<tag>.so with extra -D flags, e.g.  halo_exp_build.sh timing -DSTP_TIMING
set -e
cd "$(dirname "$0")/.."
C=segmentation_training_pipeline_amd/csrc
tag=$1; shift
mkdir -p scratch/_exp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on "$@" -c $C/conv_halo.hip -o scratch/_exp/conv_halo_$tag.o
objs=$(ls $C/_obj/*.o | grep -v -e conv_halo.o -e "\.f16\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/_exp/libstp_halo_$tag.so scratch/_exp/conv_halo_$tag.o $objs
ls -la scratch/_exp/libstp_halo_$tag.so

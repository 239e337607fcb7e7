#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-stem_ab}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "stem" 2>&1 | tail -5 | tee $O/tests.txt
for v in 0 1; do STP_STEM_LEAN=$v python scratch/launch_table.py 2>&1 | grep -E "^conv0 |^total" | sed "s/^/STEM_LEAN=$v /"; done | tee $O/rows.txt
STP_LIB=$PWD/scratch/_exp/libstp_sc_expwl.so python scratch/launch_table.py 2>&1 | grep -E "^conv0 |^total" | sed "s/^/late-wait /" | tee -a $O/rows.txt
bash scratch/r03_ab.sh $(basename $O) <<EOF
single STP_STEM_LEAN=0
persistent STP_STEM_LEAN=1
latewait STP_LIB=$PWD/scratch/_exp/libstp_sc_expwl.so
EOF

#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-stem_ab}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "stem or weight_gradient" 2>&1 | tail -5 | tee $O/tests.txt
for v in 0 1; do STP_STEM_LEAN=$v STP_STEM_WG_LEAN=$v python scratch/launch_table.py 2>&1 | grep -E "^conv0 |^total|wgrad_reduce " | sed "s/^/STEM_LEAN=$v /"; done | tee $O/rows.txt
bash scratch/r03_ab.sh $(basename $O) <<EOF
single STP_STEM_LEAN=0 STP_STEM_WG_LEAN=0
persistent STP_STEM_LEAN=1 STP_STEM_WG_LEAN=1
EOF

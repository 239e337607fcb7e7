#!/bin/bash
# quick check of a lean-kernel change: small-channel op tests, sc_bench, launch-table rows, 40-step A/B against the generic kernels
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-lean_q}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt; cat $O/tests.txt
timeout 300 python scratch/sc_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/sc_bench.txt
python scratch/launch_table.py > $O/lt_lean.txt 2>&1
grep -hE "^(decoder_stage4|final_conv|decoder_stage3_conv2)" $O/lt_lean.txt
bash scratch/r03_ab.sh $(basename $O) <<'EOF'
generic STP_SC_LEAN=0
lean STP_SC_LEAN=1
EOF

#!/bin/bash
# weight-gradient op tests + launch tables with the lean small-channel kernels off / on (same box), full GPU suite tail
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-lean_wg}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/tests_wg.txt
cat $O/tests_wg.txt
STP_SC_LEAN=0 python scratch/launch_table.py > $O/lt_generic.txt 2>&1
STP_SC_LEAN=1 python scratch/launch_table.py > $O/lt_lean.txt 2>&1
for f in generic lean; do echo "== $f"; grep -E "^(decoder_stage4|final_conv|decoder_stage3_conv2) " $O/lt_$f.txt; done | tee $O/rows.txt
bash scratch/r03_ab.sh $(basename $O) <<'EOF'
generic STP_SC_LEAN=0
lean STP_SC_LEAN=1
EOF
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > $O/gputest_tail.txt; grep -E "passed|failed|error" $O/gputest_tail.txt

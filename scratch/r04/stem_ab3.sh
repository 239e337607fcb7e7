#!/bin/bash
cd "$(dirname "$0")/../.."
for rep in 1 2; do
python scratch/launch_table.py 2>&1 | grep -E "^conv0 .*fwd" | sed "s/^/default  /"
STP_LIB=$PWD/scratch/_exp/libstp_sc_expct.so python scratch/launch_table.py 2>&1 | grep -E "^conv0 .*fwd" | sed "s/^/counted  /"
done

#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "loss or step or bce or dice" 2>&1 | grep -E "passed|failed" | tail -2
python scratch/launch_table.py 2>&1 | grep -E "^total|stp_sigmoid_bce_dice"

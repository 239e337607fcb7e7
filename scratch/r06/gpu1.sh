# round 6, GPU call 1: new parity tests, the three bench configs (short CPU legs), GEMM floor tables
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests/test_abi.py -q -x > $O/abi.txt 2>&1; tail -2 $O/abi.txt
python -m pytest tests/test_model_gpu.py -q -x -s -k "other_graphs or storage_quantised_oracle" > $O/parity16.txt 2>&1; tail -3 $O/parity16.txt
python -m pytest tests/test_ops_gpu.py -q -x -k "space_to_depth or folded_shortcut" > $O/ops_s2d.txt 2>&1; tail -2 $O/ops_s2d.txt
python bench.py --cpu-baseline short > $O/bench_config1.json 2> $O/bench_config1.err; tail -c 600 $O/bench_config1.json
python bench.py --config 3 --cpu-baseline short > $O/bench_config3.json 2> $O/bench_config3.err; tail -c 300 $O/bench_config3.err
python bench.py --config 4 --cpu-baseline short > $O/bench_config4.json 2> $O/bench_config4.err; tail -c 300 $O/bench_config4.err
python scratch/r06/gemm_floor_table.py 4 > $O/floor_config4.txt 2>&1
python scratch/r06/gemm_floor_table.py 3 > $O/floor_config3.txt 2>&1
python scratch/r06/gemm_floor_table.py 1 > $O/floor_config1.txt 2>&1
head -c 1500 $O/bench_config4.json

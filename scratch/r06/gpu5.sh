cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
timeout 300 python scratch/r06/pw_debug.py 2>&1 | grep -v amdgpu | grep "bad" | head -30
timeout 1200 python -m pytest tests/test_ops_gpu.py -q -k "pointwise or batched_reduce" > $O/ops_pw.txt 2>&1; tail -5 $O/ops_pw.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -q -x -k "other_graphs or pspnet or fpn or resnet50 or storage_quantised or hipgraph" > $O/model_pw.txt 2>&1; tail -5 $O/model_pw.txt
STP_PW=1 timeout 300 python scratch/r06/pw_bench.py 2>&1 | grep -v amdgpu > $O/pw_bench_on.txt; cat $O/pw_bench_on.txt
for rb in 0; do
  for c in 4 3 1; do
    STP_WGRAD_REDUCE_BATCH=$rb timeout 600 python bench.py --config $c --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('reduce batch $rb config $c', d['ms_per_step'])" | tee -a $O/step_ab.txt
  done
done
for t in; do STP_1X1_DEEPK_TILE=$t timeout 600 python bench.py --config 4 --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deepk tile $t config 4', d['ms_per_step'])" | tee -a $O/step_ab.txt; done
STP_WGRAD_LONE_GROUP_GFLOP=600 timeout 600 python bench.py --config 3 --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lone group 600 config 3', d['ms_per_step'])" | tee -a $O/step_ab.txt

# round 6, GPU call 2: the pointwise streaming kernel - op tests, layer bench A/B, step A/B on configs 3 / 4
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "pointwise" > $O/ops_pw.txt 2>&1; tail -5 $O/ops_pw.txt
STP_PW=1 timeout 300 python scratch/r06/pw_bench.py > $O/pw_bench_on.txt 2>&1
STP_PW=0 timeout 300 python scratch/r06/pw_bench.py > $O/pw_bench_off.txt 2>&1
paste -d'\n' $O/pw_bench_on.txt $O/pw_bench_off.txt | grep -v amdgpu
for pw in 1 0 1 0; do
  for c in 4 3; do
    STP_PW=$pw timeout 600 python bench.py --config $c --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('STP_PW=$pw config $c', d['ms_per_step'])" | tee -a $O/step_ab.txt
  done
done
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "other_graphs or pspnet or fpn or resnet50" > $O/model_pw.txt 2>&1; tail -5 $O/model_pw.txt

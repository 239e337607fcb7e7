# round 6, GPU call 3: pointwise kernel after the wait fix - op tests (all), model tests, layer bench, step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py -q -k "pointwise" > $O/ops_pw.txt 2>&1; tail -5 $O/ops_pw.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -q -k "other_graphs or pspnet or fpn or resnet50 or storage_quantised or hipgraph" > $O/model_pw.txt 2>&1; tail -5 $O/model_pw.txt
STP_PW=1 timeout 300 python scratch/r06/pw_bench.py > $O/pw_bench_on.txt 2>&1
grep -v amdgpu $O/pw_bench_on.txt
for pw in 1 0 1 0; do
  for c in 4 3; do
    STP_PW=$pw timeout 600 python bench.py --config $c --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('STP_PW=$pw config $c', d['ms_per_step'])" | tee -a $O/step_ab.txt
  done
done
for v in plain nt; do for g in 2048 4096 8192; do STP_CALIB_COPY=$v STP_CALIB_COPY_WGS=$g python - <<PY
import sys; sys.path.insert(0,'.')
import torch, bench
print("copy $v $g", bench.box_calibration(torch.device("cuda:0"), seconds=0.3)["copy_gbs"])
PY
done; done 2>&1 | grep copy | tee $O/copy_variants.txt
STP_WGRAD_LONE_GROUP_GFLOP=600 timeout 600 python bench.py --config 3 --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lone group 600 config 3', d['ms_per_step'])" | tee -a $O/step_ab.txt

cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
for t in 0 133 129 97 70; do
  STP_1X1_SMALLM_TILE=$t timeout 600 python bench.py --config 3 --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('small-M tile $t config 3', d['ms_per_step'], 'without augmentation', d['ms_per_step_without_augmentation'])" | tee -a $O/step_ab.txt
done
bash scratch/r06/prof_r06.sh r06h 4 short
bash scratch/r06/prof_r06.sh r06h 3 short
bash scratch/r06/prof_r06.sh r06h 1 short
ls $O

# A/B of a compile-time switch in conv_wgrad.hip on one box: current build, then rebuild conv_wgrad.o with the macro and relink
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_pipe}; MACRO=${2:-STP_T9_NOPIPE}; mkdir -p $O; cd $R
run() { for rep in 1 2; do python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile --sustain 0 2>>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1 rep$rep %.3f ms' % d['ms_per_step'])" | tee -a $O/ab.txt; done; }
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "grouped_weight" 2>&1 | tail -2 | tee -a $O/ab.txt
run default_build
python scratch/launch_table.py bf16 2>/dev/null | grep "group\[" | grep wgrad | tee -a $O/ab.txt
cd segmentation_training_pipeline_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-result -D$MACRO -c conv_wgrad.hip -o _obj/conv_wgrad.o 2>>$O/err.txt
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libstp_hip.so _obj/augment.o _obj/bn_pool.o _obj/conv_halo.o _obj/conv_igemm.o _obj/conv_sc.o _obj/conv_wgrad.o _obj/deeplab.o _obj/loss_optim.o _obj/lovasz.o 2>>$O/err.txt
cd $R
run with_$MACRO
python scratch/launch_table.py bf16 2>/dev/null | grep "group\[" | grep wgrad | tee -a $O/ab.txt

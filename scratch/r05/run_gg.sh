R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_gg; mkdir -p $O; cd $R
for sw in 128 256 128 256; do
  STP_DGRAD1X1_BNB_K=$sw timeout 900 python scratch/other_configs_bench.py 2>&1 | grep "FPN/resnet50 1024x1024 3-class bs4 bf16\|PSPNet" | cut -c1-120 | sed "s/^/bnbK=$sw /" >> $O/other.txt
done
cat $O/other.txt

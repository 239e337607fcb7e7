# round-4 profile set: bench lines, rocprofv3 kernel stats (graph-only + eager), PMC traffic (FETCH / WRITE, separate passes), SQ counters,
# launch table, other configs.   usage (on the GPU box): bash scratch/r05/prof_r05.sh <tag>      -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r05z}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
Q="--no-cpu-baseline --sustain 0 --no-feed"
python bench.py > $O/bench_default.json 2> $O/bench.err
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench.json 2>> $O/bench.err
python bench.py --dtype fp32 --steps 10 --warmup 3 $Q --no-kernel-profile > $O/bench_fp32.json 2>> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktg -o kt -- python $R/bench.py --steps 10 --warmup 3 $Q --no-kernel-profile > $O/bench_profiled_graph_only.json 2> $O/ktg.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kte -o kte -- python $R/bench.py --eager --steps 10 --warmup 3 $Q > $O/bench_eager_profiled.json 2> $O/kte.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --eager $Q --no-kernel-profile > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --eager $Q --no-kernel-profile > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -o q -- python $R/bench.py --steps 2 --warmup 1 --eager $Q --no-kernel-profile > $O/pmc_sq.json 2> $O/pmc_sq.err
cd $R
python scratch/pmc_aggregate.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json
python scratch/pmc_aggregate_sq.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) $(find $O/pmc_sq -name "*kernel_trace.csv" | head -1) $O/pmc_sq.json
cp $(find $O/ktg -name "*kernel_stats.csv" | head -1) $O/kernel_stats_graph_only.csv
cp $(find $O/kte -name "*kernel_stats.csv" | head -1) $O/kernel_stats_eager.csv
python - <<PY > $O/graph_gaps.txt
import csv,glob
f=glob.glob("$O/ktg/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "augment_kernel" in r["Kernel_Name"]]
idx=idx[-10:]
seg=rows[idx[0]:]
# wall clock per step from augmentation kernel to augmentation kernel (9 intervals; kernels launched after the run do not count)
wall=(int(rows[idx[-1]]["Start_Timestamp"])-int(rows[idx[0]]["Start_Timestamp"]))/(len(idx)-1)
inner=rows[idx[0]:idx[-1]]
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in inner)/(len(idx)-1)
print("last %d steps: kernels/step %.1f  wall/step %.1f us  sum of kernel durations/step %.1f us  gaps/step %.1f us"%(len(idx),len(inner)/(len(idx)-1),wall/1e3,busy/1e3,(wall-busy)/1e3))
fam={}
for r in seg:
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    fam[k]=fam.get(k,[0,0]); fam[k][0]+=1; fam[k][1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,v in sorted(fam.items(),key=lambda kv:-kv[1][1]): print("%-70s %6.1f launches/step %9.1f us/step"%(k[:70],v[0]/len(idx),v[1]/1e3/len(idx)))
PY
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/ktg $O/kte
python scratch/r04/family_table.py $O/graph_gaps.txt > $O/family_table.txt 2>&1
python scratch/launch_table.py > $O/launch_table.txt 2>&1
python scratch/other_configs_bench.py > $O/other_configs.txt 2>&1
